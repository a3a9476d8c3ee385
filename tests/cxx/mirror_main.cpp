// A caller written against the reference's C++ interface (cf. reference src/main.cpp:261-300):
// LoadScene -> Camera -> Scene::Init -> BeginRender -> Render x N -> EndRender, then dump the accumulator.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../gpu_pathtracer_amd/csrc/pathtracer.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const char *scene_file = argv[1];
    const unsigned spp = (unsigned)std::atoi(argv[2]);
    const char *out_file = argv[3];
    GlobalConfig config;
    Scene scene;
    if (!LoadScene(scene_file, config, scene)) {
        std::fprintf(stderr, "Failed to load scene: %s\n", gpt_last_error());
        return 1;
    }
    Camera cam = config.camera;
    float2_t res;
    res.x = (float)config.width;
    res.y = (float)config.height;
    Camera *camera = new Camera(cam.position, cam.u, cam.v, cam.w, res, 0.1f, cam.fov, cam.apertureRadius,
                                cam.focalDistance, cam.filmic != 0, cam.medium);
    camera->environment = cam.environment;
    scene.Init(camera, scene_file);
    if (argc > 4 && argv[4][0] == 'h') { std::printf("host-only ok: %d prims %d nodes\n", (int)scene.bvh.prims.size(), scene.bvh.total_nodes); return 0; }

    BeginRender(scene, config.width, config.height, config.epsilon);
    if (!CurrentRenderContext()) return 3;
    std::vector<float> acc((size_t)config.width * config.height * 3);
    if (argc > 4 && argv[4][0] == 'o') {
        // The reference's display loop (src/main.cpp:134-144): Render() into a device `output` buffer, then use that
        // buffer from the DEFAULT stream straight away - no synchronisation call exists in the reference's interface.
        float3_t *output = nullptr;
        if (hipMalloc((void **)&output, acc.size() * sizeof(float)) != hipSuccess) return 5;
        if (hipMemset(output, 0, acc.size() * sizeof(float)) != hipSuccess) return 5;
        for (unsigned iter = 1; iter <= spp; ++iter)
            Render(scene, config.width, config.height, camera, iter, iter == 1, output);
        if (hipMemcpy(acc.data(), output, acc.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 6;
        EndRender();
        FILE *fo = std::fopen(out_file, "wb");
        std::fwrite(acc.data(), sizeof(float), acc.size(), fo);
        std::fclose(fo);
        return 0;
    }
    for (unsigned iter = 1; iter <= spp; ++iter)
        Render(scene, config.width, config.height, camera, iter, iter == 1, nullptr);
    if (gpt_read_accum(CurrentRenderContext(), acc.data()) != GPT_OK) return 4;
    EndRender();
    FILE *f = std::fopen(out_file, "wb");
    std::fwrite(acc.data(), sizeof(float), acc.size(), f);
    std::fclose(f);
    return 0;
}

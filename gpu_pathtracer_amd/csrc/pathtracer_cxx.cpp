// pathtracer_cxx.cpp — BeginRender / Render / EndRender with the reference's C++ signatures (reference
// src/pathtracer.h:10-12, src/pathtracer.cu:2568-2750), and the gpt_scene_* entry points of the C ABI,
// both thin layers over gpt_begin / gpt_render / gpt_end.
//
// Like the reference, the three C++ calls keep ONE renderer per process in file-scope state
// (src/pathtracer.cu:9-20); callers that need several renderers use the gpt_ctx API directly.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>

#include "host_util.h"
#include "pathtracer.h"

static gpt_ctx *g_ctx = nullptr;

gpt_ctx *CurrentRenderContext() { return g_ctx; }

static void report(const char *where)
{
    // the reference prints "<msg> in <file> at line <n>" and breaks into the debugger (src/common.h:29-39)
    std::fprintf(stderr, "%s: %s\n", where, gpt_last_error());
}

void BeginRender(Scene &scene, unsigned width, unsigned height, float ep)
{
    if (g_ctx) EndRender();
    gpt_scene_desc desc;
    std::vector<gpt_texture> tex;
    scene.Describe(desc, tex);
    int device = 0;
    if (gpt_begin(&desc, width, height, ep, device, &g_ctx) != GPT_OK) {
        g_ctx = nullptr;
        report("BeginRender");
    }
}

void Render(Scene &scene, unsigned width, unsigned height, Camera *camera, unsigned iter, bool reset, float3_t *output)
{
    (void)width; (void)height;                // fixed at BeginRender, as in the reference
    if (!g_ctx) { std::fprintf(stderr, "Render: BeginRender has not succeeded\n"); return; }
    // the integrator is read from the scene on every call (pathtracer.cu:2711-2715)
    if (gpt_set_integrator(g_ctx, (int32_t)scene.integrator.type, scene.integrator.maxDepth, scene.integrator.maxDist) != GPT_OK) {
        report("Render");
        return;
    }
    if (gpt_render(g_ctx, camera, iter, 1, reset ? 1 : 0, reinterpret_cast<float *>(output)) != GPT_OK) report("Render");
}

void EndRender()
{
    if (g_ctx) gpt_end(g_ctx);
    g_ctx = nullptr;
}

// ---- C ABI: LoadScene + InitScene (src/main.cpp:261-278) behind an opaque handle ------------------
struct gpt_scene {
    Scene scene;
    GlobalConfig config;
    Camera *camera = nullptr;
    std::vector<gpt_texture> tex;
    ~gpt_scene() { delete camera; }
};

extern "C" {

int gpt_scene_load(const char *json_path, gpt_scene **out) { return gpt_scene_load_cached(json_path, 0, out); }

int gpt_scene_load_cached(const char *json_path, int use_bvh_cache, gpt_scene **out)
{
    return gpt_scene_load_ex(json_path, use_bvh_cache ? GPT_LOAD_BVH_CACHE : 0, out);
}

int gpt_scene_load_ex(const char *json_path, int flags, gpt_scene **out)
{
    if (!json_path || !out) { gpt_set_error("gpt_scene_load: null argument"); return GPT_ERR_INVALID_ARG; }
    *out = nullptr;
    gpt_scene *s = new gpt_scene();
    // the reference always reads/writes <scene dir>/bvh.cache (src/bvh.cpp:189-218); here it is opt-in
    s->scene.use_bvh_cache = (flags & GPT_LOAD_BVH_CACHE) != 0;
    s->scene.use_sbvh = (flags & GPT_LOAD_SBVH) != 0;
    s->scene.reference_bvh = (flags & GPT_LOAD_REFERENCE_BVH) != 0;
    try {                                            // (no exception crosses the C ABI: a file that asks for more memory than there is)
        if (!LoadScene(json_path, s->config, s->scene)) {
            delete s;
            return std::strstr(gpt_last_error(), "Parse scene error") ? GPT_ERR_PARSE : GPT_ERR_IO;
        }
        // InitScene, src/main.cpp:267-272: distance is the literal 0.1f
        const Camera &c = s->config.camera;
        gpt_float2 res;
        res.x = (float)s->config.width;
        res.y = (float)s->config.height;
        s->camera = new Camera(c.position, c.u, c.v, c.w, res, 0.1f, c.fov, c.apertureRadius, c.focalDistance, c.filmic != 0, c.medium);
        s->camera->environment = c.environment;
        s->scene.Init(s->camera, json_path);
        // Scene::Init is void like the reference's; a builder that refused the primitives (a non-finite vertex after a singular
        // transform ...) leaves them where they were and the tree empty: an error here, not a scene that renders the background
        if (!s->scene.primitives.empty() && s->scene.bvh.prims.empty()) {
            gpt_set_error("gpt_scene_load: the BVH builder refused the primitives of %s (a non-finite vertex?)", json_path);
            delete s;
            return GPT_ERR_INVALID_ARG;
        }
    } catch (const std::exception &e) {
        gpt_set_error("gpt_scene_load: %s while loading %s", e.what(), json_path);
        delete s;
        return GPT_ERR_IO;
    }
    *out = s;
    return GPT_OK;
}

int gpt_scene_get_desc(const gpt_scene *scene, gpt_scene_desc *desc_out)
{
    if (!scene || !desc_out) { gpt_set_error("gpt_scene_get_desc: null argument"); return GPT_ERR_INVALID_ARG; }
    gpt_scene *s = const_cast<gpt_scene *>(scene);
    s->scene.Describe(*desc_out, s->tex);
    return GPT_OK;
}

int gpt_scene_get_config(const gpt_scene *scene, int32_t *width, int32_t *height, float *epsilon, gpt_camera *camera_out)
{
    if (!scene) { gpt_set_error("gpt_scene_get_config: null scene"); return GPT_ERR_INVALID_ARG; }
    if (width) *width = scene->config.width;
    if (height) *height = scene->config.height;
    if (epsilon) *epsilon = scene->config.epsilon;
    if (camera_out) *camera_out = *static_cast<const gpt_camera *>(scene->camera);
    return GPT_OK;
}

int gpt_scene_set_integrator(gpt_scene *scene, int32_t integrator_type, int32_t max_depth)
{
    if (!scene || integrator_type < 0 || integrator_type > 7) { gpt_set_error("gpt_scene_set_integrator: invalid argument"); return GPT_ERR_INVALID_ARG; }
    scene->scene.integrator.type = (IntegratorType)integrator_type;
    scene->scene.integrator.maxDepth = max_depth;
    return GPT_OK;
}

int gpt_scene_free(gpt_scene *scene)
{
    delete scene;
    return GPT_OK;
}

}  // extern "C"

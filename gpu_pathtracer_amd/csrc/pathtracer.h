// pathtracer.h — the reference's C++ host interface for this path, on top of the C ABI (include/gpt.h).
//
// Same names, argument meaning and call order as the reference:
//   bool LoadScene(const char* filename, GlobalConfig& config, Scene& scene);   reference src/parsescene.h:26
//   void Scene::Init(Camera* cam, std::string file);                            reference src/scene.h:50
//   void BeginRender(Scene& scene, unsigned width, unsigned height, float ep);  reference src/pathtracer.h:11
//   void Render(Scene&, unsigned w, unsigned h, Camera*, unsigned iter, bool reset, float3* output);  :10
//   void EndRender();                                                            :12
// The record types are the reference's layouts (include/gpt_types.h), so a caller written against the
// reference's headers (src/main.cpp:261-300) compiles against this one by swapping the include.
//
// Deliberate differences: LoadScene returns false instead of exit(1) for a missing material / mesh /
// environment map (the message is in gpt_last_error()); only "pt" scenes can be rendered; `output` stays a
// DEVICE pointer (it may be NULL when the caller reads the accumulator instead).
#pragma once

#include <string>
#include <vector>

#include "../../include/gpt.h"

#pragma GCC visibility push(default)

typedef gpt_float2 float2_t;
typedef gpt_float3 float3_t;
typedef gpt_vertex Vertex;
typedef gpt_triangle Triangle;
typedef gpt_primitive Primitive;
typedef gpt_bvh_node LinearBVHNode;
typedef gpt_material Material;
typedef gpt_area Area;
typedef gpt_infinite Infinite;

struct Texture {                       // reference src/texture.h:9-28
    std::vector<gpt_uchar4> data;
    int width = 0, height = 0;
};

class Camera : public gpt_camera {     // reference src/camera.h:8-129
public:
    Camera();
    Camera(float3_t pos, float3_t uu, float3_t vv, float3_t ww, float2_t res, float dis, float angle, float radius,
           float focal, bool filmic, int medium);
    void Lookat(const float3_t &eye_pos, const float3_t &dest, const float3_t &up);
};

struct BBox {
    float3_t fmin, fmax;
};

class BVH {                            // reference src/bvh.h:31-47
public:
    LinearBVHNode *linear_root = nullptr;
    int total_nodes = 0;
    std::vector<Primitive> prims;
    BBox root_box;
    ~BVH();
    // reads <scene dir>/bvh.cache when it matches the primitives, else builds and writes it (src/bvh.cpp:189-218)
    void LoadOrBuildBVH(std::vector<Primitive> &primitives, std::string file);
    void Build(std::vector<Primitive> &primitives);
    // gpt_sbvh_build instead (spatial splits; not in the reference, whose src/sbvh.h is empty)
    void BuildSplit(std::vector<Primitive> &primitives, float alpha = 1e-5f);
    std::vector<int> prim_origin;       // BuildSplit: input index of every (possibly duplicated) primitive
};

enum IntegratorType { IT_AO = 0, IT_PT, IT_VPT, IT_LT, IT_BDPT, IT_MLT, IT_SPPM, IT_IR };   // src/scene.h:15-24

class Scene {                          // reference src/scene.h:26-84
public:
    std::vector<Primitive> primitives;
    std::vector<Material> materials;
    std::vector<Area> lights;
    std::vector<Texture> textures;
    std::vector<gpt_medium> mediums;         // parsescene.cpp:72-137
    std::vector<std::vector<float>> density_grids;   // own the heterogeneous media's density arrays
    std::vector<float> lightDistribution;
    std::vector<float3_t> infinite_data;     // owns infinite.data
    Camera *camera = nullptr;
    Infinite infinite;
    BVH bvh;
    struct {
        IntegratorType type = IT_PT;
        union {
            float maxDist;
            int maxDepth;
        };
        float vplBias = 0.f;
        float initRadius = 0.f;
        int photonsPerIteration = 0;
    } integrator;
    bool use_bvh_cache = false;          // the reference always uses <scene dir>/bvh.cache; opt-in here
    bool use_sbvh = false;               // GPT_LOAD_SBVH: spatial-split tree instead of the reference's builder
    bool reference_bvh = false;          // GPT_LOAD_REFERENCE_BVH: the reference's builder whatever tree it makes.  Neither flag: the reference's
                                         // builder, and the split tree instead when that tree has a leaf of more than 16 primitives (the
                                         // reference makes ONE leaf of any set whose box is thinner than 1e-4 - bvh.cpp:43 - however large)

    Scene();
    void Init(Camera *cam, std::string file);
    // what BeginRender reads (include/gpt_types.h gpt_scene_desc); valid until the scene changes
    void Describe(gpt_scene_desc &desc, std::vector<gpt_texture> &texture_records) const;
};

struct GlobalConfig {                  // reference src/parsescene.h:8-24
    int width = 512, height = 512;
    Camera camera;
    float camera_move_speed = 0.1f;
    float epsilon = 0.001f;
};

bool LoadScene(const char *filename, GlobalConfig &config, Scene &scene);

void BeginRender(Scene &scene, unsigned width, unsigned height, float ep);
void Render(Scene &scene, unsigned width, unsigned height, Camera *camera, unsigned iter, bool reset, float3_t *output);
void EndRender();
// the context behind the three calls above (NULL outside BeginRender..EndRender)
gpt_ctx *CurrentRenderContext();

#pragma GCC visibility pop

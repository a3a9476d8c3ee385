/*
 * gpt_types.h — plain-old-data records exchanged across the drop-in boundary.
 *
 * Every record is layout-compatible (size and field offsets) with the struct
 * the reference uploads to the GPU in BeginRender() (reference
 * src/pathtracer.cu:2568-2676), so a reference-built `Scene` can be handed to
 * this library by pointer without conversion.  Offsets were taken from the
 * reference headers under CUDA alignment rules (float2 is 8-byte aligned,
 * float3 is 4-byte aligned); see SURVEY.md §8(a).
 *
 *   record            reference type      reference file:line
 *   gpt_vertex        Vertex        48 B  src/mesh.h:13-18
 *   gpt_triangle      Triangle     168 B  src/mesh.h:20-26
 *   gpt_primitive     Primitive    176 B  src/primitive.h:9-23
 *   gpt_bvh_node      LinearBVHNode 40 B  src/bvh.h:19-29
 *   gpt_material      Material      72 B  src/material.h:19-27
 *   gpt_area          Area         192 B  src/area.h:7-11
 *   gpt_infinite      Infinite      72 B  src/infinite.h:6-13
 *   gpt_camera        Camera       104 B  src/camera.h:8-26
 *
 * C99 and C++ compatible; no CUDA/HIP types.
 */
#ifndef GPT_TYPES_H
#define GPT_TYPES_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpt_float3 { float x, y, z; } gpt_float3;
typedef struct __attribute__((aligned(8))) gpt_float2 { float x, y; } gpt_float2;
typedef struct gpt_uchar4 { unsigned char x, y, z, w; } gpt_uchar4;

/* src/primitive.h:9-13 */
enum { GPT_GT_TRIANGLE = 0, GPT_GT_LINES = 1, GPT_GT_SPHERE = 2 };

/* src/material.h:10-17 */
enum {
    GPT_MT_LAMBERTIAN = 0,
    GPT_MT_MIRROR = 1,
    GPT_MT_DIELECTRIC = 2,
    GPT_MT_ROUGHDIELECTRIC = 3,
    GPT_MT_ROUGHCONDUCTOR = 4,
    GPT_MT_SUBSTRATE = 5
};

/* src/scene.h:15-24 */
enum {
    GPT_IT_AO = 0, GPT_IT_PT = 1, GPT_IT_VPT = 2, GPT_IT_LT = 3,
    GPT_IT_BDPT = 4, GPT_IT_MLT = 5, GPT_IT_SPPM = 6, GPT_IT_IR = 7
};

typedef struct gpt_vertex {          /* src/mesh.h:13-18 */
    gpt_float3 v;                    /* @0  position  */
    gpt_float3 n;                    /* @12 normal    */
    gpt_float2 uv;                   /* @24 texcoord  */
    gpt_float3 t;                    /* @32 tangent   */
    float _pad;                      /* @44           */
} gpt_vertex;

typedef struct gpt_triangle {        /* src/mesh.h:20-26 */
    gpt_vertex v1, v2, v3;           /* @0 @48 @96 */
    int32_t matIdx;                  /* @144 */
    int32_t bssrdfIdx;               /* @148 */
    int32_t lightIdx;                /* @152 */
    int32_t mediumInside;            /* @156 */
    int32_t mediumOutside;           /* @160 */
    int32_t _pad;                    /* @164 */
} gpt_triangle;

typedef struct gpt_primitive {       /* src/primitive.h:15-23 */
    int32_t type;                    /* @0 GPT_GT_* */
    int32_t _pad;
    gpt_triangle triangle;           /* @8 (union with Line/Sphere in the reference) */
} gpt_primitive;

typedef struct gpt_bvh_node {        /* src/bvh.h:19-29 */
    gpt_float3 fmin;                 /* @0  */
    gpt_float3 fmax;                 /* @12 */
    int32_t second_child_offset;     /* @24 absolute index of the right child, -1 for a leaf */
    uint8_t is_leaf;                 /* @28 */
    uint8_t _pad[3];
    int32_t start;                   /* @32 first primitive (leaf) */
    int32_t end;                     /* @36 last primitive, inclusive */
} gpt_bvh_node;

typedef struct gpt_material {        /* src/material.h:19-27 */
    int32_t type;                    /* @0  GPT_MT_* */
    float alphaU, alphaV;            /* @4 @8 */
    float insideIOR, outsideIOR;     /* @12 @16 */
    gpt_float3 k;                    /* @20 */
    gpt_float3 eta;                  /* @32 */
    gpt_float3 diffuse;              /* @44 */
    gpt_float3 specular;             /* @56 */
    int32_t textureIdx;              /* @68 (-1: use diffuse) */
} gpt_material;

typedef struct gpt_area {            /* src/area.h:7-11 */
    gpt_float3 radiance;             /* @0 */
    int32_t _pad0;                   /* @12 */
    gpt_triangle triangle;           /* @16 */
    int32_t medium;                  /* @184 */
    int32_t _pad1;                   /* @188 */
} gpt_area;

typedef struct gpt_infinite {        /* src/infinite.h:6-13 */
    const gpt_float3 *data;          /* @0  lat-long radiance map, width*height */
    int32_t width, height;           /* @8 @12 */
    gpt_float3 center;               /* @16 */
    float radius;                    /* @28 */
    gpt_float3 u, v, w;              /* @32 @44 @56 */
    uint8_t isvalid;                 /* @68 */
    uint8_t _pad[3];
} gpt_infinite;

typedef struct gpt_camera {          /* src/camera.h:8-26 */
    gpt_float3 position;             /* @0  */
    gpt_float3 u, v, w;              /* @12 @24 @36 */
    gpt_float2 resolution;           /* @48 */
    float distance;                  /* @56 */
    float fov;                       /* @60 */
    float apertureRadius;            /* @64 */
    float focalDistance;             /* @68 */
    uint8_t filmic;                  /* @72 */
    uint8_t environment;             /* @73 */
    uint8_t _pad[2];
    int32_t medium;                  /* @76 */
    /* private members of the reference class, filled by its constructor
     * (src/camera.h:31-46); gpt_camera_init() reproduces that arithmetic */
    float width, height;             /* @80 @84 */
    gpt_float2 pixel2screen;         /* @88 */
    float ratio;                     /* @96 */
    float area;                      /* @100 */
} gpt_camera;

typedef struct gpt_medium {          /* src/medium.h:9-13,53-62,186-193 */
    int32_t type;                    /* @0  GPT_MEDIUM_HOMOGENEOUS / GPT_MEDIUM_HETEROGENEOUS */
    float g;                         /* @4  Henyey-Greenstein asymmetry (0 = isotropic) */
    union {                          /* @8 */
        struct { gpt_float3 sigmaA, sigmaS, sigmaT; } homogeneous;
        struct {
            gpt_float3 sigmaA, sigmaS, sigmaT;
            int32_t nx, ny, nz;
            const float *density;
            float invMaxDensity;
            gpt_float3 p0, p1;
            int32_t iterMax;
            int32_t evalTransmittanceType;
        } heterogeneous;             /* density: nx*ny*nz floats, x fastest; 1 <= iterMax <= 2^20 */
    };
} gpt_medium;
#define GPT_MEDIUM_HOMOGENEOUS   0
#define GPT_MEDIUM_HETEROGENEOUS 1

typedef struct gpt_texture {         /* src/texture.h:9-28 (vector<uchar4> + size) */
    const gpt_uchar4 *data;          /* width*height texels, row 0 = bottom */
    int32_t width, height;
} gpt_texture;

/* What BeginRender() reads out of `Scene` (src/pathtracer.cu:2578-2676). */
typedef struct gpt_scene_desc {
    const gpt_primitive *prims;      /* scene.bvh.prims (after BVH reorder) */
    int32_t n_prims;
    const gpt_bvh_node *nodes;       /* scene.bvh.linear_root */
    int32_t n_nodes;
    const gpt_material *materials;   /* scene.materials */
    int32_t n_materials;
    const gpt_area *lights;          /* scene.lights */
    int32_t n_lights;
    const float *light_distribution; /* scene.lightDistribution (normalised CDF) */
    int32_t n_light_distribution;
    const gpt_infinite *infinite;    /* &scene.infinite (may be NULL = invalid) */
    const gpt_texture *textures;     /* scene.textures */
    int32_t n_textures;
    int32_t integrator_type;         /* scene.integrator.type: GPT_IT_PT, GPT_IT_VPT or GPT_IT_AO */
    union {                          /* the reference's anonymous union (src/scene.h:38-46) */
        int32_t max_depth;           /* scene.integrator.maxDepth (pt, vpt) */
        float max_dist;              /* scene.integrator.maxDist  (ao) */
    };
    const gpt_medium *mediums;       /* scene.mediums (vpt; triangles and the camera refer to them by index) */
    int32_t n_mediums;
} gpt_scene_desc;

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus)
#define GPT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define GPT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
GPT_STATIC_ASSERT(sizeof(gpt_vertex) == 48, "Vertex layout");
GPT_STATIC_ASSERT(offsetof(gpt_vertex, uv) == 24 && offsetof(gpt_vertex, t) == 32, "Vertex layout");
GPT_STATIC_ASSERT(sizeof(gpt_triangle) == 168 && offsetof(gpt_triangle, matIdx) == 144, "Triangle layout");
GPT_STATIC_ASSERT(offsetof(gpt_triangle, lightIdx) == 152, "Triangle layout");
GPT_STATIC_ASSERT(sizeof(gpt_primitive) == 176 && offsetof(gpt_primitive, triangle) == 8, "Primitive layout");
GPT_STATIC_ASSERT(sizeof(gpt_bvh_node) == 40 && offsetof(gpt_bvh_node, is_leaf) == 28, "LinearBVHNode layout");
GPT_STATIC_ASSERT(offsetof(gpt_bvh_node, start) == 32 && offsetof(gpt_bvh_node, end) == 36, "LinearBVHNode layout");
GPT_STATIC_ASSERT(sizeof(gpt_material) == 72 && offsetof(gpt_material, k) == 20, "Material layout");
GPT_STATIC_ASSERT(offsetof(gpt_material, specular) == 56 && offsetof(gpt_material, textureIdx) == 68, "Material layout");
GPT_STATIC_ASSERT(sizeof(gpt_area) == 192 && offsetof(gpt_area, triangle) == 16, "Area layout");
GPT_STATIC_ASSERT(offsetof(gpt_area, medium) == 184, "Area layout");
GPT_STATIC_ASSERT(sizeof(gpt_infinite) == 72 && offsetof(gpt_infinite, center) == 16, "Infinite layout");
GPT_STATIC_ASSERT(offsetof(gpt_infinite, u) == 32 && offsetof(gpt_infinite, isvalid) == 68, "Infinite layout");
GPT_STATIC_ASSERT(sizeof(gpt_medium) == 104 && offsetof(gpt_medium, heterogeneous.density) == 56, "Medium layout");
GPT_STATIC_ASSERT(sizeof(gpt_camera) == 104 && offsetof(gpt_camera, resolution) == 48, "Camera layout");
GPT_STATIC_ASSERT(offsetof(gpt_camera, filmic) == 72 && offsetof(gpt_camera, medium) == 76, "Camera layout");
GPT_STATIC_ASSERT(offsetof(gpt_camera, pixel2screen) == 88 && offsetof(gpt_camera, area) == 100, "Camera layout");

#endif /* GPT_TYPES_H */

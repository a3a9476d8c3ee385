// pt_layout.h — how the scene sits in HBM for the gfx950 kernel.
//
// The reference uploads its host structs verbatim (AoS: 40-B LinearBVHNode,
// 176-B Primitive fetched by value per triangle test, reference
// src/pathtracer.cu:222,231).  Here the same information is re-laid-out once in
// gpt_begin() so that every per-lane gather is a small number of aligned
// 16-byte loads:
//
//   DevNode   32 B  two dwordx4: AABB + threaded-traversal links
//   DevTri    48 B  three dwordx4: v1, e1 = v2-v1, e2 = v3-v1 (all the
//                   Moeller-Trumbore test reads; 36 B used)
//   DevShade  80 B  five dwordx4: what is needed once per ACCEPTED FINAL hit
//                   (normals, uvs, normalised dp/dv, material and light index)
//   DevLight  96 B  six dwordx4: one emissive triangle for sampling / Le / pdf
//
// Everything precomputed here (e1, e2, normalize(dpdv), triangle area) is a
// pure function of one triangle evaluated with the same IEEE operations, in the
// same order, as the per-hit code of the reference (src/mesh.h:45-98,39-43), so
// the values are bit-identical to computing them per hit.
#pragma once

#include <stdint.h>
#include "../../include/gpt_types.h"
#include "pt_vec.h"
#include "../../include/gpt_wide_bvh.h"

namespace pt {

struct alignas(16) DevNode {
    float bmin[3];
    float bmax[3];
    // Threaded preorder traversal (fixed left-first order == the reference's
    // stack discipline, src/pathtracer.cu:221-252):
    //   inner: link = where to continue when the box is missed ("escape":
    //          first node after this subtree in preorder); last = -1
    //   leaf : link = first primitive; last = last primitive (inclusive, >= 0)
    // A hit inner node continues at the next node; a leaf always does.
    // Both are stored as BYTE offsets (node index * 32, primitive index * 48)
    // so the traversal loop does no address arithmetic.
    int32_t link;
    int32_t last;
};
static_assert(sizeof(DevNode) == 32, "DevNode");

struct alignas(16) DevTri {
    float v1[3]; float e1x;
    float e1yz[2]; float e2xy[2];
    float e2z; float pad[3];
};
static_assert(sizeof(DevTri) == 48, "DevTri");

// A wide node (include/gpt_wide_bvh.h) as ONE lane reads it: the four children's boxes as a structure of arrays (six dwordx4),
// then what the walk pushes for each child, ready made:
//   another wide node: its BYTE offset (index * 128; bit 31 clear);  a leaf: bit 31 | (count - 1) << 27 | first triangle;
//   empty: 0xffffffff (GPT_WIDE_NONE; never hit)
struct alignas(16) DevWideNode {
    float lo_x[4], lo_y[4], lo_z[4];
    float hi_x[4], hi_y[4], hi_z[4];
    uint32_t entry[4];
    uint32_t pad[4];
};
static_assert(sizeof(DevWideNode) == 128, "DevWideNode");
// A wave's slice of DevParams.wide_stack, in dwords: first one 8-dword suspend record per lane (a ray that is still being walked when
// a drain stops; read and written once per drain: the hot 2 KB), then the overflow levels of its 64 per-lane stacks (level-major:
// level l of lane i at 512 + 64 l + i)
constexpr int kWideSpillLevels = GPT_WIDE_STACK_MAX + 8;
constexpr int kWideWaveSliceDwords = 64 * kWideSpillLevels + 64 * 8;

struct alignas(16) DevShade {
    float n1[3], n2[3], n3[3];   // vertex normals
    float uv1[2], uv2[2], uv3[2];
    float ndpdv[3];              // normalize(dpdv) of src/mesh.h:71-83,91
    int32_t matIdx;
    int32_t lightIdx;
};
static_assert(sizeof(DevShade) == 80, "DevShade");

struct alignas(16) DevLight {
    float radiance[3];
    float area;                  // Triangle::GetSurfaceArea, src/mesh.h:39-43
    float v1[3], v2[3], v3[3];
    float n1[3], n2[3], n3[3];
    float pad[2];
};
static_assert(sizeof(DevLight) == 96, "DevLight");

struct DevTexture {
    const gpt_uchar4 *data;
    int32_t width, height;
};

struct DevInfinite {
    const float *data;           // float3[w*h]
    int32_t width, height;
    float radius;
    float u[3], v[3], w[3];
    int32_t isvalid;
};

// kernel arguments (by value in the kernarg segment)
struct alignas(16) DevMedium {       // a medium (src/medium.h:9-62,186-233)
    float sigmaS[3];
    float g;
    float sigmaT[3];
    int32_t type;                    // GPT_MEDIUM_HOMOGENEOUS / GPT_MEDIUM_HETEROGENEOUS; the rest is the density grid
    const float *density;            // nx*ny*nz floats in HBM, x fastest
    int32_t nx, ny;
    int32_t nz, iterMax, trType;     // trType: 0 delta, 1 ratio, 2 residual ratio tracking (medium.h:60)
    float invMaxDensity;
    float p0[3], _pad0;
    float p1[3], _pad1;
};
static_assert(sizeof(DevMedium) == 96, "DevMedium");

struct DevParams {
    const DevNode *nodes;
    const DevTri *tris;
    const DevShade *shade;
    const gpt_material *materials;
    const DevLight *lights;
    const float *light_cdf;
    const DevTexture *textures;
    DevInfinite inf;
    int32_t n_nodes;
    int32_t n_prims;
    int32_t n_materials;
    int32_t n_lights;
    int32_t n_cdf;
    int32_t max_depth;
    int32_t integrator;         // GPT_IT_PT, GPT_IT_AO or GPT_IT_VPT
    float ao_max_dist;          // scene.integrator.maxDist (ao)
    float eps;
    gpt_camera cam;
    // film
    float *acc;                  // kernel_acc_image (read/written by the output kernel)
    float *color;                // kernel_color
    float *out;                  // tonemapped output or nullptr
    float *samples;              // per-iteration sample planes of float4: plane index = iter - iter_first
    uint64_t plane;              // float4 slots per plane: 64 per tile this rank owns (tile-major: local tile * 64 + pixel in tile)
    uint32_t stride;             // 32*(W/32): row stride of the reference's pixel index
    uint32_t rows;               // 4*(H/4)
    uint32_t tiles_x;            // 8x8 tiles per row
    uint32_t n_tiles;            // all tiles
    uint32_t rank, n_ranks;      // tile ownership: t % n_ranks == rank
    uint32_t iter_first, iter_count;
    int32_t reset;
    // scheduler: independent work items w = (chunk, tile), chunk = w / n_owned
    uint32_t chunk_iters;        // iterations per work item
    uint32_t n_chunks;           // ceil(iter_count / chunk_iters)
    uint32_t *tile_counter;      // work queue heads, one per XCD (8 words, zeroed before every launch)
    unsigned long long *counters;  // work counters (counting build only)
    int32_t traversal;           // GPT_TRAVERSAL_REFERENCE / GPT_TRAVERSAL_WIDE4
    // Volpath only (new fields go at the END: the kernarg layout steers the register allocation of the headline kernel)
    const struct DevMedium *mediums;
    const int32_t *prim_media;         // per primitive (BVH order): mediumInside, mediumOutside
    int32_t vpt_walk;                  // Volpath: density grids or material-less surfaces -> the one-ray-at-a-time kernel
    // GPT_TRAVERSAL_WIDE4 only (include/gpt_wide_bvh.h)
    const struct DevWideNode *wide;    // the 4-wide tree; node w at byte offset 128 * w
    uint32_t *wide_stack;              // overflow of the per-ray LDS stacks and suspend records: kWideWaveSliceDwords per wave of the widest grid
    uint32_t wide_stack_blocks;        // workgroups that buffer was sized for (no wide kernel is launched with more)
    uint32_t wide_tris_off;            // byte offset from `wide` of the copy of `tris` that the hand-scheduled wide loop fetches from
};

}  // namespace pt

// render_api.cpp — host half of the render API: the C ABI of include/gpt.h
// (gpt_begin / gpt_render / gpt_end ...), replacing the host side of the
// reference's BeginRender / Render / EndRender (reference
// src/pathtracer.cu:2568-2750).
//
// Differences from the reference, by design:
//   * state lives in a gpt_ctx, not in file-scope globals (re-entrant, one
//     context per GPU / per process rank);
//   * the scene is re-laid-out for the GPU once (pt_layout.h) instead of
//     uploading host structs verbatim;
//   * one launch per BATCH of iterations (the reference launches Path + Output
//     and does a synchronous 104-byte camera cudaMemcpy per spp); the camera
//     travels in the kernel arguments;
//   * errors come back as codes + gpt_last_error(), never __debugbreak().
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>           // types and prototypes only: the library is opened at run time (gpt_comm_init)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/gpt.h"
#include "host_util.h"
#include "pt_layout.h"
#include "../../include/gpt_traversal.h"
#include "../../include/gpt_wide_bvh.h"

namespace pt {
hipError_t launch_render(const DevParams &P, bool count, int n_blocks, bool lds_scene, bool force_walk, hipStream_t stream);
bool render_scene_fits_lds(const DevParams &P);
hipError_t launch_output(const DevParams &P, hipStream_t stream);
hipError_t launch_trace_rays(const DevParams &P, bool lds_scene, const float4 *rays, int n, float4 *out, hipStream_t stream);
hipError_t launch_tonemap(const float *acc, float *out, uint32_t stride, uint32_t rows, uint32_t iter, int filmic,
                          hipStream_t stream);
hipError_t launch_debug_math(int fn, const float *x, const float *y, float *out, int n, hipStream_t stream);
hipError_t launch_debug_rng(uint32_t pixel, uint32_t iter, uint32_t *seed_out, float *u_out, int n, hipStream_t stream);
hipError_t launch_debug_bsdf(const gpt_material *material, const DevTexture *texture, const float *geom11, const float *in3, int n, int mode,
                             float *out7, hipStream_t stream);
int render_kernel_blocks_per_cu(bool count, bool walk, bool wide);
bool render_uses_walk_kernel(const DevParams &P, bool force);
}  // namespace pt

using namespace pt;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            gpt_set_error("%s in %s at line %d", hipGetErrorString(e_), __FILE__, __LINE__);   \
            return GPT_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

struct gpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t width = 0, height = 0;
    std::vector<void *> allocs;       // everything to hipFree in gpt_end
    DevParams P{};                    // scene pointers + constants; per-call fields filled in gpt_render
    float *acc = nullptr, *color = nullptr;
    uint32_t *tile_counter = nullptr;
    unsigned long long *counters = nullptr;
    // gpt_set_option (explicit, readable back with gpt_get_option; nothing here is read from the environment)
    uint32_t chunk_override = 0;          // "chunk_iters": iterations per work item, 0 = the cost model of gpt_render
    bool lds_scene = true;                // "lds_scene": stage a small scene in LDS (0: always traverse from global memory)
    bool force_walk = false;              // "vpt_walk_kernel": Volpath always on the one-ray-at-a-time kernel
    bool media_ok = true;                 // every medium record and medium index is valid ("vpt" can run)
    bool has_interface = false;           // some primitive has no material (matIdx -1): only "vpt" renders such scenes
    int n_mediums = 0;
    float *samples = nullptr;             // per-iteration sample planes, grown on demand
    size_t sample_bytes = 0;              // bytes allocated for them
    uint32_t last_batch_cap = 0;          // iterations per launch the last gpt_render used
    uint32_t batch_cap_limit = 0;         // a larger plane allocation failed (0: none did): not retried until the tile ownership changes
    uint32_t batch_limit_owned = 0;       // ... the owned-tile count that limit was found for
    double last_trace_ms = 0.0;           // kernel time of the last gpt_debug_trace (HIP events)
    uint32_t max_batch = 256;             // "max_batch": iterations per path-kernel launch; also capped by the memory budget
    bool max_batch_set = false;           // ... as set by the caller (else: 256 x the number of ranks sharing the frame)
    bool count_next = false;
    int n_cus = 256;
    int blocks_per_cu[2][2] = {{4, 4}, {3, 3}};      // [walk kernel][counting build]
    int blocks_per_cu_wide[2][2] = {{4, 4}, {3, 3}}; // ... of the GPT_TRAVERSAL_WIDE4 kernels
    // timing of the path kernel on its own stream
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> output_events;   // {end of the path kernel (owned by `events`), mark behind the accumulation kernel}
    std::vector<hipEvent_t> free_marks;
    uint32_t timed_launches = 0;
    double timed_ms = 0.0;
    double timed_output_ms = 0.0;        // ... and of the accumulation kernel that follows every launch ("output_kernel_us")
    // multi-GPU film reduce (gpt_comm_init / gpt_reduce_film)
    bool wide_ok = false;                 // the 4-wide tree exists (GPT_TRAVERSAL_WIDE4 can be selected)
    bool wide_fallback = false;           // gpt_begin wanted the wide order and had no room for its tree ("wide_fallback")
    int wide_depth = 0, n_wide = 0;
    std::vector<pt::DevWideNode> wide_host;   // built with the scene, uploaded by the first gpt_set_traversal_order(GPT_TRAVERSAL_WIDE4)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    float *reduced = nullptr;             // root: the whole frame after gpt_reduce_film (W*H*3); acc stays this rank's tiles
    // HIP events around the last gpt_reduce_film and the last gpt_tonemap[_from] on the stream ("last_reduce_us", "last_tonemap_us"):
    // where a multi-GPU job's time outside the path kernel goes (bench.py: config.per_rank)
    hipEvent_t ev_reduce[2] = {nullptr, nullptr}, ev_tonemap[2] = {nullptr, nullptr};
};

namespace {
std::atomic<bool> g_fail_next_wide_alloc{false};   // gpt_debug_fail_next_wide_alloc: tests of gpt_begin's low-memory path
int span_begin(gpt_ctx *ctx, hipEvent_t ev[2])
{
    if (!ev[0]) {
        HIP_TRY(hipEventCreate(&ev[0]));
        HIP_TRY(hipEventCreate(&ev[1]));
    }
    HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    return GPT_OK;
}
int64_t span_us(hipEvent_t ev[2])       // -1: nothing recorded
{
    float ms = 0.f;
    if (!ev[0] || hipEventSynchronize(ev[1]) != hipSuccess || hipEventElapsedTime(&ms, ev[0], ev[1]) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return (int64_t)(ms * 1000.0);
}
}  // namespace

namespace {

template <class T>
int dev_upload(gpt_ctx *ctx, const T *host, size_t n, const T **out)
{
    *out = nullptr;
    if (n == 0) return GPT_OK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(T)));
    ctx->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(p);
    return GPT_OK;
}

int fold_events(gpt_ctx *ctx)
{
    for (auto &ev : ctx->events) {
        HIP_TRY(hipEventSynchronize(ev.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        ctx->timed_ms += ms;
        ctx->timed_launches++;
        ctx->free_events.push_back(ev);
    }
    ctx->events.clear();
    // the accumulation kernel of launch i runs between the end of path kernel i and the mark recorded behind it
    for (auto &ev : ctx->output_events) {
        HIP_TRY(hipEventSynchronize(ev.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        ctx->timed_output_ms += ms;
        ctx->free_marks.push_back(ev.second);
    }
    ctx->output_events.clear();
    return GPT_OK;
}

// ---- scene re-layout (see pt_layout.h) ----------------------------------------
V3 f3(const gpt_float3 &a) { return V3{a.x, a.y, a.z}; }
void st3(float *d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

// wrap.h:6-16
void make_coordinate(V3 n, V3 &u, V3 &w)
{
    if (std::fabs(n.x) > std::fabs(n.y)) {
        float invLen = 1.0f / sqrt_rn(n.x * n.x + n.z * n.z);
        w = v3(n.z * invLen, 0.0f, -n.x * invLen);
    } else {
        float invLen = 1.0f / sqrt_rn(n.y * n.y + n.z * n.z);
        w = v3(0.0f, n.z * invLen, -n.y * invLen);
    }
    u = cross(w, n);
}

void pack_triangle(const gpt_triangle &t, DevTri &dt, DevShade &ds)
{
    const V3 v1 = f3(t.v1.v), v2 = f3(t.v2.v), v3_ = f3(t.v3.v);
    const V3 e1 = v2 - v1;          // mesh.h:46-47
    const V3 e2 = v3_ - v1;
    std::memset(&dt, 0, sizeof(dt));
    st3(dt.v1, v1);
    dt.e1x = e1.x; dt.e1yz[0] = e1.y; dt.e1yz[1] = e1.z;
    dt.e2xy[0] = e2.x; dt.e2xy[1] = e2.y; dt.e2z = e2.z;

    // mesh.h:69-83: dp/dv depends only on the triangle
    V3 dpdu, dpdv;
    const V2 duv1 = V2{t.v2.uv.x - t.v1.uv.x, t.v2.uv.y - t.v1.uv.y};
    const V2 duv2 = V2{t.v3.uv.x - t.v1.uv.x, t.v3.uv.y - t.v1.uv.y};
    const float det = duv1.x * duv2.y - duv1.y * duv2.x;
    if ((double)std::fabs(det) < 1e-8) {
        V3 nn = normalize(cross(e1, e2));
        make_coordinate(nn, dpdu, dpdv);
    } else {
        float invDet = 1 / det;
        dpdu = (duv2.y * e1 - duv1.y * e2) * invDet;
        dpdv = (-duv2.x * e1 + duv1.x * e2) * invDet;
    }
    (void)dpdu;
    std::memset(&ds, 0, sizeof(ds));
    st3(ds.n1, f3(t.v1.n)); st3(ds.n2, f3(t.v2.n)); st3(ds.n3, f3(t.v3.n));
    ds.uv1[0] = t.v1.uv.x; ds.uv1[1] = t.v1.uv.y;
    ds.uv2[0] = t.v2.uv.x; ds.uv2[1] = t.v2.uv.y;
    ds.uv3[0] = t.v3.uv.x; ds.uv3[1] = t.v3.uv.y;
    st3(ds.ndpdv, normalize(dpdv));  // mesh.h:91 normalize(dpdv)
    ds.matIdx = t.matIdx;
    ds.lightIdx = t.lightIdx;
}

// Threaded traversal: the nodes stay in the reference's preorder (left child first, pathtracer.cu:221-252); an inner node continues at
// the next node when its box is hit and at its "escape" - the first node after its subtree - when it is missed.  That reproduces the
// reference's push-right / push-left stack order with no stack at all.
void thread_nodes(const gpt_bvh_node *nodes, int n, std::vector<DevNode> &out)
{
    out.resize((size_t)n);
    if (n <= 0) return;
    std::vector<int> subtree_end((size_t)n, 0);      // index one past the node's subtree
    // in preorder a subtree is a contiguous range: leaf [i, i + 1); inner node [i, end of its right child's subtree)
    for (int i = n - 1; i >= 0; --i) {
        const gpt_bvh_node &nd = nodes[i];
        subtree_end[(size_t)i] = (nd.is_leaf || nd.second_child_offset <= 0) ? i + 1 : subtree_end[(size_t)nd.second_child_offset];
    }
    for (int i = 0; i < n; ++i) {
        const gpt_bvh_node &nd = nodes[i];
        DevNode &d = out[(size_t)i];
        d.bmin[0] = nd.fmin.x; d.bmin[1] = nd.fmin.y; d.bmin[2] = nd.fmin.z;
        d.bmax[0] = nd.fmax.x; d.bmax[1] = nd.fmax.y; d.bmax[2] = nd.fmax.z;
        if (nd.is_leaf) {
            d.link = nd.start * (int32_t)sizeof(DevTri);
            d.last = nd.end * (int32_t)sizeof(DevTri);
        } else {
            d.link = subtree_end[(size_t)i] * (int32_t)sizeof(DevNode);
            d.last = -1;
        }
    }
}

}  // namespace

extern "C" {

const char *gpt_version(void) { return "gpu_pathtracer_amd 0.1 (gfx950)"; }

int gpt_begin(const gpt_scene_desc *scene, uint32_t width, uint32_t height, float epsilon, int device, gpt_ctx **out)
{
    if (!scene || !out || width == 0 || height == 0) {
        gpt_set_error("gpt_begin: null scene/out or empty frame");
        return GPT_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (scene->integrator_type != GPT_IT_PT && scene->integrator_type != GPT_IT_AO && scene->integrator_type != GPT_IT_VPT) {
        gpt_set_error("gpt_begin: integrator type %d is not supported (\"pt\", \"vpt\" and \"ao\" are)", scene->integrator_type);
        return GPT_ERR_UNSUPPORTED;
    }
    if (scene->n_prims < 0 || scene->n_nodes < 0 || scene->n_materials <= 0 || scene->n_light_distribution < 1) {
        gpt_set_error("gpt_begin: inconsistent scene counts");
        return GPT_ERR_INVALID_ARG;
    }
    // traversal cursors are 32-bit byte offsets into the packed node / triangle arrays (pt_layout.h)
    if ((int64_t)scene->n_nodes * (int64_t)sizeof(DevNode) > INT32_MAX || (int64_t)scene->n_prims * (int64_t)sizeof(DevTri) > INT32_MAX) {
        gpt_set_error("gpt_begin: scene too large (%d nodes, %d primitives; limits %d / %d)", scene->n_nodes, scene->n_prims,
                      (int)(INT32_MAX / (int64_t)sizeof(DevNode)), (int)(INT32_MAX / (int64_t)sizeof(DevTri)));
        return GPT_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < scene->n_prims; ++i) {
        if (scene->prims[i].type != GPT_GT_TRIANGLE) {
            gpt_set_error("gpt_begin: primitive %d has type %d; only triangles are supported", i, scene->prims[i].type);
            return GPT_ERR_UNSUPPORTED;
        }
        const int m = scene->prims[i].triangle.matIdx;
        if (m < -1 || m >= scene->n_materials) {      // -1: a material-less surface between two media (parsescene.cpp:357)
            gpt_set_error("gpt_begin: primitive %d has material index %d outside [-1,%d)", i, m, scene->n_materials);
            return GPT_ERR_INVALID_ARG;
        }
        const int l = scene->prims[i].triangle.lightIdx;
        if (l < -1 || l >= scene->n_lights) {
            gpt_set_error("gpt_begin: primitive %d has light index %d outside [-1,%d)", i, l, scene->n_lights);
            return GPT_ERR_INVALID_ARG;
        }
    }
    // the tree is walked by index on the host (threading, wide collapse) and by offset on the device: refuse one whose links
    // leave the arrays (the reference trusts its own builder; a caller-supplied tree gets checked once)
    for (int i = 0; i < scene->n_nodes; ++i) {
        const gpt_bvh_node &nd = scene->nodes[i];
        const bool ok = nd.is_leaf ? ((nd.start == -1 && nd.end == -1) || (nd.start >= 0 && nd.end >= nd.start && nd.end < scene->n_prims))
                                   : (i + 1 < scene->n_nodes && nd.second_child_offset > i + 1 && nd.second_child_offset < scene->n_nodes);
        if (!ok) {
            gpt_set_error("gpt_begin: BVH node %d is inconsistent (leaf %d, second child %d, primitives %d..%d; %d nodes, %d primitives)", i,
                          (int)nd.is_leaf, nd.second_child_offset, nd.start, nd.end, scene->n_nodes, scene->n_prims);
            return GPT_ERR_INVALID_ARG;
        }
    }
    // ---- media and material-less surfaces, checked before any device work (same answer with and without a GPU)
    bool has_interface = false, media_ok = scene->n_mediums == 0 || scene->mediums != nullptr;
    for (int i = 0; i < scene->n_prims; ++i) {
        const gpt_triangle &t = scene->prims[i].triangle;
        if (t.matIdx == -1) has_interface = true;
        if (t.mediumInside < -1 || t.mediumInside >= scene->n_mediums || t.mediumOutside < -1 || t.mediumOutside >= scene->n_mediums)
            media_ok = false;
    }
    for (int i = 0; i < scene->n_mediums && media_ok; ++i) {
        const gpt_medium &m = scene->mediums[i];
        if (m.type == GPT_MEDIUM_HOMOGENEOUS) continue;
        const auto &h = m.heterogeneous;
        // the tracking loops end after iterMax steps at the latest (medium.h:70,138): it has to be positive
        if (m.type != GPT_MEDIUM_HETEROGENEOUS || h.nx <= 0 || h.ny <= 0 || h.nz <= 0 || (int64_t)h.nx * h.ny * h.nz > ((int64_t)1 << 30) ||
            !h.density || h.iterMax < 1 || h.iterMax > (1 << 20) || h.evalTransmittanceType < 0 || h.evalTransmittanceType > 2)
            media_ok = false;
    }
    if (scene->integrator_type == GPT_IT_VPT && !media_ok) {
        gpt_set_error("gpt_begin: \"vpt\": a medium record or a medium index of the scene is invalid (%d media)", scene->n_mediums);
        return GPT_ERR_INVALID_ARG;
    }
    if (scene->integrator_type != GPT_IT_VPT && has_interface) {
        gpt_set_error("gpt_begin: the scene has surfaces without a material (media interfaces); only \"vpt\" renders them");
        return GPT_ERR_UNSUPPORTED;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        gpt_set_error("gpt_begin: no HIP device visible (this library has no CPU fallback)");
        return GPT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n_dev) {
        gpt_set_error("gpt_begin: device %d out of range (%d visible)", device, n_dev);
        return GPT_ERR_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(device));

    gpt_ctx *ctx = new gpt_ctx();
    ctx->device = device;
    ctx->width = width;
    ctx->height = height;
    int rc = GPT_OK;
    auto fail = [&](int code) {
        gpt_end(ctx);
        return code;
    };
    // A BLOCKING stream: it is ordered with the legacy default stream like the reference's kernels, which are launched on
    // the default stream itself (pathtracer.cu:2707-2750).  A caller in the reference's shape - Render(), then a copy or
    // a GL unmap of `output` on the default stream, src/main.cpp:139-143 - therefore sees the finished frame without
    // any extra call; gpt_synchronize() is only needed before touching `output` from ANOTHER non-blocking stream.
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamDefault) != hipSuccess) {
        gpt_set_error("gpt_begin: hipStreamCreate failed");
        return fail(GPT_ERR_HIP);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
    for (int w = 0; w < 2; ++w)
        for (int c = 0; c < 2; ++c) {
            ctx->blocks_per_cu[w][c] = render_kernel_blocks_per_cu(c != 0, w != 0, false);
            ctx->blocks_per_cu_wide[w][c] = render_kernel_blocks_per_cu(c != 0, w != 0, true);
        }

    // ---- geometry
    std::vector<DevTri> tris((size_t)scene->n_prims + 1);      // + padding: a triangle trip may also read the record after the one it tests
    std::vector<DevShade> shade((size_t)scene->n_prims);
    for (int i = 0; i < scene->n_prims; ++i) pack_triangle(scene->prims[i].triangle, tris[(size_t)i], shade[(size_t)i]);
    std::vector<DevNode> nodes;
    thread_nodes(scene->nodes, scene->n_nodes, nodes);
    nodes.push_back(DevNode{});       // padding: a node trip may also read the 32 bytes after the node it visits
    // the 4-wide tree of GPT_TRAVERSAL_WIDE4 (include/gpt_wide_bvh.h), derived from the reference's tree
    std::vector<DevWideNode> wide_dev;
    if (scene->n_nodes > 0 && scene->n_prims < (1 << 27)) {
        const int32_t cap = gpt_wide_capacity(scene->n_nodes, scene->n_prims);
        std::vector<gpt_wide_node> wide((size_t)cap);
        int32_t depth = 0;
        const int32_t n_wide = gpt_wide_build(scene->nodes, scene->n_nodes, scene->prims, wide.data(), cap, &depth);
        if (n_wide > 0 && 3 * depth + 1 <= GPT_WIDE_STACK_MAX && (int64_t)n_wide * (int64_t)sizeof(DevWideNode) < INT32_MAX) {
            wide_dev.resize((size_t)n_wide);
            for (int32_t w = 0; w < n_wide; ++w) {
                DevWideNode &d = wide_dev[(size_t)w];
                std::memset(&d, 0, sizeof(d));
                for (int k = 0; k < 4; ++k) {
                    const gpt_wide_child &c = wide[(size_t)w].c[k];
                    d.lo_x[k] = c.bmin[0]; d.lo_y[k] = c.bmin[1]; d.lo_z[k] = c.bmin[2];
                    d.hi_x[k] = c.bmax[0]; d.hi_y[k] = c.bmax[1]; d.hi_z[k] = c.bmax[2];
                    d.entry[k] = c.count < 0 ? (uint32_t)c.ref * (uint32_t)sizeof(DevWideNode)
                               : c.count > 0 ? gpt_wide_leaf_entry(c.ref, c.count) : GPT_WIDE_NONE;
                }
            }
            ctx->wide_ok = true;
            ctx->wide_depth = depth;
            ctx->n_wide = n_wide;
        }
    }
    std::vector<DevLight> lights((size_t)scene->n_lights);
    for (int i = 0; i < scene->n_lights; ++i) {
        const gpt_area &a = scene->lights[i];
        DevLight &L = lights[(size_t)i];
        std::memset(&L, 0, sizeof(L));
        st3(L.radiance, f3(a.radiance));
        const V3 v1 = f3(a.triangle.v1.v), v2 = f3(a.triangle.v2.v), v3_ = f3(a.triangle.v3.v);
        L.area = length(cross(v2 - v1, v3_ - v1)) * 0.5f;       // mesh.h:39-43
        st3(L.v1, v1); st3(L.v2, v2); st3(L.v3, v3_);
        st3(L.n1, f3(a.triangle.v1.n)); st3(L.n2, f3(a.triangle.v2.n)); st3(L.n3, f3(a.triangle.v3.n));
    }

    DevParams &P = ctx->P;
    {
        // nodes and triangles share ONE allocation, triangles behind the nodes: the global-memory loop fetches for its node lanes and
        // its triangle lanes with the same instructions, a 32-bit offset per lane from the base of the nodes
        const size_t node_bytes = nodes.size() * sizeof(nodes[0]), tri_bytes = tris.size() * sizeof(DevTri);
        if (node_bytes + tri_bytes > (size_t)UINT32_MAX) {
            gpt_set_error("gpt_begin: node arrays and triangles exceed 4 GB");
            return fail(GPT_ERR_UNSUPPORTED);
        }
        void *p = nullptr;
        if (hipMalloc(&p, node_bytes + tri_bytes) != hipSuccess) { gpt_set_error("gpt_begin: hipMalloc(nodes + triangles) failed"); return fail(GPT_ERR_HIP); }
        ctx->allocs.push_back(p);
        if (hipMemcpy(p, nodes.data(), node_bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(static_cast<char *>(p) + node_bytes, tris.data(), tri_bytes, hipMemcpyHostToDevice) != hipSuccess) {
            gpt_set_error("gpt_begin: upload of nodes / triangles failed");
            return fail(GPT_ERR_HIP);
        }
        P.nodes = static_cast<const DevNode *>(p);
        P.tris = reinterpret_cast<const DevTri *>(static_cast<const char *>(p) + node_bytes);
    }
    ctx->wide_host.swap(wide_dev);          // (128 B per wide node: uploaded only if the wide walk is ever selected)
    if ((rc = dev_upload(ctx, shade.data(), shade.size(), &P.shade)) != GPT_OK) return fail(rc);
    if ((rc = dev_upload(ctx, scene->materials, (size_t)scene->n_materials, &P.materials)) != GPT_OK) return fail(rc);
    if ((rc = dev_upload(ctx, lights.data(), lights.size(), &P.lights)) != GPT_OK) return fail(rc);
    if ((rc = dev_upload(ctx, scene->light_distribution, (size_t)scene->n_light_distribution, &P.light_cdf)) != GPT_OK)
        return fail(rc);
    // ---- participating media (Volpath): uploaded whenever the scene has them, so that gpt_set_integrator can switch to
    // "vpt" later.  Density grids go to HBM as they are (x fastest).
    ctx->media_ok = media_ok;
    ctx->has_interface = has_interface;
    ctx->n_mediums = scene->n_mediums;
    bool walk = has_interface;
    if (media_ok) {
        std::vector<DevMedium> media((size_t)scene->n_mediums);
        for (int i = 0; i < scene->n_mediums; ++i) {
            const gpt_medium &m = scene->mediums[i];
            DevMedium &d = media[(size_t)i];
            std::memset(&d, 0, sizeof(d));
            d.sigmaS[0] = m.homogeneous.sigmaS.x; d.sigmaS[1] = m.homogeneous.sigmaS.y; d.sigmaS[2] = m.homogeneous.sigmaS.z;
            d.sigmaT[0] = m.homogeneous.sigmaT.x; d.sigmaT[1] = m.homogeneous.sigmaT.y; d.sigmaT[2] = m.homogeneous.sigmaT.z;
            d.g = m.g;
            d.type = m.type;
            if (m.type == GPT_MEDIUM_HETEROGENEOUS) {
                const auto &h = m.heterogeneous;
                walk = true;
                d.nx = h.nx; d.ny = h.ny; d.nz = h.nz;
                d.iterMax = h.iterMax;
                d.trType = h.evalTransmittanceType;
                d.invMaxDensity = h.invMaxDensity;
                d.p0[0] = h.p0.x; d.p0[1] = h.p0.y; d.p0[2] = h.p0.z;
                d.p1[0] = h.p1.x; d.p1[1] = h.p1.y; d.p1[2] = h.p1.z;
                if ((rc = dev_upload(ctx, h.density, (size_t)h.nx * h.ny * h.nz, &d.density)) != GPT_OK) return fail(rc);
            }
        }
        std::vector<int32_t> prim_media((size_t)scene->n_prims * 2);
        for (int i = 0; i < scene->n_prims; ++i) {
            prim_media[2 * (size_t)i] = scene->prims[i].triangle.mediumInside;
            prim_media[2 * (size_t)i + 1] = scene->prims[i].triangle.mediumOutside;
        }
        if ((rc = dev_upload(ctx, media.data(), media.size(), &P.mediums)) != GPT_OK) return fail(rc);
        if ((rc = dev_upload(ctx, prim_media.data(), prim_media.size(), &P.prim_media)) != GPT_OK) return fail(rc);
    }
    // Two Volpath kernels (pt_kernel.hip): with homogeneous media and no material-less surfaces every transmittance is
    // known from a segment length, and a bounce keeps its three rays in flight; otherwise (density grids draw random
    // numbers per segment, interfaces split shadow rays into segments) the path is a one-ray-at-a-time state machine.
    P.vpt_walk = walk ? 1 : 0;

    // ---- textures
    std::vector<DevTexture> texs((size_t)scene->n_textures);
    for (int i = 0; i < scene->n_textures; ++i) {
        const gpt_texture &t = scene->textures[i];
        if (t.width <= 0 || t.height <= 0 || !t.data) {
            gpt_set_error("gpt_begin: texture %d is empty", i);
            return fail(GPT_ERR_INVALID_ARG);
        }
        if ((rc = dev_upload(ctx, t.data, (size_t)t.width * (size_t)t.height, &texs[(size_t)i].data)) != GPT_OK) return fail(rc);
        texs[(size_t)i].width = t.width;
        texs[(size_t)i].height = t.height;
    }
    if ((rc = dev_upload(ctx, texs.data(), texs.size(), &P.textures)) != GPT_OK) return fail(rc);
    for (int i = 0; i < scene->n_materials; ++i) {
        const int ti = scene->materials[i].textureIdx;
        if (ti < -1 || ti >= scene->n_textures) {
            gpt_set_error("gpt_begin: material %d has texture index %d outside [-1,%d)", i, ti, scene->n_textures);
            return fail(GPT_ERR_INVALID_ARG);
        }
    }

    // ---- environment light
    std::memset(&P.inf, 0, sizeof(P.inf));
    if (scene->infinite && scene->infinite->isvalid) {
        const gpt_infinite &I = *scene->infinite;
        if (I.width <= 0 || I.height <= 0 || !I.data) {
            gpt_set_error("gpt_begin: infinite light has no data");
            return fail(GPT_ERR_INVALID_ARG);
        }
        const float *envd = nullptr;
        if ((rc = dev_upload(ctx, reinterpret_cast<const float *>(I.data), (size_t)I.width * (size_t)I.height * 3, &envd)) != GPT_OK)
            return fail(rc);
        P.inf.data = envd;
        P.inf.width = I.width;
        P.inf.height = I.height;
        P.inf.radius = I.radius;
        st3(P.inf.u, f3(I.u)); st3(P.inf.v, f3(I.v)); st3(P.inf.w, f3(I.w));
        P.inf.isvalid = 1;
    }
    const int expect_cdf = scene->n_lights + 1 + (P.inf.isvalid ? 1 : 0);
    if (scene->n_light_distribution != expect_cdf) {
        gpt_set_error("gpt_begin: light distribution has %d entries, expected %d", scene->n_light_distribution, expect_cdf);
        return fail(GPT_ERR_INVALID_ARG);
    }

    P.n_nodes = scene->n_nodes;
    P.n_prims = scene->n_prims;
    P.n_materials = scene->n_materials;
    P.n_lights = scene->n_lights;
    P.n_cdf = scene->n_light_distribution;
    P.integrator = scene->integrator_type;
    P.max_depth = scene->integrator_type != GPT_IT_AO ? scene->max_depth : 0;
    P.ao_max_dist = scene->integrator_type == GPT_IT_AO ? scene->max_dist : 0.f;
    P.eps = epsilon;

    // ---- film: reference launch geometry (pathtracer.cu:2707-2709, 881-883)
    P.stride = 32u * (width / 32u);
    P.rows = 4u * (height / 4u);
    P.tiles_x = (P.stride + 7u) / 8u;
    P.n_tiles = P.tiles_x * ((P.rows + 7u) / 8u);
    P.rank = 0;
    P.n_ranks = 1;
    const size_t film_bytes = (size_t)width * height * 3 * sizeof(float);
    void *p = nullptr;
    if (hipMalloc(&p, film_bytes) != hipSuccess) { gpt_set_error("gpt_begin: hipMalloc(acc) failed"); return fail(GPT_ERR_HIP); }
    ctx->allocs.push_back(p);
    ctx->acc = static_cast<float *>(p);
    if (hipMalloc(&p, film_bytes) != hipSuccess) { gpt_set_error("gpt_begin: hipMalloc(color) failed"); return fail(GPT_ERR_HIP); }
    ctx->allocs.push_back(p);
    ctx->color = static_cast<float *>(p);
    if (hipMalloc(&p, 256) != hipSuccess) { gpt_set_error("gpt_begin: hipMalloc(queue) failed"); return fail(GPT_ERR_HIP); }
    ctx->allocs.push_back(p);
    ctx->tile_counter = static_cast<uint32_t *>(p);
    ctx->counters = reinterpret_cast<unsigned long long *>(static_cast<char *>(p) + 64);
    if (hipMemset(ctx->acc, 0, film_bytes) != hipSuccess || hipMemset(ctx->color, 0, film_bytes) != hipSuccess ||
        hipMemset(p, 0, 256) != hipSuccess) {
        gpt_set_error("gpt_begin: hipMemset failed");
        return fail(GPT_ERR_HIP);
    }
    P.acc = ctx->acc;
    P.color = ctx->color;
    P.tile_counter = ctx->tile_counter;
    P.plane = 0;                                   // set per gpt_render call: 64 slots per owned tile
    P.counters = ctx->counters;
    if (hipDeviceSynchronize() != hipSuccess) { gpt_set_error("gpt_begin: device sync failed"); return fail(GPT_ERR_HIP); }
    // the default traversal order (include/gpt_traversal.h): the 4-wide tree for every scene that does not fit LDS.  If there is no ROOM
    // for the wide tree (device memory exhausted, more than 4 GB: GPT_ERR_UNSUPPORTED) the scene still renders, in the reference's order,
    // with what is allocated; the option "wide_fallback" reads 1 and gpt_last_error() keeps the reason.  Any other failure - a device
    // fault, a failed copy - is an error of gpt_begin: going on would hide it behind a slower render.
    if ((rc = gpt_set_traversal_order(ctx, GPT_TRAVERSAL_AUTO)) != GPT_OK) {
        if (rc != GPT_ERR_UNSUPPORTED) return fail(rc);
        ctx->wide_ok = false;
        ctx->wide_fallback = true;
        std::vector<DevWideNode>().swap(ctx->wide_host);
        ctx->P.traversal = GPT_TRAVERSAL_REFERENCE;
    }
    *out = ctx;
    return GPT_OK;
}

int gpt_set_tile_owner(gpt_ctx *ctx, int rank, int n_ranks)
{
    if (!ctx || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        gpt_set_error("gpt_set_tile_owner: invalid rank %d of %d", rank, n_ranks);
        return GPT_ERR_INVALID_ARG;
    }
    ctx->P.rank = (uint32_t)rank;
    ctx->P.n_ranks = (uint32_t)n_ranks;
    return GPT_OK;
}

int gpt_set_integrator(gpt_ctx *ctx, int32_t integrator_type, int32_t max_depth, float max_dist)
{
    if (!ctx) { gpt_set_error("gpt_set_integrator: null context"); return GPT_ERR_INVALID_ARG; }
    if (integrator_type != GPT_IT_PT && integrator_type != GPT_IT_AO && integrator_type != GPT_IT_VPT) {
        gpt_set_error("gpt_set_integrator: integrator type %d is not supported (\"pt\", \"vpt\" and \"ao\" are)", integrator_type);
        return GPT_ERR_UNSUPPORTED;
    }
    if (integrator_type == GPT_IT_VPT && !ctx->media_ok) {
        gpt_set_error("gpt_set_integrator: \"vpt\": a medium record or a medium index of the scene is invalid");
        return GPT_ERR_INVALID_ARG;
    }
    if (integrator_type != GPT_IT_VPT && ctx->has_interface) {
        gpt_set_error("gpt_set_integrator: the scene has surfaces without a material (media interfaces); only \"vpt\" renders them");
        return GPT_ERR_UNSUPPORTED;
    }
    ctx->P.integrator = integrator_type;
    if (integrator_type == GPT_IT_PT || integrator_type == GPT_IT_VPT) ctx->P.max_depth = max_depth;
    else ctx->P.ao_max_dist = max_dist;
    return GPT_OK;
}

int gpt_set_option(gpt_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) { gpt_set_error("gpt_set_option: null argument"); return GPT_ERR_INVALID_ARG; }
    const std::string n(name);
    if (n == "lds_scene" && (value == 0 || value == 1)) ctx->lds_scene = value != 0;
    else if (n == "vpt_walk_kernel" && (value == 0 || value == 1)) ctx->force_walk = value != 0;
    else if (n == "max_batch" && value >= 1 && value <= 65536) { ctx->max_batch = (uint32_t)value; ctx->max_batch_set = true; }
    else if (n == "chunk_iters" && value >= 0 && value <= 65536) ctx->chunk_override = (uint32_t)value;
    else {
        gpt_set_error("gpt_set_option: unknown option or value out of range: %s = %lld", name, (long long)value);
        return GPT_ERR_INVALID_ARG;
    }
    return GPT_OK;
}

int gpt_get_option(gpt_ctx *ctx, const char *name, int64_t *value)
{
    if (!ctx || !name || !value) { gpt_set_error("gpt_get_option: null argument"); return GPT_ERR_INVALID_ARG; }
    const std::string n(name);
    if (n == "lds_scene") *value = ctx->lds_scene ? 1 : 0;
    else if (n == "vpt_walk_kernel") *value = ctx->force_walk ? 1 : 0;
    else if (n == "max_batch") *value = ctx->max_batch;
    else if (n == "chunk_iters") *value = ctx->chunk_override;
    else if (n == "traversal_order") *value = ctx->P.traversal;
    // read-only: what the renderer actually does with the current scene and settings
    else if (n == "lds_scene_active") *value = (ctx->lds_scene && render_scene_fits_lds(ctx->P)) ? 1 : 0;
    else if (n == "walk_kernel_active") *value = render_uses_walk_kernel(ctx->P, ctx->force_walk) ? 1 : 0;
    else if (n == "last_batch") *value = ctx->last_batch_cap;
    else if (n == "sample_plane_bytes") *value = (int64_t)ctx->sample_bytes;
    else if (n == "owned_tiles") *value = ctx->P.n_tiles > ctx->P.rank ? (ctx->P.n_tiles - ctx->P.rank + ctx->P.n_ranks - 1) / ctx->P.n_ranks : 0;
    else if (n == "wide_fallback") *value = ctx->wide_fallback ? 1 : 0;
    else if (n == "last_trace_us") *value = (int64_t)(ctx->last_trace_ms * 1000.0);
    else if (n == "output_kernel_us") {       // since the last gpt_kernel_time_reset; synchronises like gpt_kernel_time
        const int rc = fold_events(ctx);
        if (rc != GPT_OK) return rc;
        *value = (int64_t)(ctx->timed_output_ms * 1000.0);
    }
    else if (n == "last_reduce_us") *value = span_us(ctx->ev_reduce);        // synchronises on the end of that reduce
    else if (n == "last_tonemap_us") *value = span_us(ctx->ev_tonemap);
    else {
        gpt_set_error("gpt_get_option: unknown option %s", name);
        return GPT_ERR_INVALID_ARG;
    }
    return GPT_OK;
}

int gpt_set_traversal_order(gpt_ctx *ctx, int32_t order)
{
    if (!ctx || (order != GPT_TRAVERSAL_AUTO && order != GPT_TRAVERSAL_REFERENCE && order != GPT_TRAVERSAL_WIDE4)) {
        gpt_set_error("gpt_set_traversal_order: invalid argument");
        return GPT_ERR_INVALID_ARG;
    }
    if (order == GPT_TRAVERSAL_AUTO)
        order = (ctx->wide_ok && !gpt_scene_fits_lds(ctx->P.n_nodes, ctx->P.n_prims, ctx->P.n_lights, ctx->P.n_materials)) ? GPT_TRAVERSAL_WIDE4
                                                                                                                         : GPT_TRAVERSAL_REFERENCE;
    if (order == GPT_TRAVERSAL_WIDE4) {
        if (!ctx->wide_ok) {
            gpt_set_error("gpt_set_traversal_order: the scene has no wide tree (empty scene, or deeper than %d wide levels)", (GPT_WIDE_STACK_MAX - 1) / 3);
            return GPT_ERR_UNSUPPORTED;
        }
        if (!ctx->P.wide) {
            HIP_TRY(hipSetDevice(ctx->device));
            // ONE allocation holds the wide nodes and, behind them, a copy of the triangle records: a trip of the wide loop fetches
            // for its node lanes and its leaf lanes with the same seven instructions, a 32-bit offset per lane from one base
            // (a leaf lane reads 112 bytes from its triangle on: the copy is padded by two records)
            const size_t wide_bytes = ctx->wide_host.size() * sizeof(DevWideNode);
            const size_t tri_bytes = (size_t)ctx->P.n_prims * sizeof(DevTri);
            if (wide_bytes + tri_bytes + 3 * sizeof(DevTri) > (size_t)UINT32_MAX) {
                gpt_set_error("gpt_set_traversal_order: the wide tree and its triangles exceed 4 GB");
                return GPT_ERR_UNSUPPORTED;
            }
            void *p = nullptr;
            const hipError_t ea = g_fail_next_wide_alloc.exchange(false) ? hipErrorOutOfMemory : hipMalloc(&p, wide_bytes + tri_bytes + 3 * sizeof(DevTri));
            if (ea == hipErrorOutOfMemory) {
                (void)hipGetLastError();           // not sticky: the caller may go on in the reference's order
                gpt_set_error("gpt_set_traversal_order: no device memory for the wide tree (%zu bytes)", wide_bytes + tri_bytes + 3 * sizeof(DevTri));
                return GPT_ERR_UNSUPPORTED;
            }
            HIP_TRY(ea);
            if (hipMemset(static_cast<char *>(p) + wide_bytes + tri_bytes, 0, 3 * sizeof(DevTri)) != hipSuccess ||
                hipMemcpy(p, ctx->wide_host.data(), wide_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(static_cast<char *>(p) + wide_bytes, ctx->P.tris, tri_bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
                gpt_set_error("gpt_set_traversal_order: uploading the wide tree: %s", hipGetErrorString(hipGetLastError()));
                (void)hipFree(p);                  // nothing half-uploaded stays behind
                return GPT_ERR_HIP;
            }
            ctx->allocs.push_back(p);
            ctx->P.wide = static_cast<const DevWideNode *>(p);
            ctx->P.wide_tris_off = (uint32_t)wide_bytes;
            std::vector<DevWideNode>().swap(ctx->wide_host);
        }
        if (!ctx->P.wide_stack) {          // spill space of the per-ray stacks: one slice per wave that can be resident
            int bpc = 1;
            for (int w = 0; w < 2; ++w)
                for (int c = 0; c < 2; ++c) bpc = std::max(bpc, ctx->blocks_per_cu_wide[w][c]);
            const size_t blocks = (size_t)std::max(ctx->n_cus, 1) * (size_t)bpc;
            void *p = nullptr;
            const size_t n = blocks * 4 * (size_t)kWideWaveSliceDwords;
            HIP_TRY(hipSetDevice(ctx->device));
            const hipError_t ea = hipMalloc(&p, n * sizeof(uint32_t));
            if (ea == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                gpt_set_error("gpt_set_traversal_order: no device memory for the wide walk's stack slices (%zu bytes)", n * sizeof(uint32_t));
                return GPT_ERR_UNSUPPORTED;
            }
            HIP_TRY(ea);
            ctx->allocs.push_back(p);
            ctx->P.wide_stack = static_cast<uint32_t *>(p);
            ctx->P.wide_stack_blocks = (uint32_t)blocks;
        }
    }
    ctx->P.traversal = order;
    return GPT_OK;
}

int gpt_render(gpt_ctx *ctx, const gpt_camera *camera, uint32_t iter_first, uint32_t iter_count, int reset,
               float *out_tonemapped_dev)
{
    if (!ctx || !camera) {
        gpt_set_error("gpt_render: null context or camera");
        return GPT_ERR_INVALID_ARG;
    }
    if (ctx->P.integrator == GPT_IT_VPT && (camera->medium < -1 || camera->medium >= ctx->n_mediums)) {
        gpt_set_error("gpt_render: camera medium %d outside [-1,%d)", camera->medium, ctx->n_mediums);
        return GPT_ERR_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const bool count = ctx->count_next;
    HIP_TRY(hipMemsetAsync(ctx->counters, 0, 16 * sizeof(unsigned long long), ctx->stream));     // (probe builds count without the counting kernels)
    const uint32_t n_tiles = ctx->P.n_tiles, rank = ctx->P.rank, n_ranks = ctx->P.n_ranks;
    const uint32_t n_owned = (n_tiles > rank) ? (n_tiles - rank + n_ranks - 1) / n_ranks : 0u;
    // reset clears the accumulator of EVERY pixel (Output, pathtracer.cu:2521).  With tile ownership the
    // kernels only touch their own tiles, so the rest is cleared here: the sum-reduce over ranks then sees
    // zeros outside each rank's support.
    if (reset && n_ranks > 1)
        HIP_TRY(hipMemsetAsync(ctx->acc, 0, (size_t)ctx->width * ctx->height * 3 * sizeof(float), ctx->stream));
    if (n_owned == 0 || iter_count == 0) return GPT_OK;

    // Sample planes for one launch (grown on demand).  A plane holds one float4 per pixel slot of the tiles THIS rank owns
    // (tile-major: slot = local tile * 64 + pixel in tile), so a 1/8 shard allocates an eighth.  A launch has a fixed cost
    // of 0.2-0.6 ms (start-up and the wait for the last work items), so launches are long: up to 256 iterations - 8.5 GB
    // of planes for a full 1080p frame - but never more than 16 GiB nor 80 % of the free device memory, and a failed
    // allocation is retried at half the batch (tools/gpu_batch.py: 64 -> 256 iterations per launch is +1.5 % on the full
    // frame and 7.0x -> 7.5x for a 1/8 shard).
    const size_t plane_bytes = (size_t)n_owned * 64 * 4 * sizeof(float);
    size_t budget = (size_t)16 << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // what this context already holds for planes counts as available: it is freed before a new allocation
            const size_t avail = free_b + ctx->sample_bytes;
            if (avail / 5 * 4 < budget) budget = avail / 5 * 4;
        } else {
            (void)hipGetLastError();
        }
    }
    uint32_t by_memory = (uint32_t)(budget / plane_bytes);
    if (by_memory < 1) by_memory = 1;
    // unless the caller fixed it, a rank that owns 1/N of the tiles takes N times the iterations per launch: the same plane
    // memory and the same work per launch as one GPU with the whole frame, so the fixed cost of a launch stays amortised
    const uint64_t wanted = ctx->max_batch_set ? (uint64_t)ctx->max_batch : (uint64_t)ctx->max_batch * n_ranks;
    const uint32_t max_batch = wanted < by_memory ? (uint32_t)wanted : by_memory;
    uint32_t batch_cap = iter_count < max_batch ? iter_count : max_batch;
    if (ctx->batch_cap_limit && ctx->batch_limit_owned == n_owned && batch_cap > ctx->batch_cap_limit)
        batch_cap = ctx->batch_cap_limit;     // a larger allocation failed before: do not free, fail and halve again every frame
    if (ctx->sample_bytes < plane_bytes * batch_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->samples) {
            HIP_TRY(hipFree(ctx->samples));
            for (auto &a : ctx->allocs) if (a == ctx->samples) a = nullptr;
            ctx->samples = nullptr;
            ctx->sample_bytes = 0;
        }
        void *ps = nullptr;
        for (;;) {
            const hipError_t e = hipMalloc(&ps, plane_bytes * batch_cap);
            if (e == hipSuccess) break;
            (void)hipGetLastError();          // clear the sticky error: a later launch must not report it
            if (batch_cap == 1) {
                gpt_set_error("gpt_render: cannot allocate one sample plane (%zu bytes): %s", plane_bytes, hipGetErrorString(e));
                return GPT_ERR_HIP;
            }
            batch_cap = (batch_cap + 1) / 2;
            ctx->batch_cap_limit = batch_cap;
            ctx->batch_limit_owned = n_owned;
        }
        ctx->allocs.push_back(ps);
        ctx->samples = static_cast<float *>(ps);
        ctx->sample_bytes = plane_bytes * batch_cap;
    }
    ctx->last_batch_cap = batch_cap;

    const bool wide = ctx->P.traversal == GPT_TRAVERSAL_WIDE4;
    const long resident_waves = (long)ctx->n_cus * (wide ? ctx->blocks_per_cu_wide : ctx->blocks_per_cu)[render_uses_walk_kernel(ctx->P, ctx->force_walk) ? 1 : 0][count ? 1 : 0] * 4;
    for (uint32_t done = 0; done < iter_count; done += batch_cap) {
        DevParams P = ctx->P;
        P.cam = *camera;
        P.samples = ctx->samples;
        P.plane = (uint64_t)n_owned * 64;
        P.iter_first = iter_first + done;
        P.iter_count = iter_count - done < batch_cap ? iter_count - done : batch_cap;
        P.reset = (reset && done == 0) ? 1 : 0;
        P.out = (done + P.iter_count == iter_count) ? out_tonemapped_dev : nullptr;
        // Work items = (iteration chunk, tile), all independent; a wave streams from one item into the next
        // without draining.  Two costs pull in opposite directions: every item costs a queue round trip and a
        // partly filled hand-out (~5 us of a wave's time), and at the end of the launch waves wait for the last
        // items (~half an item).  With I items per wave the sum is I*o/T + 1/(2I), minimal at I = sqrt(T/(2o));
        // T is estimated from the sample count at ~1 sample/us/wave.  1080p x 64 iterations: full frame -> chunks
        // of 8-9 iterations, a 1/8 shard -> 3-4 (measured optimum: tools/gpu_chunks.py, profiles/r01/kernel_variants.log).
        const double owned_samples = (double)n_owned * 64.0 * (double)P.iter_count;
        double items_per_wave = std::sqrt(owned_samples / (10.0 * (double)resident_waves));
        if (items_per_wave < 4.0) items_per_wave = 4.0;
        long n_chunks = (long)(items_per_wave * (double)resident_waves / (double)n_owned + 0.5);
        if (n_chunks > (long)P.iter_count) n_chunks = (long)P.iter_count;
        if (n_chunks < 1) n_chunks = 1;
        uint32_t chunk_iters = (uint32_t)((P.iter_count + n_chunks - 1) / n_chunks);
        if (ctx->chunk_override) chunk_iters = ctx->chunk_override < P.iter_count ? ctx->chunk_override : P.iter_count;
        P.chunk_iters = chunk_iters;
        P.n_chunks = (P.iter_count + chunk_iters - 1) / chunk_iters;
        // persistent grid: as many 4-wave workgroups as stay resident, no more than the work
        const long needed = ((long)n_owned * P.n_chunks + 3) / 4;
        const long resident = resident_waves / 4;
        const int n_blocks = (int)(needed < resident ? needed : resident);

        HIP_TRY(hipMemsetAsync(ctx->tile_counter, 0, 64, ctx->stream));
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (!ctx->free_events.empty()) {
            ev = ctx->free_events.back();
            ctx->free_events.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ev.first));
            HIP_TRY(hipEventCreate(&ev.second));
        }
        HIP_TRY(hipEventRecord(ev.first, ctx->stream));
        HIP_TRY(launch_render(P, count, n_blocks, ctx->lds_scene, ctx->force_walk, ctx->stream));
        HIP_TRY(hipEventRecord(ev.second, ctx->stream));
        ctx->events.push_back(ev);
        HIP_TRY(launch_output(P, ctx->stream));
        hipEvent_t mark = nullptr;
        if (!ctx->free_marks.empty()) {
            mark = ctx->free_marks.back();
            ctx->free_marks.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&mark));
        }
        HIP_TRY(hipEventRecord(mark, ctx->stream));
        ctx->output_events.push_back({ev.second, mark});
        if (ctx->events.size() > 2048) {
            int rc = fold_events(ctx);
            if (rc != GPT_OK) return rc;
        }
    }
    return GPT_OK;
}

int gpt_tonemap(gpt_ctx *ctx, uint32_t iter, int filmic, float *out_dev)
{
    return gpt_tonemap_from(ctx, ctx ? ctx->acc : nullptr, iter, filmic, out_dev);
}

int gpt_tonemap_from(gpt_ctx *ctx, const float *acc_dev, uint32_t iter, int filmic, float *out_dev)
{
    if (!ctx || !acc_dev || !out_dev || iter == 0) {
        gpt_set_error("gpt_tonemap: invalid argument");
        return GPT_ERR_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = span_begin(ctx, ctx->ev_tonemap);
    if (rc != GPT_OK) return rc;
    HIP_TRY(launch_tonemap(acc_dev, out_dev, ctx->P.stride, ctx->P.rows, iter, filmic, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_tonemap[1], ctx->stream));
    return GPT_OK;
}

// ---- multi-GPU: pixel tiles across ranks, ONE sum-reduce of the float3 accumulator ------------------------
// RCCL is opened at run time: a single-GPU caller never needs it, and a process that already holds an RCCL (a PyTorch
// process: the wheel bundles its own librccl.so.1 next to its HIP runtime) gets that same copy by soname.
namespace {
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl *rccl()
{
    static Rccl lib;
    static std::string why;                 // why it could not be loaded (kept: dlerror() can be read once only)
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            lib.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib.handle) break;
            const char *e = dlerror();
            why = e ? e : "dlopen failed";
        }
        if (lib.handle) {
            lib.GetUniqueId = reinterpret_cast<decltype(lib.GetUniqueId)>(dlsym(lib.handle, "ncclGetUniqueId"));
            lib.CommInitRank = reinterpret_cast<decltype(lib.CommInitRank)>(dlsym(lib.handle, "ncclCommInitRank"));
            lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(dlsym(lib.handle, "ncclCommDestroy"));
            lib.Reduce = reinterpret_cast<decltype(lib.Reduce)>(dlsym(lib.handle, "ncclReduce"));
            lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(dlsym(lib.handle, "ncclGetErrorString"));
            if (!lib.GetUniqueId || !lib.CommInitRank || !lib.CommDestroy || !lib.Reduce || !lib.GetErrorString) {
                dlclose(lib.handle);
                lib.handle = nullptr;
                why = "a symbol of the RCCL API is missing (ncclGetUniqueId / CommInitRank / CommDestroy / Reduce / GetErrorString)";
            }
        }
    });
    if (!lib.handle) {
        gpt_set_error("RCCL (librccl.so.1) cannot be loaded: %s", why.c_str());
        return nullptr;
    }
    return &lib;
}
}  // namespace

int gpt_comm_unique_id(void *id128)
{
    if (!id128) { gpt_set_error("gpt_comm_unique_id: null argument"); return GPT_ERR_INVALID_ARG; }
    Rccl *L = rccl();
    if (!L) return GPT_ERR_UNSUPPORTED;
    ncclUniqueId id;
    const ncclResult_t e = L->GetUniqueId(&id);
    if (e != ncclSuccess) { gpt_set_error("ncclGetUniqueId: %s", L->GetErrorString(e)); return GPT_ERR_HIP; }
    static_assert(sizeof(id) == 128, "the ABI passes the RCCL unique id as 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return GPT_OK;
}

int gpt_comm_init(gpt_ctx *ctx, int rank, int n_ranks, const void *id128)
{
    if (!ctx || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        gpt_set_error("gpt_comm_init: invalid argument (rank %d of %d)", rank, n_ranks);
        return GPT_ERR_INVALID_ARG;
    }
    if (ctx->comm) { gpt_set_error("gpt_comm_init: the context already has a communicator"); return GPT_ERR_INVALID_ARG; }
    Rccl *L = rccl();
    if (!L) return GPT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t e = L->CommInitRank(&ctx->comm, n_ranks, id, rank);
    if (e != ncclSuccess) {
        ctx->comm = nullptr;
        gpt_set_error("ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, L->GetErrorString(e));
        return GPT_ERR_HIP;
    }
    ctx->comm_rank = rank;
    ctx->comm_size = n_ranks;
    return gpt_set_tile_owner(ctx, rank, n_ranks);
}

static int ensure_reduced(gpt_ctx *ctx)
{
    if (ctx->reduced) return GPT_OK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, (size_t)ctx->width * ctx->height * 3 * sizeof(float)));
    ctx->allocs.push_back(p);
    ctx->reduced = static_cast<float *>(p);
    return GPT_OK;
}

int gpt_reduce_film(gpt_ctx *ctx, int root)
{
    if (!ctx || !ctx->comm || root < 0 || root >= ctx->comm_size) {
        gpt_set_error("gpt_reduce_film: no communicator (gpt_comm_init) or root %d out of range", root);
        return GPT_ERR_INVALID_ARG;
    }
    Rccl *L = rccl();
    if (!L) return GPT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    // The receive buffer is NOT the accumulator: acc keeps exactly this rank's tiles, so progressive rendering
    // (gpt_render with reset = 0, then another reduce) never counts the other ranks' tiles twice on the root.
    if (ctx->comm_rank == root) {
        const int rc = ensure_reduced(ctx);
        if (rc != GPT_OK) return rc;
    }
    const size_t count = (size_t)ctx->width * ctx->height * 3;
    const int rc_span = span_begin(ctx, ctx->ev_reduce);
    if (rc_span != GPT_OK) return rc_span;
    const ncclResult_t e = L->Reduce(ctx->acc, ctx->comm_rank == root ? ctx->reduced : nullptr, count, ncclFloat32, ncclSum, root, ctx->comm,
                                     ctx->stream);
    if (e != ncclSuccess) { gpt_set_error("ncclReduce: %s", L->GetErrorString(e)); return GPT_ERR_HIP; }
    HIP_TRY(hipEventRecord(ctx->ev_reduce[1], ctx->stream));
    return GPT_OK;
}

float *gpt_reduced_device_ptr(gpt_ctx *ctx)
{
    if (!ctx || ensure_reduced(ctx) != GPT_OK) return nullptr;
    return ctx->reduced;
}

int gpt_read_reduced(gpt_ctx *ctx, float *host_rgb)
{
    if (!ctx || !ctx->reduced) { gpt_set_error("gpt_read_reduced: nothing has been reduced on this rank"); return GPT_ERR_INVALID_ARG; }
    return gpt_copy_to_host(ctx, ctx->reduced, host_rgb, (size_t)ctx->width * ctx->height * 3);
}

int gpt_comm_destroy(gpt_ctx *ctx)
{
    if (!ctx || !ctx->comm) return GPT_OK;
    Rccl *L = rccl();
    if (L) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)L->CommDestroy(ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->comm_size = 1;
    ctx->comm_rank = 0;
    return GPT_OK;
}

int gpt_synchronize(gpt_ctx *ctx)
{
    if (!ctx) { gpt_set_error("gpt_synchronize: null context"); return GPT_ERR_INVALID_ARG; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GPT_OK;
}

float *gpt_accum_device_ptr(gpt_ctx *ctx) { return ctx ? ctx->acc : nullptr; }
float *gpt_color_device_ptr(gpt_ctx *ctx) { return ctx ? ctx->color : nullptr; }

int gpt_copy_to_host(gpt_ctx *ctx, const float *dev, float *host, size_t n_floats)
{
    if (!ctx || !dev || !host) { gpt_set_error("gpt_copy_to_host: null argument"); return GPT_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(host, dev, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return GPT_OK;
}

int gpt_read_accum(gpt_ctx *ctx, float *host_rgb)
{
    if (!ctx) { gpt_set_error("gpt_read_accum: null context"); return GPT_ERR_INVALID_ARG; }
    return gpt_copy_to_host(ctx, ctx->acc, host_rgb, (size_t)ctx->width * ctx->height * 3);
}

int gpt_read_color(gpt_ctx *ctx, float *host_rgb)
{
    if (!ctx) { gpt_set_error("gpt_read_color: null context"); return GPT_ERR_INVALID_ARG; }
    return gpt_copy_to_host(ctx, ctx->color, host_rgb, (size_t)ctx->width * ctx->height * 3);
}

int gpt_write_state(gpt_ctx *ctx, const float *host_acc, const float *host_color)
{
    if (!ctx || !host_acc || !host_color) { gpt_set_error("gpt_write_state: null argument"); return GPT_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t bytes = (size_t)ctx->width * ctx->height * 3 * sizeof(float);
    HIP_TRY(hipMemcpy(ctx->acc, host_acc, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->color, host_color, bytes, hipMemcpyHostToDevice));
    return GPT_OK;
}

int gpt_bind_film(gpt_ctx *ctx, float *acc_dev, float *color_dev)
{
    if (!ctx) { gpt_set_error("gpt_bind_film: null context"); return GPT_ERR_INVALID_ARG; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (acc_dev) { ctx->acc = acc_dev; ctx->P.acc = acc_dev; }
    if (color_dev) { ctx->color = color_dev; ctx->P.color = color_dev; }
    return GPT_OK;
}

int gpt_end(gpt_ctx *ctx)
{
    if (!ctx) return GPT_OK;
    (void)gpt_comm_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto &ev : ctx->free_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto &ev : ctx->output_events) (void)hipEventDestroy(ev.second);
    for (hipEvent_t e : ctx->free_marks) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ctx->ev_reduce[0], ctx->ev_reduce[1], ctx->ev_tonemap[0], ctx->ev_tonemap[1]}) if (e) (void)hipEventDestroy(e);
    for (void *p : ctx->allocs) if (p) (void)hipFree(p);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GPT_OK;
}

int gpt_kernel_time(gpt_ctx *ctx, uint32_t *launches, double *total_ms)
{
    if (!ctx) { gpt_set_error("gpt_kernel_time: null context"); return GPT_ERR_INVALID_ARG; }
    int rc = fold_events(ctx);
    if (rc != GPT_OK) return rc;
    if (launches) *launches = ctx->timed_launches;
    if (total_ms) *total_ms = ctx->timed_ms;
    return GPT_OK;
}

int gpt_kernel_time_reset(gpt_ctx *ctx)
{
    if (!ctx) { gpt_set_error("gpt_kernel_time_reset: null context"); return GPT_ERR_INVALID_ARG; }
    int rc = fold_events(ctx);
    ctx->timed_launches = 0;
    ctx->timed_ms = 0.0;
    ctx->timed_output_ms = 0.0;
    return rc;
}

int gpt_enable_counters(gpt_ctx *ctx, int enable)
{
    if (!ctx) { gpt_set_error("gpt_enable_counters: null context"); return GPT_ERR_INVALID_ARG; }
    ctx->count_next = enable != 0;
    return GPT_OK;
}

int gpt_read_probe_counters(gpt_ctx *ctx, uint64_t out16[16])
{
    if (!ctx || !out16) { gpt_set_error("gpt_read_probe_counters: null argument"); return GPT_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long tmp[16];
    HIP_TRY(hipMemcpy(tmp, ctx->counters, sizeof(tmp), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) out16[i] = tmp[i];
    return GPT_OK;
}

int gpt_read_counters(gpt_ctx *ctx, uint64_t out6[6])
{
    if (!ctx || !out6) { gpt_set_error("gpt_read_counters: null argument"); return GPT_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long tmp[6];
    HIP_TRY(hipMemcpy(tmp, ctx->counters, sizeof(tmp), hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) out6[i] = tmp[i];
    return GPT_OK;
}

int gpt_debug_trace(gpt_ctx *ctx, const float *rays8, int n, int32_t *prim_out, float *tb_out)
{
    if (!ctx || !rays8 || !prim_out || !tb_out || n <= 0) { gpt_set_error("gpt_debug_trace: invalid argument"); return GPT_ERR_INVALID_ARG; }
    if (ctx->P.n_nodes <= 0) {
        for (int i = 0; i < n; ++i) { prim_out[i] = -1; tb_out[3 * i] = rays8[8 * i + 6]; tb_out[3 * i + 1] = tb_out[3 * i + 2] = 0.f; }
        return GPT_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<float4> h((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        const float *r = rays8 + 8 * (size_t)i;
        h[2 * (size_t)i] = make_float4(r[0], r[1], r[2], r[6]);
        h[2 * (size_t)i + 1] = make_float4(r[3], r[4], r[5], r[7] != 0.f ? 1.f : 0.f);
    }
    float4 *d_rays = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_rays, h.size() * sizeof(float4)));
    if (hipMalloc((void **)&d_out, (size_t)n * sizeof(float4)) != hipSuccess) { (void)hipFree(d_rays); gpt_set_error("gpt_debug_trace: hipMalloc failed"); return GPT_ERR_HIP; }
    int rc = GPT_OK;
    std::vector<float4> res((size_t)n);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    if (hipMemcpy(d_rays, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess ||
        hipEventRecord(e0, ctx->stream) != hipSuccess ||
        launch_trace_rays(ctx->P, ctx->lds_scene, d_rays, n, d_out, ctx->stream) != hipSuccess ||
        hipEventRecord(e1, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(res.data(), d_out, res.size() * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) {
        gpt_set_error("gpt_debug_trace: %s", hipGetErrorString(hipGetLastError()));
        rc = GPT_ERR_HIP;
    }
    if (rc == GPT_OK) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ctx->last_trace_ms = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_rays);
    (void)hipFree(d_out);
    if (rc != GPT_OK) return rc;
    for (int i = 0; i < n; ++i) {
        int32_t prim;
        std::memcpy(&prim, &res[(size_t)i].x, 4);
        prim_out[i] = prim;
        tb_out[3 * (size_t)i] = res[(size_t)i].y;
        tb_out[3 * (size_t)i + 1] = res[(size_t)i].z;
        tb_out[3 * (size_t)i + 2] = res[(size_t)i].w;
    }
    return GPT_OK;
}

int gpt_debug_math(int device, int fn, const float *x, const float *y, float *out, int n)
{
    if (!x || !out || n <= 0) { gpt_set_error("gpt_debug_math: invalid argument"); return GPT_ERR_INVALID_ARG; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { gpt_set_error("gpt_debug_math: no HIP device"); return GPT_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    HIP_TRY(hipMalloc((void **)&dx, bytes));
    HIP_TRY(hipMalloc((void **)&dy, bytes));
    HIP_TRY(hipMalloc((void **)&dout, bytes));
    HIP_TRY(hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy, y ? y : x, bytes, hipMemcpyHostToDevice));
    HIP_TRY(launch_debug_math(fn, dx, dy, dout, n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return GPT_OK;
}

int gpt_debug_fail_next_wide_alloc(int enable)
{
    g_fail_next_wide_alloc.store(enable != 0);
    return GPT_OK;
}

int gpt_debug_bsdf(int device, const gpt_material *material, const gpt_texture *texture, const float *geom11, const float *in3, int n,
                   int mode, float *out7)
{
    if (!material || !geom11 || !in3 || !out7 || n <= 0 || (mode != 0 && mode != 1)) { gpt_set_error("gpt_debug_bsdf: invalid argument"); return GPT_ERR_INVALID_ARG; }
    if (material->textureIdx != -1 && (material->textureIdx != 0 || !texture || !texture->data || texture->width <= 0 || texture->height <= 0)) {
        gpt_set_error("gpt_debug_bsdf: the material's textureIdx must be -1, or 0 with a texture given");
        return GPT_ERR_INVALID_ARG;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { gpt_set_error("gpt_debug_bsdf: no HIP device"); return GPT_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    gpt_material *dm = nullptr;
    DevTexture *dt = nullptr;
    gpt_uchar4 *dtexels = nullptr;
    float *dg = nullptr, *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void **)&dm, sizeof(gpt_material)));
    HIP_TRY(hipMemcpy(dm, material, sizeof(gpt_material), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&dt, sizeof(DevTexture)));
    DevTexture ht;
    ht.data = nullptr;
    ht.width = ht.height = 0;
    if (material->textureIdx == 0) {
        const size_t texels = (size_t)texture->width * (size_t)texture->height;
        HIP_TRY(hipMalloc((void **)&dtexels, texels * sizeof(gpt_uchar4)));
        HIP_TRY(hipMemcpy(dtexels, texture->data, texels * sizeof(gpt_uchar4), hipMemcpyHostToDevice));
        ht.data = dtexels;
        ht.width = texture->width;
        ht.height = texture->height;
    }
    HIP_TRY(hipMemcpy(dt, &ht, sizeof(DevTexture), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&dg, (size_t)n * 11 * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&din, (size_t)n * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&dout, (size_t)n * 7 * sizeof(float)));
    HIP_TRY(hipMemcpy(dg, geom11, (size_t)n * 11 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(din, in3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(launch_debug_bsdf(dm, dt, dg, din, n, mode, dout, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out7, dout, (size_t)n * 7 * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dm); (void)hipFree(dt); (void)hipFree(dtexels); (void)hipFree(dg); (void)hipFree(din); (void)hipFree(dout);
    return GPT_OK;
}

int gpt_debug_rng(int device, uint32_t pixel, uint32_t iter, uint32_t *seed_out, float *u_out, int n)
{
    if (!seed_out || !u_out || n <= 0) { gpt_set_error("gpt_debug_rng: invalid argument"); return GPT_ERR_INVALID_ARG; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { gpt_set_error("gpt_debug_rng: no HIP device"); return GPT_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    uint32_t *dseed = nullptr;
    float *du = nullptr;
    HIP_TRY(hipMalloc((void **)&dseed, sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&du, (size_t)n * sizeof(float)));
    HIP_TRY(launch_debug_rng(pixel, iter, dseed, du, n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(seed_out, dseed, sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(u_out, du, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dseed); (void)hipFree(du);
    return GPT_OK;
}

}  // extern "C"

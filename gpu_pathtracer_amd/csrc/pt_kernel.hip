// pt_kernel.hip — the per-pixel-sample path-tracing kernel for gfx950 (MI355X).
//
// Replaces the reference's `Path` + `Output` kernels (reference
// src/pathtracer.cu:880-1021, 2516-2531) and everything they call:
//   camera ray            src/camera.h:48-84
//   BVH traversal         src/pathtracer.cu:214-296, src/bbox.h:77-96
//   ray/triangle          src/mesh.h:45-98
//   BSDF sample / eval    src/pathtracer.cu:491-826 (+ helpers :51-169)
//   textures              src/pathtracer.cu:324-359
//   area / env lights     src/area.h:14-41, src/infinite.h:17-94, src/pathtracer.cu:172-185
//   RNG                   src/pathtracer.cu:40-49,888-889 (WangHash + thrust minstd_rand)
//
// Design (not a translation of the CUDA kernel, which is one thread per
// pixel-sample, one launch per spp, a 64-int private stack and 176-byte AoS
// primitive fetches):
//   * ONE launch renders a whole batch of iterations.  Waves are persistent:
//     each 64-lane wavefront pulls (8x8-pixel tile, iteration chunk) work items
//     from a global atomic queue; one lane owns one pixel for the iterations of
//     its chunk.  Every finished sample is written to its iteration's plane in
//     HBM (one 16-B store per sample; a 64-iteration 1080p batch is 2.1 GB of the 288 GB), so
//     work items are completely independent of each other, and
//     pt_output_kernel then folds the planes into the accumulator in iteration
//     order — exactly the reference's fp32 summation order (its Output kernel
//     runs once per spp, pathtracer.cu:2516-2531).
//   * Path regeneration: a lane whose path ends starts its next sample in the
//     same loop trip, so every lane carries a live path into every
//     closest-hit traversal.
//   * Traversal is threaded ("escape index") preorder: identical visit order
//     to the reference's push-right/push-left stack, but stackless — no
//     scratch, no LDS traffic, and the per-ray 1/d is hoisted.
//   * Nodes are 32 B, triangles 48 B (pt_layout.h): aligned 16-byte gathers.
//   * The hit record is built once per ray from (prim, b1, b2, t), not on every
//     accepted candidate.
//   * Float contract: no FMA contraction, IEEE divide/sqrt, soft-math
//     transcendentals (include/gpt_softmath.h) — the same operation sequence as
//     oracle/pt_oracle.c, so parity is checked bit for bit.
//
// Draw order inside argument lists is left to right.

#include "pt_device.h"
#include "pt_shade.h"
#include "../../include/gpt_wide_bvh.h"
#include "../../include/gpt_traversal.h"

namespace pt {

// ------------------------------------------------------------ traversal ------
// Rays live in an LDS pool that belongs to ONE wavefront (no workgroup barrier
// is ever needed).  Each bounce, every lane (= one path) deposits up to three
// rays that share its shading point: the path ray (closest hit), the MIS light
// ray (closest hit) and the shadow ray (any hit).  __ballot/mbcnt compaction
// packs them densely — path rays first, then MIS rays, then shadow rays, i.e.
// longest first — and the traversal loop below drains the pool with dynamic
// fetch: a lane whose ray is finished writes its result into the ray's slot and
// takes the next untraced ray, whichever path it belongs to.  The reference
// traces the same rays one kind at a time with every thread waiting for the
// slowest ray of its warp (src/pathtracer.cu:905,942,960).
//
// Inside the loop the wave votes each trip between a node step and a triangle
// step (whichever more lanes are waiting for).  Per ray the visit order is
// exactly the reference's: threaded preorder == its push-right/push-left stack
// (pathtracer.cu:221-252), leaf triangles in index order; arithmetic is
// bbox.h:77-96 and mesh.h:45-67.
constexpr int kPoolSlots = 192;                    // 3 rays x 64 lanes
constexpr int kWaveLdsFloat4 = 2 * kPoolSlots + 64;   // slots (2 x float4) + one origin per lane
// Scenes in global memory ("carry" layout): rays keep FIXED slots (kind * 64 + owner lane) so that results and
// unfinished rays survive a shading round, a compacted fetch-order list says which slots are new this round, one
// pending mask per lane says which of its rays are still out, and every lane has a record for a suspended ray.
constexpr int kOrderOff = kWaveLdsFloat4;             // 192 x u16 slot indices
constexpr int kPendOff = kOrderOff + 24;              // 64 x u32: bit k = ray of kind k not finished yet
constexpr int kSuspOff = kPendOff + 16;               // 64 x {cursor, tri, tri_last, slot | result-in-progress}
constexpr int kWaveCarryFloat4 = kSuspOff + 128;
// A dry pool with <= T rays in flight ends the drain.  T trades emptier shading rounds against shorter tails: long
// rays and cheap materials want it high, expensive BSDFs low.  Measured optimum (tools/gpu_configs.py): 8-16 on the
// 2k-triangle material scene and the env-lit 22k-triangle scene, 24 on the 253k-triangle stand-in.
#ifndef PT_STOP_T
#define PT_STOP_T 24                                  // deep trees (>= 64k nodes)
#endif
#ifndef PT_STOP_T_SMALL
#define PT_STOP_T_SMALL 12                            // everything else in global memory
#endif
#ifndef PT_FETCH_T
#define PT_FETCH_T 12
#endif
#ifndef PT_XCD_QUEUES
#define PT_XCD_QUEUES 1          // 1 (scenes in global memory): one work queue per XCD (workgroup b runs on XCD b % 8), each a contiguous eighth of the items, so that
#endif                           // an XCD's private L2 serves neighbouring tiles; a wave whose queue is empty takes from the next XCD's.  0: one queue
#ifndef PT_TILE_STRIP
#define PT_TILE_STRIP 16         // N > 0 (scenes in global memory): consecutive items walk down strips of N tiles instead of along whole tile rows (512 consecutive tiles = a compact block)
#endif
#ifndef PT_VOTE_NODE_SHIFT
#define PT_VOTE_NODE_SHIFT 1      // a node trip costs half a triangle trip: 2 * node-waiters >= triangle-waiters
#endif
// scenes in global memory: a node trip waits on L2/HBM and most of a long ray's steps are node steps, so triangle
// trips are taken earlier: node trip iff node-waiters >= 2 * triangle-waiters (253k-triangle stand-in at 4K:
// node-biased 534, plain majority 604, this rule 656, 4x 650, 8x 604 Msamples/s)
#ifndef PT_GLOBAL_VOTE_TRI_SHIFT
#define PT_GLOBAL_VOTE_TRI_SHIFT 1
#endif
#define PT_GLOBAL_VOTE_WEIGHT "s_lshl_b32 s72, s72, " PT_STR(PT_GLOBAL_VOTE_TRI_SHIFT) "\n"
#ifndef PT_VOTE_TRI_SHIFT
#define PT_VOTE_TRI_SHIFT 0
#endif
constexpr int kFetchThreshold = PT_FETCH_T;        // idle lanes that trigger a refill
static_assert(PT_STOP_T < 64 - PT_FETCH_T && PT_STOP_T_SMALL < 64 - PT_FETCH_T, "a resumed drain must start with a refill (the loop header runs after it)");

struct RaySet {        // what one lane deposits
    V3 org;
    V3 dir_p, dir_m, dir_s;
    float tmax_s;
    bool has_p, has_m, has_s;
    bool mis_any;      // the MIS ray only needs hit / no hit (no emitter can be its closest hit)
};
struct PoolLayout {    // wave-uniform: where each kind starts, from the ballots
    unsigned long long m_p, m_m, m_s;
    int n_p, n_m, n_rays;
};
struct RayResult {
    int prim;          // -1 = miss
    float t, b1, b2;
};


// slot = { dir.xyz, tmax } { 1/dir.xyz, bits(owner lane | any_hit << 8) }; the result replaces the second half
__device__ __forceinline__ void pool_put(float4 *pool, int slot, V3 dir, float tmax, unsigned owner, bool any_hit)
{
    // bbox.h:79 computes 1/d at every node visit; the quotient is the same every time
    pool[2 * slot] = make_float4(dir.x, dir.y, dir.z, tmax);
    pool[2 * slot + 1] = make_float4(1.f / dir.x, 1.f / dir.y, 1.f / dir.z, __int_as_float((int)(owner | (any_hit ? 256u : 0u))));
}

// P_TMAX: the path ray's interval ends at rs.tmax_s (the one-ray-at-a-time Volpath walks bounded segments with it)
template <bool P_TMAX = false>
__device__ __forceinline__ PoolLayout pool_deposit(float4 *pool, const RaySet &rs, unsigned lane)
{
    PoolLayout L;
    L.m_p = ballot(rs.has_p);
    L.m_m = ballot(rs.has_m);
    L.m_s = ballot(rs.has_s);
    L.n_p = popc(L.m_p);
    L.n_m = popc(L.m_m);
    L.n_rays = L.n_p + L.n_m + popc(L.m_s);
    pool[2 * kPoolSlots + lane] = make_float4(rs.org.x, rs.org.y, rs.org.z, 0.f);
    if (rs.has_p) pool_put(pool, lane_rank(L.m_p), rs.dir_p, P_TMAX ? rs.tmax_s : __builtin_inff(), lane, false);
    if (rs.has_m) pool_put(pool, L.n_p + lane_rank(L.m_m), rs.dir_m, __builtin_inff(), lane, rs.mis_any);
    if (rs.has_s) pool_put(pool, L.n_p + L.n_m + lane_rank(L.m_s), rs.dir_s, rs.tmax_s, lane, true);
    return L;
}

// carry layout: the rays of the lanes in `depositing` go to their fixed slots; returns how many are new
template <bool P_TMAX = false>
__device__ __forceinline__ int pool_deposit_fixed(float4 *pool, const RaySet &rs, unsigned lane, bool depositing)
{
    const bool dp = depositing && rs.has_p, dm = depositing && rs.has_m, ds = depositing && rs.has_s;
    const unsigned long long m_p = ballot(dp), m_m = ballot(dm), m_s = ballot(ds);
    const int n_p = popc(m_p), n_m = popc(m_m);
    unsigned short *order = reinterpret_cast<unsigned short *>(pool + kOrderOff);
    unsigned *pend = reinterpret_cast<unsigned *>(pool + kPendOff);
    if (depositing) {
        pool[2 * kPoolSlots + lane] = make_float4(rs.org.x, rs.org.y, rs.org.z, 0.f);
        pend[lane] = (dp ? 1u : 0u) | (dm ? 2u : 0u) | (ds ? 4u : 0u);
    }
    if (dp) { pool_put(pool, (int)lane, rs.dir_p, P_TMAX ? rs.tmax_s : __builtin_inff(), lane, false); order[lane_rank(m_p)] = (unsigned short)lane; }
    if (dm) { pool_put(pool, 64 + (int)lane, rs.dir_m, __builtin_inff(), lane, rs.mis_any); order[n_p + lane_rank(m_m)] = (unsigned short)(64u + lane); }
    if (ds) { pool_put(pool, 128 + (int)lane, rs.dir_s, rs.tmax_s, lane, true); order[n_p + n_m + lane_rank(m_s)] = (unsigned short)(128u + lane); }
    return n_p + n_m + popc(m_s);
}

__device__ __forceinline__ RayResult pool_result(const float4 *pool, int slot)
{
    const float4 r = pool[2 * slot + 1];
    RayResult out;
    out.prim = __float_as_int(r.x);
    out.t = r.y;
    out.b1 = r.z;
    out.b2 = r.w;
    return out;
}

// Every cursor is a BYTE offset: DevNode.link/last are stored pre-scaled (pt_layout.h), so a visit is "load
// at cursor" with no address arithmetic.  Two memory spaces:
//   GlobalScene  cursors are offsets from the node / triangle arrays in HBM
//   LdsScene     the scene was staged into LDS with every link rebased to an absolute LDS address
struct GlobalScene {
    const char *nodes, *tris;
    int first, end;                        // cursor of node 0, cursor one past the last node
    int tri_bias;                          // cursor of triangle 0
    __device__ __forceinline__ int first_of(V3) const { return 0; }
    static constexpr int vote_node_shift = 0, vote_tri_shift = PT_GLOBAL_VOTE_TRI_SHIFT;
    __device__ __forceinline__ float4 node4(int c) const { return *reinterpret_cast<const float4 *>(nodes + (unsigned)c); }
    __device__ __forceinline__ float4 tri4(int c) const { return *reinterpret_cast<const float4 *>(tris + (unsigned)c); }
    __device__ __forceinline__ float tri1(int c) const { return *reinterpret_cast<const float *>(tris + (unsigned)c); }
};
struct LdsScene {
    int first, end, tri_bias;
    __device__ __forceinline__ int first_of(V3) const { return first; }
    static constexpr int vote_node_shift = PT_VOTE_NODE_SHIFT, vote_tri_shift = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float native4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ float4 node4(int c) const
    {
        const native4 v = *(const __attribute__((address_space(3))) native4 *)(unsigned)c;
        return make_float4(v.x, v.y, v.z, v.w);
    }
    __device__ __forceinline__ float4 tri4(int c) const { return node4(c); }
    __device__ __forceinline__ float tri1(int c) const { return *(const __attribute__((address_space(3))) float *)(unsigned)c; }
#else
    float4 node4(int) const { return float4(); }
    float4 tri4(int) const { return float4(); }
    float tri1(int) const { return 0.f; }
#endif
};

template <bool COUNT, bool FIXED, class Mem>
__device__ __forceinline__ void trace_pool(const DevParams &P, float4 *pool, int n_rays, Counters &cnt, const Mem mem)
{
    const int n_bytes = mem.end - mem.first;      // the node array
    const int tri_bias = mem.tri_bias;
    const float tmin_ray = P.eps;          // every ray of the integrator starts at epsilon

    // Loop-carried lane state is integers only; every predicate is a compare made in the trip that uses it,
    // so a vote is one v_cmp into an SGPR pair and the mask algebra stays on the scalar unit.  (A bool
    // carried around the loop lives in an SGPR lane mask that the compiler re-materialises through a VGPR
    // for every ballot.)  An idle lane has slot < 0, idx == end and tri > tri_last.
    int next = 0;                          // wave-uniform: first slot nobody has taken yet
    int slot = -1;
    int any_hit = 0;
    V3 o = v3(0.f), d = v3(0.f), inv = v3(0.f);
    float tmax = 0.f;
    int end = 0;                                   // per ray: one past the last node of its variant
    int idx = end, tri = 0, tri_last = -1;
    int hprim = -1;
    float hb1 = 0.f, hb2 = 0.f;

    for (;;) {
        const bool want_tri = tri <= tri_last;
        const bool more_nodes = idx < end;
        const unsigned long long m_tri = ballot(want_tri);
        const unsigned long long m_more = ballot(more_nodes);
        const unsigned long long m_has = ballot(slot >= 0);
        const unsigned long long m_fin = m_has & ~(m_tri | m_more);
        if (m_fin != 0ull) {
            if (slot >= 0 && !want_tri && !more_nodes) {
                // ray finished: the result overwrites the second half of its slot (the direction stays)
                const int prim = hprim < 0 ? -1 : (int)((unsigned)(hprim - tri_bias) / 48u);
                pool[2 * slot + 1] = make_float4(__int_as_float(prim), tmax, hb1, hb2);
                if (FIXED)      // carry layout: tell the owner (lane slot % 64) that its ray of kind slot / 64 is back
                    atomicAnd(reinterpret_cast<unsigned *>(pool + kPendOff) + (slot & 63), ~(1u << (slot >> 6)));
                slot = -1;
            }
        }
        const unsigned long long m_busy = m_has & ~m_fin;
        const int n_idle = 64 - popc(m_busy);
        if (next < n_rays && n_idle >= kFetchThreshold) {       // (an empty wave has 64 idle lanes)
            // ---- refill: idle lanes take the next slots, in lane order ------------------
            const int nth = next + lane_rank(~m_busy);
            if (slot < 0 && nth < n_rays) {
                const int mine = FIXED ? (int)reinterpret_cast<const unsigned short *>(pool + kOrderOff)[nth] : nth;
                const float4 r0 = pool[2 * mine];
                const float4 r1 = pool[2 * mine + 1];
                const int tag = __float_as_int(r1.w);
                const float4 ro = pool[2 * kPoolSlots + (tag & 255)];
                slot = mine;
                any_hit = tag & 256;
                o = V3{ro.x, ro.y, ro.z};
                d = V3{r0.x, r0.y, r0.z};
                inv = V3{r1.x, r1.y, r1.z};
                tmax = r0.w;
                idx = mem.first_of(d);
                end = idx + n_bytes;
                tri = 0;
                tri_last = -1;
                hprim = -1;
                hb1 = hb2 = 0.f;
            }
            next += n_idle;               // next trip votes with the new rays on board
        } else {
        if (m_busy == 0ull) break;
        if (COUNT && first_active_lane()) { cnt.w_trip++; cnt.l_trip += (uint32_t)popc(m_busy); }

        const unsigned long long m_node = m_more & ~m_tri;
        const bool node_trip = (popc(m_node) << Mem::vote_node_shift) >= (popc(m_tri) << Mem::vote_tri_shift);
        {
            if (node_trip && more_nodes && !want_tri) {
                // ---- one node --------------------------------------------------------------
                const float4 a = mem.node4(idx);
                const float4 b = mem.node4(idx + 16);
                if (COUNT) { cnt.node_visits++; if (first_active_lane()) cnt.w_node++; }
                const float t1 = (a.x - o.x) * inv.x;
                const float t2 = (a.w - o.x) * inv.x;
                const float t3 = (a.y - o.y) * inv.y;
                const float t4 = (b.x - o.y) * inv.y;
                const float t5 = (a.z - o.z) * inv.z;
                const float t6 = (b.y - o.z) * inv.z;
                const float tn = fmax_(fmax_(fmin_(t1, t2), fmin_(t3, t4)), fmin_(t5, t6));
                const float tf = fmin_(fmin_(fmax_(t1, t2), fmax_(t3, t4)), fmax_(t5, t6));
                // the reference's comparison senses (NaN passes)
                const bool box = !(tf <= 0.00001f) && !(tn > tf) && !(tn > tmax);
                const int link = __float_as_int(b.z);
                const int last = __float_as_int(b.w);
                const bool leaf = last >= 0;
                idx = (box || leaf) ? idx + 32 : link;
                if (box && leaf) {
                    tri = link;
                    tri_last = last;
                }
            }
        }
        {
            if (!node_trip && want_tri) {
                // ---- one triangle ----------------------------------------------------------
                const int i = tri;
                tri += 48;
                const float4 q0 = mem.tri4(i);
                const float4 q1 = mem.tri4(i + 16);
                const float e2z = mem.tri1(i + 32);
                if (COUNT) { cnt.prim_tests++; if (first_active_lane()) cnt.w_prim++; }
                const V3 v1 = V3{q0.x, q0.y, q0.z};
                const V3 e1 = V3{q0.w, q1.x, q1.y};
                const V3 e2 = V3{q1.z, q1.w, e2z};
                const V3 s1 = cross(d, e2);
                const float divisor = dot(s1, e1);
                // reference: float invDivisor = 1.0 / divisor (double divide rounded to float)
                // == the correctly rounded float quotient (53 >= 2*24+2 bits)
                const float invDivisor = 1.0f / divisor;
                const V3 s = o - v1;
                const float b1 = dot(s, s1) * invDivisor;
                const V3 s2 = cross(s, e1);
                const float b2 = dot(d, s2) * invDivisor;
                const float tt = dot(e2, s2) * invDivisor;
                const bool accept = !(fabs_(divisor) < 1e-8f) && !(b1 < 0.0f || b1 > 1.0f) &&
                                    !(b2 < 0.0f || b1 + b2 > 1.0f) && !(tt < tmin_ray || tt > tmax);
                if (accept) {
                    tmax = tt;
                    hprim = i;
                    hb1 = b1;
                    hb2 = b2;
                    if (any_hit != 0) {      // IntersectP: the first accepted triangle ends the ray
                        idx = end;
                        tri_last = -1;
                    }
                }
            }
        }
        }
    }
}

// ---- GPT_TRAVERSAL_WIDE4: one lane per ray on the 4-wide tree (include/gpt_wide_bvh.h) ------------------------------------
// The walk of one ray is the one the header specifies (and oracle/pt_oracle.c restates).  One lane owns one ray, as in the
// binary loops, but a node step loads ONE 128-byte record (seven dwordx4: six planes of four boxes and the four children's
// ready-made stack entries) and tests all four boxes; the hit children are ordered nearest first by a five-exchange sorting
// network on (key, entry) pairs, the nearest becomes the current entry and the others go to the ray's stack.  Against the
// binary loop: a quarter of the dependent fetches, about half the bytes through the vector cache, and no half-empty
// lookahead halves.  A leaf step tests one triangle (Moeller-Trumbore, the same instructions as trace_pool<>).
// A trip serves both kinds of lanes: the fetches of the lanes at a wide node and of the lanes at a leaf go out together,
// then the node block and the triangle block run under their own lane masks.
// The stack: kWideStackDepth entries per lane in LDS, level-major (level l of lane i at word 64 l + i: conflict-free for any
// mix of levels); deeper entries go to the wave's slice of P.wide_stack in global memory, in the same arrangement
// (3 * depth + 1 <= GPT_WIDE_STACK_MAX is checked by the host).
#ifndef PT_WIDE_STACK_DEPTH
#define PT_WIDE_STACK_DEPTH 9
#endif
constexpr int kWideStackDepth = PT_WIDE_STACK_DEPTH;
constexpr int kWideStackOff = kSuspOff;                           // where the binary loops keep their suspend records: the wide walk's
                                                                  // are in the wave's slice of P.wide_stack (read and written once per drain)
constexpr int kWaveWideFloat4 = kSuspOff + 16 * kWideStackDepth;  // one level of 64 lanes = 16 float4; 9 levels: 10 112 B per wave, four workgroups per CU
#ifndef PT_WIDE_FETCH_T
#define PT_WIDE_FETCH_T 12                                        // idle lanes that trigger a refill
#endif
#ifndef PT_WIDE_STOP_T
#define PT_WIDE_STOP_T 20                                         // a dry pool with at most this many rays in flight ends the drain (16 - 24 within 1 %: profiles/r06/d1_wide_stop_sweep.log)
#endif
#ifndef PT_WIDE_STOP_T_SMALL
#define PT_WIDE_STOP_T_SMALL 12                                   // ... trees of fewer than 64k binary nodes
#endif
static_assert(PT_WIDE_STOP_T < 64 - PT_WIDE_FETCH_T && PT_WIDE_STOP_T_SMALL < 64 - PT_WIDE_FETCH_T, "a resumed drain starts with a refill");
// A block of a trip costs the wave the same whether one lane or sixty-four take part.  Lanes at a leaf wait until
// PT_WIDE_LEAF_MIN of them have gathered (or no lane is at a wide node), and the other way round with PT_WIDE_NODE_MIN.
#ifndef PT_WIDE_LEAF_MIN
#define PT_WIDE_LEAF_MIN 8
#endif
#ifndef PT_WIDE_NODE_MIN
#define PT_WIDE_NODE_MIN 1
#endif

// one compare-exchange of the sorting network: afterwards ka <= kb (the entries travel with their keys)
__device__ __forceinline__ void wide_cex(unsigned &ka, unsigned &ea, unsigned &kb, unsigned &eb)
{
    const bool swap = kb < ka;
    const unsigned k0 = swap ? kb : ka, k1 = swap ? ka : kb, e0 = swap ? eb : ea, e1 = swap ? ea : eb;
    ka = k0; kb = k1; ea = e0; eb = e1;
}

template <bool COUNT>
__device__ __forceinline__ void trace_pool_wide(const DevParams &P, float4 *pool, int n_rays, Counters &cnt)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned *stk = reinterpret_cast<unsigned *>(pool + kWideStackOff) + lane;
    volatile unsigned *spill = P.wide_stack + (size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)kWideWaveSliceDwords + 512u + lane;
    const char *wnodes = reinterpret_cast<const char *>(P.wide);
    const char *tris = reinterpret_cast<const char *>(P.tris);
    const float tmin_ray = P.eps;

    int next = 0;                          // wave-uniform: first ray of the fetch order nobody has taken yet
    int slot = -1, any_hit = 0;
    V3 o = v3(0.f), d = v3(0.f), inv = v3(0.f);
    float tmax = 0.f;
    unsigned cur = GPT_WIDE_NONE;
    int sp = 0;
    int bprim = -1;
    float bt = 0.f, bb1 = 0.f, bb2 = 0.f;

    for (;;) {
        const bool has = slot >= 0;
        const bool fin = has && cur == GPT_WIDE_NONE;
        const unsigned long long m_has = ballot(has), m_fin = ballot(fin);
        if (m_fin != 0ull) {
            if (fin) {       // a miss reports the end of the interval, like the other loops
                pool[2 * slot + 1] = make_float4(__int_as_float(bprim), bprim < 0 ? tmax : bt, bb1, bb2);
                atomicAnd(reinterpret_cast<unsigned *>(pool + kPendOff) + (slot & 63), ~(1u << (slot >> 6)));
                slot = -1;
            }
        }
        const unsigned long long m_busy = m_has & ~m_fin;
        const int n_idle = 64 - popc(m_busy);
        if (next < n_rays && n_idle >= PT_WIDE_FETCH_T) {
            // ---- refill: idle lanes take the next rays of the fetch order, in lane order
            const int nth = next + lane_rank(~m_busy);
            if (slot < 0 && nth < n_rays) {
                const int mine = (int)reinterpret_cast<const unsigned short *>(pool + kOrderOff)[nth];
                const float4 r0 = pool[2 * mine];
                const float4 r1 = pool[2 * mine + 1];
                const int tag = __float_as_int(r1.w);
                const float4 ro = pool[2 * kPoolSlots + (tag & 255)];
                slot = mine;
                any_hit = tag & 256;
                o = V3{ro.x, ro.y, ro.z};
                d = V3{r0.x, r0.y, r0.z};
                inv = V3{r1.x, r1.y, r1.z};
                tmax = r0.w;
                cur = 0u;                  // wide node 0
                sp = 0;
                bprim = -1;
                bt = bb1 = bb2 = 0.f;
            }
            next += n_idle;
            continue;
        }
        if (m_busy == 0ull) break;

        bool leaf = has && !fin && (cur >> 31) != 0u;
        bool inner = has && !fin && (cur >> 31) == 0u;
        {
            const int n_leaf = popc(ballot(leaf)), n_inner = popc(ballot(inner));
            if (n_leaf < PT_WIDE_LEAF_MIN && n_inner > 0) leaf = false;                      // the leaves wait
            else if (n_inner < PT_WIDE_NODE_MIN && n_leaf > 0) inner = false;                // the wide nodes wait
        }
        if (COUNT) {                        // utilisation probes: trips, busy lanes, trips with a node block / a triangle block
            const bool any_inner = ballot(inner) != 0ull, any_leaf = ballot(leaf) != 0ull;
            if (lane == 0u) {
                cnt.w_trip++;
                cnt.l_trip += (uint32_t)popc(m_busy);
                if (any_inner) cnt.w_node++;
                if (any_leaf) cnt.w_prim++;
            }
        }
        bool pop = false;
        if (inner) {
            // ---- a wide node: four boxes, bbox.h:77-96 each ---------------------------------
            const float4 *np = reinterpret_cast<const float4 *>(wnodes + (size_t)cur);
            const float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5];
            const uint4 en = *reinterpret_cast<const uint4 *>(np + 6);
            if (COUNT) cnt.node_visits++;
            const float blx[4] = {lx.x, lx.y, lx.z, lx.w}, bly[4] = {ly.x, ly.y, ly.z, ly.w}, blz[4] = {lz.x, lz.y, lz.z, lz.w};
            const float bhx[4] = {hx.x, hx.y, hx.z, hx.w}, bhy[4] = {hy.x, hy.y, hy.z, hy.w}, bhz[4] = {hz.x, hz.y, hz.z, hz.w};
            unsigned e[4] = {en.x, en.y, en.z, en.w}, key[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t1 = (blx[k] - o.x) * inv.x;
                const float t2 = (bhx[k] - o.x) * inv.x;
                const float t3 = (bly[k] - o.y) * inv.y;
                const float t4 = (bhy[k] - o.y) * inv.y;
                const float t5 = (blz[k] - o.z) * inv.z;
                const float t6 = (bhz[k] - o.z) * inv.z;
                const float tn = fmax_(fmax_(fmin_(t1, t2), fmin_(t3, t4)), fmin_(t5, t6));
                const float tf = fmin_(fmin_(fmax_(t1, t2), fmax_(t3, t4)), fmax_(t5, t6));
                const bool hit = e[k] != GPT_WIDE_NONE && !(tf <= 0.00001f) && !(tn > tf) && !(tn > tmax);
                // order key (gpt_wide_key): the bit pattern of max(tn, +0), slot in the two lowest bits; a child that is not hit
                // gets the largest key and sorts behind every hit one
                key[k] = hit ? ((__float_as_uint(tn > 0.0f ? tn : 0.0f) & ~3u) | (unsigned)k) : 0xffffffffu;
            }
            wide_cex(key[0], e[0], key[1], e[1]);
            wide_cex(key[2], e[2], key[3], e[3]);
            wide_cex(key[0], e[0], key[2], e[2]);
            wide_cex(key[1], e[1], key[3], e[3]);
            wide_cex(key[1], e[1], key[2], e[2]);
            if (key[0] == 0xffffffffu) {
                pop = true;
            } else {
                // the nearest is visited next; the others are pushed farthest first: sorted child j ends at sp' - j, sp' the new size
                const int top = sp + 3 - (key[1] == 0xffffffffu ? 1 : 0) - (key[2] == 0xffffffffu ? 1 : 0) - (key[3] == 0xffffffffu ? 1 : 0);
#pragma unroll
                for (int j = 3; j >= 1; --j)
                    if (key[j] != 0xffffffffu) {
                        const int at = top - j;
                        if (at < kWideStackDepth) stk[64 * at] = e[j];
                        else spill[64 * at] = e[j];
                    }
                sp = top;
                cur = e[0];
            }
        }
        if (leaf) {
            // ---- a leaf: its first triangle, mesh.h:45-67 -------------------------------------
            const int prim = (int)(cur & 0x07ffffffu), left = (int)((cur >> 27) & 15u);      // left = triangles after this one
            const char *tp = tris + (size_t)prim * 48u;
            const float4 q0 = *reinterpret_cast<const float4 *>(tp);
            const float4 q1 = *reinterpret_cast<const float4 *>(tp + 16);
            const float e2z = *reinterpret_cast<const float *>(tp + 32);
            if (COUNT) cnt.prim_tests++;
            const V3 v1 = V3{q0.x, q0.y, q0.z};
            const V3 e1 = V3{q0.w, q1.x, q1.y};
            const V3 e2 = V3{q1.z, q1.w, e2z};
            const V3 s1 = cross(d, e2);
            const float divisor = dot(s1, e1);
            const float invDivisor = 1.0f / divisor;           // == (float)(1.0 / divisor), see trace_pool<>
            const V3 s = o - v1;
            const float b1 = dot(s, s1) * invDivisor;
            const V3 s2 = cross(s, e1);
            const float b2 = dot(d, s2) * invDivisor;
            const float tt = dot(e2, s2) * invDivisor;
            const bool accept = !(fabs_(divisor) < 1e-8f) && !(b1 < 0.0f || b1 > 1.0f) &&
                                !(b2 < 0.0f || b1 + b2 > 1.0f) && !(tt < tmin_ray || tt > tmax);
            bool ended = false;
            if (accept) {
                if (bprim < 0 || tt < bt || (tt == bt && prim > bprim)) {
                    bprim = prim;
                    bt = tt;
                    bb1 = b1;
                    bb2 = b2;
                }
                if (tt < tmax) tmax = tt;                      // (a NaN distance never becomes the interval's end)
                ended = any_hit != 0;                          // IntersectP: the first accepted triangle ends the ray
            }
            if (ended) {
                cur = GPT_WIDE_NONE;
                sp = 0;
            } else if (left > 0) {
                cur = 0x80000000u | ((unsigned)(left - 1) << 27) | (unsigned)(prim + 1);
            } else {
                pop = true;
            }
        }
        if (pop) {
            if (sp > 0) {
                --sp;
                cur = sp < kWideStackDepth ? stk[64 * sp] : spill[64 * sp];
            } else {
                cur = GPT_WIDE_NONE;
            }
        }
    }
}

// ---- the same loop, hand-scheduled for a scene staged in LDS ---------------------------------------------
// trace_pool<> above is the specification; this is its instruction-for-instruction twin with the control
// skeleton the compiler cannot be talked into.  Measured on gfx950 (tools/micro/issue_rate.hip): a wave issues
// one dependent instruction every ~9 cycles, VALU or SALU alike, and with 4 waves per SIMD that - not VALU
// throughput - bounds this loop.  The compiler spends ~54 scalar instructions per trip on lane-mask phis,
// saveexec pairs and uniform-bool round trips; written by hand the skeleton is 16.  All floating-point
// instructions (and their order of evaluation) are the ones the compiler emits for trace_pool<>, so the two
// are bit-identical; tests/test_gpu_parity.py runs both (the counting build uses trace_pool<>).
//
// Register map (all clobbered, nothing is live across):
//   v[0:2] origin   v[4:6] dir  v7 (tmax as loaded)   v[8:10] 1/dir  v11 tag
//   v12 node cursor  v13 triangle cursor  v14 last triangle  v15 slot address (-1 = idle lane)
//   v[20:23] result {triangle cursor / index, tmax, b1, b2}   v[24:32] node or triangle data
//   v33..v43 temporaries (40 VGPRs in all).  The block sits at the
//   bottom of the register file: with v80..v123 the allocator spilled 17 dwords per lane, here none.
//   Scenes in global memory also use v3 (one past the last node of the ray's node array) and v[44:52] (the node after
//   the one being visited, PT_NODE2_LOOKAHEAD, or the triangle after the one being tested, PT_TRI2_LOOKAHEAD).
//   s[60:61] m_tri  s[62:63] m_more / m_node  s[64:65] m_has / m_busy  s[66:69] scratch masks
//   s70 next  s71 s72 counts  s76 1e-8f  s77 2^100
// The kernel always runs full wavefronts (256-thread workgroups, wave-uniform control flow), so exec is
// restored to all ones.
// gfx950 hazards honoured by hand: >= 2 wait states between a VALU write of VCC/SGPR and a VALU read of it
// (4 for v_div_fmas), >= 1 between v_rcp_f32 and the use of its result.
#define PT_COMMA ,
#define PT_STR2(x) #x
#define PT_STR(x) PT_STR2(x)
// Hooks of the loop macro.
//  classic (LDS scenes): every lane starts idle, slot i of the compacted pool is ray i, a dry pool is drained to the end.
#define PT_ENTRY_IDLE \
        "v_mov_b32_e32 v12, %[end]\n" "v_mov_b32_e32 v13, 0\n" "v_mov_b32_e32 v14, -1\n" "v_mov_b32_e32 v15, -1\n"
#define PT_FETCH_COMPACT "v_lshl_add_u32 v15, v33, 5, %[pool]\n"
#define PT_CURSOR_FIRST "v_mov_b32_e32 v12, %[first]\n"
//  scenes in global memory: cursors are byte offsets from the node array; v3 = one past its last node
#define PT_CURSOR_VARIANT \
        "v_mov_b32_e32 v12, %[first]\n" "v_add_u32_e32 v3, %[end], v12\n"
#define PT_DRY_DRAIN "s_cmp_lg_u64 s[64:65], 0\n" "s_cbranch_scc1 TP_VOTE_%=\n"
//  carry (scenes in global memory): a lane resumes the ray it was tracing when the last drain stopped (its cursors and
//  partial result come back from its suspend record, direction and origin from the ray's slot); the i-th new ray is
//  slot order[i]; a finished ray clears its bit in the owner's pending mask; a dry pool with few rays left in flight
//  ends the drain (if this round had new rays: otherwise nothing would ever get ready) and every lane parks its state.
#define PT_ENTRY_RESUME \
        "v_mbcnt_lo_u32_b32 v33, -1, 0\n" "v_mbcnt_hi_u32_b32 v33, -1, v33\n" "v_lshl_add_u32 v34, v33, 5, %[susp]\n" \
        "ds_read_b128 v[12:15], v34\n" "ds_read_b128 v[20:23], v34 offset:16\n" "s_waitcnt lgkmcnt(0)\n" \
        "v_mov_b32_e32 v3, 0\n" "v_cmp_lt_i32_e64 s[64:65], -1, v15\n" "s_mov_b64 exec, s[64:65]\n" \
        "ds_read_b128 v[4:7], v15\n" "ds_read_b128 v[8:11], v15 offset:16\n" "s_waitcnt lgkmcnt(0)\n" \
        "v_mov_b32_e32 v34, %[first]\n" "v_add_u32_e32 v3, %[end], v34\n" \
        "v_and_b32_e32 v33, 0xff, v11\n" "v_lshl_add_u32 v33, v33, 4, %[pool]\n" "ds_read_b96 v[0:2], v33 offset:%[org]\n" \
        "s_waitcnt lgkmcnt(0)\n" "s_mov_b64 exec, -1\n" \
        "v_cmp_le_i32_e64 s[60:61], v13, v14\n" "v_cmp_gt_i32_e64 s[62:63], v3, v12\n"   /* the vote masks of the loop header: a drain that starts with resumed rays and nothing to fetch goes from the dry-pool test straight to the vote */
#define PT_FETCH_ORDERED \
        "v_lshl_add_u32 v34, v33, 1, %[order]\n" "ds_read_u16 v34, v34\n" "s_waitcnt lgkmcnt(0)\n" "v_lshl_add_u32 v15, v34, 5, %[pool]\n"
#define PT_FINISH_PENDING \
        "v_subrev_u32_e32 v34, %[pool], v15\n" "v_lshrrev_b32_e32 v35, 11, v34\n" "v_bfe_u32 v34, v34, 5, 6\n" \
        "v_lshlrev_b32_e64 v35, v35, 1\n" "v_not_b32_e32 v35, v35\n" "v_lshl_add_u32 v34, v34, 2, %[pend]\n" "ds_and_b32 v34, v35\n"
#define PT_DRY_MAY_STOP \
        "s_cmp_eq_u64 s[64:65], 0\n" "s_cbranch_scc1 TP_DONE_%=\n" \
        "s_cmp_eq_u32 %[allow], 0\n" "s_cbranch_scc1 TP_VOTE_%=\n" \
        "s_bcnt1_i32_b64 s71, s[64:65]\n" "s_cmp_gt_u32 s71, %[tstop]\n" "s_cbranch_scc1 TP_VOTE_%=\n"
#define PT_EXIT_SUSPEND \
        "v_mbcnt_lo_u32_b32 v33, -1, 0\n" "v_mbcnt_hi_u32_b32 v33, -1, v33\n" "v_lshl_add_u32 v34, v33, 5, %[susp]\n" \
        "ds_write_b128 v34, v[12:15]\n" "ds_write_b128 v34, v[20:23] offset:16\n"

//  node lookahead (scenes in global memory, opt-in): a node trip also fetches the NEXT node in memory (cursor + 32: in
//  the threaded preorder that is the left child).  A lane whose node is an inner node with a hit box descends to exactly
//  that node, so it visits it in the same trip - same visits in the same order, one memory round trip instead of two.
//  At entry: s[66:67] = box hit, vcc = box hit && leaf, v12 already advanced; the second node sits in v[44:51].
#define PT_NODE2_LOOKAHEAD \
        "s_andn2_b64 s[66:67], s[66:67], vcc\n" "s_mov_b64 exec, s[66:67]\n" "s_cbranch_execz TP_N2_%=\n" "s_waitcnt vmcnt(0)\n" \
        "v_sub_f32_e32 v33, v44, v0\n" "v_sub_f32_e32 v34, v47, v0\n" "v_sub_f32_e32 v35, v45, v1\n" "v_sub_f32_e32 v37, v46, v2\n" \
        "v_sub_f32_e32 v36, v48, v1\n" "v_sub_f32_e32 v38, v49, v2\n" \
        "v_mul_f32_e32 v33, v8, v33\n" "v_mul_f32_e32 v34, v8, v34\n" "v_mul_f32_e32 v35, v9, v35\n" "v_mul_f32_e32 v36, v9, v36\n" \
        "v_mul_f32_e32 v37, v10, v37\n" "v_mul_f32_e32 v38, v10, v38\n" \
        "v_min_f32_e32 v39, v33, v34\n" "v_min_f32_e32 v40, v35, v36\n" "v_min_f32_e32 v41, v37, v38\n" \
        "v_max_f32_e32 v33, v33, v34\n" "v_max_f32_e32 v35, v35, v36\n" "v_max_f32_e32 v37, v37, v38\n" \
        "v_min3_f32 v33, v33, v35, v37\n" "v_max3_f32 v39, v39, v40, v41\n" \
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v33\n" "v_min_f32_e32 v33, v33, v21\n" "v_cmp_nlt_f32_e64 s[66:67], v33, v39\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" "v_cmp_lt_i32_e32 vcc, -1, v51\n" "v_add_u32_e32 v33, 32, v12\n" \
        "s_or_b64 s[68:69], vcc, s[66:67]\n" "s_and_b64 vcc, vcc, s[66:67]\n" \
        "v_cndmask_b32_e64 v12, v50, v33, s[68:69]\n" "v_cndmask_b32_e32 v14, v14, v51, vcc\n" "v_cndmask_b32_e32 v13, v13, v50, vcc\n" \
        "TP_N2_%=:\n"

//  triangle lookahead (scenes in global memory): a triangle trip fetches two consecutive triangles; a lane whose leaf has
//  another one (and whose ray the first test did not end) tests it in the same trip, after the first - the order of the
//  tests and the interval they see are those of two trips.  Same code as the first test on v[44:52].
#define PT_TRI2_LOOKAHEAD(RAY_END) \
        "s_mov_b64 exec, s[60:61]\n" "v_cmp_le_i32_e32 vcc, v13, v14\n" "s_and_b64 exec, exec, vcc\n" "s_cbranch_scc0 TP_TRI_END2_%=\n" \
        "v_add_u32_e32 v13, 48, v13\n" "s_waitcnt vmcnt(0)\n" \
        "v_mul_f32_e32 v33, v5, v52\n" \
        "v_mul_f32_e32 v42, v6, v51\n" \
        "v_sub_f32_e32 v33, v33, v42\n" \
        "v_mul_f32_e32 v34, v6, v50\n" \
        "v_mul_f32_e32 v42, v4, v52\n" \
        "v_sub_f32_e32 v34, v34, v42\n" \
        "v_mul_f32_e32 v35, v4, v51\n" \
        "v_mul_f32_e32 v42, v5, v50\n" \
        "v_sub_f32_e32 v35, v35, v42\n" \
        "v_mul_f32_e32 v36, v33, v47\n" \
        "v_mul_f32_e32 v42, v34, v48\n" \
        "v_add_f32_e32 v36, v36, v42\n" \
        "v_mul_f32_e32 v42, v35, v49\n" \
        "v_add_f32_e32 v36, v36, v42\n" \
        "v_rcp_f32_e32 v38, v36\n" \
        "v_sub_f32_e32 v44, v0, v44\n" \
        "v_sub_f32_e32 v45, v1, v45\n" \
        "v_sub_f32_e32 v46, v2, v46\n" \
        "v_cmp_nle_f32_e64 s[66:67], abs(v36), s77\n" \
        "v_fma_f32 v41, -v36, v38, 1.0\n" \
        "v_fma_f32 v37, v41, v38, v38\n" \
        "s_cmp_lg_u64 s[66:67], 0\n" \
        "s_cbranch_scc1 TP_DIV_IEEE2_%=\n" \
        "TP_DIV_DONE2_%=:\n" \
        "v_mul_f32_e32 v43, v44, v33\n" \
        "v_mul_f32_e32 v42, v45, v34\n" \
        "v_add_f32_e32 v43, v43, v42\n" \
        "v_mul_f32_e32 v42, v46, v35\n" \
        "v_add_f32_e32 v43, v43, v42\n" \
        "v_mul_f32_e32 v33, v45, v49\n" \
        "v_mul_f32_e32 v42, v46, v48\n" \
        "v_sub_f32_e32 v33, v33, v42\n" \
        "v_mul_f32_e32 v34, v46, v47\n" \
        "v_mul_f32_e32 v42, v44, v49\n" \
        "v_sub_f32_e32 v34, v34, v42\n" \
        "v_mul_f32_e32 v35, v44, v48\n" \
        "v_mul_f32_e32 v42, v45, v47\n" \
        "v_sub_f32_e32 v35, v35, v42\n" \
        "v_mul_f32_e32 v43, v43, v37\n" \
        "v_mul_f32_e32 v38, v4, v33\n" \
        "v_mul_f32_e32 v42, v5, v34\n" \
        "v_add_f32_e32 v38, v38, v42\n" \
        "v_mul_f32_e32 v42, v6, v35\n" \
        "v_add_f32_e32 v38, v38, v42\n" \
        "v_mul_f32_e32 v38, v38, v37\n" \
        "v_cmp_nlt_f32_e64 s[66:67], abs(v36), s76\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v43\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v43\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v38\n" \
        "v_add_f32_e32 v42, v43, v38\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v42\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TP_TRI_END2_%=\n" \
        "v_mul_f32_e32 v39, v50, v33\n" \
        "v_mul_f32_e32 v42, v51, v34\n" \
        "v_add_f32_e32 v39, v39, v42\n" \
        "v_mul_f32_e32 v42, v52, v35\n" \
        "v_add_f32_e32 v39, v39, v42\n" \
        "v_mul_f32_e32 v39, v39, v37\n" \
        "v_cmp_ngt_f32_e32 vcc, %[eps], v39\n" \
        "v_cmp_ngt_f32_e64 s[66:67], v39, v21\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TP_TRI_END2_%=\n" \
        "v_and_b32_e32 v42, 0x100, v11\n" \
        "v_cmp_ne_u32_e32 vcc, 0, v42\n" \
        "v_mov_b32_e32 v44, " RAY_END "\n" \
        "v_mov_b32_e32 v21, v39\n" \
        "v_subrev_u32_e32 v20, 48, v13\n" \
        "v_mov_b32_e32 v22, v43\n" \
        "v_mov_b32_e32 v23, v38\n" \
        "v_cndmask_b32_e32 v12, v12, v44, vcc\n" \
        "v_cndmask_b32_e64 v14, v14, -1, vcc\n" \
        "TP_TRI_END2_%=:\n" "s_branch TP_T2X_%=\n" \
        "TP_DIV_IEEE2_%=:\n" \
        "v_div_scale_f32 v37, s[66:67], v36, v36, 1.0\n" \
        "v_div_scale_f32 v39, vcc, 1.0, v36, 1.0\n" \
        "v_rcp_f32_e32 v38, v37\n" \
        "s_nop 0\n" \
        "v_fma_f32 v41, -v37, v38, 1.0\n" \
        "v_fmac_f32_e32 v38, v41, v38\n" \
        "v_mul_f32_e32 v40, v39, v38\n" \
        "v_fma_f32 v41, -v37, v40, v39\n" \
        "v_fmac_f32_e32 v40, v41, v38\n" \
        "v_fma_f32 v37, -v37, v40, v39\n" \
        "v_div_fmas_f32 v37, v37, v38, v40\n" \
        "v_div_fixup_f32 v37, v37, v36, 1.0\n" \
        "s_branch TP_DIV_DONE2_%=\n" \
        "TP_T2X_%=:\n"

// The loop as a macro over the memory space of the scene (the only difference: how node and triangle records are
// loaded and which counter is waited on).  Comments live in the block above and in trace_pool<>.
#define PT_TRACE_ASM(LD_NODE, NODE_W1, NODE_W0, NODE2, TRI2, LD_TRI, WAIT_1, WAIT_0, VOTE_WEIGHT, ENTRY_STATE, FETCH_SLOT, CURSOR_EARLY, CURSOR_LATE, RAY_END, FINISH_EXTRA, DRY_POOL, EXIT_EXTRA, MORE_CLOBBERS, ...) \
    asm volatile( \
        "s_mov_b32 s70, 0\n" \
        "s_mov_b32 s76, 0x322bcc77\n" \
        "s_mov_b32 s77, 0x71800000\n" \
        "s_mov_b64 s[64:65], 0\n" \
        ENTRY_STATE \
        "s_branch TP_FILL_%=\n" \
        "TP_LOOP_%=:\n" \
        "v_cmp_le_i32_e64 s[60:61], v13, v14\n" \
        "v_cmp_gt_i32_e64 s[62:63], " RAY_END ", v12\n" \
        "v_cmp_lt_i32_e64 s[64:65], -1, v15\n" \
        "s_or_b64 s[66:67], s[60:61], s[62:63]\n" \
        "s_andn2_b64 s[68:69], s[64:65], s[66:67]\n" \
        "s_cbranch_scc1 TP_FIN_%=\n" \
        "TP_VOTE_%=:\n" \
        "s_andn2_b64 s[62:63], s[62:63], s[60:61]\n" \
        "s_bcnt1_i32_b64 s71, s[62:63]\n" \
        "s_bcnt1_i32_b64 s72, s[60:61]\n" \
        VOTE_WEIGHT \
        "s_cmp_ge_u32 s71, s72\n" \
        "s_cbranch_scc0 TP_TRI_%=\n" \
        "s_mov_b64 exec, s[62:63]\n" \
        LD_NODE \
        NODE_W1 \
        "v_sub_f32_e32 v33, v24, v0\n" \
        "v_sub_f32_e32 v34, v27, v0\n" \
        "v_sub_f32_e32 v35, v25, v1\n" \
        "v_sub_f32_e32 v37, v26, v2\n" \
        NODE_W0 \
        "v_sub_f32_e32 v36, v28, v1\n" \
        "v_sub_f32_e32 v38, v29, v2\n" \
        "v_mul_f32_e32 v33, v8, v33\n" \
        "v_mul_f32_e32 v34, v8, v34\n" \
        "v_mul_f32_e32 v35, v9, v35\n" \
        "v_mul_f32_e32 v36, v9, v36\n" \
        "v_mul_f32_e32 v37, v10, v37\n" \
        "v_mul_f32_e32 v38, v10, v38\n" \
        "v_min_f32_e32 v39, v33, v34\n" \
        "v_min_f32_e32 v40, v35, v36\n" \
        "v_min_f32_e32 v41, v37, v38\n" \
        "v_max_f32_e32 v33, v33, v34\n" \
        "v_max_f32_e32 v35, v35, v36\n" \
        "v_max_f32_e32 v37, v37, v38\n" \
        "v_min3_f32 v33, v33, v35, v37\n" \
        "v_max3_f32 v39, v39, v40, v41\n" \
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v33\n" \
        "v_min_f32_e32 v33, v33, v21\n" \
        "v_cmp_nlt_f32_e64 s[66:67], v33, v39\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_lt_i32_e32 vcc, -1, v31\n" \
        "v_add_u32_e32 v33, 32, v12\n" \
        "s_or_b64 s[68:69], vcc, s[66:67]\n" \
        "s_and_b64 vcc, vcc, s[66:67]\n" \
        "v_cndmask_b32_e64 v12, v30, v33, s[68:69]\n" \
        "v_cndmask_b32_e32 v14, v14, v31, vcc\n" \
        "v_cndmask_b32_e32 v13, v13, v30, vcc\n" \
        NODE2 \
        "s_mov_b64 exec, -1\n" \
        "s_branch TP_LOOP_%=\n" \
        "TP_TRI_%=:\n" \
        "s_mov_b64 exec, s[60:61]\n" \
        LD_TRI \
        "v_add_u32_e32 v13, 48, v13\n" \
        WAIT_1 \
        "v_mul_f32_e32 v33, v5, v32\n" \
        "v_mul_f32_e32 v42, v6, v31\n" \
        "v_sub_f32_e32 v33, v33, v42\n" \
        "v_mul_f32_e32 v34, v6, v30\n" \
        "v_mul_f32_e32 v42, v4, v32\n" \
        "v_sub_f32_e32 v34, v34, v42\n" \
        "v_mul_f32_e32 v35, v4, v31\n" \
        "v_mul_f32_e32 v42, v5, v30\n" \
        "v_sub_f32_e32 v35, v35, v42\n" \
        WAIT_0 \
        "v_mul_f32_e32 v36, v33, v27\n" \
        "v_mul_f32_e32 v42, v34, v28\n" \
        "v_add_f32_e32 v36, v36, v42\n" \
        "v_mul_f32_e32 v42, v35, v29\n" \
        "v_add_f32_e32 v36, v36, v42\n" \
        "v_rcp_f32_e32 v38, v36\n" \
        "v_sub_f32_e32 v24, v0, v24\n" \
        "v_sub_f32_e32 v25, v1, v25\n" \
        "v_sub_f32_e32 v26, v2, v26\n" \
        "v_cmp_nle_f32_e64 s[66:67], abs(v36), s77\n" \
        "v_fma_f32 v41, -v36, v38, 1.0\n" \
        "v_fma_f32 v37, v41, v38, v38\n" \
        "s_cmp_lg_u64 s[66:67], 0\n" \
        "s_cbranch_scc1 TP_DIV_IEEE_%=\n" \
        "TP_DIV_DONE_%=:\n" \
        "v_mul_f32_e32 v43, v24, v33\n" \
        "v_mul_f32_e32 v42, v25, v34\n" \
        "v_add_f32_e32 v43, v43, v42\n" \
        "v_mul_f32_e32 v42, v26, v35\n" \
        "v_add_f32_e32 v43, v43, v42\n" \
        "v_mul_f32_e32 v33, v25, v29\n" \
        "v_mul_f32_e32 v42, v26, v28\n" \
        "v_sub_f32_e32 v33, v33, v42\n" \
        "v_mul_f32_e32 v34, v26, v27\n" \
        "v_mul_f32_e32 v42, v24, v29\n" \
        "v_sub_f32_e32 v34, v34, v42\n" \
        "v_mul_f32_e32 v35, v24, v28\n" \
        "v_mul_f32_e32 v42, v25, v27\n" \
        "v_sub_f32_e32 v35, v35, v42\n" \
        "v_mul_f32_e32 v43, v43, v37\n" \
        "v_mul_f32_e32 v38, v4, v33\n" \
        "v_mul_f32_e32 v42, v5, v34\n" \
        "v_add_f32_e32 v38, v38, v42\n" \
        "v_mul_f32_e32 v42, v6, v35\n" \
        "v_add_f32_e32 v38, v38, v42\n" \
        "v_mul_f32_e32 v38, v38, v37\n" \
        "v_cmp_nlt_f32_e64 s[66:67], abs(v36), s76\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v43\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v43\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v38\n" \
        "v_add_f32_e32 v42, v43, v38\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v42\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TP_TRI_END_%=\n" \
        "v_mul_f32_e32 v39, v30, v33\n" \
        "v_mul_f32_e32 v42, v31, v34\n" \
        "v_add_f32_e32 v39, v39, v42\n" \
        "v_mul_f32_e32 v42, v32, v35\n" \
        "v_add_f32_e32 v39, v39, v42\n" \
        "v_mul_f32_e32 v39, v39, v37\n" \
        "v_cmp_ngt_f32_e32 vcc, %[eps], v39\n" \
        "v_cmp_ngt_f32_e64 s[66:67], v39, v21\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TP_TRI_END_%=\n" \
        "v_and_b32_e32 v42, 0x100, v11\n" \
        "v_cmp_ne_u32_e32 vcc, 0, v42\n" \
        "v_mov_b32_e32 v24, " RAY_END "\n" \
        "v_mov_b32_e32 v21, v39\n" \
        "v_subrev_u32_e32 v20, 48, v13\n" \
        "v_mov_b32_e32 v22, v43\n" \
        "v_mov_b32_e32 v23, v38\n" \
        "v_cndmask_b32_e32 v12, v12, v24, vcc\n" \
        "v_cndmask_b32_e64 v14, v14, -1, vcc\n" \
        "TP_TRI_END_%=:\n" \
        TRI2 \
        "s_mov_b64 exec, -1\n" \
        "s_branch TP_LOOP_%=\n" \
        "TP_DIV_IEEE_%=:\n" \
        "v_div_scale_f32 v37, s[66:67], v36, v36, 1.0\n" \
        "v_div_scale_f32 v39, vcc, 1.0, v36, 1.0\n" \
        "v_rcp_f32_e32 v38, v37\n" \
        "s_nop 0\n" \
        "v_fma_f32 v41, -v37, v38, 1.0\n" \
        "v_fmac_f32_e32 v38, v41, v38\n" \
        "v_mul_f32_e32 v40, v39, v38\n" \
        "v_fma_f32 v41, -v37, v40, v39\n" \
        "v_fmac_f32_e32 v40, v41, v38\n" \
        "v_fma_f32 v37, -v37, v40, v39\n" \
        "v_div_fmas_f32 v37, v37, v38, v40\n" \
        "v_div_fixup_f32 v37, v37, v36, 1.0\n" \
        "s_branch TP_DIV_DONE_%=\n" \
        "TP_FIN_%=:\n" \
        "s_mov_b64 exec, s[68:69]\n" \
        "s_mov_b32 s72, 0xaaaaaaab\n" \
        "v_cmp_gt_i32_e32 vcc, 0, v20\n" \
        "v_subrev_u32_e32 v33, %[bias], v20\n" \
        "v_mul_hi_u32 v33, v33, s72\n" \
        "v_lshrrev_b32_e32 v33, 5, v33\n" \
        "v_cndmask_b32_e64 v20, v33, -1, vcc\n" \
        "s_andn2_b64 s[64:65], s[64:65], s[68:69]\n" \
        "ds_write_b128 v15, v[20:23] offset:16\n" \
        FINISH_EXTRA \
        "v_mov_b32_e32 v15, -1\n" \
        "s_mov_b64 exec, -1\n" \
        "TP_FILL_%=:\n" \
        "s_cmp_ge_i32 s70, %[rays]\n" \
        "s_cbranch_scc1 TP_EMPTY_%=\n" \
        "s_bcnt1_i32_b64 s71, s[64:65]\n" \
        "s_cmp_gt_u32 s71, %[maxbusy]\n" \
        "s_cbranch_scc1 TP_VOTE_%=\n" \
        "s_not_b64 s[66:67], s[64:65]\n" \
        "v_mbcnt_lo_u32_b32 v33, s66, 0\n" \
        "v_mbcnt_hi_u32_b32 v33, s67, v33\n" \
        "v_add_u32_e32 v33, s70, v33\n" \
        "v_cmp_gt_i32_e32 vcc, %[rays], v33\n" \
        "s_and_b64 s[66:67], vcc, s[66:67]\n" \
        "s_sub_i32 s71, 64, s71\n" \
        "s_add_i32 s70, s70, s71\n" \
        "s_mov_b64 exec, s[66:67]\n" \
        FETCH_SLOT \
        "ds_read_b128 v[4:7], v15\n" \
        "ds_read_b128 v[8:11], v15 offset:16\n" \
        CURSOR_EARLY \
        "v_mov_b32_e32 v13, 0\n" \
        "v_mov_b32_e32 v14, -1\n" \
        "v_mov_b32_e32 v20, -1\n" \
        "v_mov_b32_e32 v22, 0\n" \
        "v_mov_b32_e32 v23, 0\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        CURSOR_LATE \
        "v_and_b32_e32 v33, 0xff, v11\n" \
        "v_lshl_add_u32 v33, v33, 4, %[pool]\n" \
        "ds_read_b96 v[0:2], v33 offset:%[org]\n" \
        "v_mov_b32_e32 v21, v7\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "s_mov_b64 exec, -1\n" \
        "s_branch TP_LOOP_%=\n" \
        "TP_EMPTY_%=:\n" \
        DRY_POOL \
        "TP_DONE_%=:\n" \
        EXIT_EXTRA \
        "s_waitcnt lgkmcnt(0)\n" \
        "s_mov_b64 exec, -1\n" \
        : \
        : [pool] "s"(s_pool), [rays] "s"(s_rays), [end] "s"(s_end), [first] "s"(s_first), [bias] "s"(s_bias), \
          [eps] "s"(s_eps), __VA_ARGS__, [maxbusy] "n"(64 - kFetchThreshold), [org] "n"(2 * kPoolSlots * 16) \
        : "memory", "vcc", "scc", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", \
          "s72", "s76", "s77", MORE_CLOBBERS "v0", "v1", "v2", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", \
          "v15", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", \
          "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43");

__device__ __forceinline__ void trace_pool_lds_asm(unsigned pool_lds, int n_rays, const LdsScene mem, float eps)
{
    const unsigned s_pool = __builtin_amdgcn_readfirstlane(pool_lds);
    const int s_rays = __builtin_amdgcn_readfirstlane(n_rays);
    const int s_end = __builtin_amdgcn_readfirstlane(mem.end);
    const int s_first = __builtin_amdgcn_readfirstlane(mem.first);
    const int s_bias = __builtin_amdgcn_readfirstlane(mem.tri_bias);
    const unsigned s_eps = __builtin_amdgcn_readfirstlane(__float_as_uint(eps));
    PT_TRACE_ASM("ds_read_b128 v[24:27], v12\n" "ds_read_b128 v[28:31], v12 offset:16\n", "s_waitcnt lgkmcnt(1)\n", "s_waitcnt lgkmcnt(0)\n", "", "",
                 "ds_read_b128 v[28:31], v13 offset:16\n" "ds_read_b32 v32, v13 offset:32\n" "ds_read_b128 v[24:27], v13\n",
                 "s_waitcnt lgkmcnt(1)\n", "s_waitcnt lgkmcnt(0)\n",
                 "s_lshl_b32 s71, s71, " PT_STR(PT_VOTE_NODE_SHIFT) "\n",
                 PT_ENTRY_IDLE, PT_FETCH_COMPACT, PT_CURSOR_FIRST, "", "%[end]", "", PT_DRY_DRAIN, "", , [unused] "n"(0))
}

// Scenes in global memory: cursors are byte offsets from the node / triangle arrays, loads use the SGPR-base +
// 32-bit VGPR-offset form.  (vmcnt also counts this wave's earlier sample stores; they are long gone.)
__device__ __forceinline__ void trace_pool_global_asm(unsigned pool_lds, int n_rays, const GlobalScene mem, float eps, bool may_stop)
{
    const unsigned s_pool = __builtin_amdgcn_readfirstlane(pool_lds);
    const int s_rays = __builtin_amdgcn_readfirstlane(n_rays);
    const int s_end = __builtin_amdgcn_readfirstlane(mem.end - mem.first);      // bytes of one node array
    const int s_first = 0, s_bias = 0;
    const unsigned s_eps = __builtin_amdgcn_readfirstlane(__float_as_uint(eps));
    const unsigned long long s_nodes = uniform64((unsigned long long)mem.nodes), s_tris = uniform64((unsigned long long)mem.tris);
    const unsigned s_order = s_pool + kOrderOff * 16, s_pend = s_pool + kPendOff * 16, s_susp = s_pool + kSuspOff * 16;
    const int s_allow = __builtin_amdgcn_readfirstlane(may_stop ? 1 : 0);
    const int s_tstop = __builtin_amdgcn_readfirstlane(mem.end - mem.first >= 65536 * 32 ? PT_STOP_T : PT_STOP_T_SMALL);
    PT_TRACE_ASM("global_load_dwordx4 v[24:27], v12, %[nodes]\n" "global_load_dwordx4 v[28:31], v12, %[nodes] offset:16\n"
                 "global_load_dwordx4 v[44:47], v12, %[nodes] offset:32\n" "global_load_dwordx4 v[48:51], v12, %[nodes] offset:48\n",
                 "s_waitcnt vmcnt(3)\n", "s_waitcnt vmcnt(2)\n", PT_NODE2_LOOKAHEAD,
                 PT_TRI2_LOOKAHEAD("v3"),
                 "global_load_dwordx4 v[28:31], v13, %[tris] offset:16\n" "global_load_dword v32, v13, %[tris] offset:32\n" "global_load_dwordx4 v[24:27], v13, %[tris]\n"
                 "global_load_dwordx4 v[48:51], v13, %[tris] offset:64\n" "global_load_dword v52, v13, %[tris] offset:80\n" "global_load_dwordx4 v[44:47], v13, %[tris] offset:48\n",
                 "s_waitcnt vmcnt(4)\n", "s_waitcnt vmcnt(3)\n", PT_GLOBAL_VOTE_WEIGHT,
                 PT_ENTRY_RESUME, PT_FETCH_ORDERED, "", PT_CURSOR_VARIANT, "v3", PT_FINISH_PENDING, PT_DRY_MAY_STOP, PT_EXIT_SUSPEND,
                 "v3" PT_COMMA "v44" PT_COMMA "v45" PT_COMMA "v46" PT_COMMA "v47" PT_COMMA "v48" PT_COMMA "v49" PT_COMMA "v50" PT_COMMA "v51" PT_COMMA "v52" PT_COMMA,
                 [nodes] "s"(s_nodes), [tris] "s"(s_tris), [order] "s"(s_order), [pend] "s"(s_pend), [susp] "s"(s_susp),
                 [allow] "s"(s_allow), [tstop] "s"(s_tstop))
}

// ---- trace_pool_wide<>, hand-scheduled -------------------------------------------------------------------------------------
// The instruction-for-instruction twin of trace_pool_wide<> above (which stays the specification and runs in the counting
// build), for the same reason trace_pool_lds_asm exists.  Floating-point instructions and their order are those of the C++
// twin (the box test and the triangle test are the blocks of PT_TRACE_ASM), so the films are bit-identical;
// tests/test_gpu_parity.py runs both.
//
// One trip serves every busy lane, whatever it is working on: the fetches of the lanes at a wide node (7 x dwordx4: one
// 128-byte record per lane) and of the lanes at a leaf (3 loads: one triangle) go out together, then the node block and
// the triangle block run under their own lane masks - on the SAME registers: a lane is at one or at the other.
//
// Register map (all clobbered):
//   v[0:2] origin  v[4:6] dir  v7 tmax as loaded  v[8:10] 1/dir  v11 tag (owner | any-hit << 8)
//   v12 current entry (wide node: byte offset; leaf: bit 31 | triangles after the first << 27 | first triangle; -1: none)
//   v13 stack size   v14 end of the ray's interval   v15 LDS address of the ray's slot (-1: idle lane)
//   v16 byte offset of the lane's column of the wave's spill slice   v17 LDS address of the lane's suspend record
//   v18 LDS address of the lane's stack column - 768 (level l at v18 + 768 + 256 l)
//   v[20:23] best hit {triangle index or -1, t, b1, b2}
//   lanes at a wide node: v[24:47] six planes of four boxes, worked on in place (child k: v24+k, v28+k, ... v44+k; its key ends in v24+k),
//                         v[48:51] the children's entries, v52 v53 temporaries; after the sort v[28:30] (push addresses)
//   lanes at a leaf:      v[24:32] the triangle record, v[44:52] the record after it (1), v[33:43] temporaries (as in PT_TRACE_ASM),
//                         v53 the triangle's index, v54 its byte offset
//   s[60:61] lanes at a leaf  s[62:63] lanes at a wide node  s[64:65] lanes with a ray / busy lanes  s[66:69],s[72:75] scratch masks
//   s70 next ray  s71 s72 counts  s76 1e-8f  s77 2^100  s[78:79] lanes that pop  s[80:81] node lanes with a hit child
#ifndef PT_WIDE_ASM
#define PT_WIDE_ASM 1
#endif
// cache-policy bits of the loop's accesses to the wave's spill slice.  None: a wave reads only what the same wave wrote, which its CU's
// write-through L1 keeps coherent, and the lines stay in L2 (" sc0 sc1" - system scope - wrote every suspend record through to HBM:
// 24 GB per launch of the c5 stand-in against 1 GB of sample planes)
#ifndef PT_WIDE_SC
#define PT_WIDE_SC ""
#endif
// measurement only (does the suspend records' write traffic cost time?): 1 writes every record a second time, to the
// slice's last stack levels - twice the records' traffic, nothing else changed (profiles/r05/c1_write_traffic.log)
#ifndef PT_WIDE_SUSP_MIRROR
#define PT_WIDE_SUSP_MIRROR 0
#endif
#if PT_WIDE_SUSP_MIRROR
#define PT_WIDE_SUSP_MIRROR_TEXT "global_store_dwordx4 v17, v[12:15], %[mirror]" PT_WIDE_SC "\n" "global_store_dwordx4 v17, v[20:23], %[mirror] offset:16" PT_WIDE_SC "\n"
#define PT_WIDE_SUSP_MIRROR_ARG [mirror] "s"(s_mirror),
#else
#define PT_WIDE_SUSP_MIRROR_TEXT
#define PT_WIDE_SUSP_MIRROR_ARG
#endif
// a lane's suspend record lives in the wave's slice of P.wide_stack (the hand-scheduled loop reads it when a drain starts and
// writes it when the drain ends, both with sc0 sc1: the same wave reads what it wrote, past its L1); "no ray" at kernel start
__device__ __forceinline__ void wide_init_suspend_record(const DevParams &P, unsigned lane)
{
    volatile unsigned *rec = P.wide_stack + (size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)kWideWaveSliceDwords + 8u * lane;
    rec[0] = 0xffffffffu; rec[1] = 0u; rec[2] = 0u; rec[3] = 0xffffffffu;
    rec[4] = 0xffffffffu; rec[5] = 0u; rec[6] = 0u; rec[7] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

__device__ __forceinline__ void trace_pool_wide_asm(unsigned pool_lds, int n_rays, const DevParams &P, unsigned lane, bool may_stop)
{
    const unsigned s_pool = __builtin_amdgcn_readfirstlane(pool_lds);
    const int s_rays = __builtin_amdgcn_readfirstlane(n_rays);
    const unsigned s_eps = __builtin_amdgcn_readfirstlane(__float_as_uint(P.eps));
    const unsigned long long s_nodes = uniform64((unsigned long long)P.wide);
    const unsigned s_trioff = __builtin_amdgcn_readfirstlane(P.wide_tris_off);
    const unsigned long long s_spill = uniform64((unsigned long long)P.wide_stack);
#if PT_WIDE_SUSP_MIRROR
    const unsigned long long s_mirror = s_spill + 4ull * (kWideWaveSliceDwords - 512);     // the slice's last, never used stack levels
#endif
    const unsigned s_order = s_pool + kOrderOff * 16, s_pend = s_pool + kPendOff * 16;
    const unsigned s_stack = s_pool + kWideStackOff * 16 - 768;
    const int s_allow = __builtin_amdgcn_readfirstlane(may_stop ? 1 : 0);
    // expensive materials want fewer, fuller shading rounds; long rays in deep trees shorter tails (as in trace_pool_global_asm)
    const int s_tstop = __builtin_amdgcn_readfirstlane(P.n_nodes >= 65536 ? PT_WIDE_STOP_T : PT_WIDE_STOP_T_SMALL);
    // this lane's column of the wave's spill slice, in bytes (level l at + 256 l)
    const unsigned wave_slice = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)kWideWaveSliceDwords;
    const unsigned v_spill = (wave_slice + 512u + lane) * 4u;
    const unsigned v_susp = (wave_slice + 8u * lane) * 4u;      // this lane's suspend record (the records lead the slice: they are its hot part)
    asm volatile(
        "s_mov_b32 s70, 0\n"
        "s_mov_b32 s76, 0x322bcc77\n"
        "s_mov_b32 s77, 0x71800000\n"
        "s_mov_b64 s[64:65], 0\n"
        "s_mov_b64 s[82:83], 0\n"
        "v_mbcnt_lo_u32_b32 v33, -1, 0\n"
        "v_mbcnt_hi_u32_b32 v33, -1, v33\n"                /* lane */
        "v_lshl_add_u32 v18, v33, 2, %[stack]\n"
        "v_mov_b32_e32 v17, %[vsusp]\n"
        "v_mov_b32_e32 v16, %[vspill]\n"
        /* every lane resumes the ray it was walking when the last drain stopped (its record: {entry, stack size, end of the
           interval, slot} {best hit}); direction and origin come back from the ray's slot */
        "global_load_dwordx4 v[12:15], v17, %[spill]" PT_WIDE_SC "\n"
        "global_load_dwordx4 v[20:23], v17, %[spill] offset:16" PT_WIDE_SC "\n"
        "s_waitcnt vmcnt(0)\n"
        "v_cmp_lt_i32_e64 s[64:65], -1, v15\n"
        "s_mov_b64 exec, s[64:65]\n"
        "ds_read_b128 v[4:7], v15\n"
        "ds_read_b128 v[8:11], v15 offset:16\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32_e32 v33, 0xff, v11\n"
        "v_lshl_add_u32 v33, v33, 4, %[pool]\n"
        "ds_read_b96 v[0:2], v33 offset:%[org]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, -1\n"
        "s_branch TW_FILL_%=\n"
        /* ---------------------------------------------------------------- loop header */
        "TW_LOOP_%=:\n"
        "v_cmp_lt_i32_e64 s[64:65], -1, v15\n"             /* lanes with a ray */
        "v_cmp_eq_u32_e64 s[66:67], -1, v12\n"             /* ... that is finished */
        "s_and_b64 s[68:69], s[64:65], s[66:67]\n"
        "s_cbranch_scc1 TW_FIN_%=\n"
        "TW_TRIP_%=:\n"
        "v_cmp_gt_i32_e64 s[60:61], 0, v12\n"
        "s_and_b64 s[60:61], s[60:61], s[64:65]\n"         /* at a leaf (bit 31 set, not -1: busy lanes only) */
        "s_andn2_b64 s[62:63], s[64:65], s[60:61]\n"       /* at a wide node */
        /* too few lanes at a leaf: they wait (unless nobody is at a wide node); else too few at a wide node: those wait */
        "s_bcnt1_i32_b64 s71, s[60:61]\n"
        "s_bcnt1_i32_b64 s72, s[62:63]\n"
        "s_cmp_ge_u32 s71, %[leafmin]\n"
        "s_cbranch_scc1 TW_VOTE_NODE_%=\n"
        "s_cmp_eq_u32 s72, 0\n"
        "s_cbranch_scc1 TW_VOTED_%=\n"
        "s_mov_b64 s[60:61], 0\n"
        "s_branch TW_VOTED_%=\n"
        "TW_VOTE_NODE_%=:\n"
        "s_cmp_ge_u32 s72, %[nodemin]\n"
        "s_cbranch_scc1 TW_VOTED_%=\n"
        "s_mov_b64 s[62:63], 0\n"
        "TW_VOTED_%=:\n"
        /* ---- fetches of both kinds (a vector-memory instruction whose exec is empty is not counted by vmcnt: both blocks
           wait for everything) */
        /* one set of fetches serves both kinds: a lane's offset from the base of the wide nodes is its node's, or that of its
           triangle in the copy behind the nodes; a node lane gets its 112-byte record, a leaf lane its triangle in v[24:32] and the
           one after it in v[36:44] (the rest of what it reads is not used) */
        "s_mov_b64 exec, s[60:61]\n"
        "v_and_b32_e32 v53, 0x7ffffff, v12\n"              /* the leaf's first triangle */
        "v_lshlrev_b32_e32 v54, 4, v53\n"
        "v_lshl_add_u32 v54, v53, 5, v54\n"                /* * 48 */
        "v_add_u32_e32 v54, %[trioff], v54\n"
        "s_mov_b64 exec, s[62:63]\n"
        "v_mov_b32_e32 v54, v12\n"
        "s_or_b64 exec, s[60:61], s[62:63]\n"
        "global_load_dwordx4 v[24:27], v54, %[nodes]\n"
        "global_load_dwordx4 v[36:39], v54, %[nodes] offset:48\n"
        "global_load_dwordx4 v[28:31], v54, %[nodes] offset:16\n"
        "global_load_dwordx4 v[40:43], v54, %[nodes] offset:64\n"
        "global_load_dwordx4 v[32:35], v54, %[nodes] offset:32\n"
        "global_load_dwordx4 v[44:47], v54, %[nodes] offset:80\n"
        "global_load_dwordx4 v[48:51], v54, %[nodes] offset:96\n"
        "s_mov_b64 s[78:79], 0\n"
        "s_mov_b64 s[82:83], 0\n"
        "s_mov_b64 exec, s[62:63]\n"
        "s_cbranch_execz TW_LEAF_%=\n"
        "s_waitcnt vmcnt(0)\n"
        /* ---------------------------------------------------------------- wide node: four boxes (exec = s[62:63]) */
        /* child k's six plane values are worked on in place (v24+k, v28+k, ... v44+k); v52 is the one temporary */
        "v_sub_f32_e32 v24, v24, v0\n"
        "v_sub_f32_e32 v36, v36, v0\n"
        "v_sub_f32_e32 v28, v28, v1\n"
        "v_sub_f32_e32 v32, v32, v2\n"
        "v_sub_f32_e32 v40, v40, v1\n"
        "v_sub_f32_e32 v44, v44, v2\n"
        "v_mul_f32_e32 v24, v8, v24\n"
        "v_mul_f32_e32 v36, v8, v36\n"
        "v_mul_f32_e32 v28, v9, v28\n"
        "v_mul_f32_e32 v40, v9, v40\n"
        "v_mul_f32_e32 v32, v10, v32\n"
        "v_mul_f32_e32 v44, v10, v44\n"
        "v_min_f32_e32 v52, v24, v36\n"
        "v_max_f32_e32 v24, v24, v36\n"
        "v_min_f32_e32 v36, v28, v40\n"
        "v_max_f32_e32 v28, v28, v40\n"
        "v_min_f32_e32 v40, v32, v44\n"
        "v_max_f32_e32 v32, v32, v44\n"
        "v_min3_f32 v24, v24, v28, v32\n"
        "v_max3_f32 v52, v52, v36, v40\n"
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v24\n"
        "v_min_f32_e32 v24, v24, v14\n"
        "v_cmp_nlt_f32_e64 s[66:67], v24, v52\n"
        "v_max_f32_e32 v52, 0, v52\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_ne_u32_e32 vcc, -1, v48\n"
        "v_and_or_b32 v52, v52, -4, 0\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cndmask_b32_e64 v24, -1, v52, s[66:67]\n"
        "v_sub_f32_e32 v25, v25, v0\n"
        "v_sub_f32_e32 v37, v37, v0\n"
        "v_sub_f32_e32 v29, v29, v1\n"
        "v_sub_f32_e32 v33, v33, v2\n"
        "v_sub_f32_e32 v41, v41, v1\n"
        "v_sub_f32_e32 v45, v45, v2\n"
        "v_mul_f32_e32 v25, v8, v25\n"
        "v_mul_f32_e32 v37, v8, v37\n"
        "v_mul_f32_e32 v29, v9, v29\n"
        "v_mul_f32_e32 v41, v9, v41\n"
        "v_mul_f32_e32 v33, v10, v33\n"
        "v_mul_f32_e32 v45, v10, v45\n"
        "v_min_f32_e32 v52, v25, v37\n"
        "v_max_f32_e32 v25, v25, v37\n"
        "v_min_f32_e32 v37, v29, v41\n"
        "v_max_f32_e32 v29, v29, v41\n"
        "v_min_f32_e32 v41, v33, v45\n"
        "v_max_f32_e32 v33, v33, v45\n"
        "v_min3_f32 v25, v25, v29, v33\n"
        "v_max3_f32 v52, v52, v37, v41\n"
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v25\n"
        "v_min_f32_e32 v25, v25, v14\n"
        "v_cmp_nlt_f32_e64 s[66:67], v25, v52\n"
        "v_max_f32_e32 v52, 0, v52\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_ne_u32_e32 vcc, -1, v49\n"
        "v_and_or_b32 v52, v52, -4, 1\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cndmask_b32_e64 v25, -1, v52, s[66:67]\n"
        "v_sub_f32_e32 v26, v26, v0\n"
        "v_sub_f32_e32 v38, v38, v0\n"
        "v_sub_f32_e32 v30, v30, v1\n"
        "v_sub_f32_e32 v34, v34, v2\n"
        "v_sub_f32_e32 v42, v42, v1\n"
        "v_sub_f32_e32 v46, v46, v2\n"
        "v_mul_f32_e32 v26, v8, v26\n"
        "v_mul_f32_e32 v38, v8, v38\n"
        "v_mul_f32_e32 v30, v9, v30\n"
        "v_mul_f32_e32 v42, v9, v42\n"
        "v_mul_f32_e32 v34, v10, v34\n"
        "v_mul_f32_e32 v46, v10, v46\n"
        "v_min_f32_e32 v52, v26, v38\n"
        "v_max_f32_e32 v26, v26, v38\n"
        "v_min_f32_e32 v38, v30, v42\n"
        "v_max_f32_e32 v30, v30, v42\n"
        "v_min_f32_e32 v42, v34, v46\n"
        "v_max_f32_e32 v34, v34, v46\n"
        "v_min3_f32 v26, v26, v30, v34\n"
        "v_max3_f32 v52, v52, v38, v42\n"
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v26\n"
        "v_min_f32_e32 v26, v26, v14\n"
        "v_cmp_nlt_f32_e64 s[66:67], v26, v52\n"
        "v_max_f32_e32 v52, 0, v52\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_ne_u32_e32 vcc, -1, v50\n"
        "v_and_or_b32 v52, v52, -4, 2\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cndmask_b32_e64 v26, -1, v52, s[66:67]\n"
        "v_sub_f32_e32 v27, v27, v0\n"
        "v_sub_f32_e32 v39, v39, v0\n"
        "v_sub_f32_e32 v31, v31, v1\n"
        "v_sub_f32_e32 v35, v35, v2\n"
        "v_sub_f32_e32 v43, v43, v1\n"
        "v_sub_f32_e32 v47, v47, v2\n"
        "v_mul_f32_e32 v27, v8, v27\n"
        "v_mul_f32_e32 v39, v8, v39\n"
        "v_mul_f32_e32 v31, v9, v31\n"
        "v_mul_f32_e32 v43, v9, v43\n"
        "v_mul_f32_e32 v35, v10, v35\n"
        "v_mul_f32_e32 v47, v10, v47\n"
        "v_min_f32_e32 v52, v27, v39\n"
        "v_max_f32_e32 v27, v27, v39\n"
        "v_min_f32_e32 v39, v31, v43\n"
        "v_max_f32_e32 v31, v31, v43\n"
        "v_min_f32_e32 v43, v35, v47\n"
        "v_max_f32_e32 v35, v35, v47\n"
        "v_min3_f32 v27, v27, v31, v35\n"
        "v_max3_f32 v52, v52, v39, v43\n"
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v27\n"
        "v_min_f32_e32 v27, v27, v14\n"
        "v_cmp_nlt_f32_e64 s[66:67], v27, v52\n"
        "v_max_f32_e32 v52, 0, v52\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_ne_u32_e32 vcc, -1, v51\n"
        "v_and_or_b32 v52, v52, -4, 3\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cndmask_b32_e64 v27, -1, v52, s[66:67]\n"
        /* five-exchange sorting network on (key, entry); registers are renamed from exchange to exchange (a child's key sits in the
           register its lo.x plane value came in) */
        "v_cmp_lt_u32_e32 vcc, v25, v24\n"
        "v_min_u32_e32 v52, v24, v25\n"
        "v_max_u32_e32 v25, v24, v25\n"
        "v_cndmask_b32_e32 v53, v48, v49, vcc\n"
        "v_cndmask_b32_e32 v49, v49, v48, vcc\n"
        "v_cmp_lt_u32_e32 vcc, v27, v26\n"
        "v_min_u32_e32 v24, v26, v27\n"
        "v_max_u32_e32 v27, v26, v27\n"
        "v_cndmask_b32_e32 v48, v50, v51, vcc\n"
        "v_cndmask_b32_e32 v51, v51, v50, vcc\n"
        "v_cmp_lt_u32_e32 vcc, v24, v52\n"
        "v_min_u32_e32 v26, v52, v24\n"
        "v_max_u32_e32 v24, v52, v24\n"
        "v_cndmask_b32_e32 v50, v53, v48, vcc\n"
        "v_cndmask_b32_e32 v48, v48, v53, vcc\n"
        "v_cmp_lt_u32_e32 vcc, v27, v25\n"
        "v_min_u32_e32 v52, v25, v27\n"
        "v_max_u32_e32 v27, v25, v27\n"
        "v_cndmask_b32_e32 v53, v49, v51, vcc\n"
        "v_cndmask_b32_e32 v51, v51, v49, vcc\n"
        "v_cmp_lt_u32_e32 vcc, v24, v52\n"
        "v_min_u32_e32 v25, v52, v24\n"
        "v_max_u32_e32 v24, v52, v24\n"
        "v_cndmask_b32_e32 v49, v53, v48, vcc\n"
        "v_cndmask_b32_e32 v48, v48, v53, vcc\n"
        /* sorted: keys v26 <= v25 <= v24 <= v27, entries v50, v49, v48, v51 (a child that is not hit: key -1, sorts last) */
        "v_cmp_ne_u32_e64 s[66:67], -1, v25\n"
        "v_cmp_ne_u32_e64 s[68:69], -1, v24\n"
        "v_cmp_ne_u32_e64 s[72:73], -1, v27\n"
        "v_cmp_ne_u32_e64 s[80:81], -1, v26\n"             /* the node has a hit child */
        "s_nop 0\n"
        "v_addc_co_u32_e64 v28, s[74:75], v13, 0, s[66:67]\n"
        "v_addc_co_u32_e64 v28, s[74:75], v28, 0, s[68:69]\n"
        "v_addc_co_u32_e64 v28, s[74:75], v28, 0, s[72:73]\n"      /* the new stack size: the nearest child is not pushed */
        /* the others are pushed farthest first: sorted child j ends at level size' - j */
        "v_cmp_lt_u32_e64 s[74:75], %[depth], v28\n"
        "v_lshl_add_u32 v29, v28, 8, v18\n"                /* address of level size' - 3 */
        "s_cmp_lg_u64 s[74:75], 0\n"
        "s_cbranch_scc1 TW_PUSH_SLOW_%=\n"
        "s_mov_b64 exec, s[72:73]\n"
        "ds_write_b32 v29, v51\n"
        "s_mov_b64 exec, s[68:69]\n"
        "ds_write_b32 v29, v48 offset:256\n"
        "s_mov_b64 exec, s[66:67]\n"
        "ds_write_b32 v29, v49 offset:512\n"
        "TW_PUSHED_%=:\n"
        "s_mov_b64 exec, s[62:63]\n"
        "v_cndmask_b32_e64 v13, v13, v28, s[80:81]\n"
        "v_cndmask_b32_e64 v12, v12, v50, s[80:81]\n"      /* the nearest hit child is the current entry */
        "s_andn2_b64 s[78:79], s[62:63], s[80:81]\n"       /* the lanes without a hit child pop */
        /* ---------------------------------------------------------------- leaf: one triangle (exec = s[60:61]) */
        "s_branch TW_LEAF_GO_%=\n"
        "TW_LEAF_%=:\n"                                     /* no lane at a wide node: the triangle fetches have not been waited for */
        "s_waitcnt vmcnt(0)\n"
        "TW_LEAF_GO_%=:\n"
        "s_mov_b64 exec, s[60:61]\n"
        "s_cbranch_execz TW_POP_%=\n"
        "v_mul_f32_e32 v33, v5, v32\n"
        "v_mul_f32_e32 v50, v6, v31\n"
        "v_sub_f32_e32 v33, v33, v50\n"
        "v_mul_f32_e32 v34, v6, v30\n"
        "v_mul_f32_e32 v50, v4, v32\n"
        "v_sub_f32_e32 v34, v34, v50\n"
        "v_mul_f32_e32 v35, v4, v31\n"
        "v_mul_f32_e32 v50, v5, v30\n"
        "v_sub_f32_e32 v35, v35, v50\n"
        "v_mul_f32_e32 v45, v33, v27\n"
        "v_mul_f32_e32 v50, v34, v28\n"
        "v_add_f32_e32 v45, v45, v50\n"
        "v_mul_f32_e32 v50, v35, v29\n"
        "v_add_f32_e32 v45, v45, v50\n"
        "v_rcp_f32_e32 v47, v45\n"
        "v_sub_f32_e32 v24, v0, v24\n"
        "v_sub_f32_e32 v25, v1, v25\n"
        "v_sub_f32_e32 v26, v2, v26\n"
        "v_cmp_nle_f32_e64 s[66:67], abs(v45), s77\n"
        "v_fma_f32 v49, -v45, v47, 1.0\n"
        "v_fma_f32 v46, v49, v47, v47\n"
        "s_cmp_lg_u64 s[66:67], 0\n"
        "s_cbranch_scc1 TW_DIV_IEEE_%=\n"
        "TW_DIV_DONE_%=:\n"
        "v_mul_f32_e32 v51, v24, v33\n"
        "v_mul_f32_e32 v50, v25, v34\n"
        "v_add_f32_e32 v51, v51, v50\n"
        "v_mul_f32_e32 v50, v26, v35\n"
        "v_add_f32_e32 v51, v51, v50\n"
        "v_mul_f32_e32 v33, v25, v29\n"
        "v_mul_f32_e32 v50, v26, v28\n"
        "v_sub_f32_e32 v33, v33, v50\n"
        "v_mul_f32_e32 v34, v26, v27\n"
        "v_mul_f32_e32 v50, v24, v29\n"
        "v_sub_f32_e32 v34, v34, v50\n"
        "v_mul_f32_e32 v35, v24, v28\n"
        "v_mul_f32_e32 v50, v25, v27\n"
        "v_sub_f32_e32 v35, v35, v50\n"
        "v_mul_f32_e32 v51, v51, v46\n"                    /* b1 */
        "v_mul_f32_e32 v47, v4, v33\n"
        "v_mul_f32_e32 v50, v5, v34\n"
        "v_add_f32_e32 v47, v47, v50\n"
        "v_mul_f32_e32 v50, v6, v35\n"
        "v_add_f32_e32 v47, v47, v50\n"
        "v_mul_f32_e32 v47, v47, v46\n"                    /* b2 */
        "v_cmp_nlt_f32_e64 s[66:67], abs(v45), s76\n"
        "v_cmp_ngt_f32_e32 vcc, 0, v51\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_nlt_f32_e32 vcc, 1.0, v51\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_ngt_f32_e32 vcc, 0, v47\n"
        "v_add_f32_e32 v50, v51, v47\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_nlt_f32_e32 vcc, 1.0, v50\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "s_and_b64 exec, exec, s[66:67]\n"
        "s_cbranch_scc0 TW_TRI_END_%=\n"
        "v_mul_f32_e32 v48, v30, v33\n"
        "v_mul_f32_e32 v50, v31, v34\n"
        "v_add_f32_e32 v48, v48, v50\n"
        "v_mul_f32_e32 v50, v32, v35\n"
        "v_add_f32_e32 v48, v48, v50\n"
        "v_mul_f32_e32 v48, v48, v46\n"                    /* tt */
        "v_cmp_ngt_f32_e32 vcc, %[eps], v48\n"
        "v_cmp_ngt_f32_e64 s[66:67], v48, v14\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "s_and_b64 exec, exec, s[66:67]\n"
        "s_cbranch_scc0 TW_TRI_END_%=\n"
        /* accepted (exec).  It replaces the best hit when it is nearer, or exactly as near with a larger triangle index; a
           distance that is not NaN and nearer than the interval's end becomes the interval's end */
        "v_cmp_gt_i32_e64 s[68:69], 0, v20\n"
        "v_cmp_lt_f32_e32 vcc, v48, v21\n"
        "s_or_b64 s[68:69], s[68:69], vcc\n"
        "v_cmp_eq_f32_e32 vcc, v48, v21\n"
        "v_cmp_gt_i32_e64 s[72:73], v53, v20\n"
        "s_and_b64 vcc, vcc, s[72:73]\n"
        "s_or_b64 s[68:69], s[68:69], vcc\n"
        "v_cmp_lt_f32_e32 vcc, v48, v14\n"
        "s_mov_b64 s[72:73], exec\n"
        "v_cndmask_b32_e32 v14, v14, v48, vcc\n"
        "v_and_b32_e32 v50, 0x100, v11\n"
        "s_and_b64 exec, exec, s[68:69]\n"
        "v_mov_b32_e32 v20, v53\n"
        "v_mov_b32_e32 v21, v48\n"
        "v_mov_b32_e32 v22, v51\n"
        "v_mov_b32_e32 v23, v47\n"
        "s_mov_b64 exec, s[72:73]\n"
        "v_cmp_ne_u32_e32 vcc, 0, v50\n"                   /* IntersectP: the first accepted triangle ends the ray */
        "v_cndmask_b32_e64 v12, v12, -1, vcc\n"
        "v_cndmask_b32_e64 v13, v13, 0, vcc\n"
        "TW_TRI_END_%=:\n"
        "s_mov_b64 exec, s[60:61]\n"
        /* a ray that goes on and whose leaf has another triangle tests it in the same trip (its record came with the first one's):
           the order of the tests and the interval they see are those of two trips */
        "v_cmp_ne_u32_e32 vcc, -1, v12\n"
        "v_bfe_u32 v50, v12, 27, 4\n"
        "v_cmp_lt_u32_e64 s[66:67], 0, v50\n"
        "s_nop 0\n"
        "s_and_b64 s[84:85], s[66:67], vcc\n"             /* second test */
        "s_andn2_b64 s[68:69], vcc, s[66:67]\n"           /* the leaf is exhausted: pop */
        "s_or_b64 s[78:79], s[78:79], s[68:69]\n"
        "s_mov_b64 exec, s[84:85]\n"
        "s_cbranch_execz TW_POP_%=\n"
        "v_add_u32_e32 v53, 1, v53\n"
        "v_mul_f32_e32 v33, v5, v44\n"
        "v_mul_f32_e32 v50, v6, v43\n"
        "v_sub_f32_e32 v33, v33, v50\n"
        "v_mul_f32_e32 v34, v6, v42\n"
        "v_mul_f32_e32 v50, v4, v44\n"
        "v_sub_f32_e32 v34, v34, v50\n"
        "v_mul_f32_e32 v35, v4, v43\n"
        "v_mul_f32_e32 v50, v5, v42\n"
        "v_sub_f32_e32 v35, v35, v50\n"
        "v_mul_f32_e32 v45, v33, v39\n"
        "v_mul_f32_e32 v50, v34, v40\n"
        "v_add_f32_e32 v45, v45, v50\n"
        "v_mul_f32_e32 v50, v35, v41\n"
        "v_add_f32_e32 v45, v45, v50\n"
        "v_rcp_f32_e32 v47, v45\n"
        "v_sub_f32_e32 v36, v0, v36\n"
        "v_sub_f32_e32 v37, v1, v37\n"
        "v_sub_f32_e32 v38, v2, v38\n"
        "v_cmp_nle_f32_e64 s[66:67], abs(v45), s77\n"
        "v_fma_f32 v49, -v45, v47, 1.0\n"
        "v_fma_f32 v46, v49, v47, v47\n"
        "s_cmp_lg_u64 s[66:67], 0\n"
        "s_cbranch_scc1 TW_DIV_IEEE2_%=\n"
        "TW_DIV_DONE2_%=:\n"
        "v_mul_f32_e32 v51, v36, v33\n"
        "v_mul_f32_e32 v50, v37, v34\n"
        "v_add_f32_e32 v51, v51, v50\n"
        "v_mul_f32_e32 v50, v38, v35\n"
        "v_add_f32_e32 v51, v51, v50\n"
        "v_mul_f32_e32 v33, v37, v41\n"
        "v_mul_f32_e32 v50, v38, v40\n"
        "v_sub_f32_e32 v33, v33, v50\n"
        "v_mul_f32_e32 v34, v38, v39\n"
        "v_mul_f32_e32 v50, v36, v41\n"
        "v_sub_f32_e32 v34, v34, v50\n"
        "v_mul_f32_e32 v35, v36, v40\n"
        "v_mul_f32_e32 v50, v37, v39\n"
        "v_sub_f32_e32 v35, v35, v50\n"
        "v_mul_f32_e32 v51, v51, v46\n"                    /* b1 */
        "v_mul_f32_e32 v47, v4, v33\n"
        "v_mul_f32_e32 v50, v5, v34\n"
        "v_add_f32_e32 v47, v47, v50\n"
        "v_mul_f32_e32 v50, v6, v35\n"
        "v_add_f32_e32 v47, v47, v50\n"
        "v_mul_f32_e32 v47, v47, v46\n"                    /* b2 */
        "v_cmp_nlt_f32_e64 s[66:67], abs(v45), s76\n"
        "v_cmp_ngt_f32_e32 vcc, 0, v51\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_nlt_f32_e32 vcc, 1.0, v51\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_ngt_f32_e32 vcc, 0, v47\n"
        "v_add_f32_e32 v50, v51, v47\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "v_cmp_nlt_f32_e32 vcc, 1.0, v50\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "s_and_b64 exec, exec, s[66:67]\n"
        "s_cbranch_scc0 TW_TRI2_END_%=\n"
        "v_mul_f32_e32 v48, v42, v33\n"
        "v_mul_f32_e32 v50, v43, v34\n"
        "v_add_f32_e32 v48, v48, v50\n"
        "v_mul_f32_e32 v50, v44, v35\n"
        "v_add_f32_e32 v48, v48, v50\n"
        "v_mul_f32_e32 v48, v48, v46\n"                    /* tt */
        "v_cmp_ngt_f32_e32 vcc, %[eps], v48\n"
        "v_cmp_ngt_f32_e64 s[66:67], v48, v14\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "s_and_b64 exec, exec, s[66:67]\n"
        "s_cbranch_scc0 TW_TRI2_END_%=\n"
        /* accepted (exec).  It replaces the best hit when it is nearer, or exactly as near with a larger triangle index; a
           distance that is not NaN and nearer than the interval's end becomes the interval's end */
        "v_cmp_gt_i32_e64 s[68:69], 0, v20\n"
        "v_cmp_lt_f32_e32 vcc, v48, v21\n"
        "s_or_b64 s[68:69], s[68:69], vcc\n"
        "v_cmp_eq_f32_e32 vcc, v48, v21\n"
        "v_cmp_gt_i32_e64 s[72:73], v53, v20\n"
        "s_and_b64 vcc, vcc, s[72:73]\n"
        "s_or_b64 s[68:69], s[68:69], vcc\n"
        "v_cmp_lt_f32_e32 vcc, v48, v14\n"
        "s_mov_b64 s[72:73], exec\n"
        "v_cndmask_b32_e32 v14, v14, v48, vcc\n"
        "v_and_b32_e32 v50, 0x100, v11\n"
        "s_and_b64 exec, exec, s[68:69]\n"
        "v_mov_b32_e32 v20, v53\n"
        "v_mov_b32_e32 v21, v48\n"
        "v_mov_b32_e32 v22, v51\n"
        "v_mov_b32_e32 v23, v47\n"
        "s_mov_b64 exec, s[72:73]\n"
        "v_cmp_ne_u32_e32 vcc, 0, v50\n"                   /* IntersectP: the first accepted triangle ends the ray */
        "v_cndmask_b32_e64 v12, v12, -1, vcc\n"
        "v_cndmask_b32_e64 v13, v13, 0, vcc\n"
        "TW_TRI2_END_%=:\n"
        "s_mov_b64 exec, s[84:85]\n"
        "v_cmp_ne_u32_e32 vcc, -1, v12\n"
        "v_bfe_u32 v50, v12, 27, 4\n"
        "v_add_u32_e32 v49, 0xf0000002, v12\n"             /* first + 2, two triangles fewer */
        "v_cmp_lt_u32_e64 s[66:67], 1, v50\n"
        "s_nop 0\n"
        "s_and_b64 s[66:67], s[66:67], vcc\n"
        "s_andn2_b64 s[68:69], vcc, s[66:67]\n"
        "v_cndmask_b32_e64 v12, v12, v49, s[66:67]\n"
        "s_or_b64 s[78:79], s[78:79], s[68:69]\n"
        /* ---------------------------------------------------------------- pop (s[78:79]) */
        "TW_POP_%=:\n"
        "s_mov_b64 exec, s[78:79]\n"
        "s_cbranch_execz TW_POP_NONE_%=\n"
        "v_cmp_lt_i32_e32 vcc, 0, v13\n"
        "v_mov_b32_e32 v12, -1\n"
        "s_and_b64 exec, exec, vcc\n"
        "s_cbranch_execz TW_POP_NONE_%=\n"
        "v_add_u32_e32 v13, -1, v13\n"
        "v_cmp_gt_i32_e32 vcc, %[depth], v13\n"
        "s_mov_b64 s[72:73], exec\n"
        "s_and_b64 exec, exec, vcc\n"
        "v_lshl_add_u32 v33, v13, 8, v18\n"
        "ds_read_b32 v12, v33 offset:768\n"
        "s_andn2_b64 exec, s[72:73], vcc\n"
        "s_cbranch_execz TW_POP_LDS_%=\n"
        "v_lshl_add_u32 v33, v13, 8, v16\n"
        "global_load_dword v12, v33, %[spill]" PT_WIDE_SC "\n"
        "s_waitcnt vmcnt(0)\n"
        "TW_POP_LDS_%=:\n"
        "s_waitcnt lgkmcnt(0)\n"
        "TW_POP_NONE_%=:\n"
        "s_mov_b64 exec, -1\n"
        "s_branch TW_LOOP_%=\n"
        /* ---------------------------------------------------------------- a push beyond the LDS levels (rare): level by level */
        "TW_PUSH_SLOW_%=:\n"
        "v_add_u32_e32 v29, -3, v28\n"
        "v_cmp_gt_i32_e32 vcc, %[depth], v29\n"
        "s_and_b64 exec, s[72:73], vcc\n"
        "v_lshl_add_u32 v30, v29, 8, v18\n"
        "ds_write_b32 v30, v51 offset:768\n"
        "s_andn2_b64 exec, s[72:73], vcc\n"
        "v_lshl_add_u32 v30, v29, 8, v16\n"
        "global_store_dword v30, v51, %[spill]" PT_WIDE_SC "\n"
        "s_mov_b64 exec, s[62:63]\n"
        "v_add_u32_e32 v29, -2, v28\n"
        "v_cmp_gt_i32_e32 vcc, %[depth], v29\n"
        "s_and_b64 exec, s[68:69], vcc\n"
        "v_lshl_add_u32 v30, v29, 8, v18\n"
        "ds_write_b32 v30, v48 offset:768\n"
        "s_andn2_b64 exec, s[68:69], vcc\n"
        "v_lshl_add_u32 v30, v29, 8, v16\n"
        "global_store_dword v30, v48, %[spill]" PT_WIDE_SC "\n"
        "s_mov_b64 exec, s[62:63]\n"
        "v_add_u32_e32 v29, -1, v28\n"
        "v_cmp_gt_i32_e32 vcc, %[depth], v29\n"
        "s_and_b64 exec, s[66:67], vcc\n"
        "v_lshl_add_u32 v30, v29, 8, v18\n"
        "ds_write_b32 v30, v49 offset:768\n"
        "s_andn2_b64 exec, s[66:67], vcc\n"
        "v_lshl_add_u32 v30, v29, 8, v16\n"
        "global_store_dword v30, v49, %[spill]" PT_WIDE_SC "\n"
        "s_waitcnt vmcnt(0)\n"
        "s_branch TW_PUSHED_%=\n"
        /* ---------------------------------------------------------------- IEEE reciprocal for divisors outside the Newton range */
        "TW_DIV_IEEE_%=:\n"
        "v_div_scale_f32 v46, s[66:67], v45, v45, 1.0\n"
        "v_div_scale_f32 v48, vcc, 1.0, v45, 1.0\n"
        "v_rcp_f32_e32 v47, v46\n"
        "s_nop 0\n"
        "v_fma_f32 v49, -v46, v47, 1.0\n"
        "v_fmac_f32_e32 v47, v49, v47\n"
        "v_mul_f32_e32 v52, v48, v47\n"
        "v_fma_f32 v49, -v46, v52, v48\n"
        "v_fmac_f32_e32 v52, v49, v47\n"
        "v_fma_f32 v46, -v46, v52, v48\n"
        "v_div_fmas_f32 v46, v46, v47, v52\n"
        "v_div_fixup_f32 v46, v46, v45, 1.0\n"
        "s_branch TW_DIV_DONE_%=\n"
        "TW_DIV_IEEE2_%=:\n"
        "v_div_scale_f32 v46, s[66:67], v45, v45, 1.0\n"
        "v_div_scale_f32 v48, vcc, 1.0, v45, 1.0\n"
        "v_rcp_f32_e32 v47, v46\n"
        "s_nop 0\n"
        "v_fma_f32 v49, -v46, v47, 1.0\n"
        "v_fmac_f32_e32 v47, v49, v47\n"
        "v_mul_f32_e32 v52, v48, v47\n"
        "v_fma_f32 v49, -v46, v52, v48\n"
        "v_fmac_f32_e32 v52, v49, v47\n"
        "v_fma_f32 v46, -v46, v52, v48\n"
        "v_div_fmas_f32 v46, v46, v47, v52\n"
        "v_div_fixup_f32 v46, v46, v45, 1.0\n"
        "s_branch TW_DIV_DONE2_%=\n"

        /* ---------------------------------------------------------------- finished rays (s[68:69]) */
        "TW_FIN_%=:\n"
        "s_mov_b64 exec, s[68:69]\n"
        /* a miss reports the end of the interval, like the other loops */
        "v_cmp_gt_i32_e32 vcc, 0, v20\n"
        "v_cndmask_b32_e32 v21, v21, v14, vcc\n"
        "ds_write_b128 v15, v[20:23] offset:16\n"
        PT_FINISH_PENDING
        "v_mov_b32_e32 v15, -1\n"
        "s_andn2_b64 s[64:65], s[64:65], s[68:69]\n"
        "s_mov_b64 exec, -1\n"
        /* ---------------------------------------------------------------- refill: idle lanes take the next rays */
        "TW_FILL_%=:\n"
        "s_cmp_ge_i32 s70, %[rays]\n"
        "s_cbranch_scc1 TW_EMPTY_%=\n"
        "s_bcnt1_i32_b64 s71, s[64:65]\n"
        "s_cmp_gt_u32 s71, %[maxbusy]\n"
        "s_cbranch_scc1 TW_TRIP_%=\n"
        "s_not_b64 s[66:67], s[64:65]\n"
        "v_mbcnt_lo_u32_b32 v33, s66, 0\n"               /* v33: a temporary of the leaf block, dead between trips (v54 is not) */
        "v_mbcnt_hi_u32_b32 v33, s67, v33\n"
        "v_add_u32_e32 v33, s70, v33\n"
        "v_cmp_gt_i32_e32 vcc, %[rays], v33\n"
        "s_and_b64 s[66:67], vcc, s[66:67]\n"
        "s_sub_i32 s71, 64, s71\n"
        "s_add_i32 s70, s70, s71\n"
        "s_mov_b64 exec, s[66:67]\n"
        PT_FETCH_ORDERED
        "ds_read_b128 v[4:7], v15\n"
        "ds_read_b128 v[8:11], v15 offset:16\n"
        "v_mov_b32_e32 v12, 0\n"
        "v_mov_b32_e32 v13, 0\n"
        "v_mov_b32_e32 v20, -1\n"
        "v_mov_b32_e32 v21, 0\n"
        "v_mov_b32_e32 v22, 0\n"
        "v_mov_b32_e32 v23, 0\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32_e32 v33, 0xff, v11\n"
        "v_lshl_add_u32 v33, v33, 4, %[pool]\n"
        "ds_read_b96 v[0:2], v33 offset:%[org]\n"
        "v_mov_b32_e32 v14, v7\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, -1\n"
        "s_branch TW_LOOP_%=\n"
        /* a dry pool: drain to the end, unless this round had new rays and only a few rays are still in flight - then the
           drain ends, the lanes park their rays and the owners of those rays sit out the shading round */
        "TW_EMPTY_%=:\n"
        "s_cmp_eq_u64 s[64:65], 0\n"
        "s_cbranch_scc1 TW_DONE_%=\n"
        "s_cmp_eq_u32 %[allow], 0\n"
        "s_cbranch_scc1 TW_TRIP_%=\n"
        "s_bcnt1_i32_b64 s71, s[64:65]\n"
        "s_cmp_gt_u32 s71, %[tstop]\n"
        "s_cbranch_scc1 TW_TRIP_%=\n"
        "TW_DONE_%=:\n"
        "s_waitcnt vmcnt(0)\n"                             /* (an early fetch must not land after the registers have been handed back) */
        "global_store_dwordx4 v17, v[12:15], %[spill]" PT_WIDE_SC "\n"
        "global_store_dwordx4 v17, v[20:23], %[spill] offset:16" PT_WIDE_SC "\n"
        PT_WIDE_SUSP_MIRROR_TEXT
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_mov_b64 exec, -1\n"
        :
        : [pool] "s"(s_pool), [rays] "s"(s_rays), [eps] "s"(s_eps), [nodes] "s"(s_nodes), [trioff] "s"(s_trioff), [spill] "s"(s_spill),
          [order] "s"(s_order), [pend] "s"(s_pend), [stack] "s"(s_stack), [allow] "s"(s_allow), [vspill] "v"(v_spill), [vsusp] "v"(v_susp),
          PT_WIDE_SUSP_MIRROR_ARG [tstop] "s"(s_tstop), [depth] "n"(kWideStackDepth), [maxbusy] "n"(64 - PT_WIDE_FETCH_T), [org] "n"(2 * kPoolSlots * 16),
          [leafmin] "n"(PT_WIDE_LEAF_MIN), [nodemin] "n"(PT_WIDE_NODE_MIN)
        : "memory", "vcc", "scc", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
          "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85",
          "v0", "v1", "v2", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37",
          "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54");
}


#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4
#endif

// Scenes that fit kSmallSceneFloat4 (nodes 2 float4 each, triangles 3, shading records 5, lights 6, materials
// 72 B) are staged in LDS once per workgroup; traversal and shading then read LDS instead of going through
// the vector-memory pipe.
#ifndef PT_SMALL_FLOAT4
#define PT_SMALL_FLOAT4 GPT_LDS_SCENE_FLOAT4
#endif
#ifndef PT_SMALL_WAVES
#define PT_SMALL_WAVES PT_MIN_WAVES
#endif
constexpr int kSmallSceneFloat4 = PT_SMALL_FLOAT4;      // 12 KB: with the ray pools exactly 4 workgroups per CU

// INTEG: GPT_IT_PT = Path (pathtracer.cu:880-1021), GPT_IT_AO = Ao (:830-876: one cosine-weighted occlusion ray
// of length maxDist per primary hit; the same pools, drain and sample planes), GPT_IT_VPT = Volpath (:1025-1242) for
// homogeneous media in scenes without material-less interface surfaces: a shadow ray is then still one any-hit ray,
// and every transmittance factor is known at shading time or from the hit distance of the BSDF-sampled light ray
// Volpath with density grids or material-less surfaces: an internal fourth value of INTEG (not an integrator type of the
// ABI; launch_render picks it when DevParams.vpt_walk is set) and the stages of its per-path state machine
#define PT_IT_VPT_WALK 8
// 4 waves per SIMD.  (Rounds 1 - 4 ran it at 3, 168 VGPRs: at 4 the if-else chain of stage bodies spilled 270 B per lane.  As one forward
// sweep the stage code needs 156 - 166 registers without scratch, and the fourth wave is worth more than the 124 - 168 B of scratch it
// costs: 212 -> 232 Msamples/s on the shipped scene's shape, profiles/r05/d1_volpath.log.)
#ifndef PT_WALK_WAVES
#define PT_WALK_WAVES 4
#endif
constexpr int kStPath = 0, kStShadow = 1, kStMisStart = 2, kStMis = 3, kStContinue = 4, kStPathB = 5, kStEmit = 6, kStShadowB = 7,
              kStShadowDone = 8, kStMisB = 9;
constexpr int kJobNone = 0, kJobSample = 1, kJobTr = 2;
#ifndef PT_TRACK_STEPS
#define PT_TRACK_STEPS 16
#endif
#ifndef PT_STEP_BATCH
#define PT_STEP_BATCH 8
#endif
#ifndef PT_TRACE_BATCH
#define PT_TRACE_BATCH 32
#endif
// a tracking turn is the most expensive of the three activities (16 steps of ~550 instructions): with fewer than PT_TRACK_MIN walks under way the
// wave rather runs the stage code or drains what it has, which brings lanes back to the walks
#ifndef PT_TRACK_MIN
#define PT_TRACK_MIN 16
#endif
constexpr int kTrackSteps = PT_TRACK_STEPS, kStepBatch = PT_STEP_BATCH, kTraceBatch = PT_TRACE_BATCH, kTrackMin = PT_TRACK_MIN;
// with no walk under way (n_track == 0 < kTrackMin) a single ready lane gets its stage pass and a single ray its drain: that is what
// guarantees progress when fewer than kStepBatch lanes are left.  0 switches both rules off and the wave spins (measured: a hung launch)
static_assert(PT_TRACK_MIN >= 1, "PT_TRACK_MIN = 0 removes the progress guarantee of the one-ray Volpath kernel's scheduler");
#ifndef PT_WIDE_WAVES
#define PT_WIDE_WAVES 4
#endif
#ifndef PT_WALK_PROBE
#define PT_WALK_PROBE 0
#endif
// WIDE: scenes in global memory walked on the 4-wide tree, one lane per ray (GPT_TRAVERSAL_WIDE4, trace_pool_wide<>)
template <bool COUNT, bool SMALL, int INTEG, bool WIDE = false>
__global__ void __launch_bounds__(256, INTEG == PT_IT_VPT_WALK ? PT_WALK_WAVES : (WIDE ? PT_WIDE_WAVES : (SMALL ? PT_SMALL_WAVES : PT_MIN_WAVES))) pt_render_kernel(const DevParams P_in)
{
    static_assert(!(WIDE && SMALL), "the wide tree is walked from global memory");
    __shared__ float4 lds_scene[SMALL ? kSmallSceneFloat4 - (0 ? 8 : 0) : 1];
    DevParams P = P_in;
    if (SMALL) {
        // layout: nodes | triangles | shading records | lights | materials (all 16-byte multiples except the
        // 72-byte materials, which come last)
        const float4 *gn = reinterpret_cast<const float4 *>(P_in.nodes);
        const float4 *gt = reinterpret_cast<const float4 *>(P_in.tris);
        const float4 *gs = reinterpret_cast<const float4 *>(P_in.shade);
        const float4 *gl = reinterpret_cast<const float4 *>(P_in.lights);
        const uint32_t *gm = reinterpret_cast<const uint32_t *>(P_in.materials);
        const int o_tri = 2 * P.n_nodes, o_shade = o_tri + 3 * P.n_prims, o_light = o_shade + 5 * P.n_prims,
                  o_mat = o_light + 6 * P.n_lights;
        const int lds_nodes = (int)lds_address(lds_scene), lds_tris = lds_nodes + o_tri * 16;
        for (int i = threadIdx.x; i < 2 * P.n_nodes; i += 256) {
            float4 v = gn[i];
            if (i & 1) {                                      // links become absolute LDS addresses
                if (__float_as_int(v.w) >= 0) {               // leaf: first / last triangle
                    v.z = __int_as_float(__float_as_int(v.z) + lds_tris);
                    v.w = __int_as_float(__float_as_int(v.w) + lds_tris);
                } else {
                    v.z = __int_as_float(__float_as_int(v.z) + lds_nodes);
                }
            }
            lds_scene[i] = v;
        }
        for (int i = threadIdx.x; i < 3 * P.n_prims; i += 256) lds_scene[o_tri + i] = gt[i];
        for (int i = threadIdx.x; i < 5 * P.n_prims; i += 256) lds_scene[o_shade + i] = gs[i];
        for (int i = threadIdx.x; i < 6 * P.n_lights; i += 256) lds_scene[o_light + i] = gl[i];
        uint32_t *lm = reinterpret_cast<uint32_t *>(lds_scene + o_mat);
        for (int i = threadIdx.x; i < 18 * P.n_materials; i += 256) lm[i] = gm[i];
        __syncthreads();
        P.shade = reinterpret_cast<const DevShade *>(lds_scene + o_shade);
        P.lights = reinterpret_cast<const DevLight *>(lds_scene + o_light);
        P.materials = reinterpret_cast<const gpt_material *>(lds_scene + o_mat);
    }
    constexpr bool CARRY = !SMALL;    // scenes in global memory: fixed slots, drains may stop early
    constexpr int kWaveFloat4 = WIDE ? kWaveWideFloat4 : (CARRY ? kWaveCarryFloat4 : kWaveLdsFloat4);
    __shared__ float4 lds_pool[4 * kWaveFloat4];        // one private ray pool per wavefront
    float4 *pool = lds_pool + (threadIdx.x >> 6) * kWaveFloat4;
    const unsigned lane = threadIdx.x & 63u;
    if (CARRY) reinterpret_cast<unsigned *>(pool + kPendOff)[lane] = 0u;     // nothing pending
    if (CARRY && !WIDE) {                               // no suspended ray (an idle lane's cursors)
        pool[kSuspOff + 2 * lane] = make_float4(__int_as_float(SMALL ? (int)lds_address(lds_scene) + 32 * P.n_nodes : 32 * P.n_nodes), __int_as_float(0), __int_as_float(-1), __int_as_float(-1));
        pool[kSuspOff + 2 * lane + 1] = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f);
    }
    if (WIDE) wide_init_suspend_record(P, lane);        // trace_pool_wide_asm: a lane's record {entry, stack size, interval end, slot} {best hit}: idle
    bool waiting = false;                               // carry: some of this path's rays are still being traced
    // Volpath: the medium the path ray travels in (-1 = none), the one the pending direct-light rays travel in, and
    // whether an occluded light sample poisons the sample (Tr = 0 times a non-finite factor is NaN, pathtracer.cu:1092,1150)
    int medium = -1, medium_ld = -1;
    bool poison_occluded = false;
    const uint32_t n_owned = (P.n_tiles > P.rank) ? (P.n_tiles - P.rank + P.n_ranks - 1) / P.n_ranks : 0u;
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long cyc_trace = 0, cyc_shade = 0, cyc_mark = COUNT ? __builtin_readcyclecounter() : 0ull;   // counting build
    unsigned long long cyc_direct = 0, cyc_hit = 0, cyc_regen = 0, cyc_sub = 0;                                // ... split of cyc_shade
#if PT_WALK_PROBE      // probe build of the one-ray Volpath kernel (tools/gpu_walk_probe.py): passes / lanes / cycles of its three activities
    unsigned long long wp_stage_pass = 0, wp_stage_lanes = 0, wp_track_turn = 0, wp_drains = 0, wp_track_steps = 0, wp_track_lane_steps = 0,
                       wp_cyc_stage = 0, wp_cyc_track = 0;
#endif
#define PT_SUBPHASE(acc)                                                                     \
    if (COUNT) {                                                                             \
        const unsigned long long now_ = __builtin_readcyclecounter();                        \
        if (lane == 0) acc += now_ - cyc_sub;                                                \
        cyc_sub = now_;                                                                      \
    }
    // 0 (probe builds only): marks inside divergent code, booked by whichever lane is first active
    // through a per-wave LDS record {mark, h[0..6]}
#if defined(PT_ISA_MARKS)        // analysis builds: comments in the .s that delimit the shading sub-phases
#define PT_MARK(i) asm volatile("; ISA_MARK " #i);
#else
#define PT_MARK(i)
#endif

    const uint32_t n_items = n_owned * P.n_chunks;
    // ---- persistent scheduler: one atomic per wave per work item.  The item's samples (64 pixels x chunk_count
    // iterations, all independent: each goes to its own slot of its iteration's plane) are handed out dynamically:
    // a lane whose path ends takes the next sample of the tile, whichever pixel it belongs to, and when the item
    // runs out the wave fetches its next item in the same round - it never drains between items, only once, at
    // the end of the launch.  sample s -> pixel s % 64 of the tile, iteration chunk_first + s / 64.
    uint32_t tile_xy = 0, slot_base = 0, chunk_first = 0, n_item_samples = 0, next_sample = 0;     // wave-uniform
    bool more_items = true;
    // Scenes in global memory: one queue per XCD (workgroup b runs on XCD b % 8), each a contiguous eighth of the items, and the
    // items walk down strips of PT_TILE_STRIP tiles - what an XCD's private L2 serves at any time is one compact block of the
    // frame.  A wave whose queue is empty takes from the next XCD's.  The LDS-resident scene has nothing to gain: one queue, tile rows.
    constexpr bool XCD_QUEUES = PT_XCD_QUEUES && !SMALL;
    constexpr uint32_t TILE_STRIP = SMALL ? 0u : (uint32_t)PT_TILE_STRIP;
    uint32_t queues_done = 0;      // queues found empty so far: own XCD's first, then the others in turn
    uint32_t x = 0, y = 0, plane_slot = 0, iter = 0;      // plane_slot: where the sample goes in its iteration's plane
    {
        // ---- per-path state ---------------------------------------------------------
        Rng rng;
        rng.x = 1;
        V3 Li = v3(0.f), beta = v3(1.f);
        bool specular = false;
        int bounces = 0;
        // direct light waiting for its shadow / MIS rays (pathtracer.cu:942-994)
        V3 beta_ld = v3(0.f);      // throughput to apply to Ld
        V3 cand = v3(0.f);         // light-sample term, added if the shadow ray is unoccluded
        V3 mis_fr = v3(0.f);       // BSDF-sample term pieces
        float mis_cos = 0.f, mis_pdf = 1.f;
        bool direct = false;       // a non-delta bounce is waiting for `Li += beta*Ld` (pathtracer.cu:994)
        bool ending = false;       // the path has no continuation; the sample ends once Ld is resolved
        bool alive = false;
        // ---- the one-ray-at-a-time Volpath (PT_IT_VPT_WALK) only: where the path's state machine stands, the surface
        // hit (or scatter point) it is working on, the shadow walk and the pieces of the pending light-sample term
        int stage = 0;
        V3 ctx_o = v3(0.f), ctx_d = v3(0.f);
        float ctx_t = 0.f, ctx_b1 = 0.f, ctx_b2 = 0.f;
        int ctx_prim = -1;
        bool ctx_scatter = false;
        V3 walk_tr = v3(1.f);
        int walk_medium = -1;
        float walk_left = 0.f;
        float sav_w = 0.f, sav_abs = 0.f, sav_den = 1.f;
        V3 sav_fr = v3(0.f), sav_rad = v3(0.f), Ld_acc = v3(0.f);
        bool sav_has = false;
        // ... and its tracking job: a walk through a density grid that did not end within kTrackSteps steps is parked
        // (job_running) and continues in the next round, while the lanes that are done move on
        int job = 0, job_medium = -1;
        float job_tmax = 0.f, trk_dist = 0.f, trk_tr = 1.f;
        int trk_iter = 0;
        bool job_running = false;

        RaySet q;
        q.org = q.dir_s = q.dir_m = q.dir_p = v3(0.f);
        q.tmax_s = 0.f;
        q.has_s = q.has_m = q.has_p = false;
        q.mis_any = false;
        RayResults res;
        res.occluded = false;
        res.prim_m = res.prim_p = -1;
        res.t_m = res.b1_m = res.b2_m = res.t_p = res.b1_p = res.b2_p = 0.f;

        for (;;) {
            bool finish = false;
            if (COUNT) cyc_sub = __builtin_readcyclecounter();
            PT_MARK(0)
            if (INTEG == PT_IT_VPT_WALK) {
                // ---- Volpath, general form (pathtracer.cu:298-322,1025-1242): one ray in flight per path.  Density
                // grids draw random numbers per traced segment and material-less surfaces split a shadow ray into
                // segments, so nothing can be sampled ahead of the ray it depends on: the loop body of the reference
                // is cut at its Intersect() calls and at its medium Sample() / Tr() calls into stages.  A lane steps
                // through them until it has a ray to trace or the sample ends; whenever it needs a medium walked it
                // posts a tracking job, and the jobs of all lanes - whatever their stage - run in ONE shared loop.
                // The ray is (q.org, q.dir_p, q.tmax_s); its closest hit arrives in res.*_p.
                const bool hit = res.prim_p >= 0;
                bool busy = alive && !waiting && !job_running;
                if (busy) q.has_p = false;
                float job_t = 0.f;
                bool job_sampled = false;
                V3 job_v = v3(1.f, 1.f, 1.f);          // the job's answer: Sample()'s weight or Tr()
                for (;;) {
                    // Three things a lane can be waiting for: its stage code (busy: it has a traced ray or a finished
                    // job to consume), a step of its tracking job (job_running) or the pool drain (q.has_p / finish).
                    // Each costs the wave the same whether 1 or 64 lanes take part, so the stage code - the longest of
                    // them, one body per stage present - runs when kStepBatch lanes have gathered or nothing is being
                    // tracked, and the round goes on to the drain when nobody is left to step and kTraceBatch rays wait
                    // (or nothing is being tracked); otherwise the wave tracks.
                    const int n_ready = popc(ballot(busy));
                    int n_track = popc(ballot(job_running));
                    if (n_ready > 0 && (n_ready >= kStepBatch || n_track < kTrackMin)) {
#if PT_WALK_PROBE
                    const unsigned long long wp_t0 = __builtin_readcyclecounter();
                    if (COUNT && lane == 0) { wp_stage_pass++; wp_stage_lanes += (unsigned)n_ready; }
#endif
                    if (busy) {
                        job = kJobNone;
                        // One forward sweep over the stages, each body at most once per pass: a lane only ever moves FORWARD in this order without
                        // tracing a ray or walking a medium in between (path ray -> medium sampled -> emitter | shadow segment -> its medium ->
                        // Tr() complete -> BSDF-sampled light ray -> ... came back -> its medium -> continuation), so the if-else chain in a loop
                        // that this replaces ran the heavy bodies up to four times per pass for lanes that reached them in different iterations.
                        bool go = true;              // still sweeping (false: the lane has a ray, a job or a finished sample)
                        if (go && stage == kStPath) do {
                            // ---- the path ray came back (pathtracer.cu:1049-1070)
                            if (!hit) {
                                if ((bounces == 0 || specular) && P.inf.isvalid)
                                    Li += beta * inf_le(P.inf, q.dir_p);
                                finish = true;
                                busy = false;
                                { go = false; break; }
                            }
                            stage = kStPathB;
                            job_sampled = false;
                            if (medium >= 0) {
                                job = kJobSample;
                                job_medium = medium;
                                job_tmax = res.t_p;
                                { go = false; break; }
                            }
                            break;
                        } while (0);
                        if (go && stage == kStPathB) do {
                            // ---- ... and its medium has been sampled (:1070-1129)
                            if (medium >= 0) beta *= job_v;
                            if (is_black(beta)) {
                                finish = true;
                                busy = false;
                                { go = false; break; }
                            }
                            Ray r;
                            r.o = q.org;
                            r.d = q.dir_p;
                            const Hit isect = make_hit(P, r, res.t_p, res.prim_p, res.b1_p, res.b2_p);
                            const V3 wo = -q.dir_p;
                            if (job_sampled) {
                                // a scattering event inside the medium (:1071-1101): light sample, then its shadow walk
                                const DevMedium &M = P.mediums[medium];
                                float u = rng_uniform(rng);
                                float choicePdf;
                                int idx = lookup_light_distribution(P, u, choicePdf);
                                bool inf = idx == P.n_lights;
                                V3 samplePos = q.org + q.dir_p * job_t;
                                float u1x = rng_uniform(rng);
                                float u1y = rng_uniform(rng);
                                V3 radiance = v3(0.f), lightNor;
                                Ray shadowRay;
                                shadowRay.o = samplePos;
                                shadowRay.d = v3(0.f);
                                shadowRay.tmin = P.eps;
                                shadowRay.tmax = 0.f;
                                float lightPdf = 0.f;
                                if (idx >= 0) {
                                    if (!inf)
                                        area_sample_light(P.lights[idx], samplePos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                                    else
                                        inf_sample_light(P.inf, samplePos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                                }
                                sav_w = medium_phase(M, wo, shadowRay.d);
                                sav_rad = radiance;
                                sav_den = lightPdf * choicePdf;
                                sav_has = !is_black(radiance);
                                ctx_scatter = true;
                                ctx_o = samplePos;
                                walk_tr = v3(1.f, 1.f, 1.f);       // Tr() is walked whatever the radiance is (:1092)
                                walk_medium = medium;
                                walk_left = shadowRay.tmax;
                                q.org = shadowRay.o;
                                q.dir_p = shadowRay.d;
                                q.tmax_s = shadowRay.tmax;
                                q.has_p = true;
                                stage = kStShadow;
                                busy = false;
                                { go = false; break; }
                            }
                            if ((bounces == 0 || specular) && isect.lightIdx != -1) {
                                stage = kStEmit;                                   // :1103-1115
                                job_v = v3(1.f, 1.f, 1.f);
                                if (medium >= 0) {
                                    job = kJobTr;
                                    job_medium = medium;
                                    job_tmax = res.t_p;
                                    { go = false; break; }
                                }
                                break;
                            }
                            if (isect.matIdx == -1) {
                                // a surface without a material only separates two media (:1117-1124): not a bounce
                                medium = dot(q.dir_p, isect.nor) > 0 ? P.prim_media[2 * res.prim_p + 1] : P.prim_media[2 * res.prim_p];
                                q.org = isect.pos;
                                q.tmax_s = __builtin_inff();
                                q.has_p = true;
                                stage = kStPath;
                                busy = false;
                                { go = false; break; }
                            }
                            ctx_scatter = false;
                            ctx_o = q.org;
                            ctx_d = q.dir_p;
                            ctx_t = res.t_p;
                            ctx_prim = res.prim_p;
                            ctx_b1 = res.b1_p;
                            ctx_b2 = res.b2_p;
                            const gpt_material &material = P.materials[isect.matIdx];
                            if (kind_is_delta(material.type)) {
                                stage = kStContinue;
                                break;
                            }
                            // direct light (:1128-1151): the light sample; its shadow ray is walked before anything else is drawn
                            Ld_acc = v3(0.f, 0.f, 0.f);
                            float u = rng_uniform(rng);
                            float choicePdf;
                            int idx = lookup_light_distribution(P, u, choicePdf);
                            bool inf = idx == P.n_lights;
                            float u1x = rng_uniform(rng);
                            float u1y = rng_uniform(rng);
                            V3 radiance = v3(0.f), lightNor;
                            Ray shadowRay;
                            shadowRay.o = isect.pos;
                            shadowRay.d = v3(0.f);
                            shadowRay.tmin = P.eps;
                            shadowRay.tmax = 0.f;
                            float lightPdf = 0.f;
                            if (idx >= 0) {
                                if (!inf)
                                    area_sample_light(P.lights[idx], isect.pos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                                else
                                    inf_sample_light(P.inf, isect.pos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                            }
                            if (is_black(radiance)) {
                                stage = kStMisStart;
                                break;
                            }
                            const Surface S = surface_prepare(P, material, wo, isect.nor, isect.dpdu, isect.uv);
                            const Scatter lit = surface_respond(S, material, shadowRay.d);
                            sav_w = mis_weight(lightPdf * choicePdf, lit.pdf);
                            sav_fr = lit.f;
                            sav_rad = radiance;
                            sav_abs = fabs_(dot(isect.nor, shadowRay.d));
                            sav_den = lightPdf * choicePdf;
                            walk_tr = v3(1.f, 1.f, 1.f);
                            walk_medium = medium;
                            walk_left = shadowRay.tmax;
                            q.org = shadowRay.o;
                            q.dir_p = shadowRay.d;
                            q.tmax_s = shadowRay.tmax;
                            q.has_p = true;
                            stage = kStShadow;
                            busy = false;
                            { go = false; break; }
                        } while (0);
                        if (go && stage == kStEmit) do {
                            // ---- a directly seen emitter, attenuated by the path's medium (:1103-1115)
                            V3 n;
                            int lightIdx;
                            make_light_hit(P, res.prim_p, res.b1_p, res.b2_p, n, lightIdx);
                            Li += job_v * beta * area_le(P.lights[lightIdx], n, -q.dir_p);
                            finish = true;
                            busy = false;
                            { go = false; break; }
                        } while (0);
                        if (go && stage == kStShadow) do {
                            // ---- one segment of Tr() (:298-322) came back
                            if (hit) {
                                const int hit_mat = __float_as_int(reinterpret_cast<const float *>(reinterpret_cast<const float4 *>(P.shade) + 5 * res.prim_p + 4)[2]);
                                if (hit_mat != -1) {
                                    walk_tr = v3(0.f, 0.f, 0.f);
                                    stage = kStShadowDone;
                                    break;
                                }
                            }
                            stage = kStShadowB;
                            if (walk_medium >= 0) {
                                job = kJobTr;
                                job_medium = walk_medium;
                                job_tmax = hit ? res.t_p : q.tmax_s;
                                { go = false; break; }
                            }
                            break;
                        } while (0);
                        if (go && stage == kStShadowB) do {
                            // ---- ... and the medium of the segment has been walked
                            if (walk_medium >= 0) walk_tr = walk_tr * job_v;
                            if (hit) {
                                V3 n;
                                int lightIdx;
                                make_light_hit(P, res.prim_p, res.b1_p, res.b2_p, n, lightIdx);
                                walk_medium = dot(q.dir_p, n) > 0 ? P.prim_media[2 * res.prim_p + 1] : P.prim_media[2 * res.prim_p];
                                walk_left -= res.t_p;
                                q.org = q.org + q.dir_p * res.t_p;
                                q.tmax_s = walk_left;
                                q.has_p = true;
                                stage = kStShadow;
                                busy = false;
                                { go = false; break; }
                            }
                            stage = kStShadowDone;
                            break;
                        } while (0);
                        if (go && stage == kStShadowDone) do {
                            // ---- Tr() is complete
                            if (ctx_scatter) {
                                if (sav_has) Li += walk_tr * beta * sav_w * sav_rad / sav_den;             // :1096-1097
                                float pux = rng_uniform(rng);
                                float puy = rng_uniform(rng);
                                const V3 dir = medium_sample_phase(P.mediums[medium], pux, puy);
                                specular = false;
                                finish = true;
                                if (bounces + 1 < P.max_depth) {
                                    bool kill = false;
                                    if (bounces > 3) {
                                        float illumate = clamp(1.f - rr_luminance(beta), 0.f, 1.f);
                                        if (rng_uniform(rng) < illumate)
                                            kill = true;
                                        else
                                            beta /= (1 - illumate);
                                    }
                                    if (!kill) {
                                        q.org = ctx_o;
                                        q.dir_p = dir;
                                        q.tmax_s = __builtin_inff();
                                        q.has_p = true;
                                        finish = false;
                                        bounces++;
                                        stage = kStPath;
                                    }
                                }
                                busy = false;
                                { go = false; break; }
                            }
                            Ld_acc += sav_w * walk_tr * sav_fr * sav_rad * sav_abs / sav_den;              // :1150
                            stage = kStMisStart;
                            break;
                        } while (0);
                        if (go && stage == kStMisStart) do {
                            // ---- the BSDF-sampled light ray (:1153-1160)
                            Ray r;
                            r.o = ctx_o;
                            r.d = ctx_d;
                            const Hit isect = make_hit(P, r, ctx_t, ctx_prim, ctx_b1, ctx_b2);
                            const gpt_material &material = P.materials[isect.matIdx];
                            float usx = rng_uniform(rng);
                            float usy = rng_uniform(rng);
                            float usz = rng_uniform(rng);
                            const Surface S = surface_prepare(P, material, -ctx_d, isect.nor, isect.dpdu, isect.uv);
                            const Scatter probe = surface_scatter(S, material, usx, usy, usz);
                            if (!(is_black(probe.f) || probe.pdf == 0)) {
                                mis_fr = probe.f;
                                mis_cos = fabs_(dot(probe.wi, isect.nor));
                                mis_pdf = probe.pdf;
                                q.org = isect.pos;
                                q.dir_p = probe.wi;
                                q.tmax_s = __builtin_inff();
                                q.has_p = true;
                                stage = kStMis;
                                busy = false;
                                { go = false; break; }
                            }
                            Li += beta * Ld_acc;
                            stage = kStContinue;
                            break;
                        } while (0);
                        if (go && stage == kStMis) do {
                            // ---- ... came back (:1161-1205)
                            bool contributes = false;
                            if (hit) {
                                V3 n;
                                int lightIdx;
                                make_light_hit(P, res.prim_p, res.b1_p, res.b2_p, n, lightIdx);
                                V3 radiance = v3(0.f, 0.f, 0.f);
                                if (lightIdx != -1) radiance = area_le(P.lights[lightIdx], n, -q.dir_p);
                                if (!is_black(radiance)) {
                                    V3 pp = q.org + res.t_p * q.dir_p;
                                    float pdfA = 1.f / P.lights[lightIdx].area;
                                    float choicePdf = pdf_from_light_distribution(P, lightIdx);
                                    float lenSquare = dot(pp - q.org, pp - q.org);
                                    float costheta = fabs_(dot(n, q.dir_p));
                                    float lPdf = pdfA * lenSquare / (costheta);
                                    sav_w = mis_weight(mis_pdf, lPdf * choicePdf);
                                    sav_rad = radiance;
                                    job_tmax = res.t_p;
                                    contributes = true;
                                }
                            } else if (P.inf.isvalid) {
                                sav_rad = inf_le(P.inf, q.dir_p);
                                float choicePdf = pdf_from_light_distribution(P, P.n_lights);
                                float lightPdf = ONE_OVER_FOUR_PI;
                                sav_w = mis_weight(mis_pdf, lightPdf * choicePdf);
                                job_tmax = __builtin_inff();
                                contributes = true;
                            }
                            if (contributes) {
                                stage = kStMisB;
                                job_v = v3(1.f, 1.f, 1.f);
                                if (medium >= 0) {
                                    job = kJobTr;
                                    job_medium = medium;
                                    { go = false; break; }
                                }
                                break;
                            }
                            Li += beta * Ld_acc;
                            stage = kStContinue;
                            break;
                        } while (0);
                        if (go && stage == kStMisB) do {
                            Ld_acc += sav_w * job_v * mis_fr * sav_rad * mis_cos / mis_pdf;
                            Li += beta * Ld_acc;
                            stage = kStContinue;
                            break;
                        } while (0);
                        if (go && stage == kStContinue) do {
                            // ---- kStContinue: the continuation (:1210-1229) and the roulette (:1232-1238).  On the last bounce
                            // nothing of it reaches Li, so it is skipped.
                            finish = true;
                            if (bounces + 1 < P.max_depth) {
                                Ray r;
                                r.o = ctx_o;
                                r.d = ctx_d;
                                const Hit isect = make_hit(P, r, ctx_t, ctx_prim, ctx_b1, ctx_b2);
                                const gpt_material &material = P.materials[isect.matIdx];
                                const V3 wo = -ctx_d;
                                float ux = rng_uniform(rng);
                                float uy = rng_uniform(rng);
                                float uz = rng_uniform(rng);
                                const Surface S = surface_prepare(P, material, wo, isect.nor, isect.dpdu, isect.uv);
                                const Scatter next = surface_scatter(S, material, ux, uy, uz);
                                const V3 out = next.wi;
                                if (!is_black(next.f)) {
                                    beta *= next.f * fabs_(dot(isect.nor, out)) / next.pdf;
                                    specular = kind_is_delta(material.type);
                                    const int m_in = P.prim_media[2 * ctx_prim], m_out = P.prim_media[2 * ctx_prim + 1];
                                    int m2 = dot(out, isect.nor) > 0 ? m_out : m_in;
                                    m2 = dot(wo, isect.nor) * dot(out, isect.nor) > 0 ? medium : m2;
                                    medium = m2;
                                    bool kill = false;
                                    if (bounces > 3) {
                                        float illumate = clamp(1.f - rr_luminance(beta), 0.f, 1.f);
                                        if (rng_uniform(rng) < illumate)
                                            kill = true;
                                        else
                                            beta /= (1 - illumate);
                                    }
                                    if (!kill) {
                                        q.org = isect.pos;
                                        q.dir_p = out;
                                        q.tmax_s = __builtin_inff();
                                        q.has_p = true;
                                        finish = false;
                                        bounces++;
                                        stage = kStPath;
                                    }
                                }
                            }
                            busy = false;
                            { go = false; break; }
                        } while (0);
                        if (busy) {          // left the stage code with a job posted: a fresh walk
                            job_running = true;
                            trk_tr = 1.f;
                            trk_dist = 0.f;
                            trk_iter = -1;
                            busy = false;
                        }
                    }
                    n_track = popc(ballot(job_running));
#if PT_WALK_PROBE
                    if (COUNT && lane == 0) wp_cyc_stage += __builtin_readcyclecounter() - wp_t0;
#endif
                    }
                    {
                        const bool any_ready = __any(busy);           // (only if the stage code was put off)
                        if (!any_ready && n_track == 0) break;
                        const int n_rays = popc(ballot(alive && (q.has_p || finish)));
                        if (!any_ready && (n_rays >= kTraceBatch || (n_rays > 0 && n_track < kTrackMin))) break;
                    }
                    // ---- the tracking jobs: Sample() / Tr() of the medium `job_medium` along the lane's ray over
                    // [0, job_tmax) (medium.h:14-50,64-157).  Homogeneous media answer in closed form; the density grids
                    // of all lanes are walked together, each lane drawing from its own path's generator, for at most
                    // kTrackSteps steps per turn: a longer walk is parked and goes on next round, so that the lanes
                    // whose walks were short can fetch new work in between instead of idling until the longest ends.
#if PT_WALK_PROBE
                    const unsigned long long wp_t1 = __builtin_readcyclecounter();
                    if (COUNT && lane == 0) wp_track_turn++;
#endif
                    if (job_running) {
                        const DevMedium M = P.mediums[job_medium];
                        if (M.type == GPT_MEDIUM_HOMOGENEOUS) {
                            if (job == kJobSample) {
                                const float u = rng_uniform(rng);
                                job_v = hom_sample(M, job_tmax, u, job_t, job_sampled);
                            } else {
                                job_v = hom_tr(M, job_tmax);
                            }
                            job_running = false;
                            busy = true;
                        } else {
                            // mode: 0 = Sample (delta tracking to the next real collision), 1..3 = Tr by delta / ratio /
                            // residual ratio tracking
                            const int mode = job == kJobSample ? 0 : 1 + M.trType;
                            const float invMax = M.invMaxDensity;
                            const float sigma = dot(V3{M.sigmaT[0], M.sigmaT[1], M.sigmaT[2]}, v3(0.212671f, 0.715160f, 0.072169f));
                            const float maxDensity = 1 / invMax;
                            const float ce = 0.5f * maxDensity;
                            const float step3 = 1 / (maxDensity - ce) / sigma;
                            float tr = trk_tr, dist = trk_dist;
                            int iter = trk_iter < 0 ? M.iterMax : trk_iter;
                            bool sampled = false, zero = false, done = false;
#pragma unroll 1
                            for (int step = 0; step < kTrackSteps; ++step) {
#if PT_WALK_PROBE
                                if (COUNT) { wp_track_lane_steps++; if (first_active_lane()) wp_track_steps++; }
#endif
                                const float l = -gpt_logf(rng_uniform(rng));
                                if (mode == 3) dist += l * step3;
                                else dist += l * invMax / sigma;
                                if (dist >= job_tmax) { done = true; break; }
                                const float dens = het_density(M, het_local(M, q.org, q.dir_p, dist));
                                if (mode <= 1) {
                                    if (dens * invMax > rng_uniform(rng)) {
                                        sampled = true;          // Sample: a real collision; Tr (delta): the ray is absorbed
                                        tr = 0;
                                        done = true;
                                        break;
                                    }
                                    if (--iter == 0) {
                                        tr = 0;
                                        done = true;
                                        break;
                                    }
                                } else {
                                    if (mode == 2) tr *= 1.f - dens * invMax;
                                    else tr *= 1.f - (dens - ce) / (maxDensity - ce);
                                    if (tr < 0.1f) {
                                        const float qq = 1.f - tr;
                                        if (rng_uniform(rng) < qq) {
                                            zero = true;
                                            done = true;
                                            break;
                                        }
                                        if (mode == 2) tr = 1;
                                        else tr /= (1.f - qq);
                                    }
                                    if (--iter == 0) { done = true; break; }
                                }
                            }
                            job_running = !done;
                            busy = done;
                            if (!done) {
                                trk_tr = tr;
                                trk_dist = dist;
                                trk_iter = iter;
                            } else if (mode == 0) {
                                job_t = dist;
                                job_sampled = sampled;
                                job_v = sampled ? v3(M.sigmaS[0] / M.sigmaT[0], M.sigmaS[1] / M.sigmaT[1], M.sigmaS[2] / M.sigmaT[2]) : v3(1.f, 1.f, 1.f);
                            } else {
                                if (mode == 3 && !zero) tr *= gpt_expf(-job_tmax * ce * sigma);
                                if (zero) tr = 0.f;
                                job_v = v3(tr, tr, tr);
                            }
                        }
                    }
#if PT_WALK_PROBE
                    if (COUNT && lane == 0) wp_cyc_track += __builtin_readcyclecounter() - wp_t1;
#endif
                }
            }
            if (INTEG != PT_IT_VPT_WALK && alive && !waiting) {
                // ---- resolve the direct light of the previous bounce ------------------
                if (direct) {
                    V3 Ld = v3(0.f, 0.f, 0.f);
                    if (q.has_s && !res.occluded) Ld += cand;
                    if (INTEG == GPT_IT_VPT && q.has_s && res.occluded && poison_occluded)
                        Ld += v3(__builtin_nanf(""));           // Tr = 0 times a non-finite factor (pathtracer.cu:298-322)
                    V3 tr_m = v3(1.f, 1.f, 1.f);                 // Volpath: transmittance along the BSDF-sampled light ray
                    if (INTEG == GPT_IT_VPT && q.has_m && medium_ld >= 0)
                        tr_m = hom_tr(P.mediums[medium_ld], res.prim_m >= 0 ? res.t_m : __builtin_inff());
                    if (q.has_m) {
                        if (res.prim_m >= 0) {
                          if (!q.mis_any) {
                            V3 n;
                            int lightIdx;
                            make_light_hit(P, res.prim_m, res.b1_m, res.b2_m, n, lightIdx);
                            V3 radiance = v3(0.f, 0.f, 0.f);
                            if (lightIdx != -1) radiance = area_le(P.lights[lightIdx], n, -q.dir_m);
                            if (!is_black(radiance)) {
                                V3 p = q.org + res.t_m * q.dir_m;
                                float pdfA = 1.f / P.lights[lightIdx].area;              // area.h:28-32
                                float choicePdf = pdf_from_light_distribution(P, lightIdx);
                                float lenSquare = dot(p - q.org, p - q.org);
                                float costheta = fabs_(dot(n, q.dir_m));
                                float lPdf = pdfA * lenSquare / (costheta);
                                float weight = mis_weight(mis_pdf, lPdf * choicePdf);
                                if (INTEG == GPT_IT_VPT) Ld += weight * tr_m * mis_fr * radiance * mis_cos / mis_pdf;
                                else Ld += weight * mis_fr * radiance * mis_cos / mis_pdf;
                            }
                          }
                        } else if (P.inf.isvalid) {
                            V3 radiance = inf_le(P.inf, q.dir_m);
                            float choicePdf = pdf_from_light_distribution(P, P.n_lights);
                            float lightPdf = ONE_OVER_FOUR_PI;                             // infinite.h:38-41
                            float weight = mis_weight(mis_pdf, lightPdf * choicePdf);
                            if (INTEG == GPT_IT_VPT) Ld += weight * tr_m * mis_fr * radiance * mis_cos / mis_pdf;
                            else Ld += weight * mis_fr * radiance * mis_cos / mis_pdf;
                        }
                    }
                    // executed even when both rays were skipped: beta * 0 is NaN for a non-finite
                    // throughput, and the reference then discards the sample (pathtracer.cu:994,1019)
                    Li += beta_ld * Ld;
                    direct = false;
                }
                if (ending) finish = true;
                PT_SUBPHASE(cyc_direct)

                // ---- the path ray came back: pathtracer.cu:905-1016 ----------------------
                if (!finish && q.has_p) {
                    if (res.prim_p < 0) {
                        if (INTEG != GPT_IT_AO && (bounces == 0 || specular) && P.inf.isvalid)
                            Li += beta * inf_le(P.inf, q.dir_p);
                        finish = true;          // Ao: the sample is 0 (pathtracer.cu:852-855)
                    } else {
                        Ray r;
                        r.o = q.org;
                        r.d = q.dir_p;
                        const Hit isect = make_hit(P, r, res.t_p, res.prim_p, res.b1_p, res.b2_p);
                        const V3 pos = isect.pos;
                        const V3 nor = isect.nor;
                        const V2 uv = isect.uv;
                        const V3 dpdu = isect.dpdu;
                        const V3 wo = -q.dir_p;
                        // (a reference: the 18 words of the record are fetched where a question reads them instead of sitting in
                        // registers from here to the continuation - 92 -> 60 B of scratch in the wide kernel, +3 % on the headline)
                        const gpt_material &material = P.materials[isect.matIdx];
                        q.has_s = q.has_m = q.has_p = false;
                        PT_MARK(1)

                        // Volpath (pathtracer.cu:1062-1070): the medium decides whether the ray gets as far as the surface
                        bool scattered = false;
                        float scatter_t = 0.f;
                        if (INTEG == GPT_IT_VPT && medium >= 0) {
                            float u = rng_uniform(rng);
                            beta *= hom_sample(P.mediums[medium], res.t_p, u, scatter_t, scattered);
                        }
                        if (INTEG == GPT_IT_VPT && is_black(beta)) {
                            finish = true;
                        } else if (INTEG == GPT_IT_VPT && scattered) {
                            // ---- a scattering event inside the medium (pathtracer.cu:1071-1101) ----
                            const DevMedium M = P.mediums[medium];
                            float u = rng_uniform(rng);
                            float choicePdf;
                            int idx = lookup_light_distribution(P, u, choicePdf);
                            bool inf = idx == P.n_lights;
                            V3 samplePos = q.org + q.dir_p * scatter_t;
                            float u1x = rng_uniform(rng);
                            float u1y = rng_uniform(rng);
                            V3 radiance = v3(0.f), lightNor;
                            Ray shadowRay;
                            shadowRay.o = samplePos;
                            shadowRay.d = v3(0.f);
                            shadowRay.tmin = P.eps;
                            shadowRay.tmax = 0.f;
                            float lightPdf = 0.f;
                            if (idx >= 0) {
                                if (!inf)
                                    area_sample_light(P.lights[idx], samplePos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                                else
                                    inf_sample_light(P.inf, samplePos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                            }
                            float phase = medium_phase(M, wo, shadowRay.d);
                            poison_occluded = false;
                            if (!is_black(radiance)) {
                                // Li += tr * beta * phase * radiance / (lightPdf * choicePdf), tr = 0 when the light is hidden
                                const V3 tr1 = hom_tr(M, shadowRay.tmax);
                                cand = tr1 * beta * phase * radiance / (lightPdf * choicePdf);
                                poison_occluded = is_nan(v3(0.f) * beta * phase * radiance / (lightPdf * choicePdf));
                                if (!is_black(cand) || poison_occluded) {
                                    q.dir_s = shadowRay.d;
                                    q.tmax_s = shadowRay.tmax;
                                    q.has_s = true;
                                }
                            }
                            beta_ld = v3(1.f, 1.f, 1.f);
                            direct = true;
                            float pux = rng_uniform(rng);
                            float puy = rng_uniform(rng);
                            const V3 dir = medium_sample_phase(M, pux, puy);
                            q.org = samplePos;
                            specular = false;
                            ending = true;
                            if (bounces + 1 < P.max_depth) {
                                bool kill = false;
                                if (bounces > 3) {
                                    float illumate = clamp(1.f - rr_luminance(beta), 0.f, 1.f);
                                    if (rng_uniform(rng) < illumate)
                                        kill = true;
                                    else
                                        beta /= (1 - illumate);
                                }
                                if (!kill) {
                                    q.dir_p = dir;
                                    q.has_p = true;
                                    ending = false;
                                    bounces++;
                                }
                            }
                            if (ending && !q.has_s) {
                                Li += beta_ld * v3(0.f, 0.f, 0.f);
                                direct = false;
                                finish = true;
                            }
                        } else if (INTEG == GPT_IT_AO) {
                            // pathtracer.cu:857-872
                            V3 n = nor;
                            if (dot(wo, nor) < 0.f)
                                n = -n;
                            float u1 = rng_uniform(rng);
                            float u2 = rng_uniform(rng);
                            float pdf;
                            V3 dir = frame_to_world(cosine_lobe(u1, u2, pdf), dpdu, n, cross(dpdu, n));
                            float cosine = dot(dir, n);
                            float v = cosine * ONE_OVER_PI / pdf;
                            cand = v3(v, v, v);                 // L += v if the occlusion ray escapes
                            beta_ld = v3(1.f, 1.f, 1.f);
                            q.org = pos;
                            if (!is_black(cand)) {               // a zero term adds nothing either way (NaN is traced)
                                q.dir_s = dir;
                                q.tmax_s = P.ao_max_dist;
                                q.has_s = true;
                            }
                            direct = true;
                            ending = true;
                            if (!q.has_s) {
                                Li += beta_ld * v3(0.f, 0.f, 0.f);
                                direct = false;
                                finish = true;
                            }
                        } else if ((bounces == 0 || specular) && isect.lightIdx != -1) {
                            if (INTEG == GPT_IT_VPT) {
                                V3 tr = v3(1.f, 1.f, 1.f);
                                if (medium >= 0) tr = hom_tr(P.mediums[medium], res.t_p);
                                Li += tr * beta * area_le(P.lights[isect.lightIdx], nor, wo);        // pathtracer.cu:1103-1115
                            } else {
                                Li += beta * area_le(P.lights[isect.lightIdx], nor, wo);
                            }
                            finish = true;
                        } else {
                            q.org = pos;
                            // direct light with multiple importance sampling: everything that
                            // does not depend on visibility is evaluated now
                            Surface S;
                            if (kind_is_delta(material.type)) {
                                S = surface_prepare(P, material, wo, nor, dpdu, uv);
                            } else {
                                poison_occluded = false;
                                float u = rng_uniform(rng);
                                float choicePdf;
                                int idx = lookup_light_distribution(P, u, choicePdf);
                                bool inf = idx == P.n_lights;
                                float u1x = rng_uniform(rng);
                                float u1y = rng_uniform(rng);
                                V2 u1 = v2(u1x, u1y);
                                V3 radiance = v3(0.f), lightNor;
                                Ray shadowRay;
                                shadowRay.o = pos;
                                shadowRay.d = v3(0.f);
                                shadowRay.tmin = P.eps;
                                shadowRay.tmax = 0.f;
                                float lightPdf = 0.f;
                                if (idx >= 0) {
                                    if (!inf)
                                        area_sample_light(P.lights[idx], pos, u1, radiance, shadowRay, lightNor, lightPdf, P.eps);
                                    else
                                        inf_sample_light(P.inf, pos, u1, radiance, shadowRay, lightNor, lightPdf, P.eps);
                                }
                                // what the bounce's three questions to the surface share (pt_bsdf.h), formed once the light's
                                // code is through with the registers
                                S = surface_prepare(P, material, wo, nor, dpdu, uv);
                                if (!is_black(radiance)) {
                                    const Scatter lit = surface_respond(S, material, shadowRay.d);
                                    const V3 fr = lit.f;
                                    float weight = mis_weight(lightPdf * choicePdf, lit.pdf);
                                    if (INTEG == GPT_IT_VPT) {
                                        // Ld += weight * tr * fr * radiance * |cos| / pdf with tr = Tr(shadowRay): the medium's
                                        // transmittance if the light is visible, 0 if not (pathtracer.cu:1146-1151)
                                        V3 tr1 = v3(1.f, 1.f, 1.f);
                                        if (medium >= 0) tr1 = hom_tr(P.mediums[medium], shadowRay.tmax);
                                        cand = weight * tr1 * fr * radiance * fabs_(dot(nor, shadowRay.d)) / (lightPdf * choicePdf);
                                        poison_occluded = is_nan(weight * v3(0.f) * fr * radiance * fabs_(dot(nor, shadowRay.d)) / (lightPdf * choicePdf));
                                    } else {
                                        cand = weight * fr * radiance * fabs_(dot(nor, shadowRay.d)) / (lightPdf * choicePdf);
                                    }
                                    // an exactly-zero term (e.g. the light is below the horizon of a lambertian
                                    // surface: Fr returns 0) adds nothing whether or not the light is visible
                                    if (!is_black(cand) || (INTEG == GPT_IT_VPT && poison_occluded)) {
                                        q.dir_s = shadowRay.d;
                                        q.tmax_s = shadowRay.tmax;
                                        q.has_s = true;
                                    }
                                }
                                PT_MARK(2)
                                float usx = rng_uniform(rng);
                                float usy = rng_uniform(rng);
                                float usz = rng_uniform(rng);
                                const Scatter probe = surface_scatter(S, material, usx, usy, usz);
                                const V3 out = probe.wi, fr = probe.f;
                                const float pdf = probe.pdf;
                                if (!(is_black(fr) || pdf == 0)) {
                                    // The BSDF-sampled light ray contributes only if its CLOSEST hit is an emitter
                                    // triangle (pathtracer.cu:964-976) or, with an environment light, if it escapes
                                    // (:978-990).  With few emitters, test their triangles first: if
                                    // Triangle::Intersect would reject all of them, no traversal order can make an
                                    // emitter the closest hit.  Then, without an environment light the ray is not
                                    // traced at all; with one, only hit / no hit matters and the ray is traced as an
                                    // any-hit ray (the first accepted triangle is the same in both traversals).
                                    bool useful = true;
                                    q.mis_any = false;
                                    if (P.n_lights <= kEmitterPretestMax) {
                                        bool emitter = false;
                                        for (int li = 0; li < P.n_lights; ++li)
                                            emitter = emitter || emitter_accepts(P.lights[li], pos, out, P.eps);
                                        if (!emitter) {
                                            useful = P.inf.isvalid != 0;
                                            q.mis_any = true;
                                        }
                                    }
                                    if (useful) {
                                        mis_fr = fr;
                                        mis_cos = fabs_(dot(out, nor));
                                        mis_pdf = pdf;
                                        q.dir_m = out;
                                        q.has_m = true;
                                    }
                                }
                                beta_ld = beta;
                                medium_ld = medium;
                                direct = true;
                                PT_MARK(3)
                            }
                            // continuation.  The reference also samples it on the last bounce and
                            // then leaves the loop; nothing of that sample reaches Li, so it is skipped.
                            ending = true;
                            if (bounces + 1 < P.max_depth) {
                                float ux = rng_uniform(rng);
                                float uy = rng_uniform(rng);
                                float uz = rng_uniform(rng);
                                const Scatter next = surface_scatter(S, material, ux, uy, uz);
                                const V3 out = next.wi;
                                if (!is_black(next.f)) {
                                    beta *= next.f * fabs_(dot(nor, out)) / next.pdf;
                                    specular = kind_is_delta(material.type);
                                    if (INTEG == GPT_IT_VPT) {
                                        // the medium on the side the new ray leaves on; a reflection stays where it was
                                        // (pathtracer.cu:1223-1227)
                                        const int m_in = P.prim_media[2 * res.prim_p], m_out = P.prim_media[2 * res.prim_p + 1];
                                        int m2 = dot(out, nor) > 0 ? m_out : m_in;
                                        m2 = dot(wo, nor) * dot(out, nor) > 0 ? medium : m2;
                                        medium = m2;
                                    }
                                    bool kill = false;
                                    if (bounces > 3) {
                                        float illumate = clamp(1.f - rr_luminance(beta), 0.f, 1.f);
                                        if (rng_uniform(rng) < illumate)
                                            kill = true;
                                        else
                                            beta /= (1 - illumate);
                                    }
                                    if (!kill) {
                                        q.dir_p = out;
                                        q.has_p = true;
                                        ending = false;
                                        bounces++;
                                    }
                                }
                            }
                            PT_MARK(4)
                            if (ending && !q.has_s && !q.has_m) {
                                if (direct) {      // nothing to wait for: Ld = 0
                                    Li += beta_ld * v3(0.f, 0.f, 0.f);
                                    direct = false;
                                }
                                finish = true;
                            }
                        }
                    }
                }
            }

            PT_SUBPHASE(cyc_hit)
            PT_MARK(8)
            if (finish) {
                // The sample goes to its iteration's plane as is; the finite-guard of pathtracer.cu:1019-1020
                // and the accumulation run in iteration order in pt_output_kernel.
                // one 16-byte store per sample: stores are written through to the fabric per request
                float4 *dst = reinterpret_cast<float4 *>(P.samples) + (uint64_t)(iter - P.iter_first) * P.plane + plane_slot;
                *dst = make_float4(Li.x, Li.y, Li.z, 0.f);
                alive = false;
                q.has_s = q.has_m = q.has_p = false;
            }
            // ---- regenerate: pathtracer.cu:881-903 ---------------------------------------
            bool start = false;
            if (next_sample >= n_item_samples && more_items && !__all(alive)) {
                uint32_t t = 0;
                if constexpr (XCD_QUEUES) {
                    t = n_items;
                    while (queues_done < 8u) {
                        const uint32_t qi = (blockIdx.x + queues_done) & 7u;
                        uint32_t k = 0;
                        if (lane == 0) k = atomicAdd(P.tile_counter + qi, 1u);
                        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                        const uint32_t lo = (uint32_t)(((uint64_t)n_items * qi) >> 3), hi = (uint32_t)(((uint64_t)n_items * (qi + 1u)) >> 3);
                        if (k < hi - lo) {
                            t = lo + k;
                            break;
                        }
                        queues_done++;
                    }
                } else {
                    if (lane == 0) t = atomicAdd(P.tile_counter, 1u);
                    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                }
                if (t >= n_items) {
                    more_items = false;
                } else {
                    const uint32_t chunk = t / n_owned;
                    uint32_t tile_local = t - chunk * n_owned;
                    if constexpr (TILE_STRIP != 0u) {   // a bijection of [0, n_owned): the owned tiles seen as a grid of gw columns (exact when tiles_x % n_ranks == 0), strip by strip
                        const uint32_t gw = (P.tiles_x + P.n_ranks - 1u) / P.n_ranks, gh = n_owned / gw;
                        if (tile_local < gw * gh) {
                            const uint32_t strip_items = TILE_STRIP * gh;
                            const uint32_t strip = tile_local / strip_items, r = tile_local - strip * strip_items;
                            const uint32_t left = gw - strip * TILE_STRIP;
                            const uint32_t w = left < TILE_STRIP ? left : TILE_STRIP;
                            const uint32_t ty = r / w;
                            tile_local = ty * gw + strip * TILE_STRIP + (r - ty * w);
                        }
                    }
                    const uint32_t tile = P.rank + tile_local * P.n_ranks;
                    tile_xy = (tile % P.tiles_x) | ((tile / P.tiles_x) << 16);      // (one SGPR: the kernel is at the SGPR limit)
                    slot_base = tile_local * 64u;          // planes are tile-major over this rank's tiles
                    chunk_first = P.iter_first + chunk * P.chunk_iters;
                    const uint32_t chunk_count = (chunk + 1u == P.n_chunks) ? P.iter_count - chunk * P.chunk_iters : P.chunk_iters;
                    n_item_samples = 64u * chunk_count;
                    next_sample = 0;
                }
            }
            if (next_sample < n_item_samples) {
                const unsigned long long m_idle = ballot(!alive);
                const uint32_t s = next_sample + (uint32_t)lane_rank(m_idle);
                next_sample += (uint32_t)popc(m_idle);
                if (!alive && s < n_item_samples) {
                    x = (tile_xy & 0xffffu) * 8u + (s & 7u);
                    y = (tile_xy >> 16) * 8u + ((s >> 3) & 7u);
                    iter = chunk_first + (s >> 6);
                    plane_slot = slot_base + (s & 63u);
                    start = x < P.stride && y < P.rows;
                }
            }
            if (start) {
                const uint32_t pixel = x + y * P.stride;          // pathtracer.cu:881-883
                rng_seed(rng, wang_hash(pixel) + wang_hash(iter));
                float offsetx = rng_uniform(rng) - 0.5f;
                float offsety = rng_uniform(rng) - 0.5f;
                float du1 = rng_uniform(rng);
                float du2 = rng_uniform(rng);
                Ray r = primary_ray(P.cam, x + offsetx, y + offsety, du1, du2);
                q.org = r.o;
                q.dir_p = r.d;
                q.has_p = true;
                q.has_s = q.has_m = false;
                Li = v3(0.f, 0.f, 0.f);
                beta = v3(1.f, 1.f, 1.f);
                specular = false;
                bounces = 0;
                ending = false;
                direct = false;
                medium = (INTEG == GPT_IT_VPT || INTEG == PT_IT_VPT_WALK) ? P.cam.medium : -1;      // pathtracer.cu:1043
                if (INTEG == PT_IT_VPT_WALK) {
                    stage = kStPath;
                    q.tmax_s = __builtin_inff();
                }
                alive = true;
                if (COUNT) cnt.samples++;
            }
            PT_SUBPHASE(cyc_regen)
            PT_MARK(5)
            if (!__any(alive)) {
                if (next_sample >= n_item_samples && !more_items) break;
                continue;                              // only samples of pixels outside the frame were drawn
            }

            if (COUNT && alive) {
                if (q.has_p) { cnt.bounce_iters++; cnt.closest_rays++; }
                if (q.has_m) cnt.closest_rays++;
                if (q.has_s) cnt.shadow_rays++;
            }
            // ---- deposit this bounce's rays, drain the pool, pick up the results -----------------
            if (!alive) q.has_p = q.has_m = q.has_s = false;
            PoolLayout L;
            int n_new = 0;
            if (CARRY) {
                n_new = pool_deposit_fixed<INTEG == PT_IT_VPT_WALK>(pool, q, lane, alive && !waiting);
                L.m_p = L.m_m = L.m_s = 0ull;
                L.n_p = L.n_m = L.n_rays = 0;
            } else {
                L = pool_deposit<INTEG == PT_IT_VPT_WALK>(pool, q, lane);
            }
            PT_MARK(6)
            wave_lds_fence();
            unsigned long long c0 = 0;
            if (COUNT) {
                c0 = __builtin_readcyclecounter();
                if (lane == 0) cyc_shade += c0 - cyc_mark;
            }
#if PT_WALK_PROBE
            if (COUNT && INTEG == PT_IT_VPT_WALK && lane == 0) wp_drains++;
#endif
            if (SMALL) {
                LdsScene mem;
                mem.first = (int)lds_address(lds_scene);
                mem.end = mem.first + 32 * P.n_nodes;
                mem.tri_bias = mem.end;
                if (COUNT)      // the counting build runs the C++ twin (it has the counters)
                    trace_pool<COUNT, false>(P, pool, L.n_rays, cnt, mem);
                else
                    trace_pool_lds_asm(lds_address(pool), L.n_rays, mem, P.eps);
            } else if (WIDE) {
                if (COUNT || !PT_WIDE_ASM)      // the counting build runs the C++ twin (it has the counters)
                    trace_pool_wide<COUNT>(P, pool, n_new, cnt);
                else
                    trace_pool_wide_asm(lds_address(pool), n_new, P, lane, n_new > 0);
            } else {
                GlobalScene mem;
                mem.nodes = reinterpret_cast<const char *>(P.nodes);
                mem.tris = reinterpret_cast<const char *>(P.tris);
                mem.first = 0;
                mem.end = 32 * P.n_nodes;
                mem.tri_bias = 0;
                if (COUNT)      // (the twin always drains to the end: nothing is ever suspended)
                    trace_pool<COUNT, true>(P, pool, n_new, cnt, mem);
                else
                    trace_pool_global_asm(lds_address(pool), n_new, mem, P.eps, n_new > 0);
            }
            if (COUNT) {
                cyc_mark = __builtin_readcyclecounter();
                if (lane == 0) cyc_trace += cyc_mark - c0;
            }
            wave_lds_fence();
            // The origin and the directions come back from the pool as well (same bits), so they do not
            // occupy registers while the pool is drained.
            {
                const float4 ro = pool[2 * kPoolSlots + lane];
                q.org = V3{ro.x, ro.y, ro.z};
            }
            if (CARRY) waiting = alive && reinterpret_cast<const volatile unsigned *>(pool + kPendOff)[lane] != 0u;
            if (q.has_p && !waiting) {
                const int sl = CARRY ? (int)lane : lane_rank(L.m_p);
                const RayResult rr = pool_result(pool, sl);
                const float4 rd = pool[2 * sl];
                q.dir_p = V3{rd.x, rd.y, rd.z};
                res.prim_p = rr.prim; res.t_p = rr.t; res.b1_p = rr.b1; res.b2_p = rr.b2;
            }
            if (q.has_m && !waiting) {
                const int sl = CARRY ? 64 + (int)lane : L.n_p + lane_rank(L.m_m);
                const RayResult rr = pool_result(pool, sl);
                const float4 rd = pool[2 * sl];
                q.dir_m = V3{rd.x, rd.y, rd.z};
                res.prim_m = rr.prim; res.t_m = rr.t; res.b1_m = rr.b1; res.b2_m = rr.b2;
            }
            if (q.has_s && !waiting) res.occluded = pool_result(pool, CARRY ? 128 + (int)lane : L.n_p + L.n_m + lane_rank(L.m_s)).prim >= 0;
            wave_lds_fence();
        }
    }

    if (COUNT) {
        atomicAdd(&P.counters[0], (unsigned long long)cnt.node_visits);
        atomicAdd(&P.counters[1], (unsigned long long)cnt.prim_tests);
        atomicAdd(&P.counters[2], (unsigned long long)cnt.bounce_iters);
        atomicAdd(&P.counters[3], (unsigned long long)cnt.shadow_rays);
        atomicAdd(&P.counters[4], (unsigned long long)cnt.closest_rays);
        atomicAdd(&P.counters[5], (unsigned long long)cnt.samples);
#if PT_WALK_PROBE
        if (INTEG == PT_IT_VPT_WALK) {
            // probe build only: counters[6..15] mean something else than in the counting build (tools/gpu_walk_probe.py reads them):
            // 6 stage passes, 7 lanes in them, 8 tracking turns, 9 pool drains, 10 tracking steps, 11 lane-steps, 12 / 13 cycles in
            // stage code / tracking, 14 / 15 cycles draining / everything else
            atomicAdd(&P.counters[6], wp_stage_pass); atomicAdd(&P.counters[7], wp_stage_lanes);
            atomicAdd(&P.counters[8], wp_track_turn); atomicAdd(&P.counters[9], wp_drains);
            atomicAdd(&P.counters[10], wp_track_steps); atomicAdd(&P.counters[11], wp_track_lane_steps);
            atomicAdd(&P.counters[12], wp_cyc_stage); atomicAdd(&P.counters[13], wp_cyc_track);
            if (lane == 0) { atomicAdd(&P.counters[14], cyc_trace); atomicAdd(&P.counters[15], cyc_shade); }
            return;
        }
#endif
        atomicAdd(&P.counters[6], (unsigned long long)cnt.w_node);
        atomicAdd(&P.counters[7], (unsigned long long)cnt.w_prim);
        atomicAdd(&P.counters[8], (unsigned long long)cnt.w_trip);
        atomicAdd(&P.counters[9], (unsigned long long)cnt.l_trip);
        if (lane == 0) {     // split of cyc_shade: direct-light resolution | hit shading | finish + regeneration | (rest: pool)
            atomicAdd(&P.counters[10], cyc_direct);
            atomicAdd(&P.counters[11], cyc_hit);
            atomicAdd(&P.counters[12], cyc_regen);
        }
        if (lane == 0) {
            atomicAdd(&P.counters[14], cyc_trace);     // shader-clock cycles this wave spent draining pools
            atomicAdd(&P.counters[15], cyc_shade);     // ... and everywhere else (shading, regeneration, deposit)
        }
    }
}

// Output (pathtracer.cu:2516-2531) for a batch: per pixel, in iteration order,
//   if the sample is finite it becomes kernel_color (:1019-1020: a non-finite sample leaves the previous
//   value there), kernel_acc_image += kernel_color; after the last iteration out = tonemap(acc / iter).
// Streaming: 16 B per pixel per iteration read, 48 B per pixel read/written once.
__global__ void __launch_bounds__(256) pt_output_kernel(const DevParams P)
{
    // one thread per slot of a sample plane: slot = local tile * 64 + pixel in tile (the path kernel's layout),
    // so a wave reads 1 KB of consecutive samples per iteration
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.plane) return;
    const uint32_t tile = P.rank + (uint32_t)(i >> 6) * P.n_ranks;
    const uint32_t x = (tile % P.tiles_x) * 8u + ((uint32_t)i & 7u), y = (tile / P.tiles_x) * 8u + (((uint32_t)i >> 3) & 7u);
    if (x >= P.stride || y >= P.rows) return;
    const uint32_t pixel = x + y * P.stride;
    V3 col = V3{P.color[3 * pixel], P.color[3 * pixel + 1], P.color[3 * pixel + 2]};
    V3 acc = v3(0.f);
    if (!P.reset) acc = V3{P.acc[3 * pixel], P.acc[3 * pixel + 1], P.acc[3 * pixel + 2]};
    const float4 *s = reinterpret_cast<const float4 *>(P.samples) + i;
    for (uint32_t k = 0; k < P.iter_count; ++k, s += P.plane) {
        const float4 sv = *s;
        const V3 Li = V3{sv.x, sv.y, sv.z};
        // Path stores a finite sample (pathtracer.cu:1019), Ao any sample that is not NaN (:874)
        if (P.integrator == GPT_IT_AO ? !is_nan(Li) : (!is_inf(Li) && !is_nan(Li))) col = Li;
        acc += col;
    }
    P.acc[3 * pixel] = acc.x;
    P.acc[3 * pixel + 1] = acc.y;
    P.acc[3 * pixel + 2] = acc.z;
    P.color[3 * pixel] = col.x;
    P.color[3 * pixel + 1] = col.y;
    P.color[3 * pixel + 2] = col.z;
    if (P.out != nullptr && P.iter_count > 0) {
        const uint32_t last = P.iter_first + P.iter_count - 1u;
        V3 o = tonemap(acc / (float)(int)last, P.cam.filmic != 0);   // Output: acc / iter, iter is int
        P.out[3 * pixel] = o.x;
        P.out[3 * pixel + 1] = o.y;
        P.out[3 * pixel + 2] = o.z;
    }
}

// Output without the accumulate: out = tonemap(acc / iter)
__global__ void __launch_bounds__(256) pt_tonemap_kernel(const float *acc, float *out, uint32_t stride, uint32_t rows,
                                                         uint32_t iter, int filmic)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stride * rows) return;
    V3 a = V3{acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]};
    V3 o = tonemap(a / (float)(int)iter, filmic != 0);
    out[3 * i] = o.x;
    out[3 * i + 1] = o.y;
    out[3 * i + 2] = o.z;
}

// The traversal operators alone - Intersect / IntersectP (pathtracer.cu:214-296) - for a list of rays, through the SAME pools and
// the SAME hand-scheduled loops as the render kernel (scene in LDS, in global memory with suspended drains, or the 4-wide walk):
// gpt_debug_trace.  Ray i = rays[2 i] {origin.xyz, tmax}, rays[2 i + 1] {direction.xyz, any_hit != 0}; out[i] = {primitive or -1,
// t, b1, b2}.  One ray per lane per round.
template <bool SMALL, bool WIDE>
__global__ void __launch_bounds__(256, WIDE ? PT_WIDE_WAVES : PT_MIN_WAVES) pt_trace_rays_kernel(const DevParams P_in, const float4 *rays, int n, float4 *out)
{
    __shared__ float4 lds_scene[SMALL ? kSmallSceneFloat4 : 1];
    DevParams P = P_in;
    if (SMALL) {         // the staging of pt_render_kernel (nodes and triangles only matter here)
        const float4 *gn = reinterpret_cast<const float4 *>(P_in.nodes);
        const float4 *gt = reinterpret_cast<const float4 *>(P_in.tris);
        const int o_tri = 2 * P.n_nodes;
        const int lds_nodes = (int)lds_address(lds_scene), lds_tris = lds_nodes + o_tri * 16;
        for (int i = threadIdx.x; i < 2 * P.n_nodes; i += 256) {
            float4 v = gn[i];
            if (i & 1) {
                if (__float_as_int(v.w) >= 0) {
                    v.z = __int_as_float(__float_as_int(v.z) + lds_tris);
                    v.w = __int_as_float(__float_as_int(v.w) + lds_tris);
                } else {
                    v.z = __int_as_float(__float_as_int(v.z) + lds_nodes);
                }
            }
            lds_scene[i] = v;
        }
        for (int i = threadIdx.x; i < 3 * P.n_prims; i += 256) lds_scene[o_tri + i] = gt[i];
        __syncthreads();
    }
    constexpr bool CARRY = !SMALL;
    constexpr int kWaveFloat4 = WIDE ? kWaveWideFloat4 : (CARRY ? kWaveCarryFloat4 : kWaveLdsFloat4);
    __shared__ float4 lds_pool[4 * kWaveFloat4];
    float4 *pool = lds_pool + (threadIdx.x >> 6) * kWaveFloat4;
    const unsigned lane = threadIdx.x & 63u;
    if (CARRY) reinterpret_cast<unsigned *>(pool + kPendOff)[lane] = 0u;
    if (CARRY && !WIDE) {
        pool[kSuspOff + 2 * lane] = make_float4(__int_as_float(32 * P.n_nodes), __int_as_float(0), __int_as_float(-1), __int_as_float(-1));
        pool[kSuspOff + 2 * lane + 1] = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f);
    }
    if (WIDE) wide_init_suspend_record(P, lane);
    Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)cnt;
    const int wave = (int)(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = (int)gridDim.x * 4;
    for (int base = wave * 64; base < n; base += n_waves * 64) {      // wave-uniform
        const int i = base + (int)lane;
        const bool valid = i < n;
        RaySet q;
        q.org = q.dir_p = q.dir_m = q.dir_s = v3(0.f);
        q.tmax_s = 0.f;
        q.has_p = q.has_m = q.has_s = false;
        q.mis_any = false;
        if (valid) {
            const float4 a = rays[2 * i], b = rays[2 * i + 1];
            q.org = V3{a.x, a.y, a.z};
            q.tmax_s = a.w;
            if (b.w != 0.f) { q.dir_s = V3{b.x, b.y, b.z}; q.has_s = true; }
            else { q.dir_p = V3{b.x, b.y, b.z}; q.has_p = true; }
        }
        PoolLayout L;
        int n_new = 0;
        if (CARRY) {
            n_new = pool_deposit_fixed<true>(pool, q, lane, valid);
            L.m_p = L.m_m = L.m_s = 0ull;
            L.n_p = L.n_m = L.n_rays = 0;
        } else {
            L = pool_deposit<true>(pool, q, lane);
        }
        wave_lds_fence();
        if (SMALL) {
            LdsScene mem;
            mem.first = (int)lds_address(lds_scene);
            mem.end = mem.first + 32 * P.n_nodes;
            mem.tri_bias = mem.end;
            trace_pool_lds_asm(lds_address(pool), L.n_rays, mem, P.eps);
        } else {
            // a drain may stop with rays parked (the render kernel shades in between): go on until every ray is back
            for (int round = 0;; ++round) {
                const int fresh = round == 0 ? n_new : 0;
                if (WIDE) {
                    if (PT_WIDE_ASM) trace_pool_wide_asm(lds_address(pool), fresh, P, lane, fresh > 0);
                    else trace_pool_wide<false>(P, pool, fresh, cnt);
                } else {
                    GlobalScene mem;
                    mem.nodes = reinterpret_cast<const char *>(P.nodes);
                    mem.tris = reinterpret_cast<const char *>(P.tris);
                    mem.first = 0;
                    mem.end = 32 * P.n_nodes;
                    mem.tri_bias = 0;
                    trace_pool_global_asm(lds_address(pool), fresh, mem, P.eps, fresh > 0);
                }
                wave_lds_fence();
                const bool pending = valid && reinterpret_cast<const volatile unsigned *>(pool + kPendOff)[lane] != 0u;
                if (!__any(pending)) break;
            }
        }
        wave_lds_fence();
        if (valid) {
            const int sl = CARRY ? (q.has_s ? 128 + (int)lane : (int)lane) : (q.has_s ? L.n_p + L.n_m + lane_rank(L.m_s) : lane_rank(L.m_p));
            const RayResult rr = pool_result(pool, sl);
            out[i] = make_float4(__int_as_float(rr.prim), rr.t, rr.b1, rr.b2);
        }
        wave_lds_fence();
    }
}

// elementary-operation probes for the parity tests (gpt_debug_math / gpt_debug_rng)
__global__ void pt_debug_math_kernel(int fn, const float *x, const float *y, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.f;
    switch (fn) {
    case 0: r = gpt_sinf(x[i]); break;
    case 1: r = gpt_cosf(x[i]); break;
    case 2: r = gpt_tanf(x[i]); break;
    case 3: r = gpt_atanf(x[i]); break;
    case 4: r = gpt_acosf(x[i]); break;
    case 5: r = gpt_powf(x[i], y[i]); break;
    case 6: r = x[i] / y[i]; break;
    case 7: r = sqrt_rn(x[i]); break;
    case 8: r = rsqrt_rn(x[i]); break;
    case 9: r = gpt_expf(x[i]); break;
    case 10: r = gpt_logf(x[i]); break;
    default: break;
    }
    out[i] = r;
}
// the surface operators on their own (gpt_debug_bsdf): one case per lane through the routines the render kernels shade with
__global__ void pt_debug_bsdf_kernel(const gpt_material *material, const DevTexture *texture, const float *geom11, const float *in3, int n, int mode,
                                     float *out7)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DevParams P;
    __builtin_memset(&P, 0, sizeof(P));
    P.textures = texture;
    const gpt_material m = *material;
    const float *g = geom11 + 11 * (size_t)i;
    const Surface S = surface_prepare(P, m, v3(g[0], g[1], g[2]), v3(g[3], g[4], g[5]), v3(g[6], g[7], g[8]), v2(g[9], g[10]));
    const float a = in3[3 * (size_t)i], b = in3[3 * (size_t)i + 1], c = in3[3 * (size_t)i + 2];
    const Scatter r = mode == 0 ? surface_respond(S, m, v3(a, b, c)) : surface_scatter(S, m, a, b, c);
    float *o = out7 + 7 * (size_t)i;
    o[0] = r.wi.x; o[1] = r.wi.y; o[2] = r.wi.z;
    o[3] = r.f.x; o[4] = r.f.y; o[5] = r.f.z;
    o[6] = r.pdf;
}
__global__ void pt_debug_rng_kernel(uint32_t pixel, uint32_t iter, uint32_t *seed_out, float *u_out, int n)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng r;
    uint32_t s = wang_hash(pixel) + wang_hash(iter);
    *seed_out = s;
    rng_seed(r, s);
    for (int i = 0; i < n; ++i) u_out[i] = rng_uniform(r);
}

}  // namespace pt

// ------------------------------------------------------------ launchers -------
namespace pt {

// resident 256-thread workgroups per CU for the persistent grid
// Volpath runs on the one-ray-at-a-time kernel when the scene has density grids or material-less surfaces
// (`force`: gpt_set_option "vpt_walk_kernel" - tests compare the two kernels)
bool render_uses_walk_kernel(const DevParams &P, bool force)
{
    return P.integrator == GPT_IT_VPT && (P.vpt_walk || force);
}

int render_kernel_blocks_per_cu(bool count, bool walk, bool wide)
{
    int n = 0;
    if (wide) {
        hipError_t ew = walk ? (count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<true, false, PT_IT_VPT_WALK, true>, 256, 0)
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<false, false, PT_IT_VPT_WALK, true>, 256, 0))
                             : (count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<true, false, GPT_IT_PT, true>, 256, 0)
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<false, false, GPT_IT_PT, true>, 256, 0));
        if (ew != hipSuccess || n < 1) n = 2;
        return n > 8 ? 8 : n;
    }
    hipError_t e = walk ? (count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<true, true, PT_IT_VPT_WALK>, 256, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<false, true, PT_IT_VPT_WALK>, 256, 0))
                        : (count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<true, true, GPT_IT_PT>, 256, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_render_kernel<false, true, GPT_IT_PT>, 256, 0));
    if (e != hipSuccess || n < 1) n = 2;
    if (n > 8) n = 8;
    return n;
}

bool render_scene_fits_lds(const DevParams &P)
{
    static_assert(kSmallSceneFloat4 == GPT_LDS_SCENE_FLOAT4, "gpt_scene_fits_lds (include/gpt_traversal.h) describes this kernel's LDS scene");
    return P.traversal == 0 && gpt_scene_fits_lds(P.n_nodes, P.n_prims, P.n_lights, P.n_materials);
}

hipError_t launch_render(const DevParams &P, bool count, int n_blocks, bool lds_scene, bool force_walk, hipStream_t stream)
{
    const bool small = lds_scene && render_scene_fits_lds(P);
    const bool ao = P.integrator == GPT_IT_AO;
#define PT_LAUNCH(C, S, I) hipLaunchKernelGGL((pt_render_kernel<C, S, I>), dim3(n_blocks), dim3(256), 0, stream, P)
#define PT_LAUNCH_WIDE(C, I) hipLaunchKernelGGL((pt_render_kernel<C, false, I, true>), dim3(n_blocks), dim3(256), 0, stream, P)
    if (P.traversal == GPT_TRAVERSAL_WIDE4) {
        if (render_uses_walk_kernel(P, force_walk)) { if (count) PT_LAUNCH_WIDE(true, PT_IT_VPT_WALK); else PT_LAUNCH_WIDE(false, PT_IT_VPT_WALK); }
        else if (P.integrator == GPT_IT_VPT) { if (count) PT_LAUNCH_WIDE(true, GPT_IT_VPT); else PT_LAUNCH_WIDE(false, GPT_IT_VPT); }
        else if (!ao) { if (count) PT_LAUNCH_WIDE(true, GPT_IT_PT); else PT_LAUNCH_WIDE(false, GPT_IT_PT); }
        else { if (count) PT_LAUNCH_WIDE(true, GPT_IT_AO); else PT_LAUNCH_WIDE(false, GPT_IT_AO); }
    } else if (render_uses_walk_kernel(P, force_walk)) {
        if (count && small) PT_LAUNCH(true, true, PT_IT_VPT_WALK);
        else if (count) PT_LAUNCH(true, false, PT_IT_VPT_WALK);
        else if (small) PT_LAUNCH(false, true, PT_IT_VPT_WALK);
        else PT_LAUNCH(false, false, PT_IT_VPT_WALK);
    } else if (P.integrator == GPT_IT_VPT) {
        if (count && small) PT_LAUNCH(true, true, GPT_IT_VPT);
        else if (count) PT_LAUNCH(true, false, GPT_IT_VPT);
        else if (small) PT_LAUNCH(false, true, GPT_IT_VPT);
        else PT_LAUNCH(false, false, GPT_IT_VPT);
    } else if (!ao) {
        if (count && small) PT_LAUNCH(true, true, GPT_IT_PT);
        else if (count) PT_LAUNCH(true, false, GPT_IT_PT);
        else if (small) PT_LAUNCH(false, true, GPT_IT_PT);
        else PT_LAUNCH(false, false, GPT_IT_PT);
    } else {
        if (count && small) PT_LAUNCH(true, true, GPT_IT_AO);
        else if (count) PT_LAUNCH(true, false, GPT_IT_AO);
        else if (small) PT_LAUNCH(false, true, GPT_IT_AO);
        else PT_LAUNCH(false, false, GPT_IT_AO);
    }
#undef PT_LAUNCH
#undef PT_LAUNCH_WIDE
    return hipGetLastError();
}

hipError_t launch_trace_rays(const DevParams &P, bool lds_scene, const float4 *rays, int n, float4 *out, hipStream_t stream)
{
    const bool small = lds_scene && render_scene_fits_lds(P);
    int n_blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    // the wide walk's spill stacks are indexed by workgroup: never more workgroups than that buffer has slices
    if (P.traversal == GPT_TRAVERSAL_WIDE4 && n_blocks > (int)P.wide_stack_blocks) n_blocks = (int)P.wide_stack_blocks;
    if (n_blocks < 1) n_blocks = 1;
    if (P.traversal == GPT_TRAVERSAL_WIDE4) hipLaunchKernelGGL((pt_trace_rays_kernel<false, true>), dim3(n_blocks), dim3(256), 0, stream, P, rays, n, out);
    else if (small) hipLaunchKernelGGL((pt_trace_rays_kernel<true, false>), dim3(n_blocks), dim3(256), 0, stream, P, rays, n, out);
    else hipLaunchKernelGGL((pt_trace_rays_kernel<false, false>), dim3(n_blocks), dim3(256), 0, stream, P, rays, n, out);
    return hipGetLastError();
}

hipError_t launch_output(const DevParams &P, hipStream_t stream)
{
    const uint64_t n = P.plane;
    hipLaunchKernelGGL(pt_output_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float *acc, float *out, uint32_t stride, uint32_t rows, uint32_t iter, int filmic,
                          hipStream_t stream)
{
    const uint32_t n = stride * rows;
    hipLaunchKernelGGL(pt_tonemap_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, acc, out, stride, rows, iter, filmic);
    return hipGetLastError();
}

hipError_t launch_debug_math(int fn, const float *x, const float *y, float *out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, x, y, out, n);
    return hipGetLastError();
}

hipError_t launch_debug_bsdf(const gpt_material *material, const DevTexture *texture, const float *geom11, const float *in3, int n, int mode,
                             float *out7, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_debug_bsdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, material, texture, geom11, in3, n, mode, out7);
    return hipGetLastError();
}

hipError_t launch_debug_rng(uint32_t pixel, uint32_t iter, uint32_t *seed_out, float *u_out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_debug_rng_kernel, dim3(1), dim3(64), 0, stream, pixel, iter, seed_out, u_out, n);
    return hipGetLastError();
}

}  // namespace pt

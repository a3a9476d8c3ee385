// pt_wavefront.h — the decoupled scheduler's device-side records, shared by pt_wavefront.hip (the kernel) and render_api.cpp
// (allocation, launch).
//
// The reference's Path loop (src/pathtracer.cu:904-1016) is cut where it calls Intersect / IntersectP (:905, :942, :960):
//   shade phase   one lane per PATH SLOT: consume the three rays' results (direct light of the previous bounce, :943-994;
//                 the surface hit, :906-941, 953-956, 997-1016), write a finished sample, start the next sample in the
//                 same slot (:881-903), emit up to three rays
//   trace phase   one lane per RAY: a lane that finishes a ray takes the next ray of the pool, whatever path it belongs to ("scheduler" 1),
//                 or a RAY STREAM: the rays live in LDS and every trip steps a batch of rays that are all at a wide node or all at a
//                 leaf ("scheduler" 2, the 4-wide walk only)
// A persistent workgroup alternates the two phases over its own POOL of path slots (kWfWgChunks x 64); everything a path
// carries between the phases lives in HBM as a structure of arrays of float4 planes (kWfStateBytes per slot) - what decouples
// ray supply and shading from one wave's 64 paths is that any wave of the workgroup may take any chunk of the pool.
#pragma once

#include <stdint.h>
#include <hip/hip_runtime.h>

namespace pt {

// control block in device memory (zeroed by the host before every batch)
struct WfCtrl {
    uint32_t next_item;                 // work items (tile, iteration chunk) handed out so far (may run past n_items)
    uint32_t pad[7];
};

// A workgroup's pool: kWfWgChunks chunks of 64 path slots.  Chunk c of the pool owns ray ids rayq[192 c ..]: its segment of the
// ray queue, filled by whichever wave shaded the chunk, with the count in LDS - no atomic, no compaction across chunks.
#ifndef PT_WF_WG_WAVES
#define PT_WF_WG_WAVES 4             // waves per workgroup (1: a wave is its own workgroup - no barrier between the phases at all)
#endif
#ifndef PT_WF_WG_CHUNKS
#define PT_WF_WG_CHUNKS (4 * PT_WF_WG_WAVES)
#endif
constexpr int kWfWgWaves = PT_WF_WG_WAVES;
constexpr int kWfWgChunks = PT_WF_WG_CHUNKS;
constexpr int kWfSegRays = 192;
// A ray the trace phase parks when its wave runs out of segments (the round must not wait for its longest ray): {id, entry, stack size |
// (round it resumes in + 1) << 8, interval end} {best hit} + the LDS levels of its stack; the same lane of the same wave resumes it in
// the next round's trace phase (levels beyond the LDS ones are in that wave's spill slice already).  The ray's result slot holds
// primitive = kWfPending meanwhile, and the shade phase lets its path sit the round out.
constexpr int kWfSaveDwords = 40;
constexpr int32_t kWfPending = -2;

// ray id in the queue: path slot | kind << 28 | any_hit << 31
constexpr uint32_t kWfPathMask = 0x0fffffffu;
constexpr int kWfKindShift = 28;
constexpr uint32_t kWfAnyHit = 0x80000000u;

// flags word of a path slot (plane s3, .w)
constexpr uint32_t kWfBouncesMask = 0xffu, kWfSpecular = 1u << 8, kWfDirect = 1u << 9, kWfEnding = 1u << 10, kWfAlive = 1u << 11,
                   kWfHasP = 1u << 12, kWfHasM = 1u << 13, kWfHasS = 1u << 14, kWfMisAny = 1u << 15, kWfPoison = 1u << 16;

struct WfParams {
    // path state, n_paths entries per plane
    float4 *s0;        // {Li.xyz, beta.x}
    float4 *s1;        // {beta.y, beta.z, mis_cos, mis_pdf}
    float4 *s2;        // {cand.xyz, rng}
    float4 *s3;        // {beta_ld.xyz, flags}
    float4 *s4;        // {mis_fr.xyz, dst}: dst = float4 index of the sample's slot in the sample planes
    float4 *org;       // {ray origin.xyz, medium | medium_ld << 16 (Volpath)}
    float4 *ray;       // [3][n_paths] {direction.xyz, tmax}: path ray, BSDF-sampled light ray, shadow ray
    float4 *hit;       // [3][n_paths] {primitive or -1, t, b1, b2}
    uint32_t *rayq;    // [3 * n_paths] ray ids of the current round, one segment of 192 per chunk
    uint2 *wave_item;  // [n_paths / 64] the work item a chunk's samples are handed out of: {item or ~0, samples taken}
    WfCtrl *ctrl;
    uint32_t *spill;   // wide walk: stack levels beyond the LDS ones, spill_levels x 64 dwords per wave
    uint32_t *save;    // wide walk: one record of kWfSaveDwords per lane of the trace grid for a ray that is parked between two rounds
    uint32_t n_items;  // work items of this batch: owned tiles x iteration chunks
    uint32_t item_iters;   // iterations per item (the last chunk may be shorter)
    uint32_t n_chunks;
    uint32_t n_paths;  // path slots: 64 kWfWgChunks per workgroup of the persistent grid
    uint32_t spill_levels;
};

constexpr int kWfStateBytes = 6 * 16 + 3 * 16 + 3 * 16 + 3 * 4;

// stream_trace: the wide walk of the trace phase as a ray stream (a wave's rays live in LDS, a batch of them is stepped per trip) instead of
// one lane per ray; the binary tree is walked one lane per ray either way
hipError_t launch_wf_render(const struct DevParams &P, const WfParams &W, int n_blocks, bool stream_trace, hipStream_t stream);
int wf_blocks_per_cu(int integrator, bool wide, bool stream_trace);
int wf_lds_stack_levels(bool stream_trace);
int wf_spill_columns(bool stream_trace);
int wf_paths_per_block();
int wf_waves_per_block();

}  // namespace pt

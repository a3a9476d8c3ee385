// pt_wavefront.h — the decoupled ("wavefront") scheduler's device-side records, shared by pt_wavefront.hip (kernels) and
// render_api.cpp (allocation, the host's round loop).
//
// The reference's Path loop (src/pathtracer.cu:904-1016) is cut where it calls Intersect / IntersectP (:905, :942, :960):
//   shade stage   one lane per PATH SLOT: consume the three rays' results (direct light of the previous bounce, :943-994;
//                 the surface hit, :906-941, 953-956, 997-1016), write a finished sample, start the next sample in the
//                 same slot (:881-903), emit up to three rays
//   trace stage   one lane per RAY: persistent waves pull ray ids from ONE device-wide queue - a lane that finishes a ray
//                 takes the next ray of any path of the frame, so a wave's lanes stay full until the queue is empty
// Stages alternate as kernel launches on the renderer's stream; everything a path carries between them lives in HBM as a
// structure of arrays of float4 planes (kWfStateBytes per slot), sized to stay inside the 256 MiB Infinity Cache.
#pragma once

#include <stdint.h>
#include <hip/hip_runtime.h>

namespace pt {

// control block in device memory (zeroed by the host before every batch)
struct WfCtrl {
    uint32_t next_item;                 // work items (tile, iteration chunk) handed out so far (may run past n_items)
    uint32_t any_rays[2];               // round r left rays in some segment: any_rays[r & 1]
    uint32_t pad;
    uint32_t head[8];                   // trace stage: next unclaimed segment group of each XCD's share (group g belongs to XCD g % 8)
    uint32_t susp[2];                   // the trace stage of round r parked unfinished rays for round r + 1: susp[(r + 1) & 1]
};

// The ray queue is segmented: wave w of the shade stage (path slots 64 w .. 64 w + 63) owns ray ids rayq[192 w ..] and
// seg_count[w] - no atomic, no compaction across waves.  The trace stage claims GROUPS of kWfGroupWaves consecutive segments
// (one shade workgroup's rays: up to 768, typically 400 - 500) with one atomic each; group g was written by shade workgroup
// g, which ran on XCD g % 8, and is claimed first by trace workgroups of the same XCD: rays and results stay in that L2.
#ifndef PT_WF_GROUP_WAVES
#define PT_WF_GROUP_WAVES 4
#endif
constexpr int kWfGroupWaves = PT_WF_GROUP_WAVES;      // 4 in the product (the hand-scheduled walk reads a group's four counts with one scalar load); 1 or 2: experiments with the C++ walk
constexpr int kWfSegRays = 192;
// A ray the trace stage parks when its wave runs out of work (the round must not wait for its longest ray): {id, entry, stack size |
// (round it resumes in + 1) << 8, interval end} {best hit} + the LDS levels of its stack; the same lane of the same wave of the
// next round's trace stage resumes it (levels beyond the LDS ones are in that wave's spill slice already).  The ray's result
// slot holds primitive = kWfPending meanwhile, and the shade stage lets its path sit the round out.
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
    uint32_t *rayq;    // [3 * n_paths] ray ids of the current round, one segment of 192 per shade wave
    uint32_t *seg_count;   // [n_paths / 64] rays in each segment
    uint2 *wave_item;  // [n_paths / 64] the work item a shade wave hands samples out of: {item or ~0, samples taken}
    WfCtrl *ctrl;
    unsigned long long *host_flag;    // pinned host memory: seq << 32 | round << 1 | done, published by the trace stage
    uint32_t *spill;   // wide walk: stack levels beyond the LDS ones, spill_levels x 64 dwords per wave
    uint32_t *save;    // wide walk: one record of kWfSaveDwords per lane of the trace grid for a ray that is parked between two rounds
    uint32_t n_items;  // work items of this batch: owned tiles x iteration chunks
    uint32_t item_iters;   // iterations per item (the last chunk may be shorter)
    uint32_t n_chunks;
    uint32_t n_paths;  // path slots (a multiple of 256)
    uint32_t round;
    uint32_t seq;      // batch number (the host tells its own batch's flags from a previous batch's)
    uint32_t spill_levels;
};

constexpr int kWfStateBytes = 6 * 16 + 3 * 16 + 3 * 16 + 3 * 4;

hipError_t launch_wf_shade(const struct DevParams &P, const WfParams &W, hipStream_t stream);
hipError_t launch_wf_trace(const struct DevParams &P, const WfParams &W, int n_blocks, hipStream_t stream);
int wf_trace_blocks_per_cu(bool wide);
int wf_lds_stack_levels();

}  // namespace pt

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
    unsigned long long next_sample;     // samples handed out so far (may run past n_samples)
    uint32_t n_rays[2];                 // rays in the queue of round r: n_rays[r & 1]
    uint32_t head;                      // trace stage: first ray id nobody has claimed yet
    uint32_t pad[3];
    uint32_t head_xcd[8];               // (reserved: one queue head per XCD)
};

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
    uint32_t *rayq;    // [3 * n_paths] ray ids of the current round
    WfCtrl *ctrl;
    unsigned long long *host_flag;    // pinned host memory: seq << 32 | round << 1 | done, published by the trace stage
    uint32_t *spill;   // wide walk: stack levels beyond the LDS ones, spill_levels x 64 dwords per wave
    unsigned long long n_samples;     // samples of this batch: owned tiles x 64 x iterations
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

// pt_device.h — device-side basics shared by the kernels of libgpt.so: the RNG (reference src/pathtracer.cu:40-49,
// 888-889: WangHash + thrust minstd_rand + uniform_real_distribution), ray / hit records, wave votes.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>
#include "pt_layout.h"

namespace pt {

// ---------------------------------------------------------------- RNG --------
struct Rng {
    uint32_t x;
};
__device__ __forceinline__ uint32_t wang_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed = seed + (seed << 3);
    seed = seed ^ (seed >> 4);
    seed = seed * 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
__device__ __forceinline__ void rng_seed(Rng &r, uint32_t s)
{
    uint32_t x = s % 2147483647u;     // minstd_rand::seed
    r.x = x == 0 ? 1u : x;
}
// x <- x * 48271 mod (2^31 - 1) without a 64-bit division: for p < 2^47,
// p mod (2^31-1) = (p & m) + (p >> 31), minus m once if that reaches m.
__device__ __forceinline__ float rng_uniform(Rng &r)
{
    uint64_t p = (uint64_t)r.x * 48271ull;
    uint32_t s = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
    if (s >= 2147483647u) s -= 2147483647u;
    r.x = s;
    // uniform_real_distribution<float>(0,1): float(x - 1) / 2^31 (exact scaling)
    return (float)(s - 1u) * 4.656612873077392578125e-10f;
}

// -------------------------------------------------------------- records ------
struct Ray {
    V3 o, d;
    float tmin, tmax;
};
struct Hit {
    V3 pos, nor;
    V2 uv;
    V3 dpdu;
    int matIdx, lightIdx;
};
struct RayResults {    // the results of a path's own three rays
    bool occluded;
    int prim_m;
    float t_m, b1_m, b2_m;
    int prim_p;
    float t_p, b1_p, b2_p;
};
struct Counters {
    uint32_t node_visits, prim_tests, bounce_iters, shadow_rays, closest_rays, samples;
    // utilisation probes (counting build): wave-level trips, incremented by one lane per wave per trip
    uint32_t w_node, w_prim, w_trip, l_trip, w_shade, l_shade, w_nee, l_nee;
};
// Wave votes.  The builtin takes the i1 directly (HIP's __ballot(int) widens the predicate to a VGPR and
// compares it again: two VALU instructions per vote in the traversal loop), and counting the halves
// separately keeps every comparison of counts on the scalar unit (a 64-bit ctpop is compared as u64 on VALU).
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int popc(unsigned long long m)
{
    return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32));
}
__device__ __forceinline__ bool first_active_lane()
{
    const unsigned long long m = ballot(true);
    return (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m);
}

__device__ __forceinline__ V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 rcp3(V3 d) { return V3{1.f / d.x, 1.f / d.y, 1.f / d.z}; }

__device__ __forceinline__ int lane_rank(unsigned long long mask)   // number of set bits below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in issue order; this only stops the compiler from
    // moving LDS accesses across the hand-off between lanes.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)      // wave-uniform value -> SGPR pair
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned lds_address(const void *p)      // LDS byte address of a __shared__ object
{
    return (unsigned)(unsigned long long)p;                          // low half of the flat (shared aperture) address
}

}  // namespace pt

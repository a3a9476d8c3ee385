// Exhaustive check (all 2^32 bit patterns): is  r = v_rcp_f32(x); e = fma(-x, r, 1); r = fma(e, r, r)
// equal to the IEEE quotient 1.0f / x whenever 2^-100 <= |x| <= 2^100 ?  (measurement tool)
// build: hipcc --offload-arch=gfx950 -O2 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -ffp-contract=off tools/micro/rcp_exact.hip -o build/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void check(unsigned long long *bad1, unsigned long long *bad2, unsigned long long *in_range, unsigned *example)
{
    const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096ull;
    unsigned long long b1 = 0, b2 = 0, n = 0;
    for (uint32_t k = 0; k < 4096; ++k) {
        const uint32_t bits = (uint32_t)(base + k);
        const float x = __uint_as_float(bits);
        const float ax = __builtin_fabsf(x);
        if (!(ax >= 0x1p-100f && ax <= 0x1p100f)) continue;
        ++n;
        const float q = 1.0f / x;
        float r;
        asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x));
        const float e = __builtin_fmaf(-x, r, 1.0f);
        const float r1 = __builtin_fmaf(e, r, r);
        const float e2 = __builtin_fmaf(-x, r1, 1.0f);
        const float r2 = __builtin_fmaf(e2, r1, r1);
        if (__float_as_uint(r1) != __float_as_uint(q)) { if (!b1) atomicExch(example, bits); ++b1; }
        if (__float_as_uint(r2) != __float_as_uint(q)) ++b2;
    }
    atomicAdd(bad1, b1); atomicAdd(bad2, b2); atomicAdd(in_range, n);
}

int main()
{
    unsigned long long *d; unsigned *ex;
    hipMalloc(&d, 24); hipMemset(d, 0, 24); hipMalloc(&ex, 4); hipMemset(ex, 0, 4);
    check<<<4096, 256>>>(d, d + 1, d + 2, ex);      // 4096*256*4096 = 2^32
    unsigned long long h[3]; unsigned hex;
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost); hipMemcpy(&hex, ex, 4, hipMemcpyDeviceToHost);
    printf("in range: %llu  one Newton step differs: %llu (e.g. 0x%08x)  two steps differ: %llu\n", h[2], h[0], hex, h[1]);
    return 0;
}

// Issue-rate microbenchmark for gfx950 (measurement tool, not part of the product): how many cycles does one
// wave64 instruction of each class cost, alone and with other waves on the same SIMD?
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/issue_rate.hip -o build/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters, int lanes)
{
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    double dd = threadIdx.x; unsigned u0 = threadIdx.x, u1 = 48271;
    unsigned s0 = blockIdx.x, s1 = 3;
    if ((int)threadIdx.x < lanes) {
        for (int i = 0; i < iters; ++i) {
            if (MODE == 0) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }                       // dependent VALU
            if (MODE == 1) { REP64(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }   // 2 independent chains (128 instrs)
            if (MODE == 2) { REP64(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");) }                      // dependent SALU
            if (MODE == 3) { REP64(asm volatile("v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, %3" : "+v"(a), "+s"(s0) : "v"(b), "s"(s1) : "scc");) }   // VALU + SALU interleaved (128)
            if (MODE == 4) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }  // cmp -> salu -> cndmask chain (192)
            if (MODE == 5) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(a));) }                                     // transcendental
            if (MODE == 6) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double*)&a) : "v"(1.0));) }
            if (MODE == 7) { REP64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a) : "v"(1.0));) }
            if (MODE == 8) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd) : "v"(1.0000001));) }
            if (MODE == 9) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd) : "v"(1.0000001));) }
            if (MODE == 10) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(u1));) }
            if (MODE == 11) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u0) : "v"(u1));) }
            if (MODE == 12) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(dd) : "v"(u1) : "vcc");) }
            if (MODE == 13) { REP64(asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));) }
            if (MODE == 14) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u0) : "v"(u1));) }
            if (MODE == 15) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
            if (MODE == 16) { REP64(asm volatile("v_cvt_f32_f64 %0, %1\n v_cvt_f64_f32 %1, %0" : "+v"(a), "+v"(dd));) }
            if (MODE == 17) { REP64(asm volatile("v_readlane_b32 %0, %1, 3\n v_writelane_b32 %1, %0, 5" : "+s"(s0), "+v"(u0));) }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + c + d + (float)s0 + (float)dd + (float)u0;
}

template <int MODE>
void run(const char *name, int instrs_per_iter, float *out, int lanes)
{
    const int iters = 4096;
    for (int w : {1, 4}) {
        const int blocks = 256 * 4 * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<blocks, 64>>>(out, 16, lanes);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<blocks, 64>>>(out, iters, lanes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double cycles = ms * 1e-3 * 2.4e9;
        const double per_wave = cycles / ((double)iters * instrs_per_iter);
        printf("%-34s lanes %2d  waves/SIMD %d: %7.3f ms  %6.2f cycles/instr/wave  %6.2f cycles/instr/SIMD\n", name, lanes, w, ms, per_wave, per_wave / w);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
    for (int lanes : {64, 32, 8}) {
        run<0>("v_add_f32 dependent", 64, out, lanes);
        run<1>("v_add_f32 two chains", 128, out, lanes);
    }
    run<2>("s_add_u32 dependent", 64, out, 64);
    run<3>("v_add + s_add interleaved", 128, out, 64);
    run<4>("v_cmp -> s_and -> v_cndmask", 192, out, 64);
    run<5>("v_rcp_f32 dependent", 64, out, 64);
    run<6>("v_mul_f64 dependent", 64, out, 64);
    run<7>("v_pk_add_f32 dependent", 64, out, 64);
    run<8>("v_fma_f64 dependent", 64, out, 64);
    run<9>("v_add_f64 dependent", 64, out, 64);
    run<10>("v_mul_lo_u32 dependent", 64, out, 64);
    run<11>("v_mul_hi_u32 dependent", 64, out, 64);
    run<12>("v_mad_u64_u32 dependent", 64, out, 64);
    run<13>("v_sqrt_f32 dependent", 64, out, 64);
    run<14>("v_mul_u32_u24 dependent", 64, out, 64);
    run<15>("v_fma_f32 dependent", 64, out, 64);
    run<16>("v_cvt f32<->f64 pair", 128, out, 64);
    run<17>("v_readlane + v_writelane pair", 128, out, 64);
    return 0;
}

// Does "global_load_dword v, v_off, s[base:base+1]" take a full 32-bit unsigned VGPR offset on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *buf, unsigned *out, unsigned off_bytes)
{
    unsigned v, o = off_bytes + threadIdx.x * 4;
    unsigned long long base = (unsigned long long)buf;
    asm volatile("global_load_dword %0, %1, %2\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(o), "s"(base) : "memory");
    out[threadIdx.x] = v;
}
int main()
{
    const size_t n = (3ull << 30) / 4;      // 3 GiB of words
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256);
    std::vector<unsigned> h(1 << 20);
    for (unsigned off : {0u, 1u << 22, (1u << 23) - 256u, 1u << 23, 12u << 20, 1u << 30, (1u << 31) + 4096u}) {
        for (int i = 0; i < 64; ++i) h[i] = off + i;
        hipMemcpy((char *)d + off, h.data(), 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d, o, off);
        hipError_t e = hipDeviceSynchronize();
        unsigned r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
        printf("offset 0x%08x: %s  got 0x%08x expect 0x%08x\n", off, hipGetErrorString(e), r[5], off + 5);
        if (e != hipSuccess) break;
    }
    return 0;
}

// Generates tests/golden/rng_table.json from the real thrust headers (rocThrust,
// /opt/rocm/include/thrust — the same linear_congruential_engine /
// uniform_real_distribution algorithm the reference gets from the CUDA toolkit;
// call sites: reference src/pathtracer.cu:888-889).  Host-only use of
// <thrust/random.h>.  Build + run: see tools/gen_rng_golden.sh.
#include <thrust/random.h>
#include <cstdio>
#include <cstdint>

static unsigned int WangHashRestated(unsigned int seed)   // arithmetic of reference src/pathtracer.cu:40-49
{
    seed = (seed ^ 61) ^ (seed >> 16);
    seed = seed + (seed << 3);
    seed = seed ^ (seed >> 4);
    seed = seed * 0x27d4eb2d;
    seed = seed ^ (seed >> 15);
    return seed;
}

int main()
{
    const unsigned pixels[] = {0u, 1u, 12345u, 262143u, 2073599u, 8294399u, 0xffffffffu};
    const unsigned iters[] = {1u, 2u, 64u, 1024u, 4096u};
    printf("[\n");
    bool first = true;
    for (unsigned p : pixels)
        for (unsigned it : iters) {
            unsigned seed = WangHashRestated(p) + WangHashRestated(it);
            thrust::default_random_engine rng(seed);
            thrust::uniform_real_distribution<float> uniform(0.0f, 1.0f);
            printf("%s {\"pixel\": %u, \"iter\": %u, \"seed\": %u, \"u_bits\": [", first ? "" : ",\n", p, it, seed);
            for (int i = 0; i < 16; ++i) {
                float u = uniform(rng);
                uint32_t bits;
                __builtin_memcpy(&bits, &u, 4);
                printf("%s%u", i ? ", " : "", bits);
            }
            printf("]}");
            first = false;
        }
    // raw engine stream from fixed seeds, including the seed==0 and seed==modulus special cases
    const unsigned seeds[] = {0u, 1u, 2147483646u, 2147483647u, 2147483648u, 4294967295u};
    for (unsigned s : seeds) {
        thrust::default_random_engine rng(s);
        thrust::uniform_real_distribution<float> uniform(0.0f, 1.0f);
        printf(",\n {\"raw_seed\": %u, \"u_bits\": [", s);
        for (int i = 0; i < 8; ++i) {
            float u = uniform(rng);
            uint32_t bits;
            __builtin_memcpy(&bits, &u, 4);
            printf("%s%u", i ? ", " : "", bits);
        }
        printf("]}");
    }
    printf("\n]\n");
    return 0;
}

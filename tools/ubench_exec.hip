// Microbenchmark: does the issue cost of a wave64 VALU instruction on gfx950 depend on how many 16-lane quarters of the
// wave have an active lane?  (The GI traversal kernels retire more VALU instructions per microsecond than 4 cycles each
// would allow: DESIGN.md 3.3.)  usage: hipcc -O3 --offload-arch=gfx950 tools/ubench_exec.hip -o tools/ubench_exec && tools/ubench_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"

__global__ __launch_bounds__(256) void k(float* out, int iters, float s, unsigned long long mask, unsigned long long mask_odd)
{
    if ((threadIdx.x >> 6) & 1u) // odd waves of the workgroup run with the second mask
        mask = mask_odd;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) { // the loop runs with EXEC = mask
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = fmaf(a0, s, 1.f); a1 = fmaf(a1, s, 1.f); a2 = fmaf(a2, s, 1.f); a3 = fmaf(a3, s, 1.f);
                a4 = fmaf(a4, s, 1.f); a5 = fmaf(a5, s, 1.f); a6 = fmaf(a6, s, 1.f); a7 = fmaf(a7, s, 1.f);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

static void run(const char* name, unsigned long long mask, int blocks_per_cu, unsigned long long mask_odd = 0, bool mixed = false)
{
    if (!mixed)
        mask_odd = mask;
    float* out;
    const int nb = 256 * blocks_per_cu, iters = 20000;
    hipMalloc(&out, nb * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<nb, 256>>>(out, 100, 0.999f, mask, mask_odd);
    hipEventRecord(e0);
    k<<<nb, 256>>>(out, iters, 0.999f, mask, mask_odd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 32 * blocks_per_cu;
    printf("%-34s waves/SIMD=%d  %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, blocks_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {1, 4}) {
        run("all 64 lanes", ~0ull, w);
        run("lanes 0-31", 0xffffffffull, w);
        run("lanes 0-15", 0xffffull, w);
        run("lanes 0-7", 0xffull, w);
        run("lanes 0-3", 0xfull, w);
        run("lanes 0-1", 0x3ull, w);
        run("lane 0 only", 1ull, w);
        run("lane 37 only", 1ull << 37, w);
        run("one lane in each quarter", 0x0001000100010001ull, w);
        run("two lanes in each quarter", 0x0101010101010101ull, w);
        run("four lanes in each quarter", 0x1111111111111111ull, w);
        run("even lanes", 0x5555555555555555ull, w);
        run("all but lane 0", ~1ull, w);
        run("even waves full, odd waves lane 0", ~0ull, w, 1ull, true);
        run("even waves full, odd waves idle", ~0ull, w, 0ull, true);
    }
    return 0;
}

// Microbenchmark: what does one 16-byte-per-lane vector load cost the texture-address path on gfx950, as a function of how many
// lanes of the wave are active?  Every load hits the L1 (a 4 KB window per wave, re-read all the time).
// usage: hipcc -O3 --offload-arch=gfx950 tools/ubench_ta.hip -o tools/ubench_ta && tools/ubench_ta
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"

template <int WIDTH>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ buf, float* out, int iters, unsigned long long mask, int scatter)
{
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const float4* base = buf + (size_t)wave * 256;       // 4 KB per wave
    float acc = 0.f;
    unsigned idx = scatter ? (lane * 37u) & 255u : lane; // scatter: every lane in its own 64-byte line; else consecutive
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4* p = base + ((idx + u * 64u) & 255u);
                float x, w;
                if (WIDTH == 4) { const float4 v = *p; x = v.x; w = v.w; }
                else if (WIDTH == 2) { const float2 v = *reinterpret_cast<const float2*>(p); x = v.x; w = v.y; }
                else { x = *reinterpret_cast<const float*>(p); w = x; }
                acc += x;
                idx = (idx + (unsigned)(w)) & 255u;       // (data == 0: keeps the address dependent on the data, loads stay in order)
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH>
static void run(const char* name, unsigned long long mask, int scatter, int blocks_per_cu)
{
    const int nb = 256 * blocks_per_cu, iters = 4000;
    float4* buf; float* out;
    hipMalloc(&buf, (size_t)nb * 4 * 256 * sizeof(float4));
    hipMemset(buf, 0, (size_t)nb * 4 * 256 * sizeof(float4));
    hipMalloc(&out, nb * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<WIDTH><<<nb, 256>>>(buf, out, 10, mask, scatter);
    hipEventRecord(e0);
    k<WIDTH><<<nb, 256>>>(buf, out, iters, mask, scatter);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)iters * 8 * 4 * blocks_per_cu; // wave-level load instructions per CU
    printf("dwordx%d %-24s %-9s waves/SIMD=%d  %.3f ms -> %.1f ns per wave-load per CU (%.1f cycles at 2.4 GHz)\n", WIDTH, name, scatter ? "scattered" : "coalesced",
           blocks_per_cu, ms, ms * 1e6 / loads_per_cu, ms * 1e6 / loads_per_cu * 2.4);
    hipFree(buf); hipFree(out);
}

int main()
{
    for (int sc : {0, 1}) {
        run<4>("all 64 lanes", ~0ull, sc, 8);
        run<2>("all 64 lanes", ~0ull, sc, 8);
        run<1>("all 64 lanes", ~0ull, sc, 8);
        run<4>("lanes 0-15", 0xffffull, sc, 8);
        run<2>("lanes 0-15", 0xffffull, sc, 8);
        run<1>("lanes 0-15", 0xffffull, sc, 8);
        run<4>("lane 0", 1ull, sc, 8);
        run<1>("lane 0", 1ull, sc, 8);
    }
    run<4>("all 64 lanes", ~0ull, 1, 2);
    run<4>("all 64 lanes", ~0ull, 1, 4);
    return 0;
}

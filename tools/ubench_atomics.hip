// tuning only: throughput of spread global atomics on MI355X (design input for an atomic-binning ray sort)
// build: hipcc -O3 --offload-arch=gfx950 -w tools/ubench_atomics.hip -o tools/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE> __global__ void k(uint32_t* bins, uint32_t mask, uint32_t* out, uint32_t n, uint32_t cluster)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // cluster consecutive threads onto nearby bins: bin = hash(i / cluster) + (i % cluster) / 4
    const uint32_t b = (hash(i / cluster) + (hash(i) % cluster) / 4) & mask;
    if (MODE == 0) atomicAdd(&bins[b], 1u);
    else out[i] = atomicAdd(&bins[b], 1u);
}
int main()
{
    const uint32_t n = 1920 * 1080;
    uint32_t *bins, *out;
    hipMalloc(&bins, 4u << 21); hipMalloc(&out, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t bits : {16u, 18u, 21u}) for (uint32_t cluster : {1u, 64u, 1024u, 16384u}) for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            hipMemset(bins, 0, 4u << 21);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((n + 255) / 256), dim3(256), 0, 0, bins, (1u << bits) - 1u, out, n, cluster);
            else hipLaunchKernelGGL(k<1>, dim3((n + 255) / 256), dim3(256), 0, 0, bins, (1u << bits) - 1u, out, n, cluster);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("bins 2^%u cluster %5u %s: %.1f us\n", bits, cluster, mode ? "return" : "noret ", best * 1000.f);
    }
    return 0;
}

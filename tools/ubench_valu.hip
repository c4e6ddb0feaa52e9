// Microbenchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 vs v_exp_f32 on gfx950 (many waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    v2f sv = {s, s};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 8 scalar fma
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = fmaf(a0, s, 1.f); a1 = fmaf(a1, s, 1.f); a2 = fmaf(a2, s, 1.f); a3 = fmaf(a3, s, 1.f);
                a4 = fmaf(a4, s, 1.f); a5 = fmaf(a5, s, 1.f); a6 = fmaf(a6, s, 1.f); a7 = fmaf(a7, s, 1.f);
            }
        } else if (MODE == 1) { // 4 packed fma (= 8 flops-lanes)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(sv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(sv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(sv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(sv));
            }
        } else if (MODE == 2) { // 8 exp
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
            }
        } else if (MODE == 3) { // 8 pk_mul
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(sv));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(sv));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(sv));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(sv));
            }
        }
    }
    if (MODE == 4) { // mixed: per unrolled step 2 exp + 8 fma (independent chains)
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1);
                a2 = fmaf(a2, s, 1.f); a3 = fmaf(a3, s, 1.f); a4 = fmaf(a4, s, 1.f); a5 = fmaf(a5, s, 1.f);
                a6 = fmaf(a6, s, 1.f); a7 = fmaf(a7, s, 1.f); p0.x = fmaf(p0.x, s, 1.f); p0.y = fmaf(p0.y, s, 1.f);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, int instr_per_iter, int blocks_per_cu)
{
    float* out;
    int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * 4);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, 256>>>(out, 100, 0.999f);
    hipEventRecord(e0);
    k<MODE><<<nb, 256>>>(out, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocks_per_cu (4 waves per block, 4 SIMDs)
    double instr_per_simd = (double)iters * instr_per_iter * blocks_per_cu;
    printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, blocks_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("fma", 32, w);
        run<1>("pk_fma", 16, w);
        run<2>("exp2", 32, w);
        run<3>("pk_muladd", 16, w);
        run<4>("2exp+8fma", 40, w);
    }
    return 0;
}

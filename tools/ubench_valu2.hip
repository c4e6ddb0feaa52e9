// Microbenchmark 2: does VALU issue cost depend on the number of distinct VGPR source operands? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f, c0 = a0 * 0.25f, c1 = a1 * .25f, c2 = a2 * .25f, c3 = a3 * .25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) { // fma: 1 VGPR + SGPR + const
                asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a0) : "s"(s)); asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a1) : "s"(s));
                asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a2) : "s"(s)); asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a3) : "s"(s));
                asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a4) : "s"(s)); asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a5) : "s"(s));
                asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a6) : "s"(s)); asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a7) : "s"(s));
            } else if (MODE == 1) { // fma: 3 distinct VGPRs
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(b0), "v"(c0)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(b1), "v"(c1));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(b2), "v"(c2)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(b3), "v"(c3));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a4) : "v"(b0), "v"(c1)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a5) : "v"(b1), "v"(c2));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(b2), "v"(c3)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a7) : "v"(b3), "v"(c0));
            } else if (MODE == 2) { // fmac (VOP2): 2 VGPR + accumulate
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b0), "v"(c0)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(b1), "v"(c1));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(b2), "v"(c2)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(b3), "v"(c3));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a4) : "v"(b0), "v"(c1)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a5) : "v"(b1), "v"(c2));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a6) : "v"(b2), "v"(c3)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a7) : "v"(b3), "v"(c0));
            } else if (MODE == 3) { // mul: 2 VGPRs
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(b1));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(b3));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(b0)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a5) : "v"(b1));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(b2)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a7) : "v"(b3));
            } else if (MODE == 4) { // log
                asm volatile("v_log_f32 %0, %0" : "+v"(a0)); asm volatile("v_log_f32 %0, %0" : "+v"(a1)); asm volatile("v_log_f32 %0, %0" : "+v"(a2)); asm volatile("v_log_f32 %0, %0" : "+v"(a3));
                asm volatile("v_log_f32 %0, %0" : "+v"(a4)); asm volatile("v_log_f32 %0, %0" : "+v"(a5)); asm volatile("v_log_f32 %0, %0" : "+v"(a6)); asm volatile("v_log_f32 %0, %0" : "+v"(a7));
            } else if (MODE == 5) { // dependent chain of fmac on ONE accumulator (8 per unrolled step)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b0), "v"(c0)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b1), "v"(c1));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b2), "v"(c2)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b3), "v"(c3));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b0), "v"(c1)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b1), "v"(c2));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b2), "v"(c3)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b3), "v"(c0));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + c0;
}
template <int MODE>
void run(const char* name, int w)
{
    float* out; int nb = 256 * w; hipMalloc(&out, nb * 256 * 4); int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, 256>>>(out, 100, 0.999f);
    hipEventRecord(e0); k<MODE><<<nb, 256>>>(out, iters, 0.999f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 32 * w;
    printf("%-22s waves/SIMD=%d %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, w, ms, ms * 1e6 / instr_per_simd);
    hipFree(out);
}
int main()
{
    for (int w : {1, 3, 6}) {
        run<0>("fma v,s,const", w); run<1>("fma v,v,v", w); run<2>("fmac v,v", w); run<3>("mul v,v", w); run<4>("log", w); run<5>("fmac dependent chain", w);
    }
}

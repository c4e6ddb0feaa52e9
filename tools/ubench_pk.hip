// diagnostics: what the packed fp32 forms used by svgf.hip's `tap_weight2` compute on gfx950 (op_sel broadcasts, the clamp bit),
// checked against the scalar instructions on a handful of values.  hipcc --offload-arch=gfx950 tools/ubench_pk.hip -o tools/ubench_pk
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out)
{
    const f2 x = {in[0], in[1]}, p = {in[2], in[3]}, z = {in[4], in[5]};
    f2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(r) : "v"(x), "v"(p), "v"(z));
    out[0] = r.x, out[1] = r.y;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(p), "v"(z));
    out[2] = r.x, out[3] = r.y;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(p), "v"(z));
    out[4] = r.x, out[5] = r.y;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(p), "v"(z));
    out[6] = r.x, out[7] = r.y;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(p), "v"(z));
    out[8] = r.x, out[9] = r.y;
    float s;
    asm volatile("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(s) : "v"(x.x), "v"(p.x), "v"(z.x));
    out[10] = s;
}
int main()
{
    float h[6] = {-3.0f, 0.25f, 2.0f, 7.0f, 1.0f, 0.125f}, *d, *o, r[11];
    hipMalloc(&d, sizeof h), hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 1>>>(d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    const float x0 = h[0], x1 = h[1], p0 = h[2], p1 = h[3], z0 = h[4], z1 = h[5];
    auto c01 = [](float v) { return fminf(fmaxf(v, 0.f), 1.f); };
    printf("pk_fma lo-bcast clamp : %g %g   expect %g %g (unclamped %g %g)\n", r[0], r[1], c01(fmaf(x0, p0, z0)), c01(fmaf(x1, p0, z1)), fmaf(x0, p0, z0), fmaf(x1, p0, z1));
    printf("pk_add hi-bcast       : %g %g   expect %g %g\n", r[2], r[3], p1 + z0, p1 + z1);
    printf("pk_fma const swapped  : %g %g   expect %g %g\n", r[4], r[5], fmaf(x0, p0, z1), fmaf(x0, p1, z0));
    printf("pk_fma const straight : %g %g   expect %g %g\n", r[6], r[7], fmaf(x0, p0, z0), fmaf(x0, p1, z1));
    printf("pk_fma lo-bcast       : %g %g   expect %g %g\n", r[8], r[9], fmaf(x0, p0, z0), fmaf(x1, p0, z1));
    printf("v_fma clamp           : %g      expect %g\n", r[10], c01(fmaf(x0, p0, z0)));
    return 0;
}

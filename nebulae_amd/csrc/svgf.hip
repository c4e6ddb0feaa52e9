// svgf.hip -- SVGF temporal accumulation and edge-stopping a-trous wavelet for gfx950.
//
// Reference arithmetic (paths relative to the reference checkout):
//   assets/shaders/svgf_temporal.hlsl:24-68  -> svgf_temporal_kernel, and the staging phase of the fused level-0 kernel
//   assets/shaders/svgf_atrous.hlsl:29-85    -> svgf_atrous_direct_kernel / svgf_atrous_lds_kernel
// This is not a transliteration of the 8x8-threadgroup texture-fetch shaders: the temporal
// pass is a coalesced stream (one pixel per lane, 16-B radiance accesses); the a-trous pass
// is tiled on the row lattice {r + step*j} so that a workgroup's taps are whole, contiguous
// image-row segments whatever the step, staged once into LDS with depth and normals already
// decoded, and each lane filters a column of R lattice rows so every staged texel is
// read from LDS once per R outputs.  A whole frame's denoise (neb_svgf_temporal followed by neb_svgf_atrous) runs
// the temporal pass INSIDE the staging phase of level 0 and never writes the accumulated radiance.  See DESIGN.md "Kernels".
#pragma clang fp contract(off) // fused operations are explicit fmaf: see neb_device.h

#include <algorithm>
#include <atomic>

#include "neb_device.h"
#include "neb_internal.h"

// Tuning / diagnostic switches (tools/build_variant.sh): the product is built with none of them set.
#ifndef NEB_ATROUS_PER_CU_R2 // most workgroups per CU of the R = 2 tiles (= waves per SIMD = the register budget: 4 -> 128 registers)
#define NEB_ATROUS_PER_CU_R2 4
#endif
#ifndef NEB_ATROUS_XGROUP // pixels per column group of the x lattice (see AtrousTile); >= 64 = consecutive columns at every step
#define NEB_ATROUS_XGROUP 8
#endif
#ifndef NEB_ATROUS_NOTRANS // timing only (wrong results): v_log / v_exp replaced by full-rate instructions
#define NEB_ATROUS_NOTRANS 0
#endif
#ifndef NEB_ATROUS_PRIO // 0: no s_setprio; 1: a workgroup's priority = the tiles it still has to do (default)
#define NEB_ATROUS_PRIO 1
#endif
#ifndef NEB_ATROUS_R_NARROW // rows per lane of the LDS kernel at steps <= 4.  2 (8-row tiles, 28 KB, four workgroups per CU at 128 registers): S = 2 / 4
#define NEB_ATROUS_R_NARROW 2 // of the fused chain 29.4 / 29.0 us against 30.9 / 30.2 with 4, 30.5 / 30.3 with 3; the separate kernels 30.5 against 33.0; a
#endif                        // 136-row strip 9.4 against 12.8.  (With five workgroups per CU and 96 registers -- the round-2 setting -- 2 lost: 34.3 against 32.4.)
#ifndef NEB_ATROUS_R_FUSED // rows per lane of the fused temporal + level-0 kernel: 2 (its staging then holds four texels per lane instead of
#define NEB_ATROUS_R_FUSED 2 // six and the kernel fits 128 registers: four workgroups per CU, 47.6 us against 50.7 with 4)
#endif
#ifndef NEB_ATROUS_STORE // how the LDS kernel stores its output: 1 write-through (sc1, default), 0 plain, 2 non-temporal (A/B arms)
#define NEB_ATROUS_STORE 1
#endif
#ifndef NEB_ATROUS_WX8 // 64-column blocks per tile at step 8 / at steps 16 and 32: 1; 2 = A/B arm (128-column tiles, eight waves)
#define NEB_ATROUS_WX8 1
#endif
#ifndef NEB_ATROUS_WX16
#define NEB_ATROUS_WX16 1
#endif
#ifndef NEB_ATROUS_PK // 1: the levels of the fused chain evaluate a texel's taps for the lane's two output rows as packed fp32 pairs (`tap2_geometry` / `tap2_weight`); 0: A/B arm
#define NEB_ATROUS_PK 1
#endif
#ifndef NEB_ATROUS_PK_FUSED // the same in the fused temporal + level-0 kernel (tap constants as scalar operands of two plain fmas, one texel at a time:
#define NEB_ATROUS_PK_FUSED 1 // it has no registers to spare; 47.9 -> 46.2 us); 0: A/B arm
#endif
#ifndef NEB_ATROUS_PK_CLASSIC // and in the separate levels (row strips, svgf_fuse = 0), which prefetch the next tile's radiance into registers
#define NEB_ATROUS_PK_CLASSIC 1
#endif
#ifndef NEB_ATROUS_FEWER_LDS_READS // timing only (wrong results), see load_group
#define NEB_ATROUS_FEWER_LDS_READS 0
#endif
#ifndef NEB_ATROUS_STAMPS // diagnostic builds only (tools/atrous_stamps.py): per-wave phase times from s_memtime
#define NEB_ATROUS_STAMPS 0
#endif
#if NEB_ATROUS_NOTRANS
#define NEB_TAP_LOG2(x) ((x) - 1.0f)
#define NEB_TAP_EXP2(x) fmaf((x), 0.001f, 1.0f)
#else
#define NEB_TAP_LOG2(x) fast_log2(x)
#define NEB_TAP_EXP2(x) fast_exp2(x)
#endif
#if NEB_ATROUS_STAMPS
#define NEB_STAMP(i)                                        \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        const uint64_t tn_ = __builtin_amdgcn_s_memtime();  \
        st[i] += tn_ - tprev;                               \
        tprev = tn_;                                        \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
#else
#define NEB_STAMP(i) \
    do {             \
    } while (0)
#endif

namespace neb {

// ------------------------------------------------------------------------------------------
// Temporal accumulation: 60 B read + 22 B written per pixel, pure stream.
// ------------------------------------------------------------------------------------------
struct TemporalArgs {
    float4* rad_cur;
    const float4* rad_hist;
    const uint32_t* depth_cur;
    const uint32_t* depth_hist;
    const uint2* normal_cur;
    const uint2* normal_hist;
    const uint32_t* mom_hist;
    uint32_t* mom_cur;
    uint16_t* variance;
    float4* geometry;               // {decoded shading normal, depth} of the current frame for the a-trous levels, or null
    uint32_t W, Wd, row_off, nrows; // row_off = row0 - row_begin
    float neg_inv_two_sigma2_log2e, alpha, varianceEps;
};

// The temporal pass's arithmetic, shared word for word by svgf_temporal_kernel and by the staging phase of the fused
// level-0 a-trous kernel (so a frame denoised through either gives the same bits).
// lerp parameter of svgf_temporal.hlsl:44-51; also returns the decoded shading normal and depth of the current frame
__device__ __forceinline__ float temporal_blend(uint32_t dc, uint32_t dh, uint32_t nc, uint32_t nh, float neg_inv_two_sigma2_log2e, float alpha_param,
                                                float3& Nc, float& zc)
{
    Nc = oct16_unpack_zw(nc);
    const float3 Nh = oct16_unpack_zw(nh);
    zc = depth_unorm24(dc);
    const float dz = fabsf(zc - depth_unorm24(dh));
    // SVGF_DWeight: exp(-dz^2 / (2 sigma^2))   (svgf_common.hlsli:11-15)
    const float wDepth = fast_exp2(dz * dz * neg_inv_two_sigma2_log2e);
    // SVGF_NWeight: saturate(dot)               (svgf_common.hlsli:4-7)
    const float wNormal = __saturatef(fmaf(Nc.z, Nh.z, fmaf(Nc.y, Nh.y, Nc.x * Nh.x)));
    const float w = wDepth * wNormal;
    return fmaf(w, alpha_param - 1.0f, 1.0f); // lerp(1, alpha, w)  (:51)
}

// Caccum = lerp(Ccurr, Chist, alpha)  (:55); alpha channel carried from the current frame
__device__ __forceinline__ float4 temporal_accumulate(float4 Cc, float4 Ch, float alpha)
{
    float4 out;
    out.x = fmaf(alpha, Ch.x - Cc.x, Cc.x);
    out.y = fmaf(alpha, Ch.y - Cc.y, Cc.y);
    out.z = fmaf(alpha, Ch.z - Cc.z, Cc.z);
    out.w = Cc.w;
    return out;
}

// moments and variance (:57-67), as the typed R16G16_FLOAT / R16_FLOAT stores leave them: {moments bits, variance bits}
__device__ __forceinline__ uint2 temporal_moments(float4 Cc, uint32_t mh, float alpha, float varianceEps)
{
    const float Y = luminance(Cc.x, Cc.y, Cc.z);
    const float Mh0 = half_bits_to_float(mh & 0xffffu), Mh1 = half_bits_to_float(mh >> 16);
    const float M1 = fmaf(alpha, Mh0 - Y, Y);
    const float Y2 = Y * Y;
    const float M2 = fmaf(alpha, Mh1 - Y2, Y2);
    const float var = fmaxf(fmaf(-M1, M1, M2), varianceEps);
    return make_uint2(float_to_half_bits(M1) | (float_to_half_bits(M2) << 16), float_to_half_bits(var));
}

__global__ __launch_bounds__(256) void svgf_temporal_kernel(TemporalArgs a)
{
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= a.Wd * a.nrows)
        return;
    const uint32_t ry = gid / a.Wd;
    const uint32_t x = gid - ry * a.Wd;
    const size_t i = (size_t)(a.row_off + ry) * a.W + x;

    const float4 Cc = a.rad_cur[i];
    const float4 Ch = a.rad_hist[i];
    const uint32_t dc = a.depth_cur[i], dh = a.depth_hist[i];
    const uint32_t nc = a.normal_cur[i].y, nh = a.normal_hist[i].y; // .zw = shading normal
    const uint32_t mh = a.mom_hist[i];

    float3 Nc;
    float zc;
    const float alpha = temporal_blend(dc, dh, nc, nh, a.neg_inv_two_sigma2_log2e, a.alpha, Nc, zc);
    if (a.geometry) // decoded once here instead of once per a-trous level and staged texel
        a.geometry[i] = make_float4(Nc.x, Nc.y, Nc.z, zc);
    const uint2 mv = temporal_moments(Cc, mh, alpha, a.varianceEps);
    a.rad_cur[i] = temporal_accumulate(Cc, Ch, alpha);
    a.mom_cur[i] = mv.x;
    a.variance[i] = (uint16_t)mv.y;
}

hipError_t launch_temporal(const SvgfLaunch& L, float4* rad_cur, const float4* rad_hist, const uint32_t* depth_cur,
                           const uint32_t* depth_hist, const uint2* normal_cur, const uint2* normal_hist,
                           const uint32_t* mom_hist, uint32_t* mom_cur, uint16_t* variance, float4* geometry, hipStream_t s)
{
    const uint32_t Wd = (L.W / 8u) * 8u, Hd = (L.H / 8u) * 8u; // Dispatch(W/8,H/8): SVGFDenoiser.cpp:116
    const uint32_t row1 = L.row1 < Hd ? L.row1 : Hd;
    if (L.row0 >= row1 || Wd == 0)
        return hipSuccess;
    TemporalArgs a;
    a.rad_cur = rad_cur;
    a.rad_hist = rad_hist;
    a.depth_cur = depth_cur;
    a.depth_hist = depth_hist;
    a.normal_cur = normal_cur;
    a.normal_hist = normal_hist;
    a.mom_hist = mom_hist;
    a.mom_cur = mom_cur;
    a.variance = variance;
    a.geometry = geometry;
    a.W = L.W;
    a.Wd = Wd;
    a.row_off = L.row0 - L.row_begin;
    a.nrows = row1 - L.row0;
    a.neg_inv_two_sigma2_log2e = -kLog2e / (2.0f * L.p.depthSigma * L.p.depthSigma);
    a.alpha = L.p.alpha;
    a.varianceEps = L.p.varianceEps;
    const uint64_t n = (uint64_t)Wd * a.nrows;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(svgf_temporal_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// The same decode for rows the temporal pass did not cover (halo rows of a strip; ragged images; a level run on its own).
__global__ __launch_bounds__(256) void svgf_decode_geometry_kernel(const uint32_t* __restrict__ depth, const uint32_t* __restrict__ normal32,
                                                                   float4* __restrict__ geometry, size_t first, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= n)
        return;
    const size_t i = first + k;
    const float3 N = oct16_unpack_zw(normal32[2 * i + 1]); // .zw = shading normal
    geometry[i] = make_float4(N.x, N.y, N.z, depth_unorm24(depth[i]));
}

// ... of two row ranges in one launch (the halo rows above and below a strip's own: one launch where two tiny ones cost their latency twice)
__global__ __launch_bounds__(256) void svgf_decode_geometry2_kernel(const uint32_t* __restrict__ depth, const uint32_t* __restrict__ normal32,
                                                                    float4* __restrict__ geometry, size_t first_a, size_t n_a, size_t first_b, size_t n_b)
{
    const size_t k = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= n_a + n_b)
        return;
    const size_t i = k < n_a ? first_a + k : first_b + (k - n_a);
    const float3 N = oct16_unpack_zw(normal32[2 * i + 1]);
    geometry[i] = make_float4(N.x, N.y, N.z, depth_unorm24(depth[i]));
}

hipError_t launch_decode_geometry2(uint32_t W, uint32_t row_begin, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, const uint32_t* depth,
                                   const uint2* normal, float4* geometry, hipStream_t s)
{
    const size_t n_a = a0 < a1 ? (size_t)(a1 - a0) * W : 0, n_b = b0 < b1 ? (size_t)(b1 - b0) * W : 0;
    if (n_a + n_b == 0)
        return hipSuccess;
    hipLaunchKernelGGL(svgf_decode_geometry2_kernel, dim3((unsigned)((n_a + n_b + 255) / 256)), dim3(256), 0, s, depth,
                       reinterpret_cast<const uint32_t*>(normal), geometry, n_a ? (size_t)(a0 - row_begin) * W : 0, n_a,
                       n_b ? (size_t)(b0 - row_begin) * W : 0, n_b);
    return hipGetLastError();
}

hipError_t launch_decode_geometry(uint32_t W, uint32_t row_begin, uint32_t row0, uint32_t row1, const uint32_t* depth, const uint2* normal,
                                  float4* geometry, hipStream_t s)
{
    if (row0 >= row1)
        return hipSuccess;
    const size_t first = (size_t)(row0 - row_begin) * W, n = (size_t)(row1 - row0) * W;
    hipLaunchKernelGGL(svgf_decode_geometry_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, depth, reinterpret_cast<const uint32_t*>(normal),
                       geometry, first, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// A-trous: 5x5 taps at spacing `step`, edge-stopped by depth, normal and luminance.
// ------------------------------------------------------------------------------------------
struct AtrousArgs {
    const float4* src;
    float4* dst;
    const uint16_t* variance;
    const float4* geometry;     // {decoded shading normal.xyz, depth}: NEB_PLANE_GEOMETRY
    const float4* alpha_src;    // kInLum with OUT_ALPHA: the plane whose centre .w the output carries (the destination itself)
    int W, H, Wd;               // image size and floor-dispatched width
    int row_begin, row_end;     // resident image rows [row_begin, row_end)
    int row0, row1;             // image rows to write (row1 already clipped to (H/8)*8)
    int step;
    int tiles_x, tiles_j;       // LDS kernel: tiles per row segment / per residue class
    uint32_t nblocks;           // real block count (grid is padded to a multiple of 8)
    float cz;                   // log2e / (phiDepth * step)
    float phiColor, phiNormal;
    // kInFused only: the temporal pass's own planes (src = radiance[cur] as rendered) and constants
    const float4* rad_hist;
    const uint32_t* depth_cur;
    const uint32_t* depth_hist;
    const uint2* normal_cur;
    const uint2* normal_hist;
    const uint32_t* mom_hist;
    uint32_t* mom_cur;
    uint16_t* variance_out;
    float4* geometry_out;
    float t_neg_inv_two_sigma2_log2e, t_alpha, t_varianceEps;
#if NEB_ATROUS_STAMPS
    unsigned long long* stamps;
#endif
};

// log2(K[abs(dx)] * K[abs(dy)]) with K = {1/16, 1/4, 3/8, 1/4, 1/16} indexed by abs(d): centre 1/16, +-1 -> 1/4, +-2 -> 3/8
// (svgf_atrous.hlsl:35,54,60 -- reproduced as written): the kernel weight is folded into the exponent
// of the edge-stopping exp2.
__host__ __device__ constexpr float atrous_log2k(int dx, int dy)
{
    // log2(1/16) = -4, log2(1/4) = -2, log2(3/8) = log2(3) - 3
    const float lx = (dx < 0 ? -dx : dx) == 0 ? -4.0f : ((dx < 0 ? -dx : dx) == 1 ? -2.0f : -1.4150374992788437f);
    const float ly = (dy < 0 ? -dy : dy) == 0 ? -4.0f : ((dy < 0 ? -dy : dy) == 1 ? -2.0f : -1.4150374992788437f);
    return lx + ly;
}

// max(0, dot(n0, n)) of svgf_atrous.hlsl:74 at no instruction: the centre normal arrives HALVED (exact), so the dot product is
// d / 2 and the free [0, 1] clamp of its last fma is max(0, d) / 2 for every d <= 2 -- normals decoded from RGBA16F are unit
// only to ~1e-3, d does exceed 1, and pow(d, 128) is then up to 1.14, which a clamp of d itself to [0, 1] would lose (rounds 1-2
// had that divergence; a bare max(x, 0) is a separate v_max per tap).  log2(d / 2) = log2(d) - 1: the tap's constant carries the
// + phiNormal that undoes it (`tap_constant`).
__device__ __forceinline__ float half_dot_max0(float dh) { return fminf(fmaxf(dh, 0.0f), 1.0f); }
__device__ __forceinline__ float tap_constant(float phiN, int dx, int dy) { return phiN + atrous_log2k(dx, dy); }

__device__ __forceinline__ float lum_scale(float var_f, float phiColor)
{
    const float varScale = phiColor * __fsqrt_rn(fmaxf(var_f, 1e-8f)); // :41
    return kLog2e * fast_rcp(fmaxf(varScale, 1e-6f));                  // :75 (as log2e / denom)
}

// One tap (svgf_atrous.hlsl:67-81): w = Kx*Ky * exp(-|dz|/(phiDepth*step)) * pow(max(0,d), phiNormal) * exp(-|dl|/denL) as ONE
// exp2: exponent = log2(Kx*Ky) + phiNormal*log2(d) - |dz|*cz - |dl|*cl   (d <= 0 -> -inf -> weight 0)
// h0 = the centre normal halved, lkp = tap_constant(phiN, dx, dy) = log2(Kx*Ky) + phiN.
__device__ __forceinline__ float tap_weight(float h0x, float h0y, float h0z, float z0, float lum0, float cl, float4 tA, float4 tB, float phiN, float cz,
                                            float lkp)
{
    const float dh = half_dot_max0(fmaf(h0z, tB.z, fmaf(h0y, tB.y, h0x * tB.x)));
    float e = fmaf(phiN, NEB_TAP_LOG2(dh), lkp);
    e = fmaf(-fabsf(z0 - tB.w), cz, e);
    e = fmaf(-fabsf(lum0 - tA.w), cl, e);
    return NEB_TAP_EXP2(e);
}

// the same with MINUS the centre depth and luminance (what the packed form keeps in its register pairs): |t + (-z0)| = |z0 - t| exactly
__device__ __forceinline__ float tap_weight_neg(float h0x, float h0y, float h0z, float nz0, float nl0, float cl, float4 tA, float4 tB, float phiN, float cz,
                                                float lkp)
{
    const float dh = half_dot_max0(fmaf(h0z, tB.z, fmaf(h0y, tB.y, h0x * tB.x)));
    float e = fmaf(phiN, NEB_TAP_LOG2(dh), lkp);
    e = fmaf(-fabsf(tB.w + nz0), cz, e);
    e = fmaf(-fabsf(tA.w + nl0), cl, e);
    return NEB_TAP_EXP2(e);
}

// ---- two output rows per instruction (round 4) ----
// A lane filters R = 2 output rows of one column, and four of the six staged rows it walks are taps of BOTH: the two taps share the
// texel and differ in the centre pixel only.  gfx950's packed fp32 forms do two IEEE operations per instruction at 2.09 ns per
// wave-instruction where a plain v_fma_f32 takes 1.24 and v_mul / v_add / v_sub 1.35-1.38 (tools/ubench_bank.hip): 16-25 % less issue
// time per operation, and every lane of a pair rounds exactly as the scalar instruction does -- `tap2_geometry` + `tap2_weight` are `tap_weight` twice,
// bit for bit (the separate kernels, the direct kernel and the strip path keep the scalar form and still agree with the chain).
// The texel's component is broadcast to both halves by op_sel; what the compiler does not match by itself (a high-half broadcast
// in v_pk_add_f32, the free [0, 1] clamp on the packed fma) is written out.
typedef float neb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ neb_f2 bc2(float s) { return (neb_f2){s, s}; }
__device__ __forceinline__ neb_f2 pk_fma(neb_f2 x, neb_f2 y, neb_f2 z) { return __builtin_elementwise_fma(x, y, z); }
// {p.y + q.x, p.y + q.y}
__device__ __forceinline__ neb_f2 pk_add_hi(neb_f2 p, neb_f2 q)
{
    neb_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(p), "v"(q));
    return r;
}
// clamp01({x.x * p.x + z.x, x.y * p.x + z.y})
__device__ __forceinline__ neb_f2 pk_fma_lo_clamp(neb_f2 x, neb_f2 p, neb_f2 z)
{
    neb_f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(r) : "v"(x), "v"(p), "v"(z));
    return r;
}
// {c.x * l.x + k.a, c.x * l.y + k.b}: (a, b) = (x, y), or (y, x) when SWAP.  l comes out of v_log_f32, and a transcendental's result needs a wait
// state before an ordinary VALU instruction may read it: the compiler inserts one for its own instructions and cannot see into an `asm` (the
// first version of this function read the registers BEFORE the logarithm had landed), so the asm carries its own s_nop.  Why asm at all: c and
// k are loop-invariant, and written as vector code the broadcast {c.x, c.x} and the swapped {k.y, k.x} are hoisted out of the tile loop as
// fourteen more registers (two of the last-level kernels then spill) instead of being the op_sel bits of this one instruction.
template <bool SWAP>
__device__ __forceinline__ neb_f2 pk_fma_const(neb_f2 c, neb_f2 l, neb_f2 k)
{
#if NEB_ATROUS_PK == 2 // A/B arm: left to the compiler
    return pk_fma(__builtin_shufflevector(c, c, 0, 0), l, SWAP ? __builtin_shufflevector(k, k, 1, 0) : k);
#else
    neb_f2 r;
    if constexpr (SWAP)
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(c), "v"(l), "v"(k));
    else
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(c), "v"(l), "v"(k));
    return r;
#endif
}
// The taps of one texel for the lane's two output rows, in three phases so that the caller can run each phase for all texels of a group before the next
// (a v_log / v_exp result needs a wait state before it is read, and the wave really waits: with a texel's phases back to back the level ran 1.1 us longer):
// h0* = the two centre normals halved, nz0 / nl0 = MINUS the centre depths / luminances (|t - z0| = |z0 - t| exactly), pc = {phiNormal, cz},
// lk = the two taps' constants in the order SWAP says.
struct Tap2 {
    neb_f2 l, dz, dl; // log2 of the halved, clamped dot products; depth and luminance differences (sign irrelevant)
};
__device__ __forceinline__ Tap2 tap2_geometry(neb_f2 h0x, neb_f2 h0y, neb_f2 h0z, neb_f2 nz0, neb_f2 nl0, float4 tA, float4 tB)
{
    const neb_f2 bxy = {tB.x, tB.y}, bzw = {tB.z, tB.w}, azw = {tA.z, tA.w};
    neb_f2 d = h0x * __builtin_shufflevector(bxy, bxy, 0, 0);
    d = pk_fma(h0y, __builtin_shufflevector(bxy, bxy, 1, 1), d);
    d = pk_fma_lo_clamp(h0z, bzw, d);
    Tap2 t;
    t.l = (neb_f2){NEB_TAP_LOG2(d.x), NEB_TAP_LOG2(d.y)};
    t.dz = pk_add_hi(bzw, nz0);
    t.dl = pk_add_hi(azw, nl0);
    return t;
}
template <bool SWAP, bool KS>
__device__ __forceinline__ neb_f2 tap2_weight(const Tap2& t, neb_f2 cl, neb_f2 pc, neb_f2 lk)
{
    neb_f2 e;
    if constexpr (KS) { // the constants are scalar operands (the fused kernel, at its register budget): two plain fmas, one SGPR source each
        e.x = fmaf(pc.x, t.l.x, SWAP ? lk.y : lk.x);
        e.y = fmaf(pc.x, t.l.y, SWAP ? lk.x : lk.y);
    } else {
        e = pk_fma_const<SWAP>(pc, t.l, lk);
    }
    e.x = fmaf(-fabsf(t.dz.x), pc.y, e.x);
    e.y = fmaf(-fabsf(t.dz.y), pc.y, e.y);
    e.x = fmaf(-fabsf(t.dl.x), cl.x, e.x);
    e.y = fmaf(-fabsf(t.dl.y), cl.y, e.y);
    return (neb_f2){NEB_TAP_EXP2(e.x), NEB_TAP_EXP2(e.y)};
}

// The a-trous output store.  A plain store leaves its line dirty in the XCD's L2, and what is dirty when the kernel ends is
// written back before the next kernel starts (MI355X_MICROARCH.md, "boundary"); written through (sc1) the 33 MB of a level
// leave beside the arithmetic: 32.4-33.5 us per level against 34.4 (non-temporal stores: 34.5).
typedef float neb_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_output(float4* p, float4 v)
{
#if NEB_ATROUS_STORE == 1
    const neb_f4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
#elif NEB_ATROUS_STORE == 2
    const neb_f4 d = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(d, reinterpret_cast<neb_f4*>(p));
#else
    *p = v;
#endif
}

// Direct kernel: one pixel per lane, 25 taps straight from global memory (L1/L2 absorb the reuse).
// Used for steps too wide for the LDS tile (>= 64) and as the in-library A/B arm (option atrous_variant = 0).
__global__ __launch_bounds__(256) void svgf_atrous_direct_kernel(AtrousArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = a.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.Wd || y >= a.row1)
        return;
    const size_t i = (size_t)(y - a.row_begin) * a.W + x;
    const float4 c0 = a.src[i];
    const float lum0 = luminance(c0.x, c0.y, c0.z);
    const float cl = lum_scale(half_bits_to_float(a.variance[i]), a.phiColor);
    const float4 g0 = a.geometry[i];
    float sr = 0.f, sg = 0.f, sb = 0.f, sw = 0.f;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        const int qy = min(max(y + dy * a.step, 0), a.H - 1); // :65 clamp to the image
        const size_t rowoff = (size_t)(qy - a.row_begin) * a.W;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int qx = min(max(x + dx * a.step, 0), a.W - 1);
            const float4 c = a.src[rowoff + qx];
            const float4 g = a.geometry[rowoff + qx];
            const float w = tap_weight(0.5f * g0.x, 0.5f * g0.y, 0.5f * g0.z, g0.w, lum0, cl, make_float4(c.x, c.y, c.z, luminance(c.x, c.y, c.z)), g,
                                       a.phiNormal, a.cz, tap_constant(a.phiNormal, dx, dy));
            sr = fmaf(w, c.x, sr);
            sg = fmaf(w, c.y, sg);
            sb = fmaf(w, c.z, sb);
            sw += w;
        }
    }
    const float inv = fast_rcp(fmaxf(sw, 1e-4f)); // :84
    a.dst[i] = make_float4(sr * inv, sg * inv, sb * inv, c0.w);
}

// LDS kernel: row-lattice tiles, persistent workgroups.
//   Workgroup = 256 lanes = 4 waves.  Output tile = BW (64) columns x BH (= 4R)
//   rows of the lattice {r + S*j}.  Taps of a lattice row are lattice rows j-2..j+2, so the
//   tile needs only BH+4 image rows for any step S; its columns are consecutive up to S = 8 (BW + 4 S staged) and a lattice of
//   8-pixel groups beyond (BW + 32 staged at any step, AtrousTile below): read amplification (1 + 4/BH)(1 + 32/BW) instead
//   of (1 + 4S/T)^2.
//   Two float4 LDS planes, lane-contiguous (ds_read_b128, conflict-free): A = {r, g, b, lum} and B = {nx, ny, nz, z}
//   (= the texels of the frame's decoded geometry plane).  Wave w filters lattice rows [w*R, w*R+R): a lane walks its
//   column's R+4 staged rows once and feeds each staged texel to every output row it is a tap of.
//
//   How a tile's texels reach LDS -- template parameter IN:
//   kInClassic  the source plane holds {r, g, b, alpha} (any radiance plane of the ABI).  The radiance of the NEXT tile is
//               loaded into registers while the current tile is filtered (its latency hides under ~1.6k VALU
//               instructions per wave); staging adds the luminance and writes A; B comes by LDS-DMA
//               (global_load_lds_dwordx4: per-lane source address, destination = wave-uniform base + lane * 16 B = exactly
//               the staging order; no registers, no VALU).
//   kInLum      the source plane holds {r, g, b, lum}: an intermediate plane of a whole-frame denoise, written by the level
//               before with OUT_ALPHA = false.  Both planes are verbatim copies: two LDS-DMAs per staged texel, no staging
//               arithmetic and no staging registers at all.
//   kInFused    level 0 of a whole-frame denoise (S = 1): src = radiance[cur] as rendered.  Staging IS the temporal pass
//               (svgf_temporal.hlsl:24-68): cur / history radiance arrive by LDS-DMA in the slots of A / B, depth and normals of
//               both frames in registers; every staged texel (halo included) is accumulated and decoded in place; the
//               tile's own pixels also write moments, variance and the decoded geometry plane -- and the accumulated
//               radiance is never written to memory (the filtered output is the next frame's history).
//   OUT_ALPHA: the output's .w is the centre pixel's alpha (the ABI's radiance planes) -- false: lum(output), for a
//   following kInLum level.  The alpha comes from src (kInClassic), from alpha_src = the destination itself (kInLum: the
//   last level writes radiance[cur], which still holds the frame's input) or from the fused staging (kInFused).
enum : int { kInClassic = 0, kInLum = 1, kInFused = 2 };

// WX: 64-column blocks per tile.  With 2, a workgroup of eight waves shares the 4 S halo columns over 128 columns (1.5 x the
// columns staged per output at S = 16 instead of 2) -- measured SLOWER, 39.2 against 36.2 us at S = 16 and 35.3 against 31.9 at
// S = 8: two large workgroups per CU leave the SIMDs idle at their barriers more than the smaller halo saves.  Product: 1.
//
// The columns of a tile.  Up to S = 8 a tile's 64 output columns are consecutive and it stages 64 + 4 S of them.  Beyond, that halo
// doubles and triples the tile (128 staged columns per 64 outputs at S = 16), so the columns form a lattice too, in GROUPS of
// XS = 8 pixels (128 bytes of a plane = one L2 line: every load and store instruction still touches whole lines): local column c is
// pixel x0 + xcol(c) = x0 + (c / XS - 2) S + c % XS.  A tap at +-S, +-2 S is then one or two groups = XS or 2 XS local columns away
// for every lane, the halo is 4 XS = 32 columns at any step (S = 16: 96 staged columns instead of 128, four workgroups per CU instead
// of three), and S / XS tiles whose x0 differ by XS interleave over the same span of (64 / XS) S pixels.  For S <= XS this is the
// plain layout.  Measured inside the frame (same box): S = 16 33.2 us against 34.9; with groups of 4 pixels (64 bytes, 80 staged
// columns) 35.6 -- an LDS-DMA instruction then touches sixteen half lines instead of eight whole ones.
template <int S, int R, int IN, int WX = 1>
struct AtrousTile {
    static constexpr int THREADS = 256 * WX;
    static constexpr int XS = S < NEB_ATROUS_XGROUP ? S : NEB_ATROUS_XGROUP; // pixels per column group = the tap stride in local columns
    static constexpr int XM = S / XS;                                        // interleaved tiles per span
    static constexpr int BW = 64 * WX, SPAN = (BW / XS) * S;                 // output columns of a tile, and the pixels they span
    static constexpr int BH = 4 * R, COLS = BW + 4 * XS, ROWS = BH + 4, TOTAL = ROWS * COLS;
    static_assert(S % XS == 0 && 64 % XS == 0, "column groups tile both the step and the wave");
    // pixel column (relative to the tile's x0) of local column c
    __host__ __device__ static constexpr int xcol(int c) { return (c / XS - 2) * S + c % XS; }
    static constexpr int NLOAD = (TOTAL + THREADS - 1) / THREADS;
    // kInFused keeps {variance, alpha} of the tile's own pixels in a third, small plane
    static constexpr int LDS_BYTES = TOTAL * 2 * 16 + (IN == kInFused ? BH * BW * 8 : 0);
    // workgroups per CU: what the 160 KB of LDS hold, at most 4 (R <= 2: 128 registers per lane) / 3 (168) -- also the register budget the kernel is compiled for
    static constexpr int PER_CU_CAP = R <= 2 ? NEB_ATROUS_PER_CU_R2 : (R == 3 ? 4 : 3);
    static constexpr int PER_CU = PER_CU_CAP < (160 * 1024) / LDS_BYTES ? PER_CU_CAP : (160 * 1024) / LDS_BYTES;
    static constexpr int WAVES_PER_SIMD = PER_CU * WX < 8 ? PER_CU * WX : 8; // launch bound: k blocks of T threads per CU <=> k T / 256 waves per SIMD
};

template <int S, int R, int IN, bool OUT_ALPHA, int WX = 1>
__global__ __launch_bounds__(256 * WX, (AtrousTile<S, R, IN, WX>::WAVES_PER_SIMD)) void svgf_atrous_lds_kernel(AtrousArgs a)
{
    using T = AtrousTile<S, R, IN, WX>;
    constexpr int BW = T::BW, BH = T::BH, COLS = T::COLS, ROWS = T::ROWS, TOTAL = T::TOTAL, NLOAD = T::NLOAD, THREADS = T::THREADS;
    constexpr int XS = T::XS, XM = T::XM, SPAN = T::SPAN;
    static_assert(IN != kInFused || (S == 1 && WX == 1), "the fused temporal staging is level 0");
    extern __shared__ float4 lds[];
    float4* __restrict__ A = lds;               // {r, g, b, lum}
    float4* __restrict__ B = lds + ROWS * COLS; // {nx, ny, nz, z}
    float2* __restrict__ V = reinterpret_cast<float2*>(lds + 2 * ROWS * COLS); // kInFused: {variance as stored (fp16), alpha} [BH][BW]

#if NEB_ATROUS_STAMPS
    uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t t_begin_real = __builtin_amdgcn_s_memrealtime();
    uint64_t tprev = __builtin_amdgcn_s_memtime();
    const uint64_t t_begin = tprev;
    uint32_t ntiles = 0;
#endif
    const float4* __restrict__ src = a.src;
    const float4* __restrict__ geometry = a.geometry;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (wave-uniform: row bases and row addresses become scalar)
    const int rg = wv & 3, cb = wv >> 2; // the wave's row group (R lattice rows) and 64-column block of the tile

    // Tile walk, XCD-aware: workgroups b, b+8, ... share an XCD (and its L2); each XCD takes a contiguous
    // run of tiles, and the workgroups of an XCD interleave inside that run.  Speed only.
    // (Handing the tiles out dynamically instead -- one ticket counter per XCD, tickets requested a tile ahead -- was
    // measured at 46 us per level against 35: a device-scope atomic is a round trip to the memory side of the fabric, and
    // the static split is already balanced where it matters, per CU: 8 tiles on all but 8 CUs against 7.97 on average.)
    const uint32_t per_xcd = (a.nblocks + 7u) >> 3;
    const uint32_t xcd = blockIdx.x & 7u, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const uint32_t t_end = min((xcd + 1u) * per_xcd, a.nblocks);

    // Byte offsets of this lane's staged texels from the element Tile::first.  A tile whose staged rectangle lies inside
    // the image (and the resident rows) -- nine in ten -- needs no clamping: the offsets are the same for every such tile
    // and its loads are "uniform tile base + offset" (scalar base, 32-bit vector offset: no address arithmetic on the VALU
    // at all).  An edge tile overwrites them with its clamped positions (the reference's edge rule), counted from the
    // plane's first element; the next interior tile puts the regular ones back.
    uint32_t toff[NLOAD];
    bool toff_regular = false;
    float4 pc[IN == kInClassic ? NLOAD : 1];

    struct Tile {
        int r, jbase, x0;
        size_t first; // element index the offsets in toff[] count from
    };
    auto tile_origin = [&](uint32_t t, Tile& o) -> bool {
        // Tile order inside an XCD's run.  The levels: along the lattice fastest, then columns, then residues -- the tiles that share
        // staged rows (j and j + 1 of one column and residue: 4 of a tile's 12 rows) and columns (neighbouring column tiles) are in
        // flight on the same XCD at the same time and its L2 serves the second reader (S = 16: 30.6 us against 32.7; 8: 28.0 / 28.8;
        // 4: 27.9 / 28.7).  Level 0 with the temporal pass, bound by memory, keeps columns fastest: narrow vertical runs cost it
        // 53.2 us against 50.8 with 16-row tiles, 50.1 against 47.6 with 8-row ones (short row segments of many rows: the DRAM pages).  Blocks of 2 / 4 / 8 tiles along the lattice with
        // the columns in between lose at the wide steps (the padded tile count unbalances the XCDs).
        int tx_tile, jt;
        if constexpr (IN == kInFused) {
            tx_tile = (int)(t % (uint32_t)a.tiles_x);
            const int rest = (int)(t / (uint32_t)a.tiles_x);
            o.r = rest % S; // residue class of the lattice rows
            jt = rest / S;  // tile index along the lattice
        } else {
            jt = (int)(t % (uint32_t)a.tiles_j);
            const int rest = (int)(t / (uint32_t)a.tiles_j);
            tx_tile = rest % a.tiles_x;
            o.r = rest / a.tiles_x;
        }
        const int jmin = (a.row0 - o.r + S - 1) / S > 0 ? (a.row0 - o.r + S - 1) / S : 0; // first lattice index inside [row0,row1)
        o.jbase = jmin + jt * BH;
        o.x0 = (tx_tile / XM) * SPAN + (tx_tile % XM) * XS;
        return o.r + S * o.jbase < a.row1; // false: whole tile below the row range
    };
    // fills toff[] / o.first for a tile about to be loaded
    auto tile_offsets = [&](Tile& o) {
        const int y_first = o.r + S * (o.jbase - 2), y_last = o.r + S * (o.jbase + ROWS - 3), x_first = o.x0 + T::xcol(0);
        const bool interior = y_first >= max(0, a.row_begin) && y_last < min(a.H, a.row_end) && x_first >= 0 && o.x0 + T::xcol(COLS - 1) < a.W;
        if (interior) {
            o.first = (size_t)(y_first - a.row_begin) * a.W + x_first;
            if (!toff_regular) {
                // (the lane index made opaque: computed from threadIdx.x these loop-invariant offsets are hoisted out of the tile
                // loop and held in NLOAD more registers for the whole kernel -- which the fused kernel does not have)
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
#pragma unroll
                for (int k = 0; k < NLOAD; ++k) {
                    const int i = tid + THREADS * k;
                    const int lr = i / COLS;
                    toff[k] = (uint32_t)(lr * S * a.W + (T::xcol(i - lr * COLS) - T::xcol(0))) * 16u;
                }
                toff_regular = true;
            }
        } else {
            o.first = 0;
            toff_regular = false;
            int tid = threadIdx.x; // (opaque for the same reason: lr and lc of every k would be held for the whole kernel)
            asm volatile("" : "+v"(tid));
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const int i = tid + THREADS * k;
                const int lr = i / COLS, lc = i - lr * COLS;
                // clamp to the image (svgf_atrous.hlsl:65), then to the resident rows: a partial tile also stages rows
                // that no valid output taps; on a row strip those may lie outside the allocation, so they are
                // redirected to a resident row (their values are never used)
                int y = min(max(o.r + S * (o.jbase + lr - 2), 0), a.H - 1);
                y = min(max(y, a.row_begin), a.row_end - 1);
                const int x = min(max(o.x0 + T::xcol(lc), 0), a.W - 1);
                toff[k] = (uint32_t)((y - a.row_begin) * a.W + x) * 16u; // (a plane is < 4 GB: checked at launch)
            }
        }
    };
    // kInClassic: the next tile's radiance, into registers
    auto issue_load = [&](int k, const Tile& o) {
        if constexpr (IN == kInClassic) {
            if (threadIdx.x + THREADS * k < TOTAL)
                pc[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(src + o.first) + toff[k]);
        }
    };
    // a 16-byte plane's texels of the tile, global -> LDS directly.  Issued once the whole workgroup has finished reading
    // the previous tile, and complete before the barrier that precedes the filter.
    auto issue_dma = [&](const float4* plane, float4* dst_lds, const Tile& o) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            if (threadIdx.x + THREADS * k < TOTAL) {
                const char* g = reinterpret_cast<const char*>(plane + o.first) + toff[k];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(dst_lds + THREADS * k + 64 * wv), 16, 0, 0);
            }
        }
    };

    // The centre pixel's variance and alpha (the per-pixel inputs that are not staged) are fetched one tile ahead as
    // well: loaded at the start of the filter phase they were a dependent global round trip in front of every tile's
    // arithmetic (2.5 us per level).  (kInFused has both in LDS.)
    uint32_t nvar[R];
    float nalpha[R];
    auto issue_centre_loads = [&](const Tile& o) {
        if constexpr (IN != kInFused) {
            const int xo = o.x0 + T::xcol(cb * 64 + lane + 2 * XS);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int yo = o.r + S * (o.jbase + rg * R + k);
                nvar[k] = 0u;
                nalpha[k] = 0.f;
                if (xo < a.Wd && yo < a.row1) {
                    const size_t i = (size_t)(yo - a.row_begin) * a.W + xo;
                    nvar[k] = a.variance[i];
                    if constexpr (OUT_ALPHA)
                        nalpha[k] = reinterpret_cast<const float*>(IN == kInClassic ? src : a.alpha_src)[4 * i + 3];
                }
            }
        }
    };

    // (Re-measured in round 3 with the co-residency known -- blocks b, b + 256, b + 512 share a CU: the 2nd / 3rd workgroup asking for
    // its first tile 1 / 2 us later than the first changes nothing measurable, 183-189 us per frame against 186.)
    uint32_t t = xcd * per_xcd + wg_in_xcd;
    Tile nt{0, 0, 0, 0};
    bool have = false;
    while (t < t_end && !(have = tile_origin(t, nt)))
        t += wgs_per_xcd;
    if (have) {
        tile_offsets(nt);
#pragma unroll
        for (int k = 0; k < NLOAD; ++k)
            issue_load(k, nt);
        issue_centre_loads(nt);
    }

    // per-launch constants of the tap arithmetic as VGPR operands (in isolation an SGPR source costs a v_fma about 0.6 ns more
    // than a third VGPR source does, tools/ubench_bank.hip: 1.83 against 1.25 ns per wave-instruction at three waves per SIMD;
    // in this kernel the difference does not show: 192.5 against 192.0 us per frame)
    float cz = a.cz, phiN = a.phiNormal;
    // log2(Kx Ky) + phiN of the six tap classes (|dx|, |dy| in {0, 1, 2}), set up once: computed per tap it would be an add each.
    // The levels of the fused chain hold all eight constants in vector registers.  The fused kernel and the kernels that prefetch
    // the next tile's radiance into registers run at their register budget (with the constants in registers the 16-row fused kernel
    // spilled five, and a kernel that needs scratch memory right after the GI kernels, which use theirs at another size, waits for
    // the queue's scratch set-up: 57.7 us per launch against 50.9): there they are scalar operands.
    float lkp[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            lkp[i][j] = tap_constant(phiN, i, j);
            if constexpr (IN != kInLum || R == 4) // (R = 4: two pairs of rows per lane leave no registers for the constants either)
                lkp[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lkp[i][j])));
            else
                asm volatile("" : "+v"(lkp[i][j]));
            lkp[j][i] = lkp[i][j];
        }
    if constexpr (IN == kInLum && R != 4)
        asm volatile("" : "+v"(cz), "+v"(phiN));
    // the packed tap's constants: {phiNormal, cz}, and per |dx| the pairs {lkp[.][0], lkp[.][1]} and {lkp[.][1], lkp[.][2]} (read straight
    // or swapped by op_sel): 14 registers where the scalar form holds 8
    constexpr bool kPacked = NEB_ATROUS_PK && (R == 2 || R == 4) && (IN == kInLum || (IN == kInFused && NEB_ATROUS_PK_FUSED) || (IN == kInClassic && NEB_ATROUS_PK_CLASSIC));
    constexpr bool kLkScalar = IN != kInLum || R == 4;
    constexpr bool kPhased = NEB_ATROUS_PK != 3 && IN == kInLum && R != 4; // (the fused kernel has no registers for a whole group's intermediate values)
    neb_f2 pcz = {phiN, cz}, lkq[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        lkq[i][0] = (neb_f2){lkp[i][0], lkp[i][1]};
        lkq[i][1] = (neb_f2){lkp[i][1], lkp[i][2]};
        if constexpr (kPacked && !kLkScalar)
            asm volatile("" : "+v"(lkq[i][0]), "+v"(lkq[i][1]));
    }
    if constexpr (kPacked && !kLkScalar)
        asm volatile("" : "+v"(pcz));

    NEB_STAMP(0);
    while (have) {
#if NEB_ATROUS_PRIO
        // The SIMD's arbiter serves its oldest wave first: of the three workgroups of a CU the first-dispatched one races
        // ahead and the youngest is left to finish alone, one wave per SIMD, at half the issue rate.  Priority = tiles still
        // to do lets the workgroups finish together.
        {
            const uint32_t remaining = (t_end - t + wgs_per_xcd - 1u) / wgs_per_xcd; // (this tile included)
            // (four levels: with 8-row tiles a workgroup has four tiles or more -- 28.7 / 27.4 / 28.0 / 30.6 us for S = 2 .. 16 against
            // 29.0 / 28.0 / 28.3 / 30.9 with "three and more" as one level; no priorities at all: 32.0 / 30.7 / 31.5 / 34.5)
            if (remaining >= 4u)
                __builtin_amdgcn_s_setprio(3);
            else if (remaining == 3u)
                __builtin_amdgcn_s_setprio(2);
            else if (remaining == 2u)
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
        }
#endif
        const int cr = nt.r, cjbase = nt.jbase, cx0 = nt.x0;
        // ---- stage the tile ----
        if constexpr (IN == kInClassic) {
            issue_dma(geometry, B, nt); // first: its flight overlaps the wait for the prefetched radiance and the luminance arithmetic
            float lum[NLOAD];
#pragma unroll
            for (int k = 0; k < NLOAD; ++k)
                lum[k] = luminance(pc[k].x, pc[k].y, pc[k].z);
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const int i = threadIdx.x + THREADS * k;
                if (i < TOTAL)
                    A[i] = make_float4(pc[k].x, pc[k].y, pc[k].z, lum[k]);
            }
        } else if constexpr (IN == kInLum) {
            issue_dma(src, A, nt);
            issue_dma(geometry, B, nt);
        } else {
            // temporal accumulation as the staging step: cur -> A's slots, history -> B's slots by DMA; depth and normals of
            // both frames (and the moments history of the tile's own pixels) through registers; then every thread turns ITS
            // texels (the ones its own wave's DMA wrote: no barrier needed, only the wave's own vmcnt) into {accumulated
            // rgb, lum} and {decoded normal, depth} in place.
            issue_dma(src, A, nt);
            issue_dma(a.rad_hist, B, nt);
            uint32_t dc[NLOAD], dh[NLOAD], nc[NLOAD], nh[NLOAD], mh[NLOAD];
            bool own[NLOAD];
            size_t gi[NLOAD];
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const int i = threadIdx.x + THREADS * k;
                own[k] = false;
                if (i < TOTAL) {
                    const size_t e = nt.first + (toff[k] >> 4);
                    gi[k] = e;
                    dc[k] = a.depth_cur[e];
                    dh[k] = a.depth_hist[e];
                    nc[k] = a.normal_cur[e].y; // .zw = shading normal
                    nh[k] = a.normal_hist[e].y;
                    const int lr = i / COLS, lc = i - lr * COLS;
                    const int x = cx0 + T::xcol(lc), y = cr + S * (cjbase + lr - 2);
                    // one of the tile's own output pixels (never a clamped position: those lie outside the image)
                    own[k] = lr >= 2 && lr < BH + 2 && lc >= 2 * XS && lc < BW + 2 * XS && x < a.Wd && y >= a.row0 && y < a.row1;
                    mh[k] = own[k] ? a.mom_hist[e] : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const int i = threadIdx.x + THREADS * k;
                if (i < TOTAL) {
                    const float4 Cc = A[i], Ch = B[i];
                    float3 Nc;
                    float zc;
                    const float alpha = temporal_blend(dc[k], dh[k], nc[k], nh[k], a.t_neg_inv_two_sigma2_log2e, a.t_alpha, Nc, zc);
                    const float4 acc = temporal_accumulate(Cc, Ch, alpha);
                    const float4 geo = make_float4(Nc.x, Nc.y, Nc.z, zc);
                    A[i] = make_float4(acc.x, acc.y, acc.z, luminance(acc.x, acc.y, acc.z));
                    B[i] = geo;
                    if (own[k]) {
                        const uint2 mv = temporal_moments(Cc, mh[k], alpha, a.t_varianceEps);
                        a.mom_cur[gi[k]] = mv.x;
                        a.variance_out[gi[k]] = (uint16_t)mv.y;
                        a.geometry_out[gi[k]] = geo;
                        const int lr = i / COLS, lc = i - lr * COLS;
                        V[(lr - 2) * BW + (lc - 2 * XS)] = make_float2(half_bits_to_float(mv.y), Cc.w); // (the variance as the levels read it back)
                    }
                }
            }
        }
        NEB_STAMP(1);
        __syncthreads(); // (also waits for this wave's DMA: an LDS-DMA is a pending LDS write on the VM counter)
        NEB_STAMP(2);

        // ---- next tile: issue its loads now, consume them after this tile is filtered ----
        uint32_t cvar[R];
        float calpha[R];
#pragma unroll
        for (int k = 0; k < R; ++k)
            cvar[k] = nvar[k], calpha[k] = nalpha[k];
        bool have_next = false;
        t += wgs_per_xcd;
        while (t < t_end && !(have_next = tile_origin(t, nt)))
            t += wgs_per_xcd;
        if (have_next) {
            tile_offsets(nt); // (the current tile's DMA has been issued: toff[] is free)
            issue_centre_loads(nt);
        }
        // The next tile's loads are spread over the row iterations of the filter below (one batch per row) so
        // that, chip-wide, memory traffic and arithmetic overlap instead of alternating in bursts; batches
        // beyond the R+4 row iterations go out first.
        if (have_next) {
#pragma unroll
            for (int k = R + 4; k < NLOAD; ++k)
                issue_load(k, nt);
        }

        NEB_STAMP(3);
        // ---- filter the current tile ----
        const int xo = cx0 + T::xcol(cb * 64 + lane + 2 * XS);
        float z0[R], n0x[R], n0y[R], n0z[R], lum0[R], cl[R], alpha0[R];
        float sr[R], sg[R], sb[R], sw[R];
        bool valid[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int lr = rg * R + k + 2;
            const float4 cA = A[lr * COLS + cb * 64 + lane + 2 * XS];
            const float4 cB = B[lr * COLS + cb * 64 + lane + 2 * XS];
            z0[k] = cB.w;
            n0x[k] = 0.5f * cB.x; // halved: see half_dot_max0
            n0y[k] = 0.5f * cB.y;
            n0z[k] = 0.5f * cB.z;
            lum0[k] = cA.w;
            const int yo = cr + S * (cjbase + rg * R + k);
            valid[k] = (xo < a.Wd) && (yo < a.row1);
            if constexpr (IN == kInFused) {
                const float2 va = V[(rg * R + k) * BW + lane]; // (a pixel that is not valid holds stale values: never stored)
                alpha0[k] = va.y;
                cl[k] = lum_scale(va.x, a.phiColor);
            } else {
                alpha0[k] = calpha[k];
                cl[k] = lum_scale(half_bits_to_float((uint16_t)cvar[k]), a.phiColor); // (an invalid pixel holds 0: never stored)
            }
            sr[k] = sg[k] = sb[k] = sw[k] = 0.f;
        }
        // LDS reads software-pipelined by hand: a row's ten ds_read_b128 go out in two groups (dx = -2..0, dx = 1..2), each
        // one group ahead of its taps, into the registers the group before last has just released -- left to itself the
        // compiler (at its register budget) issues each read right in front of its use and every wave pays the LDS round
        // trip five times per row.
        float4 gA[2][3], gB[2][3];
        auto load_group = [&](int g) {
            const int ir = g >> 1, lrow_base = (rg * R + ir) * COLS + cb * 64 + lane + 2 * XS;
            if ((g & 1) == 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
#if NEB_ATROUS_FEWER_LDS_READS == 1 // timing only (wrong results, and the compiler merges the identical taps: not a clean probe): one texel read per group instead of three / two -- what ANY saving of LDS reads could buy
                    if (j > 0) {
                        gA[0][j] = gA[0][0], gB[0][j] = gB[0][0];
                        continue;
                    }
#endif
                    gA[0][j] = A[lrow_base + (j - 2) * XS];
#if NEB_ATROUS_FEWER_LDS_READS == 2 // timing only (wrong results): the B plane read as 8 bytes instead of 16 -- a quarter of the LDS bytes gone, every VALU instruction still there
                    {
                        const float2 b2 = *reinterpret_cast<const float2*>(&B[lrow_base + (j - 2) * XS]);
                        gB[0][j] = make_float4(b2.x, b2.y, gA[0][j].z, gA[0][j].w);
                    }
#else
                    gB[0][j] = B[lrow_base + (j - 2) * XS];
#endif
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#if NEB_ATROUS_FEWER_LDS_READS == 1
                    if (j > 0) {
                        gA[1][j] = gA[1][0], gB[1][j] = gB[1][0];
                        continue;
                    }
#endif
                    gA[1][j] = A[lrow_base + (j + 1) * XS];
#if NEB_ATROUS_FEWER_LDS_READS == 2
                    {
                        const float2 b2 = *reinterpret_cast<const float2*>(&B[lrow_base + (j + 1) * XS]);
                        gB[1][j] = make_float4(b2.x, b2.y, gA[1][j].z, gA[1][j].w);
                    }
#else
                    gB[1][j] = B[lrow_base + (j + 1) * XS];
#endif
                }
            }
        };
        if constexpr (kPacked) {
            // the lane's output rows side by side in PAIRS (see tap2_geometry): pair p = rows 2 p, 2 p + 1; staged row ir is row lir = ir - 2 p of the
            // pair's own six, tap dy = lir - 2 of its first row and lir - 3 of its second.  (R = 4, round 5: two pairs per lane -- a staged texel is
            // read from LDS once for up to four output rows, 20 reads per pixel instead of 30 -- at the price of 16-row tiles and their registers.)
            constexpr int NP = R / 2;
            neb_f2 h0x[NP], h0y[NP], h0z[NP], nz0[NP], nl0[NP], cl2[NP], sr2[NP], sg2[NP], sb2[NP], sw2[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                h0x[p] = (neb_f2){n0x[2 * p], n0x[2 * p + 1]}, h0y[p] = (neb_f2){n0y[2 * p], n0y[2 * p + 1]}, h0z[p] = (neb_f2){n0z[2 * p], n0z[2 * p + 1]};
                nz0[p] = (neb_f2){-z0[2 * p], -z0[2 * p + 1]}, nl0[p] = (neb_f2){-lum0[2 * p], -lum0[2 * p + 1]}, cl2[p] = (neb_f2){cl[2 * p], cl[2 * p + 1]};
                sr2[p] = sg2[p] = sb2[p] = sw2[p] = (neb_f2){0.f, 0.f};
            }
            load_group(0);
#pragma unroll
            for (int g = 0; g < 2 * (R + 4); ++g) {
                const int ir = g >> 1;
                if ((g & 1) == 0 && ir < NLOAD && have_next)
                    issue_load(ir, nt);
                if (g + 1 < 2 * (R + 4))
                    load_group(g + 1);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int kMaxGroup = 3;
                const int ng = (g & 1) ? 2 : 3; // texels of this group: dx = -2, -1, 0 or 1, 2
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int lir = ir - 2 * p;
                    if (lir < 0 || lir > 5)
                        continue;
                    if (lir == 0 || lir == 5) { // taps of one row only (dy = -2 of the pair's first row, dy = +2 of its second)
                        const int k = lir == 0 ? 0 : 1;
#pragma unroll
                        for (int j = 0; j < ng; ++j) {
                            const int dx = (g & 1) ? j + 1 : j - 2, adx = dx < 0 ? -dx : dx;
                            const float4 tA = gA[g & 1][j];
                            const float4 tB = gB[g & 1][j];
                            const float w = tap_weight_neg(h0x[p][k], h0y[p][k], h0z[p][k], nz0[p][k], nl0[p][k], cl2[p][k], tA, tB, pcz.x, pcz.y, lkq[adx][1].y);
                            sr2[p][k] = fmaf(w, tA.x, sr2[p][k]);
                            sg2[p][k] = fmaf(w, tA.y, sg2[p][k]);
                            sb2[p][k] = fmaf(w, tA.z, sb2[p][k]);
                            sw2[p][k] += w;
                        }
                    } else if constexpr (kLkScalar) {
                        // the fused kernel: one texel at a time (its staging phase leaves no registers for a whole group's intermediate values)
#pragma unroll
                        for (int j = 0; j < ng; ++j) {
                            const int dx = (g & 1) ? j + 1 : j - 2, adx = dx < 0 ? -dx : dx;
                            const float4 tA = gA[g & 1][j];
                            const Tap2 t = tap2_geometry(h0x[p], h0y[p], h0z[p], nz0[p], nl0[p], tA, gB[g & 1][j]);
                            const neb_f2 w = (lir == 1)   ? tap2_weight<false, true>(t, cl2[p], pcz, lkq[adx][1])
                                             : (lir == 2) ? tap2_weight<false, true>(t, cl2[p], pcz, lkq[adx][0])
                                             : (lir == 3) ? tap2_weight<true, true>(t, cl2[p], pcz, lkq[adx][0])
                                                          : tap2_weight<true, true>(t, cl2[p], pcz, lkq[adx][1]);
                            const neb_f2 axy = {tA.x, tA.y}, azw = {tA.z, tA.w};
                            sr2[p] = pk_fma(w, __builtin_shufflevector(axy, axy, 0, 0), sr2[p]);
                            sg2[p] = pk_fma(w, __builtin_shufflevector(axy, axy, 1, 1), sg2[p]);
                            sb2[p] = pk_fma(w, __builtin_shufflevector(azw, azw, 0, 0), sb2[p]);
                            sw2[p] += w;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        Tap2 t[kMaxGroup];
                        neb_f2 w[kMaxGroup];
#pragma unroll
                        for (int j = 0; j < ng; ++j)
                            t[j] = tap2_geometry(h0x[p], h0y[p], h0z[p], nz0[p], nl0[p], gA[g & 1][j], gB[g & 1][j]);
                        if constexpr (kPhased)
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < ng; ++j) {
                            const int dx = (g & 1) ? j + 1 : j - 2, adx = dx < 0 ? -dx : dx;
                            // (|dy0|, |dy1|) = (1, 2), (0, 1), (1, 0), (2, 1) for lir = 1 .. 4: the constant pairs {0, 1} and {1, 2}, straight or swapped
                            w[j] = (lir == 1)   ? tap2_weight<false, kLkScalar>(t[j], cl2[p], pcz, lkq[adx][1])
                                   : (lir == 2) ? tap2_weight<false, kLkScalar>(t[j], cl2[p], pcz, lkq[adx][0])
                                   : (lir == 3) ? tap2_weight<true, kLkScalar>(t[j], cl2[p], pcz, lkq[adx][0])
                                                : tap2_weight<true, kLkScalar>(t[j], cl2[p], pcz, lkq[adx][1]);
                        }
                        if constexpr (kPhased)
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < ng; ++j) {
                            const float4 tA = gA[g & 1][j];
                            const neb_f2 axy = {tA.x, tA.y}, azw = {tA.z, tA.w};
                            sr2[p] = pk_fma(w[j], __builtin_shufflevector(axy, axy, 0, 0), sr2[p]);
                            sg2[p] = pk_fma(w[j], __builtin_shufflevector(axy, axy, 1, 1), sg2[p]);
                            sb2[p] = pk_fma(w[j], __builtin_shufflevector(azw, azw, 0, 0), sb2[p]);
                            sw2[p] += w[j];
                        }
                    }
                    asm volatile("" : "+v"(sr2[p]), "+v"(sg2[p]), "+v"(sb2[p]), "+v"(sw2[p])); // (pinned for the reason given below)
                    if constexpr (NP > 1)
                        __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                sr[k] = sr2[k >> 1][k & 1];
                sg[k] = sg2[k >> 1][k & 1];
                sb[k] = sb2[k >> 1][k & 1];
                sw[k] = sw2[k >> 1][k & 1];
            }
        } else {
            load_group(0);
#pragma unroll
            for (int g = 0; g < 2 * (R + 4); ++g) {
                const int ir = g >> 1;
                if ((g & 1) == 0 && ir < NLOAD && have_next)
                    issue_load(ir, nt);
                if (g + 1 < 2 * (R + 4))
                    load_group(g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < ((g & 1) ? 2 : 3); ++j) {
                    const int dx = (g & 1) ? j + 1 : j - 2;
                    const float4 tA = gA[g & 1][j];
                    const float4 tB = gB[g & 1][j];
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const int dy = ir - k - 2;
                        if (dy < -2 || dy > 2)
                            continue;
                        const float w = tap_weight(n0x[k], n0y[k], n0z[k], z0[k], lum0[k], cl[k], tA, tB, phiN, cz, lkp[dx < 0 ? -dx : dx][dy < 0 ? -dy : dy]);
                        sr[k] = fmaf(w, tA.x, sr[k]);
                        sg[k] = fmaf(w, tA.y, sg[k]);
                        sb[k] = fmaf(w, tA.z, sb[k]);
                        sw[k] += w;
                    }
                }
                // Pin the partial sums here: they only feed the predicated store below, so LLVM would
                // otherwise sink ALL the arithmetic under that branch and keep every staged texel live
                // (spilling ~1.3 KB per lane).  The sched_barrier keeps one group's ds_reads per region.
#pragma unroll
                for (int k = 0; k < R; ++k)
                    asm volatile("" : "+v"(sr[k]), "+v"(sg[k]), "+v"(sb[k]), "+v"(sw[k]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        NEB_STAMP(4);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (!valid[k])
                continue;
            const int yo = cr + S * (cjbase + rg * R + k);
            const float inv = fast_rcp(fmaxf(sw[k], 1e-4f)); // :84
            const float r = sr[k] * inv, g = sg[k] * inv, b = sb[k] * inv;
            store_output(a.dst + ((size_t)(yo - a.row_begin) * a.W + xo), make_float4(r, g, b, OUT_ALPHA ? alpha0[k] : luminance(r, g, b)));
        }
        NEB_STAMP(5);
        have = have_next;
        if (have)
            __syncthreads(); // everyone is done reading LDS before the next tile overwrites it
        NEB_STAMP(6);
#if NEB_ATROUS_STAMPS
        ++ntiles;
#endif
    }
#if NEB_ATROUS_STAMPS
    if (lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wv) * 16;
        for (int i = 0; i < 7; ++i)
            o[i] = st[i];
        o[7] = __builtin_amdgcn_s_memtime() - t_begin;
        o[8] = t_begin_real;
        o[9] = __builtin_amdgcn_s_memrealtime();
        o[10] = ntiles;
        o[11] = __builtin_amdgcn_s_getreg(6164 /* HW_REG_XCC_ID (20), offset 0, size 4: ((4-1)<<11)|20 */);
        o[12] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4 /* HW_REG_HW_ID */);
        o[13] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 6 /* HW_REG_LDS_ALLOC */);
    }
#endif
}

#if NEB_ATROUS_STAMPS
// one stamp block per step (log2 S = 0..5): 2048 workgroups x 4 waves x 16 words; read back by neb_debug_atrous_stamps
static unsigned long long* g_stamp_buf = nullptr;
static uint32_t g_stamp_grid[6] = {0, 0, 0, 0, 0, 0};
static constexpr size_t kStampWords = 2048 * 4 * 16;
static unsigned long long* atrous_stamp_buffer(int S, uint32_t grid)
{
    if (!g_stamp_buf && hipMalloc(&g_stamp_buf, 6 * kStampWords * 8) != hipSuccess)
        return nullptr;
    int l = 0;
    while ((1 << l) < S)
        ++l;
    g_stamp_grid[l] = grid;
    return g_stamp_buf + (size_t)l * kStampWords;
}
extern "C" int neb_debug_atrous_stamps(unsigned long long* host, uint32_t* grids)
{
    if (!g_stamp_buf)
        return -1;
    (void)hipDeviceSynchronize();
    for (int l = 0; l < 6; ++l)
        grids[l] = g_stamp_grid[l];
    return hipMemcpy(host, g_stamp_buf, 6 * kStampWords * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

// (Measured and dropped: a double-buffered LDS variant that weaves the staging of tile i+1 into the row loop of
// tile i, one barrier per tile -- 45-94 us per level against 41-49 us for the kernel above; the longer live
// ranges cost more than the barrier and the exposed decode they remove.  Round 3: the geometry texels prefetched into
// registers like the radiance instead of the DMA at the tile boundary: 45 us per level against 35.)
template <int S, int R, int IN, bool OUT_ALPHA, int WX = 1>
static hipError_t launch_lds(AtrousArgs a, int device, int num_cus, hipStream_t s)
{
    using T = AtrousTile<S, R, IN, WX>;
    constexpr size_t lds_bytes = (size_t)T::LDS_BYTES;
    // the dynamic-LDS limit is a per-device function attribute: remember which devices have it (one bit each; a
    // device ordinal beyond the mask just sets it on every launch)
    static std::atomic<uint64_t> attr_set{0};
    const uint64_t bit = (device >= 0 && device < 64) ? (1ull << device) : 0ull;
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&svgf_atrous_lds_kernel<S, R, IN, OUT_ALPHA, WX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    a.tiles_x = ((a.Wd + T::SPAN - 1) / T::SPAN) * T::XM;
    const int max_lattice_rows = (a.row1 - a.row0 + S - 1) / S; // per residue class, upper bound
    a.tiles_j = (max_lattice_rows + T::BH - 1) / T::BH;
    a.nblocks = (uint32_t)a.tiles_x * (uint32_t)S * (uint32_t)a.tiles_j;
    // persistent grid: as many workgroups as fit (LDS-limited, at most 3 per CU by the launch bounds)
    static_assert(T::PER_CU >= 1, "tile too large for the LDS");
    const uint32_t per_cu = (uint32_t)T::PER_CU;
    uint32_t grid = (uint32_t)num_cus * (per_cu ? per_cu : 1u);
    if (grid > a.nblocks)
        grid = a.nblocks;
    grid = ((grid + 7u) / 8u) * 8u;
#if NEB_ATROUS_STAMPS
    a.stamps = atrous_stamp_buffer(S, grid);
#endif
    hipLaunchKernelGGL((svgf_atrous_lds_kernel<S, R, IN, OUT_ALPHA, WX>), dim3(grid), dim3(T::THREADS), lds_bytes, s, a);
    return hipGetLastError();
}

// R = 2 rows per lane at every step (12 staged rows of 68 - 96 columns: 26 - 37 KB, 4 workgroups per CU, 128 registers per lane),
// as measured -- the fused temporal + level-0 kernel included (NEB_ATROUS_R_FUSED)
template <int IN, bool OUT_ALPHA>
static hipError_t launch_lds_step(const AtrousArgs& a, uint32_t step, int device, int num_cus, hipStream_t s)
{
    switch (step) {
    case 1: return launch_lds<1, NEB_ATROUS_R_NARROW, IN, OUT_ALPHA>(a, device, num_cus, s);
    case 2: return launch_lds<2, NEB_ATROUS_R_NARROW, IN, OUT_ALPHA>(a, device, num_cus, s);
    case 4: return launch_lds<4, NEB_ATROUS_R_NARROW, IN, OUT_ALPHA>(a, device, num_cus, s);
    case 8: return launch_lds<8, 2, IN, OUT_ALPHA, NEB_ATROUS_WX8>(a, device, num_cus, s);
    case 16: return launch_lds<16, 2, IN, OUT_ALPHA, NEB_ATROUS_WX16>(a, device, num_cus, s);
    case 32: return launch_lds<32, 2, IN, OUT_ALPHA, NEB_ATROUS_WX16>(a, device, num_cus, s);
    default: return hipErrorInvalidValue;
    }
}

bool atrous_lds_serves(const SvgfLaunch& L, int variant, uint32_t step)
{
    // wider steps do not fit the LDS tile; the LDS kernel addresses a plane with 32-bit byte offsets: a resident plane of
    // 4 GB or more -- 16 k x 16 k -- takes the direct kernel
    return variant >= 1 && step <= 32u && (step & (step - 1u)) == 0u && (uint64_t)(L.row_end - L.row_begin) * L.W < (1ull << 28);
}

static bool atrous_args(const SvgfLaunch& L, uint32_t step, const float4* src, float4* dst, const uint16_t* variance, const float4* geometry,
                        AtrousArgs& a)
{
    const uint32_t Wd = (L.W / 8u) * 8u, Hd = (L.H / 8u) * 8u; // SVGFDenoiser.cpp:185
    const uint32_t row1 = L.row1 < Hd ? L.row1 : Hd;
    if (L.row0 >= row1 || Wd == 0)
        return false;
    a = AtrousArgs{};
    a.src = src;
    a.dst = dst;
    a.variance = variance;
    a.geometry = geometry;
    a.W = (int)L.W;
    a.H = (int)L.H;
    a.Wd = (int)Wd;
    a.row_begin = (int)L.row_begin;
    a.row_end = (int)L.row_end;
    a.row0 = (int)L.row0;
    a.row1 = (int)row1;
    a.step = (int)step;
    a.cz = kLog2e / (L.p.phiDepth * (float)step);
    a.phiColor = L.p.phiColor;
    a.phiNormal = L.p.phiNormal;
    return true;
}

hipError_t launch_atrous(const SvgfLaunch& L, int variant, uint32_t step, const float4* src, float4* dst,
                         const uint16_t* variance, const float4* geometry, hipStream_t s)
{
    AtrousArgs a;
    if (!atrous_args(L, step, src, dst, variance, geometry, a))
        return hipSuccess;
    if (atrous_lds_serves(L, variant, step))
        return launch_lds_step<kInClassic, true>(a, step, L.device, L.num_cus > 0 ? L.num_cus : 256, s);
    dim3 grid((a.Wd + 63) / 64, (a.row1 - a.row0 + 3) / 4);
    hipLaunchKernelGGL(svgf_atrous_direct_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_atrous_lum(const SvgfLaunch& L, uint32_t step, bool last, const float4* src, float4* dst, const uint16_t* variance,
                             const float4* geometry, hipStream_t s)
{
    AtrousArgs a;
    if (!atrous_args(L, step, src, dst, variance, geometry, a))
        return hipSuccess;
    a.alpha_src = dst; // the last level writes the plane that still holds the frame's input: its alpha is carried
    const int num_cus = L.num_cus > 0 ? L.num_cus : 256;
    // (the last level's alpha read from its destination -- 4 of every 16 bytes -- is hidden under the arithmetic: 35.9 us with it,
    // 35.6 without)
    return last ? launch_lds_step<kInLum, true>(a, step, L.device, num_cus, s) : launch_lds_step<kInLum, false>(a, step, L.device, num_cus, s);
}

hipError_t launch_atrous_fused_temporal(const SvgfLaunch& L, bool only_level, const float4* rad_cur, const float4* rad_hist, const uint32_t* depth_cur,
                                        const uint32_t* depth_hist, const uint2* normal_cur, const uint2* normal_hist, const uint32_t* mom_hist,
                                        uint32_t* mom_cur, uint16_t* variance, float4* geometry, float4* dst, hipStream_t s)
{
    AtrousArgs a;
    if (!atrous_args(L, 1u, rad_cur, dst, variance, geometry, a))
        return hipSuccess;
    a.rad_hist = rad_hist;
    a.depth_cur = depth_cur;
    a.depth_hist = depth_hist;
    a.normal_cur = normal_cur;
    a.normal_hist = normal_hist;
    a.mom_hist = mom_hist;
    a.mom_cur = mom_cur;
    a.variance_out = variance;
    a.geometry_out = geometry;
    a.t_neg_inv_two_sigma2_log2e = -kLog2e / (2.0f * L.p.depthSigma * L.p.depthSigma);
    a.t_alpha = L.p.alpha;
    a.t_varianceEps = L.p.varianceEps;
    const int num_cus = L.num_cus > 0 ? L.num_cus : 256;
    return only_level ? launch_lds<1, NEB_ATROUS_R_FUSED, kInFused, true>(a, L.device, num_cus, s) : launch_lds<1, NEB_ATROUS_R_FUSED, kInFused, false>(a, L.device, num_cus, s);
}

} // namespace neb

// svgf.hip -- SVGF temporal accumulation and edge-stopping a-trous wavelet for gfx950.
//
// Reference arithmetic (paths relative to the reference checkout):
//   assets/shaders/svgf_temporal.hlsl:24-68  -> svgf_temporal_kernel
//   assets/shaders/svgf_atrous.hlsl:29-85    -> svgf_atrous_direct_kernel / svgf_atrous_lds_kernel
// This is not a transliteration of the 8x8-threadgroup texture-fetch shaders: the temporal
// pass is a coalesced stream (one pixel per lane, 16-B radiance accesses); the a-trous pass
// is tiled on the row lattice {r + step*j} so that a workgroup's taps are whole, contiguous
// image-row segments whatever the step, staged once into LDS with depth and normals already
// decoded, and each lane filters a column of R lattice rows so every staged texel is
// read from LDS once per R outputs.  See DESIGN.md "Kernels".
#include "neb_device.h"
#include "neb_internal.h"

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
    uint32_t W, Wd, row_off, nrows; // row_off = row0 - row_begin
    float neg_inv_two_sigma2_log2e, alpha, varianceEps;
};

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

    const float3 Nc = oct16_unpack_zw(nc);
    const float3 Nh = oct16_unpack_zw(nh);
    const float dz = fabsf(depth_unorm24(dc) - depth_unorm24(dh));
    // SVGF_DWeight: exp(-dz^2 / (2 sigma^2))   (svgf_common.hlsli:11-15)
    const float wDepth = fast_exp2(dz * dz * a.neg_inv_two_sigma2_log2e);
    // SVGF_NWeight: saturate(dot)               (svgf_common.hlsli:4-7)
    const float wNormal = __saturatef(Nc.x * Nh.x + Nc.y * Nh.y + Nc.z * Nh.z);
    const float w = wDepth * wNormal;
    const float alpha = 1.0f + w * (a.alpha - 1.0f); // lerp(1, alpha, w)  (:51)

    const float Y = luminance(Cc.x, Cc.y, Cc.z);
    const float Mh0 = half_bits_to_float(mh & 0xffffu), Mh1 = half_bits_to_float(mh >> 16);
    const float M1 = Y + alpha * (Mh0 - Y);
    const float Y2 = Y * Y;
    const float M2 = Y2 + alpha * (Mh1 - Y2);
    const float var = fmaxf(M2 - M1 * M1, a.varianceEps);

    float4 out;
    out.x = Cc.x + alpha * (Ch.x - Cc.x);
    out.y = Cc.y + alpha * (Ch.y - Cc.y);
    out.z = Cc.z + alpha * (Ch.z - Cc.z);
    out.w = Cc.w;
    a.rad_cur[i] = out;
    a.mom_cur[i] = float_to_half_bits(M1) | (float_to_half_bits(M2) << 16);
    a.variance[i] = (uint16_t)float_to_half_bits(var);
}

hipError_t launch_temporal(const SvgfLaunch& L, float4* rad_cur, const float4* rad_hist, const uint32_t* depth_cur,
                           const uint32_t* depth_hist, const uint2* normal_cur, const uint2* normal_hist,
                           const uint32_t* mom_hist, uint32_t* mom_cur, uint16_t* variance, hipStream_t s)
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

// ------------------------------------------------------------------------------------------
// A-trous: 5x5 taps at spacing `step`, edge-stopped by depth, normal and luminance.
// ------------------------------------------------------------------------------------------
struct AtrousArgs {
    const float4* src;
    float4* dst;
    const uint16_t* variance;
    const uint32_t* depth;
    const uint2* normal;
    int W, H, Wd;               // image size and floor-dispatched width
    int row_begin, row_end;     // resident image rows [row_begin, row_end)
    int row0, row1;             // image rows to write (row1 already clipped to (H/8)*8)
    int step;
    int tiles_x, tiles_j;       // LDS kernel: tiles per row segment / per residue class
    uint32_t nblocks;           // real block count (grid is padded to a multiple of 8)
    float cz;                   // log2e / (phiDepth * step)
    float phiColor, phiNormal;
};

// K[abs(d)] with K = {1/16, 1/4, 3/8, 1/4, 1/16}: centre 1/16, +-1 -> 1/4, +-2 -> 3/8
// (svgf_atrous.hlsl:35,54,60 -- reproduced as written).
__host__ __device__ constexpr float atrous_k(int d)
{
    return (d < 0 ? -d : d) == 0 ? 0.0625f : ((d < 0 ? -d : d) == 1 ? 0.25f : 0.375f);
}

// exp(-|dz|/(phiDepth*step)) * pow(max(0,d), phiNormal) * exp(-|dl|/denL) evaluated as one
// exp2: exponent = phiNormal*log2(d) - |dz|*cz - |dl|*cl (d == 0 -> log2 = -inf -> weight 0).
__device__ __forceinline__ float edge_weight(float d, float adz, float adl, float cz, float cl, float phiN)
{
    const float e = fmaf(phiN, fast_log2(fmaxf(d, 0.0f)), -(adz * cz)) - adl * cl;
    return fast_exp2(e);
}

__device__ __forceinline__ float lum_scale(float var_f, float phiColor)
{
    const float varScale = phiColor * __fsqrt_rn(fmaxf(var_f, 1e-8f)); // :41
    return kLog2e * fast_rcp(fmaxf(varScale, 1e-6f));                  // :75 (as log2e / denom)
}

// Variant 0: one pixel per lane, 25 taps straight from global memory (L1/L2 absorb the reuse).
// Used for steps too wide for the LDS tile and as the in-library A/B arm.
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
    const float z0 = depth_unorm24(a.depth[i]);
    const float3 n0 = oct16_unpack_zw(a.normal[i].y);
    float sr = 0.f, sg = 0.f, sb = 0.f, sw = 0.f;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        const int qy = min(max(y + dy * a.step, 0), a.H - 1); // :65 clamp to the image
        const size_t rowoff = (size_t)(qy - a.row_begin) * a.W;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int qx = min(max(x + dx * a.step, 0), a.W - 1);
            const float4 c = a.src[rowoff + qx];
            const float z = depth_unorm24(a.depth[rowoff + qx]);
            const float3 n = oct16_unpack_zw(a.normal[rowoff + qx].y);
            const float lum = luminance(c.x, c.y, c.z);
            const float d = n0.x * n.x + n0.y * n.y + n0.z * n.z;
            const float w = (atrous_k(dx) * atrous_k(dy)) *
                            edge_weight(d, fabsf(z0 - z), fabsf(lum0 - lum), a.cz, cl, a.phiNormal);
            sr += w * c.x;
            sg += w * c.y;
            sb += w * c.z;
            sw += w;
        }
    }
    const float inv = fast_rcp(fmaxf(sw, 1e-4f)); // :84
    a.dst[i] = make_float4(sr * inv, sg * inv, sb * inv, c0.w);
}

// Variant 1: LDS row-lattice tile.
//   Workgroup = 256 lanes = 4 waves.  Output tile = BW (64) consecutive columns x BH (= 4R)
//   rows of the lattice {r + S*j}.  Taps of a lattice row are lattice rows j-2..j+2, so the
//   tile needs only BH+4 image rows (each a contiguous, coalesced segment of BW+4S texels)
//   for any step S: read amplification (1 + 4/BH)(1 + 4S/BW) instead of (1 + 4S/T)^2.
//   Staging decodes depth (D24 -> float) and the oct shading normal ONCE per texel and
//   keeps {r,g,b,z} and {nx,ny,nz,lum} as two float4 LDS planes (ds_read_b128, lane-contiguous,
//   conflict-free).  Wave w filters lattice rows [w*R, w*R+R): a lane walks its column's R+4
//   staged rows once and feeds each staged texel to every output row it is a tap of.
template <int S, int R>
__global__ __launch_bounds__(256, 4) void svgf_atrous_lds_kernel(AtrousArgs a)
{
    constexpr int BW = 64, BH = 4 * R, COLS = BW + 4 * S, ROWS = BH + 4;
    extern __shared__ float4 lds[];
    float4* __restrict__ A = lds;               // {r, g, b, z}
    float4* __restrict__ B = lds + ROWS * COLS; // {nx, ny, nz, lum}

    // XCD-aware tile order: workgroups b, b+8, ... share an XCD (and its L2), so give each XCD a
    // contiguous run of tiles (neighbouring tiles share halo columns/rows).  Speed only.
    const uint32_t chunk = gridDim.x >> 3;
    const uint32_t t = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if (t >= a.nblocks)
        return;
    const int tx_tile = (int)(t % (uint32_t)a.tiles_x);
    const int rest = (int)(t / (uint32_t)a.tiles_x);
    const int r = rest % S;     // residue class of the lattice rows
    const int jt = rest / S;    // tile index along the lattice
    // first lattice index of the residue class inside [row0, row1)
    const int jmin = (a.row0 - r + S - 1) / S > 0 ? (a.row0 - r + S - 1) / S : 0;
    const int jbase = jmin + jt * BH;
    const int x0 = tx_tile * BW;
    if (r + S * jbase >= a.row1)
        return; // whole tile below the row range (uniform for the workgroup)

    const float4* __restrict__ src = a.src;
    const uint32_t* __restrict__ depth = a.depth;
    const uint32_t* __restrict__ normal32 = reinterpret_cast<const uint32_t*>(a.normal);

    // ---- stage (BH+4) x (BW+4S) texels, clamped to the image (svgf_atrous.hlsl:65) ----
    constexpr int TOTAL = ROWS * COLS;
#pragma unroll 4
    for (int i = threadIdx.x; i < TOTAL; i += 256) {
        const int lr = i / COLS, lc = i - lr * COLS;
        // clamp to the image (the reference's edge rule), then to the resident rows: a partial tile
        // also stages rows that no valid output taps; on a row strip those may lie outside the
        // allocation, so they are redirected to a resident row (their values are never used)
        int y = min(max(r + S * (jbase + lr - 2), 0), a.H - 1);
        y = min(max(y, a.row_begin), a.row_end - 1);
        const int x = min(max(x0 - 2 * S + lc, 0), a.W - 1);
        const size_t q = (size_t)(y - a.row_begin) * a.W + x;
        const float4 c = src[q];
        const uint32_t d = depth[q];
        const uint32_t nzw = normal32[2 * q + 1];
        const float3 n = oct16_unpack_zw(nzw);
        A[i] = make_float4(c.x, c.y, c.z, depth_unorm24(d));
        B[i] = make_float4(n.x, n.y, n.z, luminance(c.x, c.y, c.z));
    }
    __syncthreads();

    // ---- filter ----
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xo = x0 + lane;
    float z0[R], n0x[R], n0y[R], n0z[R], lum0[R], cl[R], alpha0[R];
    float sr[R], sg[R], sb[R], sw[R];
    bool valid[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int lr = wv * R + k + 2;
        const float4 cA = A[lr * COLS + lane + 2 * S];
        const float4 cB = B[lr * COLS + lane + 2 * S];
        z0[k] = cA.w;
        n0x[k] = cB.x;
        n0y[k] = cB.y;
        n0z[k] = cB.z;
        lum0[k] = cB.w;
        const int yo = r + S * (jbase + wv * R + k);
        valid[k] = (xo < a.Wd) && (yo < a.row1);
        float var_f = 0.f;
        alpha0[k] = 0.f;
        if (valid[k]) {
            const size_t i = (size_t)(yo - a.row_begin) * a.W + xo;
            var_f = half_bits_to_float(a.variance[i]);
            alpha0[k] = reinterpret_cast<const float*>(src)[4 * i + 3];
        }
        cl[k] = lum_scale(var_f, a.phiColor);
        sr[k] = sg[k] = sb[k] = sw[k] = 0.f;
    }
    const float cz = a.cz, phiN = a.phiNormal;
#pragma unroll
    for (int ir = 0; ir < R + 4; ++ir) {
        const int lrow = (wv * R + ir) * COLS + lane + 2 * S;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const float4 tA = A[lrow + dx * S];
            const float4 tB = B[lrow + dx * S];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int dy = ir - k - 2;
                if (dy < -2 || dy > 2)
                    continue;
                const float d = n0x[k] * tB.x + n0y[k] * tB.y + n0z[k] * tB.z;
                const float w = (atrous_k(dx) * atrous_k(dy)) *
                                edge_weight(d, fabsf(z0[k] - tA.w), fabsf(lum0[k] - tB.w), cz, cl[k], phiN);
                sr[k] = fmaf(w, tA.x, sr[k]);
                sg[k] = fmaf(w, tA.y, sg[k]);
                sb[k] = fmaf(w, tA.z, sb[k]);
                sw[k] += w;
            }
        }
        // Pin the partial sums here: they only feed the predicated store below, so LLVM would
        // otherwise sink ALL the arithmetic under that branch and keep every staged texel live
        // (spilling ~1.3 KB per lane).  The sched_barrier keeps one row's ds_reads per region.
#pragma unroll
        for (int k = 0; k < R; ++k)
            asm volatile("" : "+v"(sr[k]), "+v"(sg[k]), "+v"(sb[k]), "+v"(sw[k]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (!valid[k])
            continue;
        const int yo = r + S * (jbase + wv * R + k);
        const float inv = fast_rcp(fmaxf(sw[k], 1e-4f)); // :84
        a.dst[(size_t)(yo - a.row_begin) * a.W + xo] = make_float4(sr[k] * inv, sg[k] * inv, sb[k] * inv, alpha0[k]);
    }
}

template <int S, int R>
static hipError_t launch_lds(AtrousArgs a, hipStream_t s)
{
    constexpr int BW = 64, BH = 4 * R, COLS = BW + 4 * S, ROWS = BH + 4;
    constexpr size_t lds_bytes = (size_t)ROWS * COLS * 2 * sizeof(float4);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&svgf_atrous_lds_kernel<S, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return e;
        attr_set = true;
    }
    a.tiles_x = (a.Wd + BW - 1) / BW;
    const int max_lattice_rows = (a.row1 - a.row0 + S - 1) / S; // per residue class, upper bound
    a.tiles_j = (max_lattice_rows + BH - 1) / BH;
    a.nblocks = (uint32_t)a.tiles_x * (uint32_t)S * (uint32_t)a.tiles_j;
    const uint32_t grid = ((a.nblocks + 7u) / 8u) * 8u;
    hipLaunchKernelGGL((svgf_atrous_lds_kernel<S, R>), dim3(grid), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_atrous(const SvgfLaunch& L, int variant, uint32_t step, const float4* src, float4* dst,
                         const uint16_t* variance, const uint32_t* depth, const uint2* normal, hipStream_t s)
{
    const uint32_t Wd = (L.W / 8u) * 8u, Hd = (L.H / 8u) * 8u; // SVGFDenoiser.cpp:185
    const uint32_t row1 = L.row1 < Hd ? L.row1 : Hd;
    if (L.row0 >= row1 || Wd == 0)
        return hipSuccess;
    AtrousArgs a;
    a.src = src;
    a.dst = dst;
    a.variance = variance;
    a.depth = depth;
    a.normal = normal;
    a.W = (int)L.W;
    a.H = (int)L.H;
    a.Wd = (int)Wd;
    a.row_begin = (int)L.row_begin;
    a.row_end = (int)L.row_end;
    a.row0 = (int)L.row0;
    a.row1 = (int)row1;
    a.step = (int)step;
    a.tiles_x = a.tiles_j = 0;
    a.nblocks = 0;
    a.cz = kLog2e / (L.p.phiDepth * (float)step);
    a.phiColor = L.p.phiColor;
    a.phiNormal = L.p.phiNormal;
    if (variant == 1) {
        switch (step) {
        case 1: return launch_lds<1, 4>(a, s);
        case 2: return launch_lds<2, 4>(a, s);
        case 4: return launch_lds<4, 4>(a, s);
        case 8: return launch_lds<8, 4>(a, s);
        case 16: return launch_lds<16, 4>(a, s);
        case 32: return launch_lds<32, 4>(a, s);
        default: break; // wider steps do not fit the LDS tile: direct kernel
        }
    }
    dim3 grid((Wd + 63) / 64, (row1 - L.row0 + 3) / 4);
    hipLaunchKernelGGL(svgf_atrous_direct_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace neb

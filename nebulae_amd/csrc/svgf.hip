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
#pragma clang fp contract(off) // fused operations are explicit fmaf: see neb_device.h

#include <algorithm>
#include <atomic>

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
    float4* geometry;               // {decoded shading normal, depth} of the current frame for the a-trous levels, or null
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
    const float zc = depth_unorm24(dc);
    if (a.geometry) // decoded once here instead of once per a-trous level and staged texel
        a.geometry[i] = make_float4(Nc.x, Nc.y, Nc.z, zc);
    const float dz = fabsf(zc - depth_unorm24(dh));
    // SVGF_DWeight: exp(-dz^2 / (2 sigma^2))   (svgf_common.hlsli:11-15)
    const float wDepth = fast_exp2(dz * dz * a.neg_inv_two_sigma2_log2e);
    // SVGF_NWeight: saturate(dot)               (svgf_common.hlsli:4-7)
    const float wNormal = __saturatef(fmaf(Nc.z, Nh.z, fmaf(Nc.y, Nh.y, Nc.x * Nh.x)));
    const float w = wDepth * wNormal;
    const float alpha = fmaf(w, a.alpha - 1.0f, 1.0f); // lerp(1, alpha, w)  (:51)

    const float Y = luminance(Cc.x, Cc.y, Cc.z);
    const float Mh0 = half_bits_to_float(mh & 0xffffu), Mh1 = half_bits_to_float(mh >> 16);
    const float M1 = fmaf(alpha, Mh0 - Y, Y);
    const float Y2 = Y * Y;
    const float M2 = fmaf(alpha, Mh1 - Y2, Y2);
    const float var = fmaxf(fmaf(-M1, M1, M2), a.varianceEps);

    float4 out;
    out.x = fmaf(alpha, Ch.x - Cc.x, Cc.x);
    out.y = fmaf(alpha, Ch.y - Cc.y, Cc.y);
    out.z = fmaf(alpha, Ch.z - Cc.z, Cc.z);
    out.w = Cc.w;
    a.rad_cur[i] = out;
    a.mom_cur[i] = float_to_half_bits(M1) | (float_to_half_bits(M2) << 16);
    a.variance[i] = (uint16_t)float_to_half_bits(var);
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

// log2(K[abs(dx)] * K[abs(dy)]): the kernel weight folded into the exponent of the edge-stopping exp2
__host__ __device__ constexpr float atrous_log2k(int dx, int dy)
{
    // log2(1/16) = -4, log2(1/4) = -2, log2(3/8) = log2(3) - 3
    const float lx = (dx < 0 ? -dx : dx) == 0 ? -4.0f : ((dx < 0 ? -dx : dx) == 1 ? -2.0f : -1.4150374992788437f);
    const float ly = (dy < 0 ? -dy : dy) == 0 ? -4.0f : ((dy < 0 ? -dy : dy) == 1 ? -2.0f : -1.4150374992788437f);
    return lx + ly;
}

// max(0, dot(n0, n)) of svgf_atrous.hlsl:74, saturated: the [0, 1] clamp is the free output modifier of the dot
// product's last fma (a bare max(x, 0) is a separate v_max per tap).  Two unit normals can give 1 + 2 ulp, where the
// reference's pow(d, 128) would be 1 + 3e-5 and this is 1 -- a deliberate divergence, the same in both
// kernels so that every variant and every level is one function.
__device__ __forceinline__ float normal_dot_sat(float d) { return fminf(fmaxf(d, 0.0f), 1.0f); }

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
    const float4 g0 = a.geometry[i];
    const float z0 = g0.w;
    const float3 n0 = make_float3(g0.x, g0.y, g0.z);
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
            const float z = g.w;
            const float3 n = make_float3(g.x, g.y, g.z);
            const float lum = luminance(c.x, c.y, c.z);
            const float d = normal_dot_sat(fmaf(n0.z, n.z, fmaf(n0.y, n.y, n0.x * n.x)));
            float e = fmaf(a.phiNormal, fast_log2(d), atrous_log2k(dx, dy));
            e = fmaf(-fabsf(z0 - z), a.cz, e);
            e = fmaf(-fabsf(lum0 - lum), cl, e);
            const float w = fast_exp2(e);
            sr = fmaf(w, c.x, sr);
            sg = fmaf(w, c.y, sg);
            sb = fmaf(w, c.z, sb);
            sw += w;
        }
    }
    const float inv = fast_rcp(fmaxf(sw, 1e-4f)); // :84
    a.dst[i] = make_float4(sr * inv, sg * inv, sb * inv, c0.w);
}

// Variant 1: LDS row-lattice tile, persistent workgroups with register prefetch.
//   Workgroup = 256 lanes = 4 waves.  Output tile = BW (64) consecutive columns x BH (= 4R)
//   rows of the lattice {r + S*j}.  Taps of a lattice row are lattice rows j-2..j+2, so the
//   tile needs only BH+4 image rows (each a contiguous, coalesced segment of BW+4S texels)
//   for any step S: read amplification (1 + 4/BH)(1 + 4S/BW) instead of (1 + 4S/T)^2.
//   Staging reads the radiance and the frame's decoded geometry plane {n.xyz, z} (written once by the temporal pass) and
//   keeps {r,g,b,z} and {nx,ny,nz,lum} as two float4 LDS planes (ds_read_b128, lane-contiguous,
//   conflict-free).  Wave w filters lattice rows [w*R, w*R+R): a lane walks its column's R+4
//   staged rows once and feeds each staged texel to every output row it is a tap of.
//   Each workgroup walks several tiles: the global loads of the NEXT tile are issued into
//   registers before the current tile is filtered, so their latency hides under ~1.6k VALU
//   instructions per wave instead of stalling an empty SIMD (the first version spent 36 % of
//   wave time in s_waitcnt).
template <int S, int R>
struct AtrousTile {
    static constexpr int BW = 64, BH = 4 * R, COLS = BW + 4 * S, ROWS = BH + 4, TOTAL = ROWS * COLS;
    static constexpr int NLOAD = (TOTAL + 255) / 256;
    // workgroups per CU: what the 160 KB of LDS hold, at most 5 (R <= 2) / 3 -- also the register budget the kernel is compiled for
    static constexpr int LDS_BYTES = TOTAL * 2 * 16;
    static constexpr int PER_CU = (R <= 2 ? 5 : 3) < (160 * 1024) / LDS_BYTES ? (R <= 2 ? 5 : 3) : (160 * 1024) / LDS_BYTES;
};

template <int S, int R>
__global__ __launch_bounds__(256, (AtrousTile<S, R>::PER_CU)) void svgf_atrous_lds_kernel(AtrousArgs a)
{
    using T = AtrousTile<S, R>;
    constexpr int BW = T::BW, BH = T::BH, COLS = T::COLS, ROWS = T::ROWS, TOTAL = T::TOTAL, NLOAD = T::NLOAD;
    extern __shared__ float4 lds[];
    float4* __restrict__ A = lds;               // {r, g, b, lum}
    float4* __restrict__ B = lds + ROWS * COLS; // {nx, ny, nz, z}: the geometry plane's texels, copied by LDS-DMA

    const float4* __restrict__ src = a.src;
    const float4* __restrict__ geometry = a.geometry;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (wave-uniform: row bases and row addresses become scalar)

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
    float4 pc[NLOAD];

    struct Tile {
        int r, jbase, x0;
        size_t first; // element index the offsets in toff[] count from
    };
    auto tile_origin = [&](uint32_t t, Tile& o) -> bool {
        const int tx_tile = (int)(t % (uint32_t)a.tiles_x);
        const int rest = (int)(t / (uint32_t)a.tiles_x);
        o.r = rest % S;             // residue class of the lattice rows
        const int jt = rest / S;    // tile index along the lattice
        const int jmin = (a.row0 - o.r + S - 1) / S > 0 ? (a.row0 - o.r + S - 1) / S : 0; // first lattice index inside [row0,row1)
        o.jbase = jmin + jt * BH;
        o.x0 = tx_tile * BW;
        return o.r + S * o.jbase < a.row1; // false: whole tile below the row range
    };
    // fills toff[] / o.first for a tile about to be loaded
    auto tile_offsets = [&](Tile& o) {
        const int y_first = o.r + S * (o.jbase - 2), y_last = o.r + S * (o.jbase + ROWS - 3), x_first = o.x0 - 2 * S;
        const bool interior = y_first >= max(0, a.row_begin) && y_last < min(a.H, a.row_end) && x_first >= 0 && x_first + COLS <= a.W;
        if (interior) {
            o.first = (size_t)(y_first - a.row_begin) * a.W + x_first;
            if (!toff_regular) {
#pragma unroll
                for (int k = 0; k < NLOAD; ++k) {
                    const int i = threadIdx.x + 256 * k;
                    const int lr = i / COLS;
                    toff[k] = (uint32_t)(lr * S * a.W + (i - lr * COLS)) * 16u;
                }
                toff_regular = true;
            }
        } else {
            o.first = 0;
            toff_regular = false;
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const int i = threadIdx.x + 256 * k;
                const int lr = i / COLS, lc = i - lr * COLS;
                // clamp to the image (svgf_atrous.hlsl:65), then to the resident rows: a partial tile also stages rows
                // that no valid output taps; on a row strip those may lie outside the allocation, so they are
                // redirected to a resident row (their values are never used)
                int y = min(max(o.r + S * (o.jbase + lr - 2), 0), a.H - 1);
                y = min(max(y, a.row_begin), a.row_end - 1);
                const int x = min(max(o.x0 - 2 * S + lc, 0), a.W - 1);
                toff[k] = (uint32_t)((y - a.row_begin) * a.W + x) * 16u; // (a plane is < 4 GB: checked at launch)
            }
        }
    };
    auto issue_load = [&](int k, const Tile& o) {
        if (threadIdx.x + 256 * k < TOTAL)
            pc[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(src + o.first) + toff[k]);
    };
    // The geometry texels go global -> LDS directly (global_load_lds_dwordx4: per-lane source address, destination =
    // wave-uniform base + lane * 16 B, which is exactly this staging order), no registers and no VALU.  Issued once the
    // whole workgroup has finished reading the previous tile, and complete before the barrier that precedes the filter.
    auto issue_geometry_dma = [&](const Tile& o) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            if (threadIdx.x + 256 * k < TOTAL) {
                const char* g = reinterpret_cast<const char*>(geometry + o.first) + toff[k];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(B + 256 * k + 64 * wv), 16, 0, 0);
            }
        }
    };
    auto issue_loads = [&](const Tile& o) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k)
            issue_load(k, o);
    };

    // The centre pixel's variance and alpha (the only per-pixel inputs that are not staged) are fetched one tile ahead as
    // well: loaded at the start of the filter phase they were a dependent global round trip in front of every tile's
    // arithmetic (2.5 us per level).
    uint32_t nvar[R];
    float nalpha[R];
    auto issue_centre_loads = [&](const Tile& o) {
        const int xo = o.x0 + lane;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int yo = o.r + S * (o.jbase + wv * R + k);
            nvar[k] = 0u;
            nalpha[k] = 0.f;
            if (xo < a.Wd && yo < a.row1) {
                const size_t i = (size_t)(yo - a.row_begin) * a.W + xo;
                nvar[k] = a.variance[i];
                nalpha[k] = reinterpret_cast<const float*>(src)[4 * i + 3];
            }
        }
    };

    uint32_t t = xcd * per_xcd + wg_in_xcd;
    Tile nt{0, 0, 0, 0};
    bool have = false;
    while (t < t_end && !(have = tile_origin(t, nt)))
        t += wgs_per_xcd;
    if (have) {
        tile_offsets(nt);
        issue_loads(nt);
        issue_centre_loads(nt);
    }

    while (have) {
        // ---- stage the tile: radiance (prefetched into registers during the previous tile) + its luminance -> plane A;
        // geometry -> plane B by DMA (normal and depth arrive decoded: the temporal pass did that once per frame) ----
        issue_geometry_dma(nt); // first: its flight overlaps the wait for the prefetched radiance and the luminance arithmetic
        float lum[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; ++k)
            lum[k] = luminance(pc[k].x, pc[k].y, pc[k].z);
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < TOTAL)
                A[i] = make_float4(pc[k].x, pc[k].y, pc[k].z, lum[k]);
        }
        __syncthreads(); // (also waits for this wave's DMA: an LDS-DMA is a pending LDS write on the VM counter)

        // ---- next tile: issue its loads now, consume them after this tile is filtered ----
        const int cr = nt.r, cjbase = nt.jbase, cx0 = nt.x0;
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

        // ---- filter the current tile ----
        const int xo = cx0 + lane;
        float z0[R], n0x[R], n0y[R], n0z[R], lum0[R], cl[R], alpha0[R];
        float sr[R], sg[R], sb[R], sw[R];
        bool valid[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int lr = wv * R + k + 2;
            const float4 cA = A[lr * COLS + lane + 2 * S];
            const float4 cB = B[lr * COLS + lane + 2 * S];
            z0[k] = cB.w;
            n0x[k] = cB.x;
            n0y[k] = cB.y;
            n0z[k] = cB.z;
            lum0[k] = cA.w;
            const int yo = cr + S * (cjbase + wv * R + k);
            valid[k] = (xo < a.Wd) && (yo < a.row1);
            alpha0[k] = calpha[k];
            cl[k] = lum_scale(half_bits_to_float((uint16_t)cvar[k]), a.phiColor); // (an invalid pixel holds 0: never stored)
            sr[k] = sg[k] = sb[k] = sw[k] = 0.f;
        }
        const float cz = a.cz, phiN = a.phiNormal;
#pragma unroll
        for (int ir = 0; ir < R + 4; ++ir) {
            if (ir < NLOAD && have_next)
                issue_load(ir, nt);
            const int lrow_base = (wv * R + ir) * COLS + lane + 2 * S;
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const float4 tA = A[lrow_base + dx * S];
                const float4 tB = B[lrow_base + dx * S];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int dy = ir - k - 2;
                    if (dy < -2 || dy > 2)
                        continue;
                    // w = Kx*Ky * exp(-|dz|/(phiDepth*step)) * pow(max(0,d), phiNormal) * exp(-|dl|/denL) as ONE exp2:
                    // exponent = log2(Kx*Ky) + phiNormal*log2(d) - |dz|*cz - |dl|*cl   (d == 0 -> -inf -> weight 0)
                    const float lk = atrous_log2k(dx, dy);
                    const float d = normal_dot_sat(fmaf(n0z[k], tB.z, fmaf(n0y[k], tB.y, n0x[k] * tB.x)));
                    float e = fmaf(phiN, fast_log2(d), lk);
                    e = fmaf(-fabsf(z0[k] - tB.w), cz, e);
                    e = fmaf(-fabsf(lum0[k] - tA.w), cl[k], e);
                    const float w = fast_exp2(e);
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
            const int yo = cr + S * (cjbase + wv * R + k);
            const float inv = fast_rcp(fmaxf(sw[k], 1e-4f)); // :84
            a.dst[(size_t)(yo - a.row_begin) * a.W + xo] = make_float4(sr[k] * inv, sg[k] * inv, sb[k] * inv, alpha0[k]);
        }
        have = have_next;
        if (have)
            __syncthreads(); // everyone is done reading LDS before the next tile overwrites it
    }
}

// (Measured and dropped: a double-buffered LDS variant that weaves the staging of tile i+1 into the row loop of
// tile i, one barrier per tile -- 45-94 us per level against 41-49 us for the kernel above; the longer live
// ranges cost more than the barrier and the exposed decode they remove.)
template <int S, int R>
static hipError_t launch_lds(AtrousArgs a, int device, int num_cus, hipStream_t s)
{
    using T = AtrousTile<S, R>;
    constexpr size_t lds_bytes = (size_t)T::TOTAL * 2 * sizeof(float4);
    // the dynamic-LDS limit is a per-device function attribute: remember which devices have it (one bit each; a
    // device ordinal beyond the mask just sets it on every launch)
    static std::atomic<uint64_t> attr_set{0};
    const uint64_t bit = (device >= 0 && device < 64) ? (1ull << device) : 0ull;
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&svgf_atrous_lds_kernel<S, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    a.tiles_x = (a.Wd + T::BW - 1) / T::BW;
    const int max_lattice_rows = (a.row1 - a.row0 + S - 1) / S; // per residue class, upper bound
    a.tiles_j = (max_lattice_rows + T::BH - 1) / T::BH;
    a.nblocks = (uint32_t)a.tiles_x * (uint32_t)S * (uint32_t)a.tiles_j;
    // persistent grid: as many workgroups as fit (LDS-limited, at most 3 per CU by the launch bounds)
    static_assert(T::LDS_BYTES == (int)lds_bytes && T::PER_CU >= 1, "tile too large for the LDS");
    const uint32_t per_cu = (uint32_t)T::PER_CU;
    uint32_t grid = (uint32_t)num_cus * (per_cu ? per_cu : 1u);
    if (grid > a.nblocks)
        grid = a.nblocks;
    grid = ((grid + 7u) / 8u) * 8u;
    hipLaunchKernelGGL((svgf_atrous_lds_kernel<S, R>), dim3(grid), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_atrous(const SvgfLaunch& L, int variant, uint32_t step, const float4* src, float4* dst,
                         const uint16_t* variance, const float4* geometry, hipStream_t s)
{
    const int num_cus = L.num_cus > 0 ? L.num_cus : 256;
    const uint32_t Wd = (L.W / 8u) * 8u, Hd = (L.H / 8u) * 8u; // SVGFDenoiser.cpp:185
    const uint32_t row1 = L.row1 < Hd ? L.row1 : Hd;
    if (L.row0 >= row1 || Wd == 0)
        return hipSuccess;
    AtrousArgs a;
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
    a.tiles_x = a.tiles_j = 0;
    a.nblocks = 0;
    a.cz = kLog2e / (L.p.phiDepth * (float)step);
    a.phiColor = L.p.phiColor;
    a.phiNormal = L.p.phiNormal;
    // variant 1 (default): R = 4 rows per lane for steps <= 4, R = 2 (smaller LDS tile, 5 waves/SIMD) for steps >= 8,
    // as measured; 2 / 3 force R = 2 / R = 4 everywhere (A/B arms)
    // (the LDS kernel addresses a plane with 32-bit byte offsets: a resident plane of 4 GB or more -- 16 k x 16 k -- takes the direct kernel)
    if (variant >= 1 && (uint64_t)(L.row_end - L.row_begin) * L.W < (1ull << 28)) {
        switch (step) {
        case 1: return (variant == 2) ? launch_lds<1, 2>(a, L.device, num_cus, s) : launch_lds<1, 4>(a, L.device, num_cus, s);
        case 2: return (variant == 2) ? launch_lds<2, 2>(a, L.device, num_cus, s) : launch_lds<2, 4>(a, L.device, num_cus, s);
        case 4: return (variant == 2) ? launch_lds<4, 2>(a, L.device, num_cus, s) : launch_lds<4, 4>(a, L.device, num_cus, s);
        case 8: return (variant == 3) ? launch_lds<8, 4>(a, L.device, num_cus, s) : launch_lds<8, 2>(a, L.device, num_cus, s);
        case 16: return (variant == 3) ? launch_lds<16, 4>(a, L.device, num_cus, s) : launch_lds<16, 2>(a, L.device, num_cus, s);
        case 32: return (variant == 3) ? launch_lds<32, 4>(a, L.device, num_cus, s) : launch_lds<32, 2>(a, L.device, num_cus, s);
        default: break; // wider steps do not fit the LDS tile: direct kernel
        }
    }
    dim3 grid((Wd + 63) / 64, (row1 - L.row0 + 3) / 4);
    hipLaunchKernelGGL(svgf_atrous_direct_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace neb

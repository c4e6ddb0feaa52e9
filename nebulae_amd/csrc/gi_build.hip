// gi_build.hip -- scene upload and acceleration structure of the GI path (one-time setup).
//
// Reference: src/nri/GIProcessedScene.cpp:16-137 (scene tables); RTAccelerationStructureBuilder.cpp:14-130 (driver
// BVH, built on the GPU once -> replaced by an own device build: Morton sort, binned-SAH splits, collapse to a 4-wide
// tree, all in HIP kernels).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp> // (the one-time build: 64-bit key sort and prefix sums from ROCm's own primitives, not the CUB-compatible layer)

#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "gi_device.h"

namespace neb {

// neb_resize: the per-pixel GI buffers (records, debug hits, per-workgroup counters) belong to the old resolution
void gi_on_resize(GiState* g)
{
    if (!g)
        return;
    std::vector<void*> stale = {g->d_hits, g->d_direct_counts};
    for (GiState::DispatchSet& ds : g->sets) {
        for (void* p : {(void*)ds.d_records, (void*)ds.d_block_counts, (void*)ds.d_sort, ds.d_sort_temp, (void*)ds.d_list})
            stale.push_back(p);
        ds = GiState::DispatchSet{};
    }
    g->traces = g->resolves = 0;
    g->begun = g->finished = 0;
    for (void* p : stale) {
        if (!p)
            continue;
        for (size_t k = 0; k < g->allocs.size(); ++k)
            if (g->allocs[k] == p) {
                g->allocs.erase(g->allocs.begin() + (long)k);
                break;
            }
        (void)hipFree(p);
    }
    g->d_hits = nullptr;
    g->d_direct_counts = nullptr;
    g->n_block_counts = 0;
}

void gi_destroy(GiState* g)
{
    if (!g)
        return;
    for (void* p : g->allocs)
        (void)hipFree(p);
    if (g->sun_table_event)
        (void)hipEventDestroy(g->sun_table_event);
    for (hipEvent_t& e : g->sun_build_ev)
        if (e)
            (void)hipEventDestroy(e);
    for (hipEvent_t& e : g->tail_ev)
        if (e)
            (void)hipEventDestroy(e);
    delete g;
}

// one thread per sorted triangle: gather its vertices' attributes from the SoA pools into the 128-B record
__global__ void pack_shade_records_kernel(SceneView S, uint32_t n, float4* out)
{
    const uint32_t ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= n)
        return;
    const float4 ids = S.tris[3 * ti + 2];
    const uint32_t geom = __float_as_uint(ids.y), prim = __float_as_uint(ids.z);
    const DevGeom g = S.geoms[geom];
    float4 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.valid) {
        const uint32_t i0 = g.vertexBase + S.indices[g.firstIndex + 3 * prim], i1 = g.vertexBase + S.indices[g.firstIndex + 3 * prim + 1],
                       i2 = g.vertexBase + S.indices[g.firstIndex + 3 * prim + 2];
        const float3 n0 = load3(S.normals, i0), n1 = load3(S.normals, i1), n2 = load3(S.normals, i2);
        r[0] = make_float4(n0.x, n0.y, n0.z, S.uvs[2 * i0]);
        r[1] = make_float4(n1.x, n1.y, n1.z, S.uvs[2 * i0 + 1]);
        r[2] = make_float4(n2.x, n2.y, n2.z, S.uvs[2 * i1]);
        r[3] = make_float4(S.tangents[4 * i0], S.tangents[4 * i0 + 1], S.tangents[4 * i0 + 2], S.tangents[4 * i0 + 3]);
        r[4] = make_float4(S.tangents[4 * i1], S.tangents[4 * i1 + 1], S.tangents[4 * i1 + 2], S.tangents[4 * i1 + 3]);
        r[5] = make_float4(S.tangents[4 * i2], S.tangents[4 * i2 + 1], S.tangents[4 * i2 + 2], S.tangents[4 * i2 + 3]);
        r[6] = make_float4(S.uvs[2 * i1 + 1], S.uvs[2 * i2], S.uvs[2 * i2 + 1], 0.f);
    }
    r[6].w = __uint_as_float(geom);
    r[7].x = __uint_as_float(prim);
    {
        const uint32_t none[kHints] = {kNoHint, kNoHint, kNoHint, kNoHint}; // (hints: gi_sun_table.hip)
        pack_hints(none, 0u, r[7]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        out[8 * (size_t)ti + k] = r[k];
}

// Bilinear-footprint table of one RGBA8 image: entry (x, y) = the texels {(x,y), (x+1,y), (x,y+1), (x+1,y+1)} with the wrap
// addressing applied, so a filtered fetch is one 16-byte load.  `stride` (in uint4) = 1 for a standalone table, 4 for one
// slot of a material's interleaved 64-byte entries.
__global__ void texture_footprints_kernel(const uint32_t* __restrict__ px, uint32_t w, uint32_t h, uint4* out, uint32_t stride)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)w * h)
        return;
    const uint32_t y = (uint32_t)(i / w), x = (uint32_t)(i - (size_t)y * w);
    const uint32_t x1 = (x + 1) % w, y1 = (y + 1) % h;
    out[i * stride] = make_uint4(px[(size_t)y * w + x], px[(size_t)y * w + x1], px[(size_t)y1 * w + x], px[(size_t)y1 * w + x1]);
}

// One 32-byte entry per texel position of a material whose three maps share a size: the bilinear footprint
// {(x,y), (x+1,y), (x,y+1), (x+1,y+1)} (wrap applied) x the eight channels the shaders read, 8 bytes per texel:
// lo = albedo.r | albedo.g << 8 | albedo.b << 16 | normal.r << 24,  hi = normal.g | normal.b << 8 | rm.g << 16 | rm.b << 24.
__global__ void material_bundle_kernel(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ pn, const uint32_t* __restrict__ pr, uint32_t w,
                                       uint32_t h, uint4* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)w * h)
        return;
    const uint32_t y = (uint32_t)(i / w), x = (uint32_t)(i - (size_t)y * w);
    const uint32_t x1 = (x + 1) % w, y1 = (y + 1) % h;
    auto pack = [&](uint32_t xx, uint32_t yy) {
        const size_t k = (size_t)yy * w + xx;
        const uint32_t a = pa[k], n = pn[k], r = pr[k];
        return make_uint2((a & 0x00ffffffu) | (n << 24), ((n >> 8) & 0xffffu) | (((r >> 8) & 0xffffu) << 16));
    };
    const uint2 t00 = pack(x, y), t10 = pack(x1, y), t01 = pack(x, y1), t11 = pack(x1, y1);
    out[2 * i] = make_uint4(t00.x, t00.y, t10.x, t10.y);
    out[2 * i + 1] = make_uint4(t01.x, t01.y, t11.x, t11.y);
}

// ------------------------------------------------------------------------------------------------
// Spatial order: 64-bit Morton keys of the triangle centroids -> radix sort -> gather
// ------------------------------------------------------------------------------------------------
// spreads the low 21 bits of v to every third bit
__device__ __forceinline__ uint64_t expand_bits21(uint64_t v)
{
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// key = Morton code of the centroid (axis_bits per axis) above the triangle index (index_bits): the index makes every
// key unique, and the code gets all the bits the index leaves (15 per axis for 262 k triangles; on the bench scene
// 10 / 12 / 15 bits traverse equally fast, denser scenes need the resolution)
// `boxes` (2 x float4 per reference, or null): lo.w != 0 marks the box of a PIECE of a split triangle (reference splitting, below);
// every other reference takes the box of its triangle's vertices, computed here as it always was.
__device__ __forceinline__ void reference_box(const float* t, const float4* boxes, uint32_t i, float3& lo, float3& hi)
{
    if (boxes && boxes[2 * i].w != 0.0f) {
        const float4 l = boxes[2 * i], h = boxes[2 * i + 1];
        lo = f3(l.x, l.y, l.z);
        hi = f3(h.x, h.y, h.z);
        return;
    }
    const float3 v0 = f3(t[0], t[1], t[2]), v1 = f3(t[0] + t[3], t[1] + t[4], t[2] + t[5]), v2 = f3(t[0] + t[6], t[1] + t[7], t[2] + t[8]);
    lo = f3(fminf(v0.x, fminf(v1.x, v2.x)), fminf(v0.y, fminf(v1.y, v2.y)), fminf(v0.z, fminf(v1.z, v2.z)));
    hi = f3(fmaxf(v0.x, fmaxf(v1.x, v2.x)), fmaxf(v0.y, fmaxf(v1.y, v2.y)), fmaxf(v0.z, fmaxf(v1.z, v2.z)));
}

__global__ void lbvh_morton_kernel(const float* __restrict__ tris12, const float4* __restrict__ boxes, uint32_t n, float3 smin, float3 sinv,
                                   uint32_t axis_bits, uint32_t index_bits, uint64_t* keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float3 blo, bhi;
    reference_box(tris12 + 12 * (size_t)i, boxes, i, blo, bhi);
    const float cx = (blo.x + bhi.x) * 0.5f;
    const float cy = (blo.y + bhi.y) * 0.5f;
    const float cz = (blo.z + bhi.z) * 0.5f;
    const float cells = (float)(1u << axis_bits), top = cells - 1.0f;
    const uint64_t qx = (uint64_t)fminf(fmaxf((cx - smin.x) * sinv.x * cells, 0.0f), top);
    const uint64_t qy = (uint64_t)fminf(fmaxf((cy - smin.y) * sinv.y * cells, 0.0f), top);
    const uint64_t qz = (uint64_t)fminf(fmaxf((cz - smin.z) * sinv.z * cells, 0.0f), top);
    const uint64_t m = (expand_bits21(qx) << 2) | (expand_bits21(qy) << 1) | expand_bits21(qz);
    keys[i] = (m << index_bits) | i;
}

__global__ void lbvh_gather_kernel(const float* __restrict__ tris12, const float4* __restrict__ boxes, const uint64_t* __restrict__ keys, uint32_t n,
                                   uint64_t index_mask, float4* out, float4* boxes_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const size_t src = (size_t)(keys[i] & index_mask);
    const float* t = tris12 + 12 * src;
    out[3 * i] = make_float4(t[0], t[1], t[2], t[3]);
    out[3 * i + 1] = make_float4(t[4], t[5], t[6], t[7]);
    out[3 * i + 2] = make_float4(t[8], t[9], t[10], t[11]);
    if (boxes) {
        boxes_out[2 * i] = boxes[2 * src];
        boxes_out[2 * i + 1] = boxes[2 * src + 1];
    }
}

// ------------------------------------------------------------------------------------------------
// Topology: top-down binned-SAH build on the device, one level per pass.
// A segment = a run of the primitive index array that still has to be split, owned by one binary node.  Per level one
// workgroup per segment (a) reduces the centroid bounds, (b) bins the primitives' boxes into kSahBins bins per axis in
// LDS, (c) evaluates the surface-area heuristic  A_left * N_left + A_right * N_right  at every bin boundary of the three
// axes, (d) partitions the run in place order (stable) into the other index array.  A second kernel then creates the
// child nodes and next level's segments at positions given by a prefix sum, so node numbering -- hence the whole tree
// -- is deterministic (every GPU of a strip group must build the same tree: ties between coincident hits are
// resolved by traversal order).  Splitting continues down to runs of <= 2 triangles, the leaf size of the BVH4.
// The first levels have few, long segments and use a fraction of the chip; all levels together take a few milliseconds
// at 262 k triangles -- against ~0.3 s for the host sweep-SAH pass this replaces.
// (Measured first and dropped: PLOC, agglomerative clustering over the Morton order.  At search radius 8 / 16 / 64 the
// closest-hit pass took 464 / 481 / 519 us, no better than the plain Karras LBVH (464 us) and far from the SAH
// sweep (381 us): on the stand-in's regular tessellations every neighbouring pair ties, and a wider window only adds
// irregular merges.)
// Binary nodes: [0, n) = the triangles in Morton order, [n, 2n - 1) = inner nodes in creation order; node n is the root.
// ------------------------------------------------------------------------------------------------
#ifndef NEB_SAH_BINS
#define NEB_SAH_BINS 32
#endif
#ifndef NEB_SAH_BIG
#define NEB_SAH_BIG 1 // the first levels' splits spread over the chip (sah_big_* kernels); 0: one workgroup per segment at every level
#endif
constexpr int kSahBins = NEB_SAH_BINS;

struct BinaryNodes { // the binary tree under construction
    float4* lo;      // {min.xyz, left child as int bits}   (triangles: children = -1)
    float4* hi;      // {max.xyz, right child as int bits}
    uint32_t* size;  // triangles below the node
};

struct SahSegment {
    uint32_t begin, end, node;
};
struct SahSplit { // result of one segment's split
    uint32_t mid; // left = [begin, mid), right = [mid, end)
    float lbox[6], rbox[6];
};

__global__ void sah_init_kernel(const float4* __restrict__ tris, const float4* __restrict__ boxes, uint32_t n, BinaryNodes N, uint32_t* idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float3 blo, bhi;
    reference_box(reinterpret_cast<const float*>(tris + 3 * (size_t)i), boxes, i, blo, bhi);
    N.lo[i] = make_float4(blo.x, blo.y, blo.z, __int_as_float(-1));
    N.hi[i] = make_float4(bhi.x, bhi.y, bhi.z, __int_as_float(-1));
    N.size[i] = 1u;
    idx[i] = i;
}

__device__ __forceinline__ float box_half_area(float3 lo, float3 hi)
{
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}

// order-preserving float <-> uint map (atomicMin / atomicMax on LDS words)
__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ int sah_bin(float c, float cmin, float scale) { return min(kSahBins - 1, max(0, (int)((c - cmin) * scale))); }

// kSahThreads: 1024 for the first levels (few, long runs: a workgroup walks its run in strides of its size), 256 below
template <int kSahThreads>
__global__ __launch_bounds__(kSahThreads) void sah_split_kernel(BinaryNodes N, const SahSegment* __restrict__ segs, const uint32_t* __restrict__ idx_in,
                                                                uint32_t* __restrict__ idx_out, SahSplit* __restrict__ splits)
{
    // bins: axis 0..2 = x, y, z by centroid; axis 3 = by position (first / second half of the run): the fallback when all
    // centroids coincide.  Per bin: box lo.xyz, hi.xyz as ordered uints, and a count.
    __shared__ uint32_t s_lo[4][kSahBins][3], s_hi[4][kSahBins][3], s_cnt[4][kSahBins];
    __shared__ uint32_t s_cmin[3], s_cmax[3];
    __shared__ float s_cost[kSahThreads];
    __shared__ int s_pick[kSahThreads];
    __shared__ uint32_t s_scan[kSahThreads];
    __shared__ uint32_t s_run[2];
    const SahSegment sg = segs[blockIdx.x];
    const uint32_t begin = sg.begin, end = sg.end, cnt = end - begin, half = begin + cnt / 2;
    const int tid = threadIdx.x;
    for (int k = tid; k < 4 * kSahBins; k += kSahThreads) {
        const int ax = k / kSahBins, b = k % kSahBins;
        for (int q = 0; q < 3; ++q) {
            s_lo[ax][b][q] = 0xffffffffu;
            s_hi[ax][b][q] = 0u;
        }
        s_cnt[ax][b] = 0u;
    }
    if (tid < 3) {
        s_cmin[tid] = 0xffffffffu;
        s_cmax[tid] = 0u;
    }
    __syncthreads();
    // (a) centroid bounds
    {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = begin + tid; i < end; i += kSahThreads) {
            const uint32_t p = idx_in[i];
            const float4 lo = N.lo[p], hi = N.hi[p];
            const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
            for (int q = 0; q < 3; ++q) {
                mn[q] = fminf(mn[q], c[q]);
                mx[q] = fmaxf(mx[q], c[q]);
            }
        }
        for (int q = 0; q < 3; ++q) {
            atomicMin(&s_cmin[q], float_to_ordered(mn[q]));
            atomicMax(&s_cmax[q], float_to_ordered(mx[q]));
        }
    }
    __syncthreads();
    float cmin[3], scale[3];
    for (int q = 0; q < 3; ++q) {
        cmin[q] = ordered_to_float(s_cmin[q]);
        const float ext = ordered_to_float(s_cmax[q]) - cmin[q];
        scale[q] = ext > 0.0f ? (float)kSahBins / ext : 0.0f; // a flat axis puts everything into bin 0: no split there
    }
    // (b) binning
    for (uint32_t i = begin + tid; i < end; i += kSahThreads) {
        const uint32_t p = idx_in[i];
        const float4 lo = N.lo[p], hi = N.hi[p];
        const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
        const uint32_t l[3] = {float_to_ordered(lo.x), float_to_ordered(lo.y), float_to_ordered(lo.z)};
        const uint32_t h[3] = {float_to_ordered(hi.x), float_to_ordered(hi.y), float_to_ordered(hi.z)};
        for (int ax = 0; ax < 4; ++ax) {
            const int b = ax < 3 ? sah_bin(c[ax], cmin[ax], scale[ax]) : (i < half ? 0 : 1);
            for (int q = 0; q < 3; ++q) {
                atomicMin(&s_lo[ax][b][q], l[q]);
                atomicMax(&s_hi[ax][b][q], h[q]);
            }
            atomicAdd(&s_cnt[ax][b], 1u);
        }
    }
    __syncthreads();
    // (c) SAH at every bin boundary: candidate = (axis, first bin of the right side)
    {
        float cost = INFINITY;
        int pick = -1;
        for (int cand = tid; cand < 3 * (kSahBins - 1); cand += kSahThreads) {
            const int ax = cand / (kSahBins - 1), s = cand % (kSahBins - 1) + 1;
            float3 llo = f3(INFINITY, INFINITY, INFINITY), lhi = f3(-INFINITY, -INFINITY, -INFINITY), rlo = llo, rhi = lhi;
            uint32_t nl = 0, nr = 0;
            for (int b = 0; b < kSahBins; ++b) {
                const uint32_t c = s_cnt[ax][b];
                if (!c)
                    continue;
                const float3 blo = f3(ordered_to_float(s_lo[ax][b][0]), ordered_to_float(s_lo[ax][b][1]), ordered_to_float(s_lo[ax][b][2]));
                const float3 bhi = f3(ordered_to_float(s_hi[ax][b][0]), ordered_to_float(s_hi[ax][b][1]), ordered_to_float(s_hi[ax][b][2]));
                if (b < s) {
                    llo = f3(fminf(llo.x, blo.x), fminf(llo.y, blo.y), fminf(llo.z, blo.z));
                    lhi = f3(fmaxf(lhi.x, bhi.x), fmaxf(lhi.y, bhi.y), fmaxf(lhi.z, bhi.z));
                    nl += c;
                } else {
                    rlo = f3(fminf(rlo.x, blo.x), fminf(rlo.y, blo.y), fminf(rlo.z, blo.z));
                    rhi = f3(fmaxf(rhi.x, bhi.x), fmaxf(rhi.y, bhi.y), fmaxf(rhi.z, bhi.z));
                    nr += c;
                }
            }
            if (nl && nr) {
                const float cst = box_half_area(llo, lhi) * (float)nl + box_half_area(rlo, rhi) * (float)nr;
                if (cst < cost) {
                    cost = cst;
                    pick = cand;
                }
            }
        }
        s_cost[tid] = cost;
        s_pick[tid] = pick;
    }
    __syncthreads();
    for (int off = kSahThreads / 2; off > 0; off >>= 1) { // argmin; ties -> the smaller candidate index (deterministic)
        if (tid < off) {
            const float c2 = s_cost[tid + off];
            const int p2 = s_pick[tid + off];
            if (p2 >= 0 && (s_pick[tid] < 0 || c2 < s_cost[tid] || (c2 == s_cost[tid] && p2 < s_pick[tid]))) {
                s_cost[tid] = c2;
                s_pick[tid] = p2;
            }
        }
        __syncthreads();
    }
    const int pick = s_pick[0];
    const int axis = pick >= 0 ? pick / (kSahBins - 1) : 3;
    const int split = pick >= 0 ? pick % (kSahBins - 1) + 1 : 1;
    uint32_t n_left = 0;
    for (int b = 0; b < split; ++b)
        n_left += s_cnt[axis][b];
    if (tid == 0) {
        s_run[0] = 0u;
        s_run[1] = 0u;
        SahSplit sp;
        sp.mid = begin + n_left;
        float l[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int b = 0; b < kSahBins; ++b) {
            if (!s_cnt[axis][b])
                continue;
            float* d = b < split ? l : r;
            for (int q = 0; q < 3; ++q) {
                d[q] = fminf(d[q], ordered_to_float(s_lo[axis][b][q]));
                d[3 + q] = fmaxf(d[3 + q], ordered_to_float(s_hi[axis][b][q]));
            }
        }
        for (int q = 0; q < 6; ++q) {
            sp.lbox[q] = l[q];
            sp.rbox[q] = r[q];
        }
        splits[blockIdx.x] = sp;
    }
    __syncthreads();
    // (d) stable partition of the run into idx_out
    for (uint32_t base = begin; base < end; base += kSahThreads) {
        const uint32_t i = base + tid;
        uint32_t p = 0, flag = 0;
        const bool in = i < end;
        if (in) {
            p = idx_in[i];
            int b;
            if (axis < 3) {
                const float4 lo = N.lo[p], hi = N.hi[p];
                const float c = axis == 0 ? 0.5f * (lo.x + hi.x) : (axis == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z));
                b = sah_bin(c, cmin[axis], scale[axis]);
            } else {
                b = i < half ? 0 : 1;
            }
            flag = b < split ? 1u : 0u;
        }
        s_scan[tid] = flag;
        __syncthreads();
        for (int off = 1; off < kSahThreads; off <<= 1) { // inclusive scan of the flags
            const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const uint32_t incl = s_scan[tid], lefts_before = incl - flag, chunk_lefts = s_scan[kSahThreads - 1];
        const uint32_t run_l = s_run[0], run_r = s_run[1];
        if (in) {
            const uint32_t dst = flag ? begin + run_l + lefts_before : begin + n_left + run_r + ((uint32_t)tid - lefts_before);
            idx_out[dst] = p;
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t chunk = min((uint32_t)kSahThreads, end - base);
            s_run[0] = run_l + chunk_lefts;
            s_run[1] = run_r + (chunk - chunk_lefts);
        }
        __syncthreads();
    }
}

// ---- the same split for the FIRST levels, spread over the chip (round 4) ----
// At the top of the tree a level has a handful of segments of 10^4 .. 10^5 primitives; one workgroup per segment (above) walks
// such a run three times in strides of 1024 with LDS atomics in its way -- the root split alone took 6.4 ms of a 16-ms build.
// Here a segment is cut into SLICES of kSahSlice primitives, one workgroup each: centroid bounds and bins are reduced per
// slice in LDS and merged into per-segment global words with ordered-uint atomicMin / atomicMax and integer adds (all
// order-independent), one small workgroup per segment evaluates the very same SAH candidates in the very same order, and the
// stable partition becomes count -> per-segment scan over the slices -> scatter.  Same arithmetic, same tie-breaks, same
// primitive order: the tree is bit-identical to the one-workgroup-per-segment build (tests/test_gi_gpu.py compares node
// counts, depth and whole frames against the numbers of that build).
constexpr uint32_t kSahSlice = 2048u;      // primitives per slice
constexpr uint32_t kSahBigMaxSegs = 1024u; // segments of a level the slice index is built for (more: the level is not "big")
struct SahBig {
    uint32_t* slice_begin;  // [n_segs + 1] exclusive prefix of the segments' slice counts
    uint32_t* cb;           // [n_segs][6] centroid bounds, ordered uints {min xyz, max xyz}
    uint32_t* bins;         // [n_segs][4][kSahBins][7] {lo xyz, hi xyz (ordered uints), count}
    uint32_t* slice_left;   // [n_slices] lefts in the slice, then (after the scan) lefts in the segment's slices before it
};
struct SahBigPick {
    int axis, split;
    uint32_t n_left, pad;
    float cmin[3], scale[3];
};

__global__ __launch_bounds__(1024) void sah_big_index_kernel(const SahSegment* __restrict__ segs, uint32_t n_segs, SahBig B)
{
    __shared__ uint32_t s_scan[kSahBigMaxSegs];
    const uint32_t tid = threadIdx.x;
    uint32_t mine = 0;
    if (tid < n_segs)
        mine = (segs[tid].end - segs[tid].begin + kSahSlice - 1u) / kSahSlice;
    s_scan[tid] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < kSahBigMaxSegs; off <<= 1) {
        const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    if (tid < n_segs) {
        B.slice_begin[tid + 1] = s_scan[tid];
        if (tid == 0)
            B.slice_begin[0] = 0u;
        for (int q = 0; q < 3; ++q) {
            B.cb[6 * tid + q] = 0xffffffffu;
            B.cb[6 * tid + 3 + q] = 0u;
        }
    }
    // the bins of every segment: lo = +max, hi = 0, count = 0
    for (uint32_t k = tid; k < n_segs * 4u * kSahBins; k += 1024u) {
        uint32_t* b = B.bins + 7u * k;
        b[0] = b[1] = b[2] = 0xffffffffu;
        b[3] = b[4] = b[5] = 0u;
        b[6] = 0u;
    }
}

// slice -> (segment, first primitive, one-past-last primitive); false: this block has no slice
__device__ __forceinline__ bool sah_big_slice(const SahSegment* segs, uint32_t n_segs, const SahBig& B, uint32_t& seg, uint32_t& first, uint32_t& last, SahSegment& sg)
{
    const uint32_t total = B.slice_begin[n_segs];
    if (blockIdx.x >= total)
        return false;
    uint32_t lo = 0, hi = n_segs - 1u; // the segment whose slices contain blockIdx.x
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (B.slice_begin[mid + 1] > blockIdx.x)
            hi = mid;
        else
            lo = mid + 1u;
    }
    seg = lo;
    sg = segs[seg];
    first = sg.begin + (blockIdx.x - B.slice_begin[seg]) * kSahSlice;
    last = min(sg.end, first + kSahSlice);
    return true;
}

__global__ __launch_bounds__(256) void sah_big_bounds_kernel(BinaryNodes N, const SahSegment* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ idx_in, SahBig B)
{
    __shared__ uint32_t s_cmin[3], s_cmax[3];
    uint32_t seg, first, last;
    SahSegment sg;
    if (!sah_big_slice(segs, n_segs, B, seg, first, last, sg))
        return;
    const int tid = threadIdx.x;
    if (tid < 3) {
        s_cmin[tid] = 0xffffffffu;
        s_cmax[tid] = 0u;
    }
    __syncthreads();
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = first + tid; i < last; i += 256u) {
        const uint32_t p = idx_in[i];
        const float4 lo = N.lo[p], hi = N.hi[p];
        const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
        for (int q = 0; q < 3; ++q) {
            mn[q] = fminf(mn[q], c[q]);
            mx[q] = fmaxf(mx[q], c[q]);
        }
    }
    for (int q = 0; q < 3; ++q) {
        atomicMin(&s_cmin[q], float_to_ordered(mn[q]));
        atomicMax(&s_cmax[q], float_to_ordered(mx[q]));
    }
    __syncthreads();
    if (tid < 3) {
        atomicMin(&B.cb[6 * seg + tid], s_cmin[tid]);
        atomicMax(&B.cb[6 * seg + 3 + tid], s_cmax[tid]);
    }
}

__device__ __forceinline__ void sah_big_scale(const SahBig& B, uint32_t seg, float cmin[3], float scale[3])
{
    for (int q = 0; q < 3; ++q) {
        cmin[q] = ordered_to_float(B.cb[6 * seg + q]);
        const float ext = ordered_to_float(B.cb[6 * seg + 3 + q]) - cmin[q];
        scale[q] = ext > 0.0f ? (float)kSahBins / ext : 0.0f;
    }
}

__global__ __launch_bounds__(256) void sah_big_bin_kernel(BinaryNodes N, const SahSegment* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ idx_in, SahBig B)
{
    __shared__ uint32_t s_lo[4][kSahBins][3], s_hi[4][kSahBins][3], s_cnt[4][kSahBins];
    uint32_t seg, first, last;
    SahSegment sg;
    if (!sah_big_slice(segs, n_segs, B, seg, first, last, sg))
        return;
    const int tid = threadIdx.x;
    for (int k = tid; k < 4 * kSahBins; k += 256) {
        const int ax = k / kSahBins, b = k % kSahBins;
        for (int q = 0; q < 3; ++q) {
            s_lo[ax][b][q] = 0xffffffffu;
            s_hi[ax][b][q] = 0u;
        }
        s_cnt[ax][b] = 0u;
    }
    __syncthreads();
    float cmin[3], scale[3];
    sah_big_scale(B, seg, cmin, scale);
    const uint32_t half = sg.begin + (sg.end - sg.begin) / 2;
    for (uint32_t i = first + tid; i < last; i += 256u) {
        const uint32_t p = idx_in[i];
        const float4 lo = N.lo[p], hi = N.hi[p];
        const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
        const uint32_t l[3] = {float_to_ordered(lo.x), float_to_ordered(lo.y), float_to_ordered(lo.z)};
        const uint32_t h[3] = {float_to_ordered(hi.x), float_to_ordered(hi.y), float_to_ordered(hi.z)};
        for (int ax = 0; ax < 4; ++ax) {
            const int b = ax < 3 ? sah_bin(c[ax], cmin[ax], scale[ax]) : (i < half ? 0 : 1);
            for (int q = 0; q < 3; ++q) {
                atomicMin(&s_lo[ax][b][q], l[q]);
                atomicMax(&s_hi[ax][b][q], h[q]);
            }
            atomicAdd(&s_cnt[ax][b], 1u);
        }
    }
    __syncthreads();
    for (int k = tid; k < 4 * kSahBins; k += 256) {
        const int ax = k / kSahBins, b = k % kSahBins;
        if (!s_cnt[ax][b])
            continue;
        uint32_t* g = B.bins + 7u * ((size_t)seg * 4u * kSahBins + (uint32_t)k);
        for (int q = 0; q < 3; ++q) {
            atomicMin(&g[q], s_lo[ax][b][q]);
            atomicMax(&g[3 + q], s_hi[ax][b][q]);
        }
        atomicAdd(&g[6], s_cnt[ax][b]);
    }
}

// one workgroup per segment: the SAH candidates of sah_split_kernel, evaluated on the merged bins, in the same order
__global__ __launch_bounds__(128) void sah_big_pick_kernel(const SahSegment* __restrict__ segs, uint32_t n_segs, SahBig B, SahSplit* __restrict__ splits, SahBigPick* __restrict__ picks)
{
    constexpr int kT = 128;
    __shared__ float s_cost[kT];
    __shared__ int s_pick[kT];
    const uint32_t seg = blockIdx.x;
    const SahSegment sg = segs[seg];
    const uint32_t* bins = B.bins + 7u * ((size_t)seg * 4u * kSahBins);
    auto cnt = [&](int ax, int b) { return bins[7 * (ax * kSahBins + b) + 6]; };
    auto blo = [&](int ax, int b, int q) { return ordered_to_float(bins[7 * (ax * kSahBins + b) + q]); };
    auto bhi = [&](int ax, int b, int q) { return ordered_to_float(bins[7 * (ax * kSahBins + b) + 3 + q]); };
    const int tid = threadIdx.x;
    {
        float cost = INFINITY;
        int pick = -1;
        for (int cand = tid; cand < 3 * (kSahBins - 1); cand += kT) {
            const int ax = cand / (kSahBins - 1), sp = cand % (kSahBins - 1) + 1;
            float3 llo = f3(INFINITY, INFINITY, INFINITY), lhi = f3(-INFINITY, -INFINITY, -INFINITY), rlo = llo, rhi = lhi;
            uint32_t nl = 0, nr = 0;
            for (int b = 0; b < kSahBins; ++b) {
                const uint32_t c = cnt(ax, b);
                if (!c)
                    continue;
                const float3 l3 = f3(blo(ax, b, 0), blo(ax, b, 1), blo(ax, b, 2)), h3 = f3(bhi(ax, b, 0), bhi(ax, b, 1), bhi(ax, b, 2));
                if (b < sp) {
                    llo = f3(fminf(llo.x, l3.x), fminf(llo.y, l3.y), fminf(llo.z, l3.z));
                    lhi = f3(fmaxf(lhi.x, h3.x), fmaxf(lhi.y, h3.y), fmaxf(lhi.z, h3.z));
                    nl += c;
                } else {
                    rlo = f3(fminf(rlo.x, l3.x), fminf(rlo.y, l3.y), fminf(rlo.z, l3.z));
                    rhi = f3(fmaxf(rhi.x, h3.x), fmaxf(rhi.y, h3.y), fmaxf(rhi.z, h3.z));
                    nr += c;
                }
            }
            if (nl && nr) {
                const float cst = box_half_area(llo, lhi) * (float)nl + box_half_area(rlo, rhi) * (float)nr;
                if (cst < cost) {
                    cost = cst;
                    pick = cand;
                }
            }
        }
        s_cost[tid] = cost;
        s_pick[tid] = pick;
    }
    __syncthreads();
    for (int off = kT / 2; off > 0; off >>= 1) { // argmin; ties -> the smaller candidate index (as sah_split_kernel)
        if (tid < off) {
            const float c2 = s_cost[tid + off];
            const int p2 = s_pick[tid + off];
            if (p2 >= 0 && (s_pick[tid] < 0 || c2 < s_cost[tid] || (c2 == s_cost[tid] && p2 < s_pick[tid]))) {
                s_cost[tid] = c2;
                s_pick[tid] = p2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int pick = s_pick[0];
        const int axis = pick >= 0 ? pick / (kSahBins - 1) : 3;
        const int split = pick >= 0 ? pick % (kSahBins - 1) + 1 : 1;
        uint32_t n_left = 0;
        for (int b = 0; b < split; ++b)
            n_left += cnt(axis, b);
        SahSplit sp;
        sp.mid = sg.begin + n_left;
        float l[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int b = 0; b < kSahBins; ++b) {
            if (!cnt(axis, b))
                continue;
            float* d = b < split ? l : r;
            for (int q = 0; q < 3; ++q) {
                d[q] = fminf(d[q], blo(axis, b, q));
                d[3 + q] = fmaxf(d[3 + q], bhi(axis, b, q));
            }
        }
        for (int q = 0; q < 6; ++q) {
            sp.lbox[q] = l[q];
            sp.rbox[q] = r[q];
        }
        splits[seg] = sp;
        SahBigPick pk;
        pk.axis = axis;
        pk.split = split;
        pk.n_left = n_left;
        pk.pad = 0;
        sah_big_scale(B, seg, pk.cmin, pk.scale);
        picks[seg] = pk;
    }
}

__device__ __forceinline__ uint32_t sah_big_flag(const BinaryNodes& N, uint32_t p, uint32_t i, uint32_t half, const SahBigPick& pk)
{
    int b;
    if (pk.axis < 3) {
        const float4 lo = N.lo[p], hi = N.hi[p];
        const float c = pk.axis == 0 ? 0.5f * (lo.x + hi.x) : (pk.axis == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z));
        b = sah_bin(c, pk.cmin[pk.axis], pk.scale[pk.axis]);
    } else {
        b = i < half ? 0 : 1;
    }
    return b < pk.split ? 1u : 0u;
}

// lefts per slice
__global__ __launch_bounds__(256) void sah_big_count_kernel(BinaryNodes N, const SahSegment* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ idx_in, SahBig B,
                                                            const SahBigPick* __restrict__ picks)
{
    __shared__ uint32_t s_n;
    uint32_t seg, first, last;
    SahSegment sg;
    if (!sah_big_slice(segs, n_segs, B, seg, first, last, sg))
        return;
    if (threadIdx.x == 0)
        s_n = 0u;
    __syncthreads();
    const SahBigPick pk = picks[seg];
    const uint32_t half = sg.begin + (sg.end - sg.begin) / 2;
    uint32_t mine = 0;
    for (uint32_t i = first + threadIdx.x; i < last; i += 256u)
        mine += sah_big_flag(N, idx_in[i], i, half, pk);
    atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0)
        B.slice_left[blockIdx.x] = s_n;
}

// per segment: exclusive scan of its slices' left counts (a segment has at most n / kSahSlice slices: one thread walks them)
__global__ void sah_big_scan_kernel(uint32_t n_segs, SahBig B)
{
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= n_segs)
        return;
    uint32_t run = 0;
    for (uint32_t k = B.slice_begin[seg]; k < B.slice_begin[seg + 1]; ++k) {
        const uint32_t c = B.slice_left[k];
        B.slice_left[k] = run;
        run += c;
    }
}

// stable partition of a slice into idx_out
__global__ __launch_bounds__(256) void sah_big_scatter_kernel(BinaryNodes N, const SahSegment* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ idx_in,
                                                              uint32_t* __restrict__ idx_out, SahBig B, const SahBigPick* __restrict__ picks)
{
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_run[2];
    uint32_t seg, first, last;
    SahSegment sg;
    if (!sah_big_slice(segs, n_segs, B, seg, first, last, sg))
        return;
    const int tid = threadIdx.x;
    const SahBigPick pk = picks[seg];
    const uint32_t half = sg.begin + (sg.end - sg.begin) / 2;
    const uint32_t lefts_before = B.slice_left[blockIdx.x];            // lefts of the segment in the slices before this one
    const uint32_t rights_before = (first - sg.begin) - lefts_before;  // ... and rights
    if (tid == 0) {
        s_run[0] = 0u;
        s_run[1] = 0u;
    }
    __syncthreads();
    for (uint32_t base = first; base < last; base += 256u) {
        const uint32_t i = base + tid;
        uint32_t p = 0, flag = 0;
        const bool in = i < last;
        if (in) {
            p = idx_in[i];
            flag = sah_big_flag(N, p, i, half, pk);
        }
        s_scan[tid] = flag;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const uint32_t incl = s_scan[tid], lb = incl - flag, chunk_lefts = s_scan[255];
        const uint32_t run_l = s_run[0], run_r = s_run[1];
        if (in) {
            const uint32_t dst = flag ? sg.begin + lefts_before + run_l + lb : sg.begin + pk.n_left + rights_before + run_r + ((uint32_t)tid - lb);
            idx_out[dst] = p;
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t chunk = min(256u, last - base);
            s_run[0] = run_l + chunk_lefts;
            s_run[1] = run_r + (chunk - chunk_lefts);
        }
        __syncthreads();
    }
}

// per segment: how many child nodes it creates (children of >= 2 triangles) and how many of them go on (>= 3 triangles),
// packed (nodes << 32 | segments) for one prefix sum
__global__ void sah_count_kernel(const SahSegment* __restrict__ segs, const SahSplit* __restrict__ splits, uint32_t n_segs, unsigned long long* counts)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_segs)
        return;
    const uint32_t cl = splits[k].mid - segs[k].begin, cr = segs[k].end - splits[k].mid;
    const unsigned long long nodes = (cl >= 2 ? 1u : 0u) + (cr >= 2 ? 1u : 0u), more = (cl >= 3 ? 1u : 0u) + (cr >= 3 ? 1u : 0u);
    counts[k] = (nodes << 32) | more;
}

// state[0] = segments of the next level, state[1] = next free node id
__global__ void sah_emit_kernel(BinaryNodes N, const SahSegment* __restrict__ segs, const SahSplit* __restrict__ splits, uint32_t n_segs,
                                const unsigned long long* __restrict__ counts, const unsigned long long* __restrict__ scan, const uint32_t* __restrict__ idx,
                                const uint32_t* __restrict__ state, uint32_t* __restrict__ state_out, SahSegment* __restrict__ next_segs)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_segs)
        return;
    const SahSegment sg = segs[k];
    const SahSplit sp = splits[k];
    uint32_t node = state[1] + (uint32_t)(scan[k] >> 32), slot = (uint32_t)scan[k];
    if (k == n_segs - 1) {
        const unsigned long long tot = scan[k] + counts[k];
        state_out[0] = (uint32_t)tot;
        state_out[1] = state[1] + (uint32_t)(tot >> 32);
    }
    int ref[2];
    const uint32_t b[2] = {sg.begin, sp.mid}, e[2] = {sp.mid, sg.end};
    for (int side = 0; side < 2; ++side) {
        const uint32_t c = e[side] - b[side];
        if (c == 1u) {
            ref[side] = (int)idx[b[side]];
            continue;
        }
        const float* bx = side == 0 ? sp.lbox : sp.rbox;
        const uint32_t id = node++;
        ref[side] = (int)id;
        N.size[id] = c;
        if (c == 2u) { // a pair: its children are the two triangles; it is not split further
            N.lo[id] = make_float4(bx[0], bx[1], bx[2], __int_as_float((int)idx[b[side]]));
            N.hi[id] = make_float4(bx[3], bx[4], bx[5], __int_as_float((int)idx[b[side] + 1]));
        } else {     // children are filled in when its own segment is split, one level down
            N.lo[id] = make_float4(bx[0], bx[1], bx[2], __int_as_float(-1));
            N.hi[id] = make_float4(bx[3], bx[4], bx[5], __int_as_float(-1));
            next_segs[slot++] = SahSegment{b[side], e[side], id};
        }
    }
    // the split node itself: keep its box, set its children
    float4 lo = N.lo[sg.node], hi = N.hi[sg.node];
    lo.w = __int_as_float(ref[0]);
    hi.w = __int_as_float(ref[1]);
    N.lo[sg.node] = lo;
    N.hi[sg.node] = hi;
}

// ------------------------------------------------------------------------------------------------
// Collapse of the binary tree into the 128-byte BVH4 nodes the traverser walks, level by level on the device.
// A queue entry = (binary node that becomes a wide node, first slot of its triangles in the final order); a level's
// entries are consecutive, and an entry's position in the queue IS its wide node index (breadth-first layout: the
// children of a node are neighbours).  Per level: `open` picks the (up to) four children -- repeatedly opening the inner
// child with the largest box -- and counts the inner ones; a prefix sum places them in the next level; `emit` writes
// the node, copies leaf triangles into their final (leaf-order) slots and appends the next level's entries.
// The number of levels is the depth of the BVH4.
// ------------------------------------------------------------------------------------------------
struct CollapseArgs {
    BinaryNodes N;
    uint32_t n_tris;
    const float4* tris_in;  // Morton order
    float4* tris_out;       // leaf order of the final tree
    Bvh4Node* wide;
    uint32_t* q_node;       // per wide node: its binary node
    uint32_t* q_off;        // per wide node: first final triangle slot of its subtree
    int4* opened;           // per wide node: the chosen children (binary node ids, -1 = unused)
    uint32_t* inner_count;  // per wide node of the current level
    uint32_t* inner_scan;   // exclusive prefix sum of inner_count over the level
    uint32_t* level_state;  // [0] = entries of the next level
    uint32_t level_start, level_count;
    const float4* plan;     // per binary node: {C(n,1), C(n,2), C(n,3), decision bits} of collapse_plan_kernel, or null (greedy)
};

__device__ __forceinline__ bool collapse_is_leaf(const BinaryNodes& N, int node) { return N.size[node] <= (uint32_t)kMaxLeafTris; }

// Which descendants of a binary node become the (up to four) children of its wide node: chosen by cost, not by the
// "open the largest box" rule -- the optimal conversion of a binary BVH into a wide one by dynamic programming over the
// tree (Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017, section
// 3.1, for width 4).  For a binary node n with children l and r:
//   C(n, 1)      = cost of the subtree with n as the root of ONE wide node  = A(n) c_node + min_k [ C(l, k) + C(r, 4 - k) ]
//                  (a run of <= kMaxLeafTris triangles is a leaf: A(n) c_leaf)
//   C(n, i), i>1 = cost of the subtree as a forest of at most i entries of an ancestor's wide node
//                = min( C(n, i - 1),  min_k [ C(l, k) + C(r, i - k) ] )
// A = half the surface area of the node's box (the chance a ray visits it), c_node / c_leaf = the traversal's cost of an
// inner visit / a leaf visit (a leaf of one or two triangles costs the same six loads).  Children come one SAH pass after
// their parent, so the tree is swept bottom-up one pass at a time (node ids of a pass are contiguous).  The decisions ride
// in .w: bits 0-1 = k of C(n,1); bit 2 = C(n,2) splits (1,1), else it is C(n,1); bits 3-4 = C(n,3): 0 = take C(n,2), else k.
#ifndef NEB_COLLAPSE_CNODE
#define NEB_COLLAPSE_CNODE 1.0f
#endif
#ifndef NEB_COLLAPSE_CLEAF
#define NEB_COLLAPSE_CLEAF 1.2f
#endif
__global__ void collapse_plan_kernel(BinaryNodes N, uint32_t first, uint32_t count, float4* plan)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count)
        return;
    const uint32_t id = first + k;
    const float4 lo = N.lo[id], hi = N.hi[id];
    const float area = box_half_area(f3(lo.x, lo.y, lo.z), f3(hi.x, hi.y, hi.z));
    if (N.size[id] <= (uint32_t)kMaxLeafTris) {
        const float c = area * NEB_COLLAPSE_CLEAF;
        plan[id] = make_float4(c, c, c, __uint_as_float(0u));
        return;
    }
    const float4 pl = plan[__float_as_int(lo.w)], pr = plan[__float_as_int(hi.w)];
    const float cl[4] = {0.f, pl.x, pl.y, pl.z}, cr[4] = {0.f, pr.x, pr.y, pr.z};
    uint32_t k4 = 1u;
    float d4 = cl[1] + cr[3];
    if (cl[2] + cr[2] < d4)
        d4 = cl[2] + cr[2], k4 = 2u;
    if (cl[3] + cr[1] < d4)
        d4 = cl[3] + cr[1], k4 = 3u;
    const float c1 = area * NEB_COLLAPSE_CNODE + d4;
    const float d2 = cl[1] + cr[1];
    const float c2 = fminf(c1, d2);
    uint32_t k3 = 0u;
    float c3 = c2;
    if (cl[1] + cr[2] < c3)
        c3 = cl[1] + cr[2], k3 = 1u;
    if (cl[2] + cr[1] < c3)
        c3 = cl[2] + cr[1], k3 = 2u;
    plan[id] = make_float4(c1, c2, c3, __uint_as_float(k4 | (d2 < c1 ? 4u : 0u) | (k3 << 3)));
}

// the entries (binary node ids) that cover `node` when it may take up to `budget` slots of a wide node, left to right
__device__ __forceinline__ void collapse_expand(const CollapseArgs& a, int node, int budget, int* out, int& n_out)
{
    int st_node[8], st_budget[8];
    int sp = 0;
    st_node[sp] = node, st_budget[sp++] = budget;
    while (sp) {
        --sp;
        const int x = st_node[sp];
        int b = st_budget[sp];
        if (b <= 1 || collapse_is_leaf(a.N, x)) {
            out[n_out++] = x;
            continue;
        }
        const uint32_t bits = __float_as_uint(a.plan[x].w);
        int kl = 0; // slots of the left child; 0 = the node stays whole at this budget
        if (b >= 3) {
            const uint32_t k3 = (bits >> 3) & 3u;
            if (k3)
                kl = (int)k3;
            else
                b = 2; // C(n,3) == C(n,2)
        }
        if (b == 2 && kl == 0)
            kl = (bits & 4u) ? 1 : 0;
        if (kl == 0) {
            out[n_out++] = x;
            continue;
        }
        // right below left on the stack: the left subtree comes out first
        st_node[sp] = __float_as_int(a.N.hi[x].w), st_budget[sp++] = b - kl;
        st_node[sp] = __float_as_int(a.N.lo[x].w), st_budget[sp++] = kl;
    }
}

__global__ void collapse_open_kernel(CollapseArgs a)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.level_count)
        return;
    const uint32_t e = a.level_start + k;
    const int bn = (int)a.q_node[e];
    int c[4] = {__float_as_int(a.N.lo[bn].w), __float_as_int(a.N.hi[bn].w), -1, -1};
    int nc = 2;
    if (a.plan) { // the cost-optimal choice (collapse_plan_kernel)
        const int k4 = (int)(__float_as_uint(a.plan[bn].w) & 3u);
        const int l = c[0], r = c[1];
        nc = 0;
        collapse_expand(a, l, k4, c, nc);
        collapse_expand(a, r, 4 - k4, c, nc);
        for (int q = nc; q < 4; ++q)
            c[q] = -1;
    }
    while (!a.plan && nc < 4) { // (A/B arm, NEB_COLLAPSE_DP = 0) open the inner child with the largest surface area
        int best = -1;
        float best_area = -1.0f;
        for (int q = 0; q < nc; ++q) {
            if (collapse_is_leaf(a.N, c[q]))
                continue;
            const float4 lo = a.N.lo[c[q]], hi = a.N.hi[c[q]];
            const float ar = box_half_area(f3(lo.x, lo.y, lo.z), f3(hi.x, hi.y, hi.z));
            if (ar > best_area) {
                best_area = ar;
                best = q;
            }
        }
        if (best < 0)
            break;
        const int o = c[best];
        c[best] = __float_as_int(a.N.lo[o].w);
        c[nc++] = __float_as_int(a.N.hi[o].w);
    }
    uint32_t inner = 0;
    for (int q = 0; q < nc; ++q)
        inner += collapse_is_leaf(a.N, c[q]) ? 0u : 1u;
    a.opened[e] = make_int4(c[0], c[1], c[2], c[3]);
    a.inner_count[k] = inner;
}

// copies the (<= 4) triangles below binary node `node` into consecutive final slots starting at `slot`
__device__ void collapse_copy_leaf(const CollapseArgs& a, int node, uint32_t slot)
{
    int stack[8];
    int sp = 0;
    stack[sp++] = node;
    while (sp) {
        const int x = stack[--sp];
        if (x < (int)a.n_tris) {
            a.tris_out[3 * slot] = a.tris_in[3 * x];
            a.tris_out[3 * slot + 1] = a.tris_in[3 * x + 1];
            a.tris_out[3 * slot + 2] = a.tris_in[3 * x + 2];
            ++slot;
        } else {
            stack[sp++] = __float_as_int(a.N.hi[x].w); // right below left: the left subtree comes out first
            stack[sp++] = __float_as_int(a.N.lo[x].w);
        }
    }
}

__global__ void collapse_emit_kernel(CollapseArgs a)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.level_count)
        return;
    const uint32_t e = a.level_start + k;
    const int4 o = a.opened[e];
    const int c[4] = {o.x, o.y, o.z, o.w};
    uint32_t off = a.q_off[e];
    uint32_t next = a.level_start + a.level_count + a.inner_scan[k]; // wide index of this entry's first inner child
    float lo[3][4], hi[3][4];
    int ch[4];
    for (int q = 0; q < 4; ++q) {
        if (c[q] < 0) {
            for (int ax = 0; ax < 3; ++ax) {
                lo[ax][q] = INFINITY; // inverted box: never hit
                hi[ax][q] = -INFINITY;
            }
            ch[q] = ~0;
            continue;
        }
        const float4 l = a.N.lo[c[q]], h = a.N.hi[c[q]];
        lo[0][q] = l.x, lo[1][q] = l.y, lo[2][q] = l.z;
        hi[0][q] = h.x, hi[1][q] = h.y, hi[2][q] = h.z;
        const uint32_t cnt = a.N.size[c[q]];
        if (cnt <= (uint32_t)kMaxLeafTris) {
            collapse_copy_leaf(a, c[q], off);
            ch[q] = ~(int)((off << 2) | (cnt - 1u));
        } else {
            ch[q] = (int)next;
            a.q_node[next] = (uint32_t)c[q];
            a.q_off[next] = off;
            ++next;
        }
        off += cnt;
    }
    Bvh4Node nd;
    nd.lox = make_float4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
    nd.loy = make_float4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]);
    nd.loz = make_float4(lo[2][0], lo[2][1], lo[2][2], lo[2][3]);
    nd.hix = make_float4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]);
    nd.hiy = make_float4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]);
    nd.hiz = make_float4(hi[2][0], hi[2][1], hi[2][2], hi[2][3]);
    nd.child = make_int4(ch[0], ch[1], ch[2], ch[3]);
    nd.pad = make_int4(0, 0, 0, 0);
    a.wide[e] = nd;
    if (k == a.level_count - 1)
        a.level_state[0] = a.inner_scan[k] + a.inner_count[k];
}

// The 64-byte copy of the wide nodes that every ray walks (Bvh4NodeQ, gi_internal.h): per node and axis, origin = the
// lower corner of the used children moved out by a pad of 2^-20 of the node's largest coordinate (16 ulps), scale =
// padded extent / 255 nudged up until 255 steps reach the padded upper corner; every child plane is rounded away from
// the child (floor for lower planes, ceil for upper ones) and then checked in the reference decoding fmaf(q, scale,
// origin) with STRICT inequalities, so every decoded plane lies strictly outside the exact one -- by at least the pad
// at the node's own faces (q = 0 / 255), by a rounding step of the coordinate elsewhere.  What that proves: the decoded
// box contains the exact box in fmaf(q, scale, origin) arithmetic, with margin.  The traversal does NOT evaluate that
// expression: it folds the ray in, t = fmaf(q, scale * inv, fmaf(origin, inv, -o * inv)) (gi_device.h), three more
// roundings of about 2^-24 of max(|node coordinate|, |ray origin coordinate|) each.  The pad covers them for rays that
// start within the scene's own coordinate range (every ray of this path does: origins are G-buffer positions and hit
// points); like any two fp32 slab tests -- the 128-byte walk against the oracle's, say -- this one and an exact walk can
// still disagree on a ray that grazes a box face at the rounding level, which is also where the triangle test itself is
// decided by an ulp.
__global__ void quantise_nodes_kernel(const Bvh4Node* __restrict__ nodes, uint32_t n, Bvh4NodeQ* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const Bvh4Node nd = nodes[i];
    const float lo[3][4] = {{nd.lox.x, nd.lox.y, nd.lox.z, nd.lox.w}, {nd.loy.x, nd.loy.y, nd.loy.z, nd.loy.w}, {nd.loz.x, nd.loz.y, nd.loz.z, nd.loz.w}};
    const float hi[3][4] = {{nd.hix.x, nd.hix.y, nd.hix.z, nd.hix.w}, {nd.hiy.x, nd.hiy.y, nd.hiy.z, nd.hiy.w}, {nd.hiz.x, nd.hiz.y, nd.hiz.z, nd.hiz.w}};
    float org[3], scl[3];
    uint32_t qlo[3], qhi[3];
    for (int ax = 0; ax < 3; ++ax) {
        float mn = INFINITY, mx = -INFINITY;
        for (int q = 0; q < 4; ++q)
            if (lo[ax][q] <= hi[ax][q]) { // (unused slots are inverted boxes)
                mn = fminf(mn, lo[ax][q]);
                mx = fmaxf(mx, hi[ax][q]);
            }
        if (!(mn <= mx))
            mn = mx = 0.0f; // (a node without children does not occur; keep the record finite)
        {
            const float pad = fmaxf(fmaxf(fabsf(mn), fabsf(mx)), mx - mn) * 0x1p-20f;
            mn -= pad;
            mx += pad;
        }
        // scale: 255 steps must reach the upper corner in the decoder's arithmetic.  (mx - mn) and the division round by
        // 2^-24 each, so a 2^-21 margin makes 255 * sc >= mx - mn exactly, and fmaf rounds monotonically: no search
        // (the loop is a bounded safeguard -- a node at 1e4 with an extent of 1e-4 must not spin here)
        float sc = fmaxf((mx - mn) / 255.0f * (1.0f + 0x1p-21f), 1e-30f);
        for (int guard = 0; guard < 8 && fmaf(255.0f, sc, mn) < mx; ++guard)
            sc *= 1.0f + 0x1p-20f;
        org[ax] = mn;
        scl[ax] = sc;
        qlo[ax] = qhi[ax] = 0u;
        for (int q = 0; q < 4; ++q) {
            uint32_t l = 255u, h = 0u; // unused: inverted
            if (lo[ax][q] <= hi[ax][q]) {
                l = (uint32_t)fminf(floorf((lo[ax][q] - mn) / sc), 255.0f);
                while (l > 0u && fmaf((float)l, sc, mn) >= lo[ax][q]) // strictly outside (l == 0 decodes to the padded corner)
                    --l;
                h = (uint32_t)fminf(ceilf((hi[ax][q] - mn) / sc), 255.0f);
                while (h < 255u && fmaf((float)h, sc, mn) <= hi[ax][q])
                    ++h;
            }
            qlo[ax] |= l << (8 * q);
            qhi[ax] |= h << (8 * q);
        }
    }
    Bvh4NodeQ o;
    o.ox = org[0], o.oy = org[1], o.oz = org[2];
    o.sx = scl[0], o.sy = scl[1], o.sz = scl[2];
    o.qlox = qlo[0], o.qloy = qlo[1], o.qloz = qlo[2];
    o.qhix = qhi[0], o.qhiy = qhi[1], o.qhiz = qhi[2];
    o.child = nd.child;
    out[i] = o;
}

} // namespace neb

using namespace neb;

extern "C" {

int neb_gi_set_scene(neb_ctx* ctx, const neb_geometry_desc* geoms, uint32_t n_geoms, const neb_material_desc* mats,
                     uint32_t n_mats, const neb_texture_desc* texs, uint32_t n_texs)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if ((n_geoms && !geoms) || (n_mats && !mats) || (n_texs && !texs))
        return gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_set_scene: null table");
    GI_GUARD(ctx);
    GI_HIP(ctx, hipDeviceSynchronize());
    gi_destroy(ctx->gi);
    ctx->gi = nullptr;
    GiState* g = new GiState();
    std::vector<DevGeom> dgeoms(n_geoms);
    std::vector<DevMat> dmats(n_mats);
    std::vector<DevTex> dtexs(n_texs);
    std::vector<uint32_t> indices;
    std::vector<float> normals, uvs, tangents;
    uint32_t vertex_base = 0;
    float smin[3] = {3.4e38f, 3.4e38f, 3.4e38f}, smax[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t gi = 0; gi < n_geoms; ++gi) {
        const neb_geometry_desc& s = geoms[gi];
        if (s.indices && s.indexStride != 2 && s.indexStride != 4) {
            delete g;
            return gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_set_scene: indexStride must be 2 or 4");
        }
        DevGeom& d = dgeoms[gi];
        const float* m = s.surfaceToWorld;
        const float m3[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
        memcpy(d.m, m3, sizeof(m3));
        d.material = (s.materialIndex >= 0 && (uint32_t)s.materialIndex < n_mats) ? s.materialIndex : -1;
        d.firstIndex = (uint32_t)indices.size();
        d.vertexBase = vertex_base;
        d.valid = (s.indices && s.attributes[0] && s.attributes[1] && s.attributes[2] && s.attributes[3]) ? 1u : 0u;
        d.pad[0] = d.pad[1] = d.pad[2] = 0;
        auto rd_index = [&](uint32_t i) -> uint32_t {
            const uint8_t* p = (const uint8_t*)s.indices + (size_t)i * s.indexStride;
            if (s.indexStride == 2) {
                uint16_t v;
                memcpy(&v, p, 2);
                return v;
            }
            uint32_t v;
            memcpy(&v, p, 4);
            return v;
        };
        auto rd_attr = [&](int a, uint32_t vtx, float* out, int n) {
            if (s.attributes[a])
                memcpy(out, (const uint8_t*)s.attributes[a] + (size_t)vtx * s.attributeStrides[a], sizeof(float) * n);
            else
                for (int k = 0; k < n; ++k)
                    out[k] = 0.f;
        };
        const uint32_t ntri = s.indices ? s.numIndices / 3 : 0;
        for (uint32_t i = 0; i < ntri * 3; ++i) {
            const uint32_t v = rd_index(i);
            if (v >= s.numVertices) {
                delete g;
                return gi_fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_gi_set_scene: index beyond numVertices");
            }
            indices.push_back(v);
        }
        for (uint32_t v = 0; v < s.numVertices; ++v) {
            float t[4];
            rd_attr(1, v, t, 3);
            normals.insert(normals.end(), t, t + 3);
            rd_attr(2, v, t, 2);
            uvs.insert(uvs.end(), t, t + 2);
            rd_attr(3, v, t, 4);
            tangents.insert(tangents.end(), t, t + 4);
        }
        vertex_base += s.numVertices;
        // bake world-space triangles: world = (p,1) * M (the correct instance transform)
        if (s.attributes[0]) {
            for (uint32_t p = 0; p < ntri; ++p) {
                float w[3][3];
                for (int k = 0; k < 3; ++k) {
                    float a[3];
                    rd_attr(0, rd_index(3 * p + k), a, 3);
                    w[k][0] = a[0] * m[0] + a[1] * m[4] + a[2] * m[8] + m[12];
                    w[k][1] = a[0] * m[1] + a[1] * m[5] + a[2] * m[9] + m[13];
                    w[k][2] = a[0] * m[2] + a[1] * m[6] + a[2] * m[10] + m[14];
                    for (int q = 0; q < 3; ++q) {
                        // (a position that is not a finite number has no place in a tree: min / max drop NaNs silently, the SAH's areas would not)
                        if (!(fabsf(w[k][q]) <= 3.0e38f)) {
                            delete g;
                            return gi_fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_gi_set_scene: a vertex position (after the instance transform) is not a finite number");
                        }
                        smin[q] = fminf(smin[q], w[k][q]);
                        smax[q] = fmaxf(smax[q], w[k][q]);
                    }
                }
                float t12[12] = {w[0][0], w[0][1], w[0][2], w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2],
                                 w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2], 0.f, 0.f, 0.f};
                memcpy(&t12[9], &gi, 4);
                memcpy(&t12[10], &p, 4);
                g->h_tris.insert(g->h_tris.end(), t12, t12 + 12);
            }
        }
    }
    for (uint32_t i = 0; i < n_mats; ++i) {
        DevMat& d = dmats[i];
        for (int k = 0; k < 3; ++k)
            d.tex[k] = (mats[i].textureIndices[k] >= 0 && (uint32_t)mats[i].textureIndices[k] < n_texs) ? mats[i].textureIndices[k] : -1;
        d.albedo[0] = mats[i].albedo[0];
        d.albedo[1] = mats[i].albedo[1];
        d.albedo[2] = mats[i].albedo[2];
        d.rough = mats[i].roughnessMetalness[0];
        d.metal = mats[i].roughnessMetalness[1];
    }
    // ---- textures: the raw RGBA8 images go up once; the bilinear-footprint tables are built from them on the device ----
    // (a footprint entry holds the 4 texels of a bilinear fetch, wrap applied: 16 B per texel position and map, or 32 B per
    // texel position for a material's three maps together.  Sponza's 69 maps of 1024^2 make 0.8 GB of material bundles --
    // nothing to assemble on the host.)
    std::vector<size_t> raw_off(n_texs);
    size_t raw_total = 0;
    for (uint32_t i = 0; i < n_texs; ++i) {
        if (!texs[i].rgba8 || !texs[i].width || !texs[i].height) {
            delete g;
            return gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_set_scene: empty texture");
        }
        dtexs[i].offset = 0xffffffffu; // no standalone table unless a material needs one (below)
        dtexs[i].w = texs[i].width;
        dtexs[i].h = texs[i].height;
        dtexs[i].pad = 0;
        raw_off[i] = raw_total;
        raw_total += (size_t)texs[i].width * texs[i].height;
    }
    // materials whose three maps share one size keep their footprints interleaved (DevMat::bundle, 32 B per texel position);
    // every other map a material uses gets a standalone table
    constexpr size_t kBundleBudget = (size_t)4 << 30; // bytes; beyond it the remaining materials sample their maps separately
    size_t bundle_entries = 0, table_entries = 0;     // in 32-byte / 16-byte units
    std::vector<char> standalone(n_texs, 0);
    for (uint32_t i = 0; i < n_mats; ++i) {
        DevMat& d = dmats[i];
        d.bundle = d.bundle_w = d.bundle_h = d.pad = 0;
        bool bundled = false;
        if (d.tex[0] >= 0 && d.tex[1] >= 0 && d.tex[2] >= 0) {
            const DevTex &ta = dtexs[d.tex[0]], &tn = dtexs[d.tex[1]], &tr = dtexs[d.tex[2]];
            const size_t n_pos = (size_t)ta.w * ta.h;
            if (ta.w == tn.w && ta.w == tr.w && ta.h == tn.h && ta.h == tr.h && (bundle_entries + n_pos) * 32 <= kBundleBudget &&
                bundle_entries + n_pos <= 0xffffffffull) {
                d.bundle = (uint32_t)bundle_entries;
                d.bundle_w = ta.w;
                d.bundle_h = ta.h;
                bundle_entries += n_pos;
                bundled = true;
            }
        }
        if (!bundled)
            for (int k = 0; k < 3; ++k)
                if (d.tex[k] >= 0)
                    standalone[d.tex[k]] = 1;
    }
    for (uint32_t i = 0; i < n_texs; ++i)
        if (standalone[i]) {
            if (table_entries + (size_t)dtexs[i].w * dtexs[i].h > 0xfffffffeull) {
                delete g;
                return gi_fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_gi_set_scene: more than 2^32 texels in standalone texture tables");
            }
            dtexs[i].offset = (uint32_t)table_entries;
            table_entries += (size_t)dtexs[i].w * dtexs[i].h;
        }
    g->texture_table_bytes = table_entries * 16 + bundle_entries * 32;
    {
        uint32_t* d_raw = nullptr;
        uint4* d_tables = nullptr;
        uint4* d_bundles = nullptr;
        hipError_t te = hipSuccess;
        auto talloc = [&](void** p, size_t bytes, bool keep) {
            if (te != hipSuccess)
                return;
            te = hipMalloc(p, bytes ? bytes : 16);
            if (te == hipSuccess && keep)
                g->allocs.push_back(*p);
        };
        talloc((void**)&d_raw, raw_total * 4, false);
        talloc((void**)&d_tables, table_entries * 16, true);
        talloc((void**)&d_bundles, bundle_entries * 32, true);
        for (uint32_t i = 0; i < n_texs && te == hipSuccess; ++i)
            te = hipMemcpy(d_raw + raw_off[i], texs[i].rgba8, (size_t)texs[i].width * texs[i].height * 4, hipMemcpyHostToDevice);
        for (uint32_t i = 0; i < n_texs && te == hipSuccess; ++i)
            if (standalone[i]) {
                const size_t n_pos = (size_t)dtexs[i].w * dtexs[i].h;
                hipLaunchKernelGGL(texture_footprints_kernel, dim3((unsigned)((n_pos + 255) / 256)), dim3(256), 0, nullptr, d_raw + raw_off[i],
                                   dtexs[i].w, dtexs[i].h, d_tables + dtexs[i].offset, 1u);
                te = hipGetLastError();
            }
        for (uint32_t i = 0; i < n_mats && te == hipSuccess; ++i) {
            const DevMat& d = dmats[i];
            if (!d.bundle_w)
                continue;
            const size_t n_pos = (size_t)d.bundle_w * d.bundle_h;
            hipLaunchKernelGGL(material_bundle_kernel, dim3((unsigned)((n_pos + 255) / 256)), dim3(256), 0, nullptr, d_raw + raw_off[d.tex[0]],
                               d_raw + raw_off[d.tex[1]], d_raw + raw_off[d.tex[2]], d.bundle_w, d.bundle_h, d_bundles + 2 * (size_t)d.bundle);
            te = hipGetLastError();
        }
        if (te == hipSuccess)
            te = hipDeviceSynchronize();
        if (d_raw)
            (void)hipFree(d_raw);
        if (te != hipSuccess) {
            gi_destroy(g);
            return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_set_scene: texture tables", te);
        }
        g->view.texels = reinterpret_cast<const uint32_t*>(d_tables);
        g->view.bundles = d_bundles;
    }
    g->n_tris = (uint32_t)(g->h_tris.size() / 12);
    memcpy(g->scene_min, smin, sizeof(smin));
    memcpy(g->scene_max, smax, sizeof(smax));
    hipError_t e = hipSuccess;
    if ((e = upload(g, dgeoms, &g->view.geoms)) != hipSuccess || (e = upload(g, dmats, &g->view.mats)) != hipSuccess ||
        (e = upload(g, dtexs, &g->view.texs)) != hipSuccess || (e = upload(g, indices, &g->view.indices)) != hipSuccess ||
        (e = upload(g, normals, &g->view.normals)) != hipSuccess || (e = upload(g, uvs, &g->view.uvs)) != hipSuccess ||
        (e = upload(g, tangents, &g->view.tangents)) != hipSuccess) {
        gi_destroy(g);
        return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_set_scene: upload", e);
    }
    void* ctr = nullptr;
    if ((e = hipMalloc(&ctr, 16 * sizeof(unsigned long long))) != hipSuccess || (e = hipMemset(ctr, 0, 16 * sizeof(unsigned long long))) != hipSuccess) {
        gi_destroy(g);
        return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_set_scene: counter", e);
    }
    g->allocs.push_back(ctr);
    g->d_ray_counter = (unsigned long long*)ctr;
    g->view.n_tris = g->n_tris;
    ctx->gi = g;
    return NEB_OK;
}

// ------------------------------------------------------------------------------------------------
// Reference splitting (round 4).  A triangle far larger than its neighbours -- a wall or floor strip that crosses the atrium, a beam
// -- has a box that overlaps everything along its length, and a builder that sorts primitives by centroid can only put it high in
// the tree, where every ray meets it (measured on the long-thin variant of the stand-in: 17.9 node visits per bounce ray against
// 15.2).  The remedy of the split-BVH builders (Stich et al., "Spatial splits in bounding volume hierarchies", HPG 2009; as a
// pre-pass: Ernst & Greiner's early split clipping, Karras & Aila 2013) is to let such a triangle appear in SEVERAL leaves, each
// with the box of the piece of the triangle inside it: the triangle itself is never cut -- every copy is the whole triangle, tested
// by the same arithmetic, so hits cannot change and no crack can open -- only its REFERENCES multiply.
// Here, on the host, before the device build: references whose box area exceeds kSplitAreaFactor x the mean are halved along their
// longest axis at the midpoint (largest first, until at most kSplitBudget x n extra references exist); a child's box is the
// bounds of the triangle clipped to its half (Sutherland-Hodgman in double, rounded outwards).  A scene without such triangles
// gets no extra reference and the very tree it had before (the stand-in, Cornell boxes, the helmet: asserted in the tests).
// ------------------------------------------------------------------------------------------------
#ifndef NEB_SPLIT_AREA_FACTOR
#define NEB_SPLIT_AREA_FACTOR 16.0 // 0: no reference splitting (A/B arm).  Measured on the long-thin stand-in (profiles/r04_split_factors.txt),
                                   // node visits per bounce ray / GI dispatch: off 17.9 / 580 us, 64: 17.5 / 581, 16: 16.2 / 575, 4: 16.5 / 574 (and the
                                   // plain stand-in starts splitting its roof quads: 15.4 against 15.2, +4 % nodes).  At 16 the plain stand-in, the
                                   // Cornell boxes and the helmet get no extra reference: their trees are the ones they always had.
#endif
constexpr double kSplitAreaFactor = NEB_SPLIT_AREA_FACTOR, kSplitBudget = 0.25;

struct HostRef {
    uint32_t tri;
    float lo[3], hi[3];
    bool clipped; // false: the whole triangle (the device computes its box from the vertices, as it always did)
};

static double half_area(const float lo[3], const float hi[3])
{
    const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// bounds of triangle `t` (12 floats: v0, e1, e2, ...) clipped to the box [lo, hi]; false if nothing of it is inside
static bool clipped_bounds(const float* t, const float lo[3], const float hi[3], float out_lo[3], float out_hi[3])
{
    double poly[16][3], tmp[16][3];
    int n = 3;
    for (int c = 0; c < 3; ++c) {
        poly[0][c] = t[c];
        poly[1][c] = (double)t[c] + t[3 + c];
        poly[2][c] = (double)t[c] + t[6 + c];
    }
    for (int axis = 0; axis < 3 && n > 0; ++axis)
        for (int side = 0; side < 2 && n > 0; ++side) {
            const double plane = side == 0 ? (double)lo[axis] : (double)hi[axis], sgn = side == 0 ? 1.0 : -1.0;
            int m = 0;
            for (int i = 0; i < n; ++i) {
                const int j = (i + 1) % n;
                const double di = sgn * (poly[i][axis] - plane), dj = sgn * (poly[j][axis] - plane);
                if (di >= 0.0) {
                    for (int c = 0; c < 3; ++c)
                        tmp[m][c] = poly[i][c];
                    ++m;
                }
                if ((di >= 0.0) != (dj >= 0.0)) {
                    const double u = di / (di - dj);
                    for (int c = 0; c < 3; ++c)
                        tmp[m][c] = poly[i][c] + u * (poly[j][c] - poly[i][c]);
                    tmp[m][axis] = plane;
                    ++m;
                }
            }
            n = m;
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < 3; ++c)
                    poly[i][c] = tmp[i][c];
        }
    if (n == 0)
        return false;
    for (int c = 0; c < 3; ++c) {
        double mn = poly[0][c], mx = poly[0][c];
        for (int i = 1; i < n; ++i) {
            mn = std::min(mn, poly[i][c]);
            mx = std::max(mx, poly[i][c]);
        }
        // outwards to float, and never beyond the box the piece was cut from
        float fl = (float)mn, fh = (float)mx;
        if ((double)fl > mn)
            fl = std::nextafter(fl, -INFINITY);
        if ((double)fh < mx)
            fh = std::nextafter(fh, INFINITY);
        out_lo[c] = std::max(fl, lo[c]);
        out_hi[c] = std::min(fh, hi[c]);
        if (out_lo[c] > out_hi[c])
            out_lo[c] = out_hi[c] = 0.5f * (out_lo[c] + out_hi[c]);
    }
    return true;
}

// -> the references of the build, in triangle order (a split triangle's pieces follow one another); refs.size() == n_tris: no splits
static void split_references(const std::vector<float>& tris12, std::vector<HostRef>& refs)
{
    const uint32_t n = (uint32_t)(tris12.size() / 12);
    refs.resize(n);
    double sum_area = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        const float* t = &tris12[12 * (size_t)i];
        HostRef& r = refs[i];
        r.tri = i;
        r.clipped = false;
        for (int c = 0; c < 3; ++c) {
            const float v0 = t[c], v1 = t[c] + t[3 + c], v2 = t[c] + t[6 + c];
            r.lo[c] = std::min(v0, std::min(v1, v2));
            r.hi[c] = std::max(v0, std::max(v1, v2));
        }
        sum_area += half_area(r.lo, r.hi);
    }
    if (kSplitAreaFactor <= 0.0 || n < 8)
        return;
    const double limit = kSplitAreaFactor * sum_area / n;
    size_t budget = (size_t)(kSplitBudget * n);
    // work list of the references above the limit, largest first (a heap of (area, index into `extra` or `refs`))
    std::vector<std::pair<double, uint32_t>> heap;
    for (uint32_t i = 0; i < n; ++i) {
        const double a = half_area(refs[i].lo, refs[i].hi);
        if (a > limit)
            heap.emplace_back(a, i);
    }
    if (heap.empty())
        return;
    std::make_heap(heap.begin(), heap.end());
    std::vector<std::vector<HostRef>> pieces(n); // per split triangle: its current pieces
    std::vector<std::pair<uint32_t, uint32_t>> where; // heap payload >= n: (triangle, piece index)
    auto payload_ref = [&](uint32_t k) -> HostRef& { return k < n ? refs[k] : pieces[where[k - n].first][where[k - n].second]; };
    while (!heap.empty() && budget > 0) {
        std::pop_heap(heap.begin(), heap.end());
        const uint32_t k = heap.back().second;
        heap.pop_back();
        HostRef cur = payload_ref(k);
        int axis = 0;
        for (int c = 1; c < 3; ++c)
            if (cur.hi[c] - cur.lo[c] > cur.hi[axis] - cur.lo[axis])
                axis = c;
        const float mid = 0.5f * (cur.lo[axis] + cur.hi[axis]);
        if (!(mid > cur.lo[axis] && mid < cur.hi[axis]))
            continue; // cannot be halved any further
        HostRef a = cur, b = cur;
        float alo[3], ahi[3], blo[3], bhi[3];
        for (int c = 0; c < 3; ++c)
            alo[c] = blo[c] = cur.lo[c], ahi[c] = bhi[c] = cur.hi[c];
        ahi[axis] = mid;
        blo[axis] = mid;
        const float* t = &tris12[12 * (size_t)cur.tri];
        const bool ha = clipped_bounds(t, alo, ahi, a.lo, a.hi), hb = clipped_bounds(t, blo, bhi, b.lo, b.hi);
        if (!ha || !hb)
            continue; // (the triangle only touches one half: nothing to gain)
        a.clipped = b.clipped = true;
        std::vector<HostRef>& pc = pieces[cur.tri];
        uint32_t ia, ib;
        if (k < n) { // first split of this triangle: its two pieces replace the whole
            pc.push_back(a);
            pc.push_back(b);
            ia = 0, ib = 1;
        } else {
            ia = where[k - n].second;
            pc[ia] = a;
            pc.push_back(b);
            ib = (uint32_t)pc.size() - 1u;
        }
        --budget;
        for (uint32_t which = 0; which < 2; ++which) {
            const HostRef& r = which == 0 ? a : b;
            const double area = half_area(r.lo, r.hi);
            if (area > limit) {
                uint32_t id;
                if (which == 0 && k >= n) {
                    id = k; // the slot of the piece that was split is reused by its first half
                } else {
                    where.emplace_back(cur.tri, which == 0 ? ia : ib);
                    id = n + (uint32_t)where.size() - 1u;
                }
                heap.emplace_back(area, id);
                std::push_heap(heap.begin(), heap.end());
            }
        }
    }
    std::vector<HostRef> out;
    out.reserve(n + (size_t)(kSplitBudget * n) + 8);
    for (uint32_t i = 0; i < n; ++i) {
        if (pieces[i].empty())
            out.push_back(refs[i]);
        else
            for (const HostRef& r : pieces[i])
                out.push_back(r);
    }
    refs.swap(out);
}

int neb_gi_build_bvh(neb_ctx* ctx, neb_stream stream_)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    if (!g)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_build_bvh: no scene (call neb_gi_set_scene first)");
    hipStream_t stream = (hipStream_t)stream_;
    GI_GUARD(ctx);
    const auto t_build0 = std::chrono::steady_clock::now(); // (the build ends with a stream synchronisation: wall time = device time + launches)
    if (g->n_tris == 0) { // empty scene: every ray misses
        g->built = true;
        g->n_nodes = 0;
        g->bvh_depth = 0;
        g->view.root = -1;
        return NEB_OK;
    }
    // Everything is built into locals and committed to g->view only at the very end, on success: a failed (re)build
    // leaves the scene exactly as it was -- still unbuilt, or still holding the previous, valid tree.
    // The whole build runs on the device: Morton keys -> radix sort -> binned-SAH splits level by level -> BVH4 collapse +
    // leaf-order triangle permutation -> shading records.  The host only reads back one counter per pass (segments / nodes of
    // the next level).
    // The primitives of the build are REFERENCES: one per triangle, except that a triangle far larger than the rest is referenced
    // by several pieces with clipped boxes (split_references above).  A reference carries a copy of its triangle.
    std::vector<HostRef> refs;
    split_references(g->h_tris, refs);
    const uint32_t n = (uint32_t)refs.size();
    const bool have_pieces = n != g->n_tris;
    std::vector<float> h_refs;   // 12 floats per reference (the triangle), only when some triangle was split
    std::vector<float> h_boxes;  // 2 x float4 per reference: {lo, flag}, {hi, 0}
    if (have_pieces) {
        h_refs.resize((size_t)n * 12);
        h_boxes.resize((size_t)n * 8);
        for (uint32_t i = 0; i < n; ++i) {
            memcpy(&h_refs[12 * (size_t)i], &g->h_tris[12 * (size_t)refs[i].tri], 48);
            float* b = &h_boxes[8 * (size_t)i];
            b[0] = refs[i].lo[0], b[1] = refs[i].lo[1], b[2] = refs[i].lo[2], b[3] = refs[i].clipped ? 1.0f : 0.0f;
            b[4] = refs[i].hi[0], b[5] = refs[i].hi[1], b[6] = refs[i].hi[2], b[7] = 0.0f;
        }
    }
    std::vector<void*> temps, fresh; // freed at the end / device arrays that outlive the build (freed again on failure)
    bool oom = false;
    auto dalloc = [&](size_t bytes, bool keep) -> void* {
        void* p = nullptr;
        if (oom || hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            oom = true;
            return nullptr;
        }
        (keep ? fresh : temps).push_back(p);
        return p;
    };
    auto release = [&](std::vector<void*>& v) {
        for (void* p : v)
            (void)hipFree(p);
        v.clear();
    };
    const size_t n2 = 2 * (size_t)n;
    float* d_tris12 = (float*)dalloc((size_t)n * 48, false);
    float4* d_sorted = (float4*)dalloc((size_t)n * 48, false); // Morton order
    float4* d_boxes = have_pieces ? (float4*)dalloc((size_t)n * 32, false) : nullptr;        // per reference, input order
    float4* d_boxes_sorted = have_pieces ? (float4*)dalloc((size_t)n * 32, false) : nullptr; // ... Morton order
    float4* d_final = (float4*)dalloc((size_t)n * 48, true);   // leaf order of the final tree
    uint64_t* d_keys = (uint64_t*)dalloc((size_t)n * 8, false);
    uint64_t* d_keys2 = (uint64_t*)dalloc((size_t)n * 8, false);
    BinaryNodes N;
    N.lo = (float4*)dalloc(n2 * 16, false);
    N.hi = (float4*)dalloc(n2 * 16, false);
    N.size = (uint32_t*)dalloc(n2 * 4, false);
    uint32_t* d_idx[2] = {(uint32_t*)dalloc((size_t)n * 4, false), (uint32_t*)dalloc((size_t)n * 4, false)};
    SahSegment* d_segs[2] = {(SahSegment*)dalloc((size_t)n * sizeof(SahSegment), false), (SahSegment*)dalloc((size_t)n * sizeof(SahSegment), false)};
    SahSplit* d_splits = (SahSplit*)dalloc((size_t)n * sizeof(SahSplit), false);
    // the first levels, spread over the chip (sah_big_*): per-segment words for up to kSahBigMaxSegs segments, one word per slice
    const uint32_t max_slices = n / kSahSlice + kSahBigMaxSegs + 1u;
    SahBig big;
    big.slice_begin = (uint32_t*)dalloc((kSahBigMaxSegs + 1u) * 4, false);
    big.cb = (uint32_t*)dalloc(kSahBigMaxSegs * 6 * 4, false);
    big.bins = (uint32_t*)dalloc((size_t)kSahBigMaxSegs * 4 * kSahBins * 7 * 4, false);
    big.slice_left = (uint32_t*)dalloc((size_t)max_slices * 4, false);
    SahBigPick* d_picks = (SahBigPick*)dalloc(kSahBigMaxSegs * sizeof(SahBigPick), false);
    unsigned long long* d_flags = (unsigned long long*)dalloc((size_t)n * 8, false);
    unsigned long long* d_scan = (unsigned long long*)dalloc((size_t)n * 8, false);
    uint32_t* d_state = (uint32_t*)dalloc(8 * 4, false); // two {segments of the next level, next node id} pairs + the collapse's level counter
    Bvh4Node* d_wide_tmp = (Bvh4Node*)dalloc((size_t)n * sizeof(Bvh4Node), false); // (at most n - 1 wide nodes)
    uint32_t* d_qnode = (uint32_t*)dalloc((size_t)n * 4, false);
    uint32_t* d_qoff = (uint32_t*)dalloc((size_t)n * 4, false);
    int4* d_opened = (int4*)dalloc((size_t)n * 16, false);
    uint32_t* d_icount = (uint32_t*)dalloc((size_t)n * 4, false);
    uint32_t* d_iscan = (uint32_t*)dalloc((size_t)n * 4, false);
    float4* d_shade = (float4*)dalloc((size_t)n * 128, true);
#ifndef NEB_COLLAPSE_DP
#define NEB_COLLAPSE_DP 1 // 1: cost-optimal BVH4 collapse (collapse_plan_kernel); 0: the greedy largest-box rule (A/B arm)
#endif
    float4* d_plan = NEB_COLLAPSE_DP ? (float4*)dalloc(n2 * 16, false) : nullptr;
    size_t cub_bytes = 0, cub_b2 = 0, cub_b3 = 0;
    (void)rocprim::radix_sort_keys(nullptr, cub_bytes, d_keys, d_keys2, (size_t)n, 0u, 64u, stream);
    (void)rocprim::exclusive_scan(nullptr, cub_b2, d_flags, d_scan, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), stream);
    (void)rocprim::exclusive_scan(nullptr, cub_b3, d_icount, d_iscan, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
    cub_bytes = std::max(cub_bytes, std::max(cub_b2, cub_b3));
    void* d_cub = dalloc(cub_bytes, false);
    if (oom) {
        release(temps);
        release(fresh);
        return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_build_bvh: out of device memory");
    }
    auto bail = [&](int code, const char* what, hipError_t e = hipSuccess) {
        (void)hipStreamSynchronize(stream);
        release(temps);
        release(fresh);
        return gi_fail(ctx, code, what, e);
    };
#define BUILD_HIP(call)                                   \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess)                             \
            return bail(NEB_ERR_HIP, #call, e_);          \
    } while (0)
    // ---- Morton order ----
    BUILD_HIP(hipMemcpyAsync(d_tris12, have_pieces ? h_refs.data() : g->h_tris.data(), (size_t)n * 48, hipMemcpyHostToDevice, stream));
    if (have_pieces)
        BUILD_HIP(hipMemcpyAsync(d_boxes, h_boxes.data(), (size_t)n * 32, hipMemcpyHostToDevice, stream));
    const float3 smin = make_float3(g->scene_min[0], g->scene_min[1], g->scene_min[2]);
    // (per-axis normalisation: cubic cells -- all axes scaled by the longest extent -- traversed 12 % slower on the bench scene)
    const float3 sinv = make_float3(1.0f / fmaxf(g->scene_max[0] - g->scene_min[0], 1e-20f), 1.0f / fmaxf(g->scene_max[1] - g->scene_min[1], 1e-20f),
                                    1.0f / fmaxf(g->scene_max[2] - g->scene_min[2], 1e-20f));
    const uint32_t nb = (n + 255) / 256;
    uint32_t index_bits = 1;
    while (index_bits < 32 && (1ull << index_bits) < (unsigned long long)n)
        ++index_bits;
#ifdef NEB_MORTON_AXIS_BITS
    const uint32_t axis_bits = NEB_MORTON_AXIS_BITS;
#else
    const uint32_t axis_bits = (64 - index_bits) / 3 < 21 ? (64 - index_bits) / 3 : 21;
#endif
    hipLaunchKernelGGL(lbvh_morton_kernel, dim3(nb), dim3(256), 0, stream, d_tris12, (const float4*)d_boxes, n, smin, sinv, axis_bits, index_bits, d_keys);
    BUILD_HIP(hipGetLastError());
    BUILD_HIP(rocprim::radix_sort_keys(d_cub, cub_bytes, d_keys, d_keys2, (size_t)n, 0u, 64u, stream));
    hipLaunchKernelGGL(lbvh_gather_kernel, dim3(nb), dim3(256), 0, stream, d_tris12, (const float4*)d_boxes, d_keys2, n, (1ull << index_bits) - 1ull, d_sorted,
                       d_boxes_sorted);
    BUILD_HIP(hipGetLastError());
    // ---- binary topology: binned SAH, one level per pass ----
    hipLaunchKernelGGL(sah_init_kernel, dim3(nb), dim3(256), 0, stream, (const float4*)d_sorted, (const float4*)d_boxes_sorted, n, N, d_idx[0]);
    BUILD_HIP(hipGetLastError());
    uint32_t passes = 0;
    const uint32_t root_node = n; // (only meaningful when n > kMaxLeafTris)
    std::vector<uint32_t> pass_first{n, n + 1u}; // node ids [pass_first[p], pass_first[p + 1]) were created together: the root, then one SAH pass each
    if (n > (uint32_t)kMaxLeafTris) {
        const SahSegment root_seg{0u, n, root_node};
        float minus_one;
        const int m1 = -1;
        memcpy(&minus_one, &m1, 4);
        const float4 root_lo = make_float4(g->scene_min[0], g->scene_min[1], g->scene_min[2], minus_one);
        const float4 root_hi = make_float4(g->scene_max[0], g->scene_max[1], g->scene_max[2], minus_one);
        const uint32_t h_state[2] = {1u, n + 1u};
        BUILD_HIP(hipMemcpyAsync(d_segs[0], &root_seg, sizeof(root_seg), hipMemcpyHostToDevice, stream));
        BUILD_HIP(hipMemcpyAsync(N.lo + root_node, &root_lo, 16, hipMemcpyHostToDevice, stream));
        BUILD_HIP(hipMemcpyAsync(N.hi + root_node, &root_hi, 16, hipMemcpyHostToDevice, stream));
        BUILD_HIP(hipMemcpyAsync(N.size + root_node, &n, 4, hipMemcpyHostToDevice, stream));
        BUILD_HIP(hipMemcpyAsync(d_state, h_state, sizeof(h_state), hipMemcpyHostToDevice, stream));
        BUILD_HIP(hipStreamSynchronize(stream)); // (the small host buffers above live on this stack frame)
        uint32_t n_segs = 1;
        while (n_segs > 0) {
            if (passes > 4096u)
                return bail(NEB_ERR_HIP, "neb_gi_build_bvh: the SAH build does not terminate (internal error)");
            const uint32_t* rs = d_state + 2 * (passes & 1u);
            uint32_t* ws = d_state + 2 * ((passes + 1u) & 1u);
            const SahSegment* segs = d_segs[passes & 1u];
            const uint32_t* idx_in = d_idx[passes & 1u];
            uint32_t* idx_out = d_idx[(passes + 1u) & 1u];
            if (NEB_SAH_BIG && (size_t)n_segs * 2048 <= (size_t)n && n_segs <= kSahBigMaxSegs) {
                // runs of 2048 primitives and more on average: slices of kSahSlice primitives, one workgroup each (a run shorter than a
                // slice is one slice); at most n / kSahSlice + n_segs slices exist, blocks beyond the real count leave at once
                const dim3 sg_grid(n / kSahSlice + n_segs);
                hipLaunchKernelGGL(sah_big_index_kernel, dim3(1), dim3(1024), 0, stream, segs, n_segs, big);
                hipLaunchKernelGGL(sah_big_bounds_kernel, sg_grid, dim3(256), 0, stream, N, segs, n_segs, idx_in, big);
                hipLaunchKernelGGL(sah_big_bin_kernel, sg_grid, dim3(256), 0, stream, N, segs, n_segs, idx_in, big);
                hipLaunchKernelGGL(sah_big_pick_kernel, dim3(n_segs), dim3(128), 0, stream, segs, n_segs, big, d_splits, d_picks);
                hipLaunchKernelGGL(sah_big_count_kernel, sg_grid, dim3(256), 0, stream, N, segs, n_segs, idx_in, big, (const SahBigPick*)d_picks);
                hipLaunchKernelGGL(sah_big_scan_kernel, dim3((n_segs + 63) / 64), dim3(64), 0, stream, n_segs, big);
                hipLaunchKernelGGL(sah_big_scatter_kernel, sg_grid, dim3(256), 0, stream, N, segs, n_segs, idx_in, idx_out, big, (const SahBigPick*)d_picks);
            } else if ((size_t)n_segs * 2048 <= (size_t)n) // (A/B arm NEB_SAH_BIG=0, or more long runs than the slice index holds)
                hipLaunchKernelGGL(sah_split_kernel<1024>, dim3(n_segs), dim3(1024), 0, stream, N, segs, idx_in, idx_out, d_splits);
            else
                hipLaunchKernelGGL(sah_split_kernel<256>, dim3(n_segs), dim3(256), 0, stream, N, segs, idx_in, idx_out, d_splits);
            const dim3 grid((n_segs + 255) / 256);
            hipLaunchKernelGGL(sah_count_kernel, grid, dim3(256), 0, stream, segs, (const SahSplit*)d_splits, n_segs, d_flags);
            BUILD_HIP(hipGetLastError());
            BUILD_HIP(rocprim::exclusive_scan(d_cub, cub_bytes, d_flags, d_scan, 0ull, (size_t)n_segs, rocprim::plus<unsigned long long>(), stream));
            hipLaunchKernelGGL(sah_emit_kernel, grid, dim3(256), 0, stream, N, segs, (const SahSplit*)d_splits, n_segs, (const unsigned long long*)d_flags,
                               (const unsigned long long*)d_scan, (const uint32_t*)idx_out, rs, ws, d_segs[(passes + 1u) & 1u]);
            BUILD_HIP(hipGetLastError());
            uint32_t next_state[2] = {0u, 0u}; // {segments of the next level, next free node id}
            BUILD_HIP(hipMemcpyAsync(next_state, ws, 8, hipMemcpyDeviceToHost, stream));
            BUILD_HIP(hipStreamSynchronize(stream));
            const uint32_t next = next_state[0];
            if (next > n || next_state[1] > 2u * n || next_state[1] < pass_first.back())
                return bail(NEB_ERR_HIP, "neb_gi_build_bvh: segment bound exceeded (internal error)");
            pass_first.push_back(next_state[1]);
            n_segs = next;
            ++passes;
        }
    }
    g->build_passes = passes;
    // ---- collapse to BVH4, triangles into leaf order ----
    int root_code = 0, max_depth = 0;
    uint32_t n_wide = 0;
    if (n <= (uint32_t)kMaxLeafTris) { // the whole scene is one leaf
        BUILD_HIP(hipMemcpyAsync(d_final, d_sorted, (size_t)n * 48, hipMemcpyDeviceToDevice, stream));
        root_code = ~(int)((0u << 2) | (n - 1u));
    } else {
        CollapseArgs a;
        a.N = N;
        a.n_tris = n;
        a.tris_in = d_sorted;
        a.tris_out = d_final;
        a.wide = d_wide_tmp;
        a.q_node = d_qnode;
        a.q_off = d_qoff;
        a.opened = d_opened;
        a.inner_count = d_icount;
        a.inner_scan = d_iscan;
        a.level_state = d_state + 4;
        a.plan = d_plan;
        if (d_plan) { // cost of every subtree as 1, 2 or 3 entries of a wide node: triangles first, then bottom-up, one SAH pass at a time
            hipLaunchKernelGGL(collapse_plan_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, N, 0u, n, d_plan);
            for (size_t lv = pass_first.size() - 1; lv-- > 0;) {
                const uint32_t first = pass_first[lv], count = pass_first[lv + 1] - first;
                if (count)
                    hipLaunchKernelGGL(collapse_plan_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, N, first, count, d_plan);
            }
            BUILD_HIP(hipGetLastError());
        }
        BUILD_HIP(hipMemcpyAsync(d_qnode, &root_node, 4, hipMemcpyHostToDevice, stream));
        BUILD_HIP(hipMemsetAsync(d_qoff, 0, 4, stream));
        BUILD_HIP(hipStreamSynchronize(stream));
        uint32_t level_start = 0, level_count = 1;
        while (level_count > 0) {
            ++max_depth;
            if ((size_t)level_start + level_count > (size_t)n)
                return bail(NEB_ERR_HIP, "neb_gi_build_bvh: wide-node bound exceeded (internal error)");
            a.level_start = level_start;
            a.level_count = level_count;
            const dim3 grid((level_count + 127) / 128);
            hipLaunchKernelGGL(collapse_open_kernel, grid, dim3(128), 0, stream, a);
            BUILD_HIP(hipGetLastError());
            BUILD_HIP(rocprim::exclusive_scan(d_cub, cub_bytes, d_icount, d_iscan, 0u, (size_t)level_count, rocprim::plus<uint32_t>(), stream));
            hipLaunchKernelGGL(collapse_emit_kernel, grid, dim3(128), 0, stream, a);
            BUILD_HIP(hipGetLastError());
            uint32_t next = 0;
            BUILD_HIP(hipMemcpyAsync(&next, d_state + 4, 4, hipMemcpyDeviceToHost, stream));
            BUILD_HIP(hipStreamSynchronize(stream));
            level_start += level_count;
            level_count = next;
        }
        n_wide = level_start;
    }
    Bvh4Node* d_wide = nullptr;
    Bvh4NodeQ* d_wide_q = nullptr;
    if (n_wide) {
        d_wide = (Bvh4Node*)dalloc((size_t)n_wide * sizeof(Bvh4Node), true);
        if (!d_wide)
            return bail(NEB_ERR_HIP, "neb_gi_build_bvh: out of device memory");
        BUILD_HIP(hipMemcpyAsync(d_wide, d_wide_tmp, (size_t)n_wide * sizeof(Bvh4Node), hipMemcpyDeviceToDevice, stream));
        d_wide_q = (Bvh4NodeQ*)dalloc((size_t)n_wide * sizeof(Bvh4NodeQ), true);
        if (!d_wide_q)
            return bail(NEB_ERR_HIP, "neb_gi_build_bvh: out of device memory");
        hipLaunchKernelGGL(quantise_nodes_kernel, dim3((n_wide + 127) / 128), dim3(128), 0, stream, d_wide, n_wide, d_wide_q);
        BUILD_HIP(hipGetLastError());
    }
    {
        SceneView sv = g->view;
        sv.tris = d_final;
        hipLaunchKernelGGL(pack_shade_records_kernel, dim3(nb), dim3(256), 0, stream, sv, n, d_shade);
        BUILD_HIP(hipGetLastError());
    }
    BUILD_HIP(hipStreamSynchronize(stream)); // the temporaries are freed below; the build is a one-time setup step
#undef BUILD_HIP
    release(temps);
    // The traverser keeps at most kLdsStack + kSpillStack pending nodes per ray; a closest-hit descent stacks up to 3
    // siblings per level, so a tree deeper than that bound could lose hits.  Refuse it here instead.
    if ((uint32_t)max_depth > g->max_bvh_depth) {
        release(fresh);
        char msg[200];
        snprintf(msg, sizeof(msg), "neb_gi_build_bvh: BVH4 depth %d exceeds the limit %u (the traversal stack holds %d entries, 3 per level)",
                 max_depth, g->max_bvh_depth, kLdsStack + kSpillStack);
        return gi_fail(ctx, NEB_ERR_OUT_OF_RANGE, msg);
    }
    // ---- commit: release the previous build's arrays (a rebuild), adopt the new ones ----
    const void* old[] = {g->view.tris, g->view.shade, g->view.nodes, g->view.qnodes};
    if (g->built)
        (void)hipDeviceSynchronize(); // no launch may still be walking the tree that is about to be freed
    for (const void* o : old) {
        if (!o)
            continue;
        for (size_t k = 0; k < g->allocs.size(); ++k)
            if (g->allocs[k] == o) {
                g->allocs.erase(g->allocs.begin() + (long)k);
                (void)hipFree(const_cast<void*>(o));
                break;
            }
    }
    g->allocs.insert(g->allocs.end(), fresh.begin(), fresh.end());
    g->view.n_tris = n; // references: the length of the triangle and shading-record arrays (g->n_tris stays the scene's triangle count)
    g->view.tris = d_final;
    g->view.shade = d_shade;
    g->view.nodes = d_wide;
    g->view.qnodes = d_wide_q;
    g->view.n_qnodes = n_wide;
    g->view.root = root_code;
    g->n_nodes = n_wide;
    g->bvh_depth = (uint32_t)max_depth;
    g->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
    g->sun_table_state = 0; // fresh shading records carry no sun-visibility flags yet
    g->built = true; // (h_tris stays: the scene can be rebuilt)
    return NEB_OK;
}

int neb_gi_scene_bytes(const neb_ctx* ctx, uint64_t out[3])
{
    if (!ctx || !ctx->gi || !out)
        return NEB_ERR_STATE;
    const neb::GiState* g = ctx->gi;
    out[0] = g->texture_table_bytes;
    out[1] = g->built ? (uint64_t)g->view.n_tris * (48 + 128) : 0;
    out[2] = (uint64_t)g->n_nodes * (sizeof(neb::Bvh4Node) + sizeof(neb::Bvh4NodeQ));
    return NEB_OK;
}

int neb_gi_bvh_depth(const neb_ctx* ctx, uint32_t* depth)
{
    if (!ctx || !ctx->gi || !depth)
        return NEB_ERR_STATE;
    *depth = ctx->gi->bvh_depth;
    return NEB_OK;
}

int neb_gi_build_passes(const neb_ctx* ctx, uint32_t* passes)
{
    if (!ctx || !ctx->gi || !passes)
        return NEB_ERR_STATE;
    *passes = ctx->gi->build_passes;
    return NEB_OK;
}

int neb_gi_build_ms(const neb_ctx* ctx, float* ms)
{
    if (!ctx || !ctx->gi || !ms)
        return NEB_ERR_STATE;
    *ms = ctx->gi->build_ms;
    return NEB_OK;
}

int neb_gi_scene_info(const neb_ctx* ctx, uint32_t* n_triangles, uint32_t* n_nodes)
{
    if (!ctx || !ctx->gi)
        return NEB_ERR_STATE;
    if (n_triangles)
        *n_triangles = ctx->gi->n_tris;
    if (n_nodes)
        *n_nodes = ctx->gi->n_nodes;
    return NEB_OK;
}

} // extern "C"

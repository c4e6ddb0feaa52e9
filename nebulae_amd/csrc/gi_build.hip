// gi_build.hip -- scene upload and acceleration structure of the GI path (one-time setup).
//
// Reference: src/nri/GIProcessedScene.cpp:16-137 (scene tables); RTAccelerationStructureBuilder.cpp:14-130 (driver
// BVH -> replaced by a Karras LBVH built on the device, its upper levels re-linked by SAH and collapsed to a 4-wide
// tree in a host pass).
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "gi_device.h"

namespace neb {

// neb_resize: the per-pixel GI buffers (records, debug hits, per-workgroup counters) belong to the old resolution
void gi_on_resize(GiState* g)
{
    if (!g)
        return;
    void* stale[] = {g->d_records, g->d_hits, g->d_block_counts, g->d_sort, g->d_sort_temp};
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
    g->d_records = nullptr;
    g->d_hits = nullptr;
    g->d_block_counts = nullptr;
    g->n_block_counts = 0;
    g->d_sort = nullptr;
    g->d_sort_temp = nullptr;
}

void gi_destroy(GiState* g)
{
    if (!g)
        return;
    for (void* p : g->allocs)
        (void)hipFree(p);
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

// ------------------------------------------------------------------------------------------------
// LBVH build (Karras 2012): Morton keys -> radix sort -> hierarchy -> bottom-up refit
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
__global__ void lbvh_morton_kernel(const float* __restrict__ tris12, uint32_t n, float3 smin, float3 sinv, uint32_t axis_bits,
                                   uint32_t index_bits, uint64_t* keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float* t = tris12 + 12 * (size_t)i;
    const float3 v0 = f3(t[0], t[1], t[2]), v1 = f3(t[0] + t[3], t[1] + t[4], t[2] + t[5]), v2 = f3(t[0] + t[6], t[1] + t[7], t[2] + t[8]);
    const float cx = (fminf(v0.x, fminf(v1.x, v2.x)) + fmaxf(v0.x, fmaxf(v1.x, v2.x))) * 0.5f;
    const float cy = (fminf(v0.y, fminf(v1.y, v2.y)) + fmaxf(v0.y, fmaxf(v1.y, v2.y))) * 0.5f;
    const float cz = (fminf(v0.z, fminf(v1.z, v2.z)) + fmaxf(v0.z, fmaxf(v1.z, v2.z))) * 0.5f;
    const float cells = (float)(1u << axis_bits), top = cells - 1.0f;
    const uint64_t qx = (uint64_t)fminf(fmaxf((cx - smin.x) * sinv.x * cells, 0.0f), top);
    const uint64_t qy = (uint64_t)fminf(fmaxf((cy - smin.y) * sinv.y * cells, 0.0f), top);
    const uint64_t qz = (uint64_t)fminf(fmaxf((cz - smin.z) * sinv.z * cells, 0.0f), top);
    const uint64_t m = (expand_bits21(qx) << 2) | (expand_bits21(qy) << 1) | expand_bits21(qz);
    keys[i] = (m << index_bits) | i;
}

__global__ void lbvh_gather_kernel(const float* __restrict__ tris12, const uint64_t* __restrict__ keys, uint32_t n, uint64_t index_mask,
                                   float4* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float* t = tris12 + 12 * (size_t)(keys[i] & index_mask);
    out[3 * i] = make_float4(t[0], t[1], t[2], t[3]);
    out[3 * i + 1] = make_float4(t[4], t[5], t[6], t[7]);
    out[3 * i + 2] = make_float4(t[8], t[9], t[10], t[11]);
}

__device__ __forceinline__ int lbvh_delta(const uint64_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n)
        return -1;
    return __clzll(keys[i] ^ keys[j]);
}

// one thread per inner node i in [0, n-2]
__global__ void lbvh_hierarchy_kernel(const uint64_t* __restrict__ keys, int n, int2* children, int* parent_inner, int* parent_leaf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1)
        return;
    const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin)
        lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin)
            l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode)
            s += t;
        if (t <= 1)
            break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int c0 = (lo == gamma) ? ~gamma : gamma;             // leaf codes are ~index
    const int c1 = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    children[i] = make_int2(c0, c1);
    if (c0 < 0)
        parent_leaf[gamma] = i;
    else
        parent_inner[gamma] = i;
    if (c1 < 0)
        parent_leaf[gamma + 1] = i;
    else
        parent_inner[gamma + 1] = i;
}

// one thread per leaf: walk up; the second arrival at a node owns it (boxes of both children are then visible)
__global__ void lbvh_refit_kernel(const float4* __restrict__ tris, int n, const int2* __restrict__ children, const int* __restrict__ parent_inner,
                                  const int* __restrict__ parent_leaf, float* node_min, float* node_max, uint32_t* visit, BvhNode* nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    int node = parent_leaf[i];
    while (true) {
        __threadfence();
        if (atomicAdd(&visit[node], 1u) == 0u)
            return; // first arrival: the sibling subtree is not finished yet
        __threadfence();
        const int2 ch = children[node];
        float bmin[2][3], bmax[2][3];
        const int cc[2] = {ch.x, ch.y};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (cc[k] < 0) {
                const uint32_t ti = (uint32_t)~cc[k];
                const float4 a = tris[3 * ti], b = tris[3 * ti + 1], c = tris[3 * ti + 2];
                const float3 v0 = f3(a.x, a.y, a.z), v1 = f3(a.x + a.w, a.y + b.x, a.z + b.y), v2 = f3(a.x + b.z, a.y + b.w, a.z + c.x);
                bmin[k][0] = fminf(v0.x, fminf(v1.x, v2.x));
                bmin[k][1] = fminf(v0.y, fminf(v1.y, v2.y));
                bmin[k][2] = fminf(v0.z, fminf(v1.z, v2.z));
                bmax[k][0] = fmaxf(v0.x, fmaxf(v1.x, v2.x));
                bmax[k][1] = fmaxf(v0.y, fmaxf(v1.y, v2.y));
                bmax[k][2] = fmaxf(v0.z, fmaxf(v1.z, v2.z));
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    bmin[k][q] = __hip_atomic_load(&node_min[3 * cc[k] + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bmax[k][q] = __hip_atomic_load(&node_max[3 * cc[k] + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        BvhNode out;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            out.c0min[q] = bmin[0][q];
            out.c0max[q] = bmax[0][q];
            out.c1min[q] = bmin[1][q];
            out.c1max[q] = bmax[1][q];
            __hip_atomic_store(&node_min[3 * node + q], fminf(bmin[0][q], bmin[1][q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&node_max[3 * node + q], fmaxf(bmax[0][q], bmax[1][q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out.c0 = ch.x;
        out.c1 = ch.y;
        out.pad0 = out.pad1 = 0;
        nodes[node] = out;
        if (node == 0)
            return; // root
        node = parent_inner[node];
    }
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
    // (a footprint entry holds the 4 texels of a bilinear fetch, wrap applied: 16 B per texel position and map.  Sponza's
    // 69 maps of 1024^2 make 1.2 GB of per-map tables / 1.6 GB of material bundles -- nothing to assemble on the host.)
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
    // materials whose three maps share one size keep their footprints interleaved (DevMat::bundle, 64 B per texel position);
    // every other map a material uses gets a standalone table
    constexpr size_t kBundleBudget = (size_t)4 << 30; // bytes; beyond it the remaining materials sample their maps separately
    size_t bundle_entries = 0, table_entries = 0;     // in 64-byte / 16-byte units
    std::vector<char> standalone(n_texs, 0);
    for (uint32_t i = 0; i < n_mats; ++i) {
        DevMat& d = dmats[i];
        d.bundle = d.bundle_w = d.bundle_h = d.pad = 0;
        bool bundled = false;
        if (d.tex[0] >= 0 && d.tex[1] >= 0 && d.tex[2] >= 0) {
            const DevTex &ta = dtexs[d.tex[0]], &tn = dtexs[d.tex[1]], &tr = dtexs[d.tex[2]];
            const size_t n_pos = (size_t)ta.w * ta.h;
            if (ta.w == tn.w && ta.w == tr.w && ta.h == tn.h && ta.h == tr.h && (bundle_entries + n_pos) * 64 <= kBundleBudget &&
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
    g->texture_table_bytes = table_entries * 16 + bundle_entries * 64;
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
        talloc((void**)&d_bundles, bundle_entries * 64, true);
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
            for (int k = 0; k < 3 && te == hipSuccess; ++k) { // slot k of every 64-byte entry
                hipLaunchKernelGGL(texture_footprints_kernel, dim3((unsigned)((n_pos + 255) / 256)), dim3(256), 0, nullptr, d_raw + raw_off[d.tex[k]],
                                   d.bundle_w, d.bundle_h, d_bundles + 4 * (size_t)d.bundle + k, 4u);
                te = hipGetLastError();
            }
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
    if ((e = hipMalloc(&ctr, 8 * sizeof(unsigned long long))) != hipSuccess || (e = hipMemset(ctr, 0, 8 * sizeof(unsigned long long))) != hipSuccess) {
        gi_destroy(g);
        return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_set_scene: counter", e);
    }
    g->allocs.push_back(ctr);
    g->d_ray_counter = (unsigned long long*)ctr;
    g->view.n_tris = g->n_tris;
    ctx->gi = g;
    return NEB_OK;
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
    const uint32_t n = g->n_tris;
    if (n == 0) { // empty scene: every ray misses
        g->built = true;
        g->n_nodes = 0;
        g->view.root = -1;
        return NEB_OK;
    }
    // Everything is built into locals and committed to g->view only at the very end, on success: a failed (re)build
    // leaves the scene exactly as it was -- still unbuilt, or still holding the previous, valid tree.
    std::vector<void*> fresh; // device arrays of THIS build that outlive it (freed again on failure)
    auto dalloc = [&](size_t bytes, bool keep) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess)
            return nullptr;
        if (keep)
            fresh.push_back(p);
        return p;
    };
    auto drop_fresh = [&]() {
        for (void* p : fresh)
            (void)hipFree(p);
        fresh.clear();
    };
    float* d_tris12 = (float*)dalloc((size_t)n * 48, false);
    float4* d_sorted = (float4*)dalloc((size_t)n * 48, true);
    uint64_t* d_keys = (uint64_t*)dalloc((size_t)n * 8, false);
    uint64_t* d_keys2 = (uint64_t*)dalloc((size_t)n * 8, false);
    const uint32_t n_inner = n > 1 ? n - 1 : 1;
    BvhNode* d_nodes = (BvhNode*)dalloc((size_t)n_inner * sizeof(BvhNode), false);
    int2* d_children = (int2*)dalloc((size_t)n_inner * sizeof(int2), false);
    int* d_parent_inner = (int*)dalloc((size_t)n_inner * 4, false);
    int* d_parent_leaf = (int*)dalloc((size_t)n * 4, false);
    float* d_nmin = (float*)dalloc((size_t)n_inner * 12, false);
    float* d_nmax = (float*)dalloc((size_t)n_inner * 12, false);
    uint32_t* d_visit = (uint32_t*)dalloc((size_t)n_inner * 4, false);
    void* temps[] = {d_tris12, d_keys, d_keys2, d_nodes, d_children, d_parent_inner, d_parent_leaf, d_nmin, d_nmax, d_visit};
    auto free_temps = [&]() {
        for (void* p : temps)
            if (p)
                (void)hipFree(p);
    };
    if (!d_tris12 || !d_sorted || !d_keys || !d_keys2 || !d_nodes || !d_children || !d_parent_inner || !d_parent_leaf || !d_nmin ||
        !d_nmax || !d_visit) {
        free_temps();
        drop_fresh();
        return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_build_bvh: out of device memory");
    }
    hipError_t e = hipMemcpyAsync(d_tris12, g->h_tris.data(), (size_t)n * 48, hipMemcpyHostToDevice, stream);
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
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lbvh_morton_kernel, dim3(nb), dim3(256), 0, stream, d_tris12, n, smin, sinv, axis_bits, index_bits, d_keys);
        e = hipGetLastError();
    }
    size_t temp_bytes = 0;
    void* d_temp = nullptr;
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_keys2, (int)n, 0, 64, stream);
    if (e == hipSuccess)
        e = hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16);
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys2, (int)n, 0, 64, stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lbvh_gather_kernel, dim3(nb), dim3(256), 0, stream, d_tris12, d_keys2, n, (1ull << index_bits) - 1ull, d_sorted);
        e = hipGetLastError();
    }
    if (e == hipSuccess && n > 1) {
        e = hipMemsetAsync(d_visit, 0, (size_t)n_inner * 4, stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(lbvh_hierarchy_kernel, dim3((n - 1 + 255) / 256), dim3(256), 0, stream, d_keys2, (int)n, d_children,
                               d_parent_inner, d_parent_leaf);
            hipLaunchKernelGGL(lbvh_refit_kernel, dim3(nb), dim3(256), 0, stream, d_sorted, (int)n, d_children, d_parent_inner, d_parent_leaf,
                               d_nmin, d_nmax, d_visit, d_nodes);
            e = hipGetLastError();
        }
    }
    float4* d_shade = nullptr;
    if (e == hipSuccess) {
        d_shade = (float4*)dalloc((size_t)n * 128, true);
        if (!d_shade) {
            e = hipErrorOutOfMemory;
        } else {
            SceneView sv = g->view;
            sv.tris = d_sorted;
            hipLaunchKernelGGL(pack_shade_records_kernel, dim3(nb), dim3(256), 0, stream, sv, n, d_shade);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(stream); // the temporaries are freed below; the build is a one-time setup step
    // ---- collapse the binary LBVH into BVH4 nodes with leaves of up to kMaxLeafTris triangles ----
    // (host pass over the device-built hierarchy: topology and boxes are the LBVH's; one-time setup)
    std::vector<Bvh4Node> wide;
    int root_code = ~0; // leaf {first 0, count 1}
    int max_depth = 0;  // inner-node levels of the BVH4
    if (e == hipSuccess && n > 1) {
        std::vector<BvhNode> bin(n - 1);
        e = hipMemcpy(bin.data(), d_nodes, (size_t)(n - 1) * sizeof(BvhNode), hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            // triangle range of every binary inner node (LBVH subtrees cover contiguous sorted ranges)
            std::vector<uint32_t> first(n - 1), count(n - 1);
            {
                std::vector<int> order; // children before parents
                order.reserve(n - 1);
                std::vector<int> stk{0};
                while (!stk.empty()) {
                    const int i = stk.back();
                    stk.pop_back();
                    order.push_back(i);
                    if (bin[i].c0 >= 0)
                        stk.push_back(bin[i].c0);
                    if (bin[i].c1 >= 0)
                        stk.push_back(bin[i].c1);
                }
                for (size_t k = order.size(); k-- > 0;) {
                    const int i = order[k];
                    const uint32_t f0 = bin[i].c0 >= 0 ? first[bin[i].c0] : (uint32_t)~bin[i].c0;
                    const uint32_t n0 = bin[i].c0 >= 0 ? count[bin[i].c0] : 1u;
                    const uint32_t f1 = bin[i].c1 >= 0 ? first[bin[i].c1] : (uint32_t)~bin[i].c1;
                    const uint32_t n1 = bin[i].c1 >= 0 ? count[bin[i].c1] : 1u;
                    first[i] = f0 < f1 ? f0 : f1;
                    count[i] = n0 + n1;
                }
            }
            struct Ref {
                int id;        // binary child code: >= 0 inner, < 0 ~triangle
                float lo[3], hi[3];
            };
            int bin_root = 0;
#if NEB_TOP_SAH
            // ---- HLBVH-style top level: the LBVH subtrees of at most NEB_TOP_SAH triangles stay as built on the device;
            // the levels above them are re-linked here by a sweep-SAH build over those subtrees' boxes.  Morton splits
            // are blind to box overlap and hurt most near the root, where every ray pays for them. ----
            {
                struct Cluster {
                    Ref ref;
                    uint32_t cnt;
                    float c[3];
                };
                std::vector<Cluster> cl;
                {
                    std::vector<Ref> stk;
                    Ref root{0, {0, 0, 0}, {0, 0, 0}};
                    for (int q = 0; q < 3; ++q) {
                        root.lo[q] = fminf(bin[0].c0min[q], bin[0].c1min[q]);
                        root.hi[q] = fmaxf(bin[0].c0max[q], bin[0].c1max[q]);
                    }
                    stk.push_back(root);
                    while (!stk.empty()) {
                        const Ref r = stk.back();
                        stk.pop_back();
                        const uint32_t c = r.id >= 0 ? count[r.id] : 1u;
                        if (r.id < 0 || c <= (uint32_t)NEB_TOP_SAH) {
                            Cluster k{r, c, {0.5f * (r.lo[0] + r.hi[0]), 0.5f * (r.lo[1] + r.hi[1]), 0.5f * (r.lo[2] + r.hi[2])}};
                            cl.push_back(k);
                            continue;
                        }
                        Ref a, b;
                        a.id = bin[r.id].c0;
                        b.id = bin[r.id].c1;
                        memcpy(a.lo, bin[r.id].c0min, 12);
                        memcpy(a.hi, bin[r.id].c0max, 12);
                        memcpy(b.lo, bin[r.id].c1min, 12);
                        memcpy(b.hi, bin[r.id].c1max, 12);
                        stk.push_back(a);
                        stk.push_back(b);
                    }
                }
                if (cl.size() > 1) {
                    auto area = [](const float* lo, const float* hi) {
                        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                        return dx * dy + dy * dz + dz * dx;
                    };
                    std::vector<uint32_t> idx(cl.size());
                    for (size_t k = 0; k < idx.size(); ++k)
                        idx[k] = (uint32_t)k;
                    std::vector<float> suffix_area;
                    std::vector<uint32_t> suffix_cnt;
                    // explicit work stack: {l, r, slot to patch}; a patch slot is (node index << 1 | child) or -1 for the root
                    struct Job {
                        size_t l, r;
                        long patch;
                    };
                    std::vector<Job> jobs{{0, cl.size(), -1}};
                    auto set_child = [&](long patch, const Ref& rf) {
                        if (patch < 0) {
                            bin_root = rf.id;
                            return;
                        }
                        BvhNode& nd = bin[(size_t)(patch >> 1)];
                        if (patch & 1) {
                            nd.c1 = rf.id;
                            memcpy(nd.c1min, rf.lo, 12);
                            memcpy(nd.c1max, rf.hi, 12);
                        } else {
                            nd.c0 = rf.id;
                            memcpy(nd.c0min, rf.lo, 12);
                            memcpy(nd.c0max, rf.hi, 12);
                        }
                    };
                    while (!jobs.empty()) {
                        const Job jb = jobs.back();
                        jobs.pop_back();
                        const size_t m = jb.r - jb.l;
                        if (m == 1) {
                            set_child(jb.patch, cl[idx[jb.l]].ref);
                            continue;
                        }
                        int best_axis = 0;
                        size_t best_k = jb.l + m / 2;
                        float best_cost = INFINITY;
                        for (int ax = 0; ax < 3; ++ax) {
                            std::sort(idx.begin() + (long)jb.l, idx.begin() + (long)jb.r,
                                      [&](uint32_t a, uint32_t b) { return cl[a].c[ax] < cl[b].c[ax] || (cl[a].c[ax] == cl[b].c[ax] && a < b); });
                            suffix_area.assign(m + 1, 0.f);
                            suffix_cnt.assign(m + 1, 0u);
                            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                            for (size_t k = m; k-- > 0;) {
                                const Cluster& c = cl[idx[jb.l + k]];
                                for (int q = 0; q < 3; ++q) {
                                    lo[q] = fminf(lo[q], c.ref.lo[q]);
                                    hi[q] = fmaxf(hi[q], c.ref.hi[q]);
                                }
                                suffix_area[k] = area(lo, hi);
                                suffix_cnt[k] = suffix_cnt[k + 1] + c.cnt;
                            }
                            float plo[3] = {INFINITY, INFINITY, INFINITY}, phi[3] = {-INFINITY, -INFINITY, -INFINITY};
                            uint32_t pc = 0;
                            for (size_t k = 1; k < m; ++k) { // split before element k
                                const Cluster& c = cl[idx[jb.l + k - 1]];
                                for (int q = 0; q < 3; ++q) {
                                    plo[q] = fminf(plo[q], c.ref.lo[q]);
                                    phi[q] = fmaxf(phi[q], c.ref.hi[q]);
                                }
                                pc += c.cnt;
                                const float cost = area(plo, phi) * (float)pc + suffix_area[k] * (float)suffix_cnt[k];
                                if (cost < best_cost) {
                                    best_cost = cost;
                                    best_axis = ax;
                                    best_k = jb.l + k;
                                }
                            }
                        }
                        if (best_axis != 2)
                            std::sort(idx.begin() + (long)jb.l, idx.begin() + (long)jb.r, [&](uint32_t a, uint32_t b) {
                                return cl[a].c[best_axis] < cl[b].c[best_axis] || (cl[a].c[best_axis] == cl[b].c[best_axis] && a < b);
                            });
                        // new inner node over [l, best_k) and [best_k, r)
                        Ref self;
                        self.id = (int)bin.size();
                        for (int q = 0; q < 3; ++q) {
                            self.lo[q] = INFINITY;
                            self.hi[q] = -INFINITY;
                        }
                        uint32_t total = 0;
                        for (size_t k = jb.l; k < jb.r; ++k) {
                            const Cluster& c = cl[idx[k]];
                            for (int q = 0; q < 3; ++q) {
                                self.lo[q] = fminf(self.lo[q], c.ref.lo[q]);
                                self.hi[q] = fmaxf(self.hi[q], c.ref.hi[q]);
                            }
                            total += c.cnt;
                        }
                        bin.emplace_back();
                        count.push_back(total);
                        first.push_back(0); // (never a leaf: it spans more than one cluster)
                        set_child(jb.patch, self);
                        jobs.push_back({jb.l, best_k, ((long)self.id << 1) | 0});
                        jobs.push_back({best_k, jb.r, ((long)self.id << 1) | 1});
                    }
                }
            }
#endif
            auto leaf_code = [&](const Ref& r) -> int {
                const uint32_t f = r.id >= 0 ? first[r.id] : (uint32_t)~r.id;
                const uint32_t c = r.id >= 0 ? count[r.id] : 1u;
                return ~(int)((f << 2) | (c - 1u));
            };
            // (only a device-built LBVH subtree covers a contiguous run of the sorted triangles; the top nodes linked
            // above never do, however few triangles they hold)
            auto is_leaf = [&](const Ref& r) { return r.id < 0 || (r.id < (int)(n - 1) && count[r.id] <= (uint32_t)kMaxLeafTris); };
            auto area = [](const Ref& r) {
                const float dx = r.hi[0] - r.lo[0], dy = r.hi[1] - r.lo[1], dz = r.hi[2] - r.lo[2];
                return dx * dy + dy * dz + dz * dx;
            };
            auto children_of = [&](int i, Ref* out) {
                out[0].id = bin[i].c0;
                out[1].id = bin[i].c1;
                memcpy(out[0].lo, bin[i].c0min, 12);
                memcpy(out[0].hi, bin[i].c0max, 12);
                memcpy(out[1].lo, bin[i].c1min, 12);
                memcpy(out[1].hi, bin[i].c1max, 12);
            };
            if (count[0] <= (uint32_t)kMaxLeafTris) {
                root_code = ~(int)((0u << 2) | (count[0] - 1u));
            } else {
                root_code = 0;
                // work list of (binary node, wide slot index); wide nodes are emitted in DFS order
                struct Work {
                    int bi, wi, depth;
                };
                std::vector<Work> work{{bin_root, 0, 1}};
                wide.emplace_back();
                while (!work.empty()) {
                    const auto [bi, wi, depth] = work.back();
                    work.pop_back();
                    max_depth = std::max(max_depth, depth);
                    Ref c[4];
                    int nc = 2;
                    children_of(bi, c);
                    while (nc < 4) { // open the inner child with the largest surface area
                        int best = -1;
                        float best_area = -1.0f;
                        for (int k = 0; k < nc; ++k)
                            if (!is_leaf(c[k]) && area(c[k]) > best_area) {
                                best_area = area(c[k]);
                                best = k;
                            }
                        if (best < 0)
                            break;
                        Ref two[2];
                        children_of(c[best].id, two);
                        c[best] = two[0];
                        c[nc++] = two[1];
                    }
                    Bvh4Node nd;
                    float lo[3][4], hi[3][4];
                    int ch[4];
                    for (int k = 0; k < 4; ++k) {
                        if (k < nc) {
                            for (int q = 0; q < 3; ++q) {
                                lo[q][k] = c[k].lo[q];
                                hi[q][k] = c[k].hi[q];
                            }
                            if (is_leaf(c[k])) {
                                ch[k] = leaf_code(c[k]);
                            } else {
                                ch[k] = (int)wide.size();
                                wide.emplace_back();
                                work.push_back({c[k].id, ch[k], depth + 1});
                            }
                        } else {
                            for (int q = 0; q < 3; ++q) {
                                lo[q][k] = INFINITY;
                                hi[q][k] = -INFINITY;
                            }
                            ch[k] = ~0;
                        }
                    }
                    nd.lox = make_float4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
                    nd.loy = make_float4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]);
                    nd.loz = make_float4(lo[2][0], lo[2][1], lo[2][2], lo[2][3]);
                    nd.hix = make_float4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]);
                    nd.hiy = make_float4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]);
                    nd.hiz = make_float4(hi[2][0], hi[2][1], hi[2][2], hi[2][3]);
                    nd.child = make_int4(ch[0], ch[1], ch[2], ch[3]);
                    nd.pad = make_int4(0, 0, 0, 0);
                    wide[wi] = nd;
                }
            }
        }
    }
    Bvh4Node* d_wide = nullptr;
    if (e == hipSuccess && !wide.empty()) {
        d_wide = (Bvh4Node*)dalloc(wide.size() * sizeof(Bvh4Node), true);
        e = d_wide ? hipMemcpy(d_wide, wide.data(), wide.size() * sizeof(Bvh4Node), hipMemcpyHostToDevice) : hipErrorOutOfMemory;
    }
    if (d_temp)
        (void)hipFree(d_temp);
    free_temps();
    if (e != hipSuccess) {
        drop_fresh();
        return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_build_bvh", e);
    }
    // The traverser keeps at most kLdsStack + kSpillStack pending nodes per ray; a closest-hit descent stacks up to 3
    // siblings per level, so a tree deeper than that bound could lose hits.  Refuse it here instead.
    if ((uint32_t)max_depth > g->max_bvh_depth) {
        drop_fresh();
        char msg[200];
        snprintf(msg, sizeof(msg), "neb_gi_build_bvh: BVH4 depth %d exceeds the limit %u (the traversal stack holds %d entries, 3 per level)",
                 max_depth, g->max_bvh_depth, kLdsStack + kSpillStack);
        return gi_fail(ctx, NEB_ERR_OUT_OF_RANGE, msg);
    }
    // ---- commit: release the previous build's arrays (a rebuild), adopt the new ones ----
    const void* old[] = {g->view.tris, g->view.shade, g->view.nodes};
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
    g->view.tris = d_sorted;
    g->view.shade = d_shade;
    g->view.nodes = d_wide;
    g->view.root = root_code;
    g->n_nodes = (uint32_t)wide.size();
    g->bvh_depth = (uint32_t)max_depth;
    g->built = true; // (h_tris stays: the scene can be rebuilt)
    return NEB_OK;
}

int neb_gi_scene_bytes(const neb_ctx* ctx, uint64_t out[3])
{
    if (!ctx || !ctx->gi || !out)
        return NEB_ERR_STATE;
    const neb::GiState* g = ctx->gi;
    out[0] = g->texture_table_bytes;
    out[1] = g->built ? (uint64_t)g->n_tris * (48 + 128) : 0;
    out[2] = (uint64_t)g->n_nodes * sizeof(neb::Bvh4Node);
    return NEB_OK;
}

int neb_gi_bvh_depth(const neb_ctx* ctx, uint32_t* depth)
{
    if (!ctx || !ctx->gi || !depth)
        return NEB_ERR_STATE;
    *depth = ctx->gi->bvh_depth;
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

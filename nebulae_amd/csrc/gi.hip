// gi.hip -- one-bounce indirect-diffuse GI on gfx950: scene tables, on-device LBVH, traversal + shading.
//
// Reference behaviour (paths relative to the reference checkout):
//   assets/shaders/pathtracer.hlsl:397-625 (PathtracerRG, query variant, NRC stubbed per
//   rtxgi/Nrc.hlsli:579-621, nrcMaxPathVertices = 2), :299-395 ReconstructSurfaceData, :209-228
//   EvaluateDirectBRDF; brdf.hlsli; rand.hlsli; sun_disk_sampling.hlsli:45-52;
//   src/nri/GIProcessedScene.cpp:16-137 (scene tables); RTAccelerationStructureBuilder.cpp:14-130
//   (driver BVH -> replaced by a Karras LBVH built here, its upper levels re-linked by SAH and collapsed to a
//   4-wide tree); deferred_gbuffers.hlsl:36-104 (G-buffer encodings).
// Not a DXR transliteration: there is no ray-gen/miss/closest-hit pipeline and no driver BVH.  The path runs as
// wavefront stages over per-pixel records: a wave owns an 8x8 pixel tile, each lane generates its ray from the
// G-buffer and walks the BVH4 with a per-lane stack kept in LDS (lane-contiguous, conflict-free); a second kernel
// shades the hits from one 128-byte record per triangle; the sun shadow rays are sorted by origin (raysort.hip) and
// go through the same traverser in any-hit mode.
//
// Floating-point contraction is OFF in this file so that ray setup and the Moeller-Trumbore test
// round exactly like the scalar CPU oracle (hit/miss decisions at triangle edges then agree).
#pragma clang fp contract(off)

#include <hipcub/hipcub.hpp>
#include <algorithm>

#include <cmath>
#include <cstring>
#include <vector>

#include "neb_device.h"
#include "neb_internal.h"

namespace neb {

// ------------------------------------------------------------------------------------------------
// Device-side scene
// ------------------------------------------------------------------------------------------------
struct DevGeom {
    float m[9];          // upper 3x3 of surfaceToWorld (row-vector convention)
    int32_t material;
    uint32_t firstIndex; // into the uint32 index pool
    uint32_t vertexBase; // into the SoA vertex pools
    uint32_t valid;      // all four attribute streams + indices present
    uint32_t pad[3];
};
struct DevMat {
    int32_t tex[3];
    float albedo[3];
    float rough, metal;
    // When the material has all three maps and they share one size, their bilinear footprints are also stored
    // interleaved, one 64-byte entry per texel position {albedo 4 texels, normal 4, roughness/metalness 4, pad}: the
    // three filtered fetches of a hit (same uv) then touch one line instead of three.  bundle_w == 0: not bundled.
    uint32_t bundle;   // first entry, in 64-byte units, into SceneView::bundles
    uint32_t bundle_w, bundle_h;
    uint32_t pad;
};
struct DevTex {
    uint32_t offset; // in footprint entries (16 B), into the texel pool
    uint32_t w, h, pad;
};
// 64-byte BVH2 node: both children's boxes live in the parent, so one node fetch decides both.
struct BvhNode {
    float c0min[3];
    int32_t c0; // >= 0: inner node, < 0: leaf, triangle = ~c
    float c0max[3];
    int32_t c1;
    float c1min[3];
    uint32_t pad0;
    float c1max[3];
    uint32_t pad1;
};

// 128-byte BVH4 node produced by collapsing the LBVH (SoA per axis: one float4 per bound and axis).
// child >= 0: inner node index; child < 0: leaf, code = ~child = (first_triangle << 2) | (count - 1), count <= 4;
// unused slots carry an inverted (never-hit) box.  The lower and the upper plane of an axis sit 64 bytes apart, so
// a ray picks its near plane with a per-ray byte offset (0 or 64 by the sign of its direction) and the far plane
// with offset ^ 64: no min / max per slab.
struct Bvh4Node {
    float4 lox, loy, loz; //  0, 16, 32
    int4 child;           // 48
    float4 hix, hiy, hiz; // 64, 80, 96
    int4 pad;
};
static_assert(sizeof(Bvh4Node) == 128 && offsetof(Bvh4Node, hix) == offsetof(Bvh4Node, lox) + 64, "node layout");
#ifndef NEB_TRACE_WAVES
#define NEB_TRACE_WAVES 8 // waves per SIMD the traversal kernels are register-budgeted for
#endif
#ifndef NEB_SHADE_WAVES
#define NEB_SHADE_WAVES 4 // waves per SIMD gi_shade_kernel is register-budgeted for
#endif
#ifndef NEB_LEAF_BATCH
#define NEB_LEAF_BATCH 12
#endif
constexpr int kLeafBatch = NEB_LEAF_BATCH;
#ifndef NEB_TOP_SAH
#define NEB_TOP_SAH 2 // > 0: LBVH subtrees of up to this many triangles are re-linked by a sweep-SAH top level (host pass); 0 / 512 / 64 / 16 / 8 / 4 / 2 / 1 measured 464 / 454 / 437 / 427 / 419 / 403 / 400 / 474 us for the closest-hit pass
#endif
#ifndef NEB_FAST_SHADE
#define NEB_FAST_SHADE 1 // gi_shade_kernel uses the 1-ulp hardware rcp / rsq / sqrt (see fdiv)
#endif
constexpr bool kFastShade = NEB_FAST_SHADE != 0;
#ifndef NEB_MAX_LEAF_TRIS
#define NEB_MAX_LEAF_TRIS 2 // 1..4 (the leaf code keeps count - 1 in two bits); measured 1/2/3/4: 1407 / 1390 / 1403 / 1500 us of GI per 1080p frame
#endif
constexpr int kMaxLeafTris = NEB_MAX_LEAF_TRIS;

struct SceneView {
    const float4* tris;      // 3 x float4 per triangle: {v0.xyz, e1.x} {e1.yz, e2.xy} {e2.z, geom, prim, -}
    const Bvh4Node* nodes;
    const DevGeom* geoms;
    const DevMat* mats;
    const DevTex* texs;
    const uint32_t* indices;
    const float* normals;    // 3 per vertex
    const float* uvs;        // 2 per vertex
    const float* tangents;   // 4 per vertex
    const uint32_t* texels;  // RGBA8 bilinear footprint table: 4 texels (16 B) per texel position, see sample_texture
    const uint4* bundles;    // per-material interleaved footprints (4 x uint4 per texel position), see DevMat::bundle
    const float4* shade;     // 8 x float4 (one 128-B line) per sorted triangle: see pack_shade_records_kernel
    uint32_t n_tris;
    int32_t root;            // root node index, or a leaf code (< 0) for a single-triangle scene
};

struct GiState {
    SceneView view{};
    std::vector<void*> allocs;
    // host copies kept for the build
    std::vector<float> h_tris; // 12 floats per triangle
    float scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};
    uint32_t n_tris = 0, n_nodes = 0;
    bool built = false;
    unsigned long long* d_ray_counter = nullptr;
    neb_gi_hit* d_hits = nullptr;
    bool debug_hits = false;
    float4* d_records = nullptr; // 5 float4 planes + one 4 x float4 record plane over the resident pixels (GiRecords)
    unsigned long long last_stats[8] = {};
    bool defer_resolve = false;
    bool sort_shadow = true;  // "gi_sort_rays" bit 0
    bool sort_bounce = false; // "gi_sort_rays" bit 1
    uint32_t* d_sort = nullptr;      // 4 x npx uint32: keys, vals, keys_out, vals_out
    void* d_sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    uint32_t pending_spp = 1, pending_row0 = 0, pending_row1 = 0;
    uint32_t* d_block_counts = nullptr; // [2][n_block_counts]: bounce / shadow rays per workgroup
    size_t n_block_counts = 0;
};

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

// ------------------------------------------------------------------------------------------------
// Small vector helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 operator*(float3 a, float3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float3 operator-(float3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b)
{
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Arithmetic policy of the shading code.  FAST = false is the C arithmetic of oracle/trace_ref.cpp (IEEE division,
// square root and powf: 11-45 instructions each on gfx950); FAST = true uses the 1-ulp hardware forms an HLSL
// compiler emits for the same source (rcp, rsq, sqrt; x^5 by multiplication; UNORM8 * (1/255)).  Only gi_shade_kernel
// uses it: its inputs (the hit) are fixed by then, so the result moves by ~1e-7 relative.  Ray generation keeps the
// exact forms -- a ray direction that moves by an ulp lands on a slightly different texel footprint, which showed up as
// 3e-5 relative L2 against the oracle -- and so do the G-buffer / direct-light producers.
template <bool FAST> __device__ __forceinline__ float fdiv(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <bool FAST> __device__ __forceinline__ float fsqrt(float x) { return FAST ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); }
template <bool FAST> __device__ __forceinline__ float fpow5(float x)
{
    if (!FAST)
        return powf(x, 5.0f);
    const float x2 = x * x;
    return x2 * x2 * x;
}
template <bool FAST = false> __device__ __forceinline__ float3 normalize3(float3 a)
{
    if (FAST) {
        const float r = __builtin_amdgcn_rsqf(dot3(a, a));
        return f3(a.x * r, a.y * r, a.z * r);
    }
    const float l = sqrtf(dot3(a, a));
    return f3(a.x / l, a.y / l, a.z / l);
}
__device__ __forceinline__ float saturate1(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + t * (b - a); }

constexpr float kPi = 3.14159265f;      // brdf.hlsli:28
constexpr float kPiInv = 1.0f / kPi;
constexpr float kPiTwo = 2.0f * kPi;
constexpr float kTraceMax = 10000.0f;   // TRACING_MAX_DISTANCE, pathtracer.hlsl:9

// rand.hlsli:6-55
__device__ __forceinline__ uint32_t jenkins(uint32_t x)
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}
__device__ __forceinline__ float rand01(uint32_t& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return __uint_as_float(0x3f800000u | (s >> 9)) - 1.0f;
}

// octahedron_encoding.hlsli:16-34
__device__ __forceinline__ float3 oct_unpack(float ex, float ey)
{
    float3 v = f3(ex, ey, 1.0f - fabsf(ex) - fabsf(ey));
    if (v.z < 0.0f) {
        const float sx = (v.x > 0.f) ? 1.f : -1.f, sy = (v.y > 0.f) ? 1.f : -1.f;
        const float nx = (1.0f - fabsf(v.y)) * sx, ny = (1.0f - fabsf(v.x)) * sy;
        v.x = nx;
        v.y = ny;
    }
    return normalize3(v);
}
__device__ __forceinline__ float2 oct_pack(float3 v)
{
    const float s = 1.0f / (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
    const float px = v.x * s, py = v.y * s;
    if (v.z <= 0.0f) {
        const float sx = (px > 0.f) ? 1.f : -1.f, sy = (py > 0.f) ? 1.f : -1.f;
        return make_float2((1.0f - fabsf(py)) * sx, (1.0f - fabsf(px)) * sy);
    }
    return make_float2(px, py);
}

// R11G11B10_FLOAT: unsigned small floats, 5-bit exponent (bias 15), 6/6/5-bit mantissa.
__device__ __forceinline__ float small_float_decode(uint32_t bits, int mbits)
{
    const uint32_t e = bits >> mbits, m = bits & ((1u << mbits) - 1u);
    const float scale = (float)(1u << mbits);
    if (e == 0)
        return ldexpf((float)m / scale, -14);
    if (e == 31)
        return m ? __uint_as_float(0x7fc00000u) : __uint_as_float(0x7f800000u);
    return ldexpf(1.0f + (float)m / scale, (int)e - 15);
}
__device__ __forceinline__ uint32_t small_float_encode(float f, int mbits)
{
    // negatives / NaN -> 0, round-to-nearest-even, overflow -> largest finite (DESIGN.md "G-buffer encodings")
    if (!(f > 0.0f))
        return 0;
    const uint32_t max_bits = (30u << mbits) | ((1u << mbits) - 1u);
    int e;
    const float m = frexpf(f, &e);
    e -= 1;
    if (e > 15)
        return max_bits;
    if (e < -14)
        return (uint32_t)rintf(ldexpf(f, 14 + mbits));
    const float q = rintf(ldexpf(2.0f * m - 1.0f, mbits));
    const uint32_t bits = ((uint32_t)(e + 15) << mbits) + (uint32_t)q;
    return bits > max_bits ? max_bits : bits;
}
__device__ __forceinline__ float3 unpack_r11g11b10(uint32_t v)
{
    return f3(small_float_decode(v & 0x7ffu, 6), small_float_decode((v >> 11) & 0x7ffu, 6),
              small_float_decode((v >> 22) & 0x3ffu, 5));
}

// brdf.hlsli
__device__ __forceinline__ float luminance3(float3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }
__device__ __forceinline__ float3 specular_f0(float3 albedo, float metal)
{
    return f3(lerp1(0.04f, albedo.x, metal), lerp1(0.04f, albedo.y, metal), lerp1(0.04f, albedo.z, metal));
}
template <bool FAST = false> __device__ __forceinline__ float3 fresnel_schlick(float3 f0, float vdoth) // brdf.hlsli:22-25, as written
{
    const float k = 1.0f - fpow5<FAST>(vdoth);
    return f3(f0.x + (1.0f - f0.x) * k, f0.y + (1.0f - f0.y) * k, f0.z + (1.0f - f0.z) * k);
}
template <bool FAST = false> __device__ __forceinline__ float specular_probability(float vdotn, float3 f0, float3 albedo) // brdf.hlsli:129-143
{
    const float dr = luminance3(albedo);
    const float fres = saturate1(luminance3(fresnel_schlick<FAST>(f0, saturate1(vdotn))));
    const float diff = dr * (1.0f - fres);
    const float p = fdiv<FAST>(diff, fmaxf(0.0001f, fres + diff));
    return fminf(fmaxf(p, 0.1f), 0.9f);
}
template <bool FAST = false> __device__ __forceinline__ float3 cosine_hemisphere_aligned(float u0, float u1, float3 sn) // brdf.hlsli:166-185
{
    const float a = fsqrt<FAST>(u0), b = kPiTwo * u1;
    const float3 z = f3(a * cosf(b), a * sinf(b), fsqrt<FAST>(1.0f - u0));
    const float3 up = fabsf(sn.z) < 0.999f ? f3(0, 0, 1) : f3(1, 0, 0);
    const float3 tx = normalize3<FAST>(cross3(up, sn));
    const float3 ty = cross3(sn, tx);
    return normalize3<FAST>(tx * z.x + ty * z.y + sn * z.z);
}
__device__ __forceinline__ float3 perpendicular(float3 u) // sun_disk_sampling.hlsli:45-52
{
    const float3 a = f3(fabsf(u.x), fabsf(u.y), fabsf(u.z));
    const uint32_t xm = ((a.x - a.y) < 0 && (a.x - a.z) < 0) ? 1 : 0;
    const uint32_t ym = (a.y - a.z) < 0 ? (1 ^ xm) : 0;
    const uint32_t zm = 1 ^ (xm | ym);
    return cross3(u, f3((float)xm, (float)ym, (float)zm));
}

struct Surface {
    float3 GN, SN, albedo;
    float roughness, metalness;
};

// EvaluateDirectBRDF (pathtracer.hlsl:209-228).  A zero Cook-Torrance denominator gives 0 instead of
// the reference's 0 * inf = NaN (which NRC discards there) -- DESIGN.md "Deliberate divergences".
template <bool FAST = false> __device__ float3 evaluate_direct_brdf(const Surface& s, float3 V, float3 L)
{
    const float3 N = s.SN;
    const float3 Hv = normalize3<FAST>(V + L);
    const float LdotN = dot3(L, N), VdotH = saturate1(dot3(V, Hv)), VdotN = dot3(V, N), NdotH = dot3(N, Hv);
    const float3 F0 = specular_f0(s.albedo, s.metalness);
    const float3 F = fresnel_schlick<FAST>(F0, saturate1(VdotH));
    const float3 Kd = f3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z);
    const float3 diff = Kd * (s.albedo * kPiInv);
    const float vn = saturate1(VdotN), ln = saturate1(LdotN), nh = saturate1(NdotH);
    float3 spec = f3(0, 0, 0);
    const float den = 4.0f * vn * ln;
    if (den > 0.0f) {
        const float alpha = s.roughness * s.roughness;
        const float a2 = alpha * alpha;
        const float dd = (nh * nh) * (a2 - 1.0f) + 1.0f;
        const float ndf = fdiv<FAST>(a2, kPi * dd * dd);
        const float k = alpha * 0.5f;
        const float gv = vn * fdiv<FAST>(1.0f, vn * (1.0f - k) + k);
        const float gl = ln * fdiv<FAST>(1.0f, ln * (1.0f - k) + k);
        const float c = ndf * (gv * gl);
        const float inv = fdiv<FAST>(1.0f, den);
        spec = f3(c * F.x * inv, c * F.y * inv, c * F.z * inv);
    }
    return diff + spec;
}

// ------------------------------------------------------------------------------------------------
// Traversal
// ------------------------------------------------------------------------------------------------
#ifndef NEB_LDS_STACK
#define NEB_LDS_STACK 16
#endif
constexpr int kLdsStack = NEB_LDS_STACK; // per-lane entries kept in LDS (4 KB per wave at 16)
constexpr int kSpillStack = 64 - kLdsStack; // deeper entries go to a private (scratch) array; rarely touched

struct Hit {
    float t, u, v;
    uint32_t tri;
    uint32_t node_visits, tri_tests; // traversal statistics (neb_gi_traversal_stats)
};

// Moeller-Trumbore in the operation order of oracle/trace_ref.cpp.  Before the oracle's own tests run, the undivided
// numerators (U = u * det, ...) are screened against det with its sign and a few ulps of slack, so the IEEE division
// -- 11 instructions -- is only paid by a wave in which some lane (nearly) hits; a leaf step usually runs with few lanes
// active and whole waves leave at the first rejection.  The screen only rejects what the exact tests reject too.
__device__ __forceinline__ bool intersect_tri_regs(float4 a, float4 b, float4 c, float3 o, float3 d, float tmin, float tmax, float& t,
                                                   float& u, float& v)
{
    const float3 v0 = f3(a.x, a.y, a.z), e1 = f3(a.w, b.x, b.y), e2 = f3(b.z, b.w, c.x);
    const float3 p = cross3(d, e2);
    const float det = dot3(e1, p);
    if (det == 0.0f)
        return false;
    const float3 tv = o - v0;
    const float ads = fabsf(det) * 1.000002f; // |det| plus ~16 ulps
    const uint32_t sgn = __float_as_uint(det) & 0x80000000u;
    const float U = dot3(tv, p);
    const float Us = __uint_as_float(__float_as_uint(U) ^ sgn); // U * sign(det)
    if (Us < 0.0f || Us > ads)
        return false;
    const float3 q = cross3(tv, e1);
    const float V = dot3(d, q);
    const float Vs = __uint_as_float(__float_as_uint(V) ^ sgn);
    if (Vs < 0.0f || Us + Vs > ads)
        return false;
    const float T = dot3(e2, q);
    const float Ts = __uint_as_float(__float_as_uint(T) ^ sgn);
    if (!(Ts > 0.0f && Ts <= tmax * ads)) // (tmin >= 0 everywhere)
        return false;
    const float inv = 1.0f / det;
    u = U * inv;
    v = V * inv;
    t = T * inv;
    return u >= 0.0f && u <= 1.0f && v >= 0.0f && u + v <= 1.0f && t > tmin && t < tmax;
}

__device__ __forceinline__ bool intersect_tri(const float4* __restrict__ tris, uint32_t ti, float3 o, float3 d, float tmin,
                                              float tmax, float& t, float& u, float& v)
{
    return intersect_tri_regs(tris[3 * ti], tris[3 * ti + 1], tris[3 * ti + 2], o, d, tmin, tmax, t, u, v);
}

// Entry distance of the ray into box k of a BVH4 node, as an ordered uint key (misses = 0xffffffff).
// n* / f* are the planes the ray meets first / last on each axis (picked by the sign of the direction when the node
// is loaded).  One fma per plane: t = plane * (1/d) - o/d.  fminf/fmaxf drop NaNs (inf - inf for axis-parallel
// rays), which only makes the interval more conservative; hits themselves are decided by the triangle test.
__device__ __forceinline__ uint32_t slab_key(float nx, float ny, float nz, float fx, float fy, float fz, float3 inv, float3 oinv,
                                             float tmin, float tmax, uint32_t slot)
{
    const float ax = fmaf(nx, inv.x, -oinv.x), bx = fmaf(fx, inv.x, -oinv.x);
    const float ay = fmaf(ny, inv.y, -oinv.y), by = fmaf(fy, inv.y, -oinv.y);
    const float az = fmaf(nz, inv.z, -oinv.z), bz = fmaf(fz, inv.z, -oinv.z);
    const float t0 = fmaxf(fmaxf(ax, ay), fmaxf(az, tmin));
    const float t1 = fminf(fminf(bx, by), fminf(bz, tmax));
    // t0 >= tmin >= 0: its bit pattern orders like an unsigned integer; the low 2 bits carry the slot
    return (t0 <= t1) ? ((__float_as_uint(t0) & ~3u) | slot) : 0xffffffffu;
}

__device__ __forceinline__ void cswap(uint32_t& a, uint32_t& b)
{
    const uint32_t lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}

// The stack pointer and the LDS column are plain scalars and the spill array is its own object: when all three
// sat in one struct the dynamically indexed array kept the whole struct (stack pointer included) in scratch memory,
// and every push / pop paid a scratch round trip behind an s_waitcnt vmcnt(0).
struct TravStack {
    int* lds;   // this lane's column of an LDS array [kLdsStack][64]
    int* spill; // private array of kSpillStack entries
    int sp;
    __device__ __forceinline__ void push(int v)
    {
        if (sp < kLdsStack)
            lds[64 * sp] = v;
        else if (sp < kLdsStack + kSpillStack)
            spill[sp - kLdsStack] = v;
        else
            return; // deeper than 64 pending nodes: drop (cannot happen for a BVH4 over 64-bit Morton keys)
        sp++;
    }
    __device__ __forceinline__ int pop()
    {
        sp--;
        return sp < kLdsStack ? lds[64 * sp] : spill[sp - kLdsStack];
    }
};

// Closest-hit (ANY_HIT = false) or first-hit (ANY_HIT = true) traversal of the BVH4.
// A step handles an inner node and then, if the lane lands on a leaf, the leaf in the same iteration
// (if-if), so lanes in the node phase and lanes in the leaf phase of a wave do not serialise two memory
// round trips per iteration.  Shadow rays skip the front-to-back ordering of the children.
constexpr int kTravDone = (int)0x80000000;

template <bool ANY_HIT, bool STATS>
__device__ bool traverse_t(const SceneView& S, float3 o, float3 d, float tmin, float tmax, int* lds_stack, Hit& hit)
{
    hit.t = tmax;
    hit.tri = ~0u;
    hit.node_visits = hit.tri_tests = 0;
    if (S.n_tris == 0)
        return false;
    const float3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float3 oinv = f3(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    // byte offset of the plane the ray enters through, per axis, inside a Bvh4Node (the exit plane is offset ^ 64)
    const uint32_t onx = d.x < 0.0f ? 64u : 0u, ony = d.y < 0.0f ? 80u : 16u, onz = d.z < 0.0f ? 96u : 32u;
    bool found = false;
    int spill_mem[kSpillStack];
    TravStack st{lds_stack, spill_mem, 0};
    int node = S.root;
    constexpr uint32_t kMiss = 0xffffffffu;
    while (node != kTravDone) {
        if (node >= 0) {
            if (STATS)
                hit.node_visits++;
            const char* nodes = reinterpret_cast<const char*>(S.nodes);
            const uint32_t nb = (uint32_t)node << 7; // 32-bit byte offset (scalar base + vector offset addressing)
            const float4 nx = *reinterpret_cast<const float4*>(nodes + (nb + onx)), fx = *reinterpret_cast<const float4*>(nodes + (nb + (onx ^ 64u)));
            const float4 ny = *reinterpret_cast<const float4*>(nodes + (nb + ony)), fy = *reinterpret_cast<const float4*>(nodes + (nb + (ony ^ 64u)));
            const float4 nz = *reinterpret_cast<const float4*>(nodes + (nb + onz)), fz = *reinterpret_cast<const float4*>(nodes + (nb + (onz ^ 64u)));
            const int4 ch = *reinterpret_cast<const int4*>(nodes + (nb + 48u));
            uint32_t k0 = slab_key(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, inv, oinv, tmin, hit.t, 0u);
            uint32_t k1 = slab_key(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, inv, oinv, tmin, hit.t, 1u);
            uint32_t k2 = slab_key(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, inv, oinv, tmin, hit.t, 2u);
            uint32_t k3 = slab_key(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, inv, oinv, tmin, hit.t, 3u);
            // select by the slot bits without branches (two levels of v_cndmask)
            auto child_of = [&](uint32_t key) -> int {
                const bool b0 = (key & 1u) != 0u, b1 = (key & 2u) != 0u;
                const int lo = b0 ? ch.y : ch.x, hi = b0 ? ch.w : ch.z;
                return b1 ? hi : lo;
            };
            if (!ANY_HIT) { // sorting network: k0 <= k1 <= k2 <= k3 (nearest first, misses last)
                cswap(k0, k1);
                cswap(k2, k3);
                cswap(k0, k2);
                cswap(k1, k3);
                cswap(k1, k2);
                node = kTravDone;
                if (k0 != kMiss) {
                    if (k3 != kMiss)
                        st.push(child_of(k3));
                    if (k2 != kMiss)
                        st.push(child_of(k2));
                    if (k1 != kMiss)
                        st.push(child_of(k1));
                    node = child_of(k0);
                }
            } else { // any order: continue with the first hit child, stack the others
                node = kTravDone;
                if (k3 != kMiss)
                    node = ch.w;
                if (k2 != kMiss) {
                    if (node != kTravDone)
                        st.push(node);
                    node = ch.z;
                }
                if (k1 != kMiss) {
                    if (node != kTravDone)
                        st.push(node);
                    node = ch.y;
                }
                if (k0 != kMiss) {
                    if (node != kTravDone)
                        st.push(node);
                    node = ch.x;
                }
            }
            if (node == kTravDone && st.sp)
                node = st.pop();
        }
        // Any-hit rays batch their leaf steps: a lane that holds a leaf waits until kLeafBatch lanes of the wave do (or
        // none has a node left), so the triangle code runs with fuller waves (shadow pass 213 -> 206 us; the closest-hit
        // pass, whose lanes need the shrunk hit.t at once, measured no gain at 4 / 12 and lost at 24).
        const bool holds_leaf = node < 0 && node != kTravDone;
        bool run_leaves = true;
        if (ANY_HIT && kLeafBatch > 1)
            run_leaves = __popcll(__ballot(holds_leaf)) >= kLeafBatch || __ballot(node >= 0) == 0ull;
        if (holds_leaf && run_leaves) {
            const uint32_t code = (uint32_t)~node;
            const uint32_t first = code >> 2, count = (code & 3u) + 1u;
            if (STATS)
                hit.tri_tests += count;
            if constexpr (kMaxLeafTris <= 2) {
                // both triangles are fetched before the first test (one memory round trip per leaf).  A one-triangle
                // leaf tests its triangle twice: the second test cannot pass t < hit.t again.
                const uint32_t second = first + count - 1u;
                float4 a0 = S.tris[3 * first];
                const float4 b0 = S.tris[3 * first + 1], c0 = S.tris[3 * first + 2];
                const float4 a1 = S.tris[3 * second], b1 = S.tris[3 * second + 1], c1 = S.tris[3 * second + 2];
                // keep the v0 load in this batch: left alone, the compiler sinks it behind the det == 0 test of the
                // first triangle, a second dependent memory access per leaf
                asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a0.z));
                float t, u, v;
                if (intersect_tri_regs(a0, b0, c0, o, d, tmin, hit.t, t, u, v)) {
                    hit.t = t, hit.u = u, hit.v = v, hit.tri = first, found = true;
                }
                if (intersect_tri_regs(a1, b1, c1, o, d, tmin, hit.t, t, u, v)) {
                    hit.t = t, hit.u = u, hit.v = v, hit.tri = second, found = true;
                }
            } else {
                for (uint32_t k = 0; k < count; ++k) {
                    float t, u, v;
                    if (intersect_tri(S.tris, first + k, o, d, tmin, hit.t, t, u, v)) {
                        hit.t = t, hit.u = u, hit.v = v, hit.tri = first + k, found = true;
                    }
                }
            }
            if (ANY_HIT && found)
                return true;
            node = st.sp ? st.pop() : kTravDone;
        }
    }
    return found;
}

// `stats` (wave-uniform, diagnostics) selects the instantiation that also counts node visits and triangle tests.
__device__ __forceinline__ bool traverse(const SceneView& S, float3 o, float3 d, float tmin, float tmax, bool any_hit, int* lds_stack,
                                         Hit& hit, bool stats = false)
{
    if (stats)
        return any_hit ? traverse_t<true, true>(S, o, d, tmin, tmax, lds_stack, hit) : traverse_t<false, true>(S, o, d, tmin, tmax, lds_stack, hit);
    return any_hit ? traverse_t<true, false>(S, o, d, tmin, tmax, lds_stack, hit) : traverse_t<false, false>(S, o, d, tmin, tmax, lds_stack, hit);
}

// SampleLevel(linear, wrap, mip 0) of an RGBA8 UNORM texture (pathtracer.hlsl:359,377,390).
// wrap-addressed texel position and bilinear fractions of (u, v) in a w x h texture
__device__ __forceinline__ void texel_position(uint32_t w, uint32_t h, float u, float v, int& x0, int& y0, float& fx, float& fy)
{
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    fx = x - fx0;
    fy = y - fy0;
    const int W = (int)w, H = (int)h;
    x0 = (int)fx0 % W;
    y0 = (int)fy0 % H;
    if (x0 < 0)
        x0 += W;
    if (y0 < 0)
        y0 += H;
}
// bilinear filter of one footprint {(x0,y0), (x0+1,y0), (x0,y0+1), (x0+1,y0+1)} of RGBA8 UNORM texels
template <bool FAST = false> __device__ __forceinline__ float4 filter_footprint(uint4 fp, float fx, float fy)
{
    const uint32_t p00 = fp.x, p10 = fp.y, p01 = fp.z, p11 = fp.w;
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // UNORM8 -> float: exact division as the oracle, or one v_cvt_f32_ubyte + one multiply
        auto un = [&](uint32_t p) { const float q = (float)((p >> (8 * c)) & 0xffu); return FAST ? q * (1.0f / 255.0f) : q / 255.0f; };
        const float a = un(p00), b = un(p10), cc = un(p01), dd = un(p11);
        const float top = a + fx * (b - a), bot = cc + fx * (dd - cc);
        r[c] = top + fy * (bot - top);
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}
template <bool FAST = false> __device__ float4 sample_texture(const SceneView& S, int ti, float u, float v)
{
    const DevTex t = S.texs[ti];
    int x0, y0;
    float fx, fy;
    texel_position(t.w, t.h, u, v, x0, y0, fx, fy);
    // bilinear footprint table: entry (x0, y0) holds the four texels {(x0,y0), (x0+1,y0), (x0,y0+1), (x0+1,y0+1)} with the
    // wrap already applied, so a filtered fetch is ONE 16-byte load instead of four scattered dwords
    const uint4 fp = reinterpret_cast<const uint4*>(S.texels)[(size_t)t.offset + (size_t)y0 * (int)t.w + x0];
    return filter_footprint<FAST>(fp, fx, fy);
}

__device__ __forceinline__ float3 load3(const float* p, uint32_t i) { return f3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ float3 xform_dir(const float* m, float3 p) // (p,0) * M, row-vector convention
{
    return f3(p.x * m[0] + p.y * m[3] + p.z * m[6], p.x * m[1] + p.y * m[4] + p.z * m[7], p.x * m[2] + p.y * m[5] + p.z * m[8]);
}

// ReconstructSurfaceData (pathtracer.hlsl:299-395)
// Per-triangle shading record (128 B, one cache line) so that a hit costs one line instead of ~10 scattered
// ones (3 indices, 3 x normal/uv/tangent in three SoA pools):
//   r0 {n0.xyz, uv0.x} r1 {n1.xyz, uv0.y} r2 {n2.xyz, uv1.x} r3..r5 tangent0..2 r6 {uv1.y, uv2.x, uv2.y, geometry} r7 {primitive,-,-,-}
struct TriShade {
    float3 n0, n1, n2;
    float2 uv0, uv1, uv2;
    float4 t0, t1, t2;
    uint32_t geom; // GeometryIndex() of the triangle (r6.w); PrimitiveIndex() is r7.x (debug records only)
};
__device__ __forceinline__ TriShade load_tri_shade(const SceneView& S, uint32_t tri)
{
    const float4* r = S.shade + 8 * (size_t)tri;
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r6 = r[6];
    TriShade t;
    t.n0 = f3(r0.x, r0.y, r0.z);
    t.n1 = f3(r1.x, r1.y, r1.z);
    t.n2 = f3(r2.x, r2.y, r2.z);
    t.uv0 = make_float2(r0.w, r1.w);
    t.uv1 = make_float2(r2.w, r6.x);
    t.uv2 = make_float2(r6.y, r6.z);
    t.geom = __float_as_uint(r6.w);
    t.t0 = r[3];
    t.t1 = r[4];
    t.t2 = r[5];
    return t;
}

// ReconstructSurfaceData (pathtracer.hlsl:299-395); `tri` is the sorted triangle index of the hit.
template <bool FAST = false> __device__ bool reconstruct_surface(const SceneView& S, uint32_t tri, float bu, float bv, Surface& out, uint32_t& geom)
{
    // the record is fetched first and names its geometry itself: one gathered line per hit, and the geometry / material
    // table reads hang off it instead of off a second gather into the triangle array
    const TriShade ts = load_tri_shade(S, tri);
    geom = ts.geom;
    const DevGeom g = S.geoms[geom];
    const float b0 = 1.0f - (bu + bv), b1 = bu, b2 = bv;
    if (!g.valid)
        return false; // :313-318
    const float3 n0 = ts.n0, n1 = ts.n1, n2 = ts.n2;
    const float3 gn = normalize3<FAST>(f3(n0.x * b0 + n1.x * b1 + n2.x * b2, n0.y * b0 + n1.y * b1 + n2.y * b2, n0.z * b0 + n1.z * b1 + n2.z * b2));
    out.GN = normalize3<FAST>(xform_dir(g.m, gn)); // :340
    const float u = ts.uv0.x * b0 + ts.uv1.x * b1 + ts.uv2.x * b2;
    const float v = ts.uv0.y * b0 + ts.uv1.y * b1 + ts.uv2.y * b2;
    if (g.material < 0)
        return false; // :349
    const DevMat m = S.mats[g.material];
    float4 t_albedo, t_normal, t_rm;
    const bool bundled = m.bundle_w != 0; // then all three maps exist
    if (bundled) {
        int x0, y0;
        float fx, fy;
        texel_position(m.bundle_w, m.bundle_h, u, v, x0, y0, fx, fy);
        const uint4* e = S.bundles + 4 * ((size_t)m.bundle + (size_t)y0 * m.bundle_w + x0);
        const uint4 fa = e[0], fn = e[1], fr = e[2]; // one 64-byte line
        t_albedo = filter_footprint<FAST>(fa, fx, fy);
        t_normal = filter_footprint<FAST>(fn, fx, fy);
        t_rm = filter_footprint<FAST>(fr, fx, fy);
    }
    if (m.tex[0] < 0) {
        out.albedo = f3(m.albedo[0], m.albedo[1], m.albedo[2]);
    } else {
        const float4 t = bundled ? t_albedo : sample_texture<FAST>(S, m.tex[0], u, v);
        out.albedo = f3(t.x, t.y, t.z);
    }
    if (m.tex[1] < 0) {
        out.SN = out.GN;
    } else {
        float tg[4];
        tg[0] = ts.t0.x * b0 + ts.t1.x * b1 + ts.t2.x * b2;
        tg[1] = ts.t0.y * b0 + ts.t1.y * b1 + ts.t2.y * b2;
        tg[2] = ts.t0.z * b0 + ts.t1.z * b1 + ts.t2.z * b2;
        tg[3] = ts.t0.w * b0 + ts.t1.w * b1 + ts.t2.w * b2;
        const float l4 = fsqrt<FAST>(tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2] + tg[3] * tg[3]); // normalize(float4), :371
#pragma unroll
        for (int k = 0; k < 4; ++k)
            tg[k] = fdiv<FAST>(tg[k], l4);
        const float3 T = f3(tg[0], tg[1], tg[2]);
        const float3 B = normalize3<FAST>(cross3(out.GN, T) * tg[3]);
        const float4 t = bundled ? t_normal : sample_texture<FAST>(S, m.tex[1], u, v);
        const float3 N = f3(t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f, t.z * 2.0f - 1.0f);
        out.SN = normalize3<FAST>(T * N.x + B * N.y + out.GN * N.z); // mul(N, float3x3(T, B, GN))
    }
    if (m.tex[2] < 0) {
        out.roughness = m.rough;
        out.metalness = m.metal;
    } else {
        const float4 t = bundled ? t_rm : sample_texture<FAST>(S, m.tex[2], u, v);
        out.roughness = t.y; // .g
        out.metalness = t.z; // .b
    }
    return true;
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

// ------------------------------------------------------------------------------------------------
// GI as three wavefront stages per sample (one wave per 8x8 pixel tile, records indexed by pixel):
//   gi_raygen_trace_kernel : G-buffer -> RNG -> throughput -> cosine ray -> closest-hit traversal
//   gi_shade_kernel        : miss -> sky; hit -> ReconstructSurfaceData, sun-disk shadow ray + BRDF contribution
//   gi_shadow_trace_kernel : any-hit traversal of the shadow ray, accumulate; last sample adds into radiance[cur]
// Splitting keeps the two traversal kernels at <= 64 VGPRs (8 waves/SIMD) and the register-hungry shading
// away from them.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t expand_bits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ uint32_t morton30(float3 p, const float* smin, const float* sinv)
{
    const uint32_t qx = (uint32_t)fminf(fmaxf((p.x - smin[0]) * sinv[0] * 1024.0f, 0.0f), 1023.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf((p.y - smin[1]) * sinv[1] * 1024.0f, 0.0f), 1023.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf((p.z - smin[2]) * sinv[2] * 1024.0f, 0.0f), 1023.0f);
    return (expand_bits10(qx) << 2) | (expand_bits10(qy) << 1) | expand_bits10(qz);
}

// Radix-sort key width for ray ordering: the top kSortBits bits of the 30-bit Morton code of the ray origin
// (each 8 bits are one pass of raysort.hip over the pairs).
#ifndef NEB_SORT_BITS
#define NEB_SORT_BITS 16
#endif
constexpr int kSortBits = NEB_SORT_BITS;
static_assert(kSortBits >= 1 && kSortBits <= 16, "raysort.hip sorts at most two 8-bit digits");

__device__ __forceinline__ uint32_t bounce_sort_key(float3 o, float3 d, const float* smin, const float* sinv)
{
    const uint32_t oct = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
    const uint32_t key = (oct << (kSortBits - 3)) | (morton30(o, smin, sinv) >> (30 - (kSortBits - 3)));
    return min(key, (1u << kSortBits) - 2u);
}

struct GiRecords {
    float4* ray_o;   // {origin.xyz, tmin}      bounce ray of the path
    float4* ray_d;   // {direction.xyz, alive}  alive = 1: the path continues with this ray
    float4* hit;     // {t (<0 miss), u, v, tri bits}
    float4* path;    // {throughput.xyz, rng bits}
    float4* state;   // {V.xyz, rng bits}: survives across the samples of a pixel
    // Everything the shadow pass needs about a pixel sits in ONE 64-byte line, because that pass visits the pixels in
    // sorted (scattered) order and every separate plane would cost it another line per ray:
    //   [0] {origin.xyz, tmin}  [1] {direction.xyz, valid}  sun shadow ray of the current vertex (valid = 1: trace it)
    //   [2] {BRDF * sunRadiance * throughput, traversal iterations (diagnostics)}
    //   [3] {sum of the samples' radiance so far, -}
    float4* srec;
};
constexpr int kSrO = 0, kSrD = 1, kSrContrib = 2, kSrSum = 3;

struct GiArgs {
    SceneView S;
    neb_gi_constants c;
    GiRecords R;
    const uint32_t* albedo;
    const uint32_t* rough_metal; // 2 x fp16
    const uint2* world_pos;      // 4 x fp16
    const uint2* normal;         // 4 x fp16 (.zw = shading normal)
    float4* radiance;
    neb_gi_hit* hits;            // may be null
    unsigned long long* ray_counter; // diagnostics: [1..4] traversal steps (only touched when stats != 0)
    uint32_t* bounce_counts;     // per-workgroup bounce-ray counts
    uint32_t* shadow_counts;     // per-workgroup shadow-ray counts
    uint32_t W, row_begin, row0, row1, tiles_x;
    uint32_t sample;             // index of the sample this launch handles
    uint32_t bounce;             // path vertex this launch handles: 1 .. maxPathVertices - 1
    uint32_t stats;              // 1: also count shadow-ray traversal steps (slow path, diagnostics)
    uint32_t defer_resolve;      // 1: leave the frame's sum in the record plane; neb_gi_resolve adds it into radiance[cur] later
    uint32_t* sort_keys;         // shadow-ray sorting ("gi_sort_shadow_rays"): Morton key of the ray origin per pixel, or null
    uint32_t* sort_vals;         // pixel index per key
    const uint32_t* sort_order;  // pixel indices in key order (after the radix sort), or null: pixel order
    uint32_t* bsort_keys;        // same for the bounce rays ("gi_sort_rays" bit 1): key = direction octant | origin Morton code
    uint32_t* bsort_vals;
    uint32_t raygen_only;        // 1: gi_raygen_trace_kernel only writes the ray record and its key (a sorted trace follows)
    float smin[3], sinv[3];      // scene box for the Morton keys
    uint32_t first_px, n_px;     // dispatched pixel range [first_px, first_px + n_px) of the resident planes
};

__device__ __forceinline__ bool gi_pixel(const GiArgs& a, uint32_t& x, uint32_t& y, size_t& i)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t tile_x = blockIdx.x % a.tiles_x, tile_y = blockIdx.x / a.tiles_x;
    x = tile_x * 8 + (lane & 7);
    y = a.row0 + tile_y * 8 + (lane >> 3);
    i = (size_t)(y - a.row_begin) * a.W + x;
    return x < a.W && y < a.row1;
}

// Ray accounting without same-address atomics (one hot word saturates at ~90 atomics/us on MI355X, which cost
// more than the traversal itself): every workgroup owns one slot of a per-kernel count array; the host sums them.
__device__ __forceinline__ void count_rays(uint32_t* block_counts, uint32_t mine)
{
    uint32_t total = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        total += __shfl_down(total, off);
    if (threadIdx.x == 0 && total)
        block_counts[blockIdx.x] += total; // slot owned by this workgroup; launches on one stream are ordered
}

__global__ __launch_bounds__(64, NEB_TRACE_WAVES) void gi_raygen_trace_kernel(GiArgs a)
{
    __shared__ int stack_mem[kLdsStack * 64];
    uint32_t x, y;
    size_t i;
    const bool active = gi_pixel(a, x, y, i);
    uint32_t rays = 0;
    if (active) {
        const float3 albedo = unpack_r11g11b10(a.albedo[i]);
        const uint2 wp = a.world_pos[i];
        const float3 worldPos = f3(half_bits_to_float(wp.x & 0xffffu), half_bits_to_float(wp.x >> 16), half_bits_to_float(wp.y & 0xffffu));
        const uint32_t nzw = a.normal[i].y;
        const float3 SN = oct_unpack(half_bits_to_float(nzw & 0xffffu), half_bits_to_float(nzw >> 16));
        const float metalness = half_bits_to_float(a.rough_metal[i] >> 16);
        uint32_t rng;
        float3 V;
        if (a.sample == 0) {
            rng = jenkins((x + y * a.W) ^ jenkins(a.c.frameIndex)); // InitRNG, rand.hlsli:26-30
            V = f3(a.c.cameraWorldPos[0], a.c.cameraWorldPos[1], a.c.cameraWorldPos[2]) - worldPos; // :431
        } else { // V survives across samples (overwritten at :522), and so does the RNG stream
            const float4 st = a.R.state[i];
            V = f3(st.x, st.y, st.z);
            rng = __float_as_uint(st.w);
        }
        (void)rand01(rng); // consumed by NrcCreatePathState (:438)
        const float3 F0 = specular_f0(albedo, metalness);
        float3 throughput = f3(1, 1, 1) * (albedo * (1.0f - metalness)); // :474
        const float pd = 1.0f - specular_probability(saturate1(dot3(normalize3(V), SN)), F0, albedo);
        if (rand01(rng) < pd)
            throughput = f3(throughput.x / pd, throughput.y / pd, throughput.z / pd); // :476-479
        const float u0 = rand01(rng), u1 = rand01(rng);
        const float3 dir = cosine_hemisphere_aligned(u0, u1, SN);
        const float3 org = worldPos + SN * 1e-2f; // :138
        const bool bounce = a.c.maxPathVertices > 1; // for (bounce = 1; bounce < nrcMaxPathVertices; ...)
        a.R.path[i] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(rng));
        a.R.ray_o[i] = make_float4(org.x, org.y, org.z, 0.01f);
        a.R.ray_d[i] = make_float4(dir.x, dir.y, dir.z, bounce ? 1.0f : 0.0f);
        if (a.sample + 1 < a.c.samplesPerPixel) // only the next sample of this pixel reads it
            a.R.state[i] = make_float4(V.x, V.y, V.z, __uint_as_float(rng));
        float4 h = make_float4(bounce ? -1.0f : -2.0f, 0.f, 0.f, 0.f); // -2: no bounce at all, nothing is added
        rays = bounce ? 1u : 0u;
        if (a.bsort_keys) {
            a.bsort_keys[i] = bounce ? bounce_sort_key(org, dir, a.smin, a.sinv) : (1u << kSortBits) - 1u;
            a.bsort_vals[i] = (uint32_t)i;
        }
        if (bounce && !a.raygen_only) {
            Hit hit;
            if (traverse(a.S, org, dir, 0.01f, kTraceMax, false, stack_mem + threadIdx.x, hit, a.stats != 0))
                h = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.tri));
            if (a.stats) { // diagnostics only
                atomicAdd(a.ray_counter + 1, (unsigned long long)hit.node_visits);
                atomicAdd(a.ray_counter + 2, (unsigned long long)hit.tri_tests);
                a.R.srec[4 * i + kSrContrib].w = __uint_as_float(hit.node_visits + ((hit.tri_tests + 3u) >> 2)); // loop iterations of this ray
            }
        }
        a.R.hit[i] = h;
    }
    count_rays(a.bounce_counts, rays);
}

// Closest-hit traversal of the bounce rays of path vertices >= 2 (vertex 1 is fused into ray generation).
__global__ __launch_bounds__(64, NEB_TRACE_WAVES) void gi_bounce_trace_kernel(GiArgs a)
{
    __shared__ int stack_mem[kLdsStack * 64];
    uint32_t x, y;
    size_t i = 0;
    bool active;
    if (a.sort_order) { // one lane per entry of the sorted order (every dispatched pixel appears exactly once)
        const uint32_t j = blockIdx.x * 64u + threadIdx.x;
        active = j < a.n_px;
        if (active)
            i = a.sort_order[j];
    } else {
        active = gi_pixel(a, x, y, i);
    }
    uint32_t rays = 0;
    if (active) {
        const float4 rd = a.R.ray_d[i];
        if (rd.w != 0.0f) {
            const float4 ro = a.R.ray_o[i];
            Hit hit;
            rays = 1;
            float4 h = make_float4(-1.0f, 0.f, 0.f, 0.f);
            if (traverse(a.S, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), ro.w, kTraceMax, false, stack_mem + threadIdx.x, hit))
                h = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.tri));
            a.R.hit[i] = h;
        }
    }
    if (a.bounce > 1) // vertex-1 rays are counted by the ray generator
        count_rays(a.bounce_counts, rays);
}

__global__ __launch_bounds__(64, NEB_SHADE_WAVES) void gi_shade_kernel(GiArgs a)
{
    uint32_t x, y;
    size_t i;
    const bool active = gi_pixel(a, x, y, i);
    uint32_t rays = 0;
    // the pixel's shadow record (GiRecords::srec), built in registers and stored through an LDS transpose below
    float4 rec_o = make_float4(0.f, 0.f, 0.f, 0.f), rec_c = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 shadow_d = make_float4(0.f, 0.f, 0.f, 0.f); // valid = 0: no shadow ray
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const float4 h = a.R.hit[i];
        const float4 pth = a.R.path[i];
        const float4 rd = a.R.ray_d[i];
        const uint32_t trav_iters = a.stats ? __float_as_uint(a.R.srec[4 * i + kSrContrib].w) : 0u; // diagnostics (written by the tracer)
        float3 throughput = f3(pth.x, pth.y, pth.z);
        if (!(a.sample == 0 && a.bounce == 1))
            sum = a.R.srec[4 * i + kSrSum];
        float4 next_d = make_float4(0.f, 0.f, 0.f, 0.f);   // alive = 0: the path ends here
        neb_gi_hit dbg = {-1.0f, ~0u, ~0u, 0u};
        const bool alive = rd.w != 0.0f; // (vertex 1 with maxPathVertices <= 1: hit.x == -2, nothing is added)
        if (alive && h.x == -1.0f) { // miss: radiance += skyColor * throughput (:508)
            sum.x += a.c.skyColor[0] * throughput.x;
            sum.y += a.c.skyColor[1] * throughput.y;
            sum.z += a.c.skyColor[2] * throughput.z;
        } else if (alive && h.x >= 0.0f) {
            const uint32_t tri = __float_as_uint(h.w);
            Surface surf;
            uint32_t geom;
            const bool shaded = reconstruct_surface<kFastShade>(a.S, tri, h.y, h.z, surf, geom);
            dbg.t = h.x;
            dbg.geometry = geom;
            if (a.hits)
                dbg.primitive = __float_as_uint(a.S.shade[8 * (size_t)tri + 7].x);
            if (shaded) {
                const float4 ro = a.R.ray_o[i];
                const float3 org = f3(ro.x, ro.y, ro.z), dir = f3(rd.x, rd.y, rd.z);
                const float3 hitP = org + dir * h.x;
                const float3 V = normalize3<kFastShade>(-dir); // :522
                uint32_t rng = __float_as_uint(pth.w);
                const float a0 = rand01(rng), a1 = rand01(rng);
                const float angle = a0 * 2.0f * 3.1415926535f, dist = fsqrt<kFastShade>(a1);
                const float3 sun_dir = f3(a.c.sunLightDirection[0], a.c.sunLightDirection[1], a.c.sunLightDirection[2]);
                const float3 sun_rad = f3(a.c.sunLightRadiance[0], a.c.sunLightRadiance[1], a.c.sunLightRadiance[2]);
                const float3 L = normalize3<kFastShade>(-sun_dir);
                const float3 Bv = normalize3<kFastShade>(perpendicular(L));
                const float3 T = cross3(Bv, L);
                // (the disk offset is scaled by tan(0.29 deg) = 0.005: the ~1e-6 error of v_sin / v_cos moves the direction by
                // less than an ulp, so the hardware forms are safe here; the hemisphere sampler keeps sinf / cosf)
                const float sn_a = kFastShade ? __sinf(angle) : sinf(angle), cs_a = kFastShade ? __cosf(angle) : cosf(angle);
                const float3 inc = normalize3<kFastShade>(L + (Bv * sn_a + T * cs_a) * a.c.sunTanHalfAngle * dist);
                const bool transition = dot3(surf.GN, inc) <= 0.0f;
                const float3 so = hitP + (transition ? -surf.GN : surf.GN) * 1e-2f;
                const float3 O = evaluate_direct_brdf<kFastShade>(surf, V, L) * sun_rad * throughput; // :573-574
                rec_o = make_float4(so.x, so.y, so.z, 0.001f);
                shadow_d = make_float4(inc.x, inc.y, inc.z, 1.0f);
                rec_c = make_float4(O.x, O.y, O.z, 0.f);
                rays = 1;
                if (a.bounce + 1 < a.c.maxPathVertices) { // not the last vertex (:579-583): sample the next bounce
                    // EvaluateIndirectBRDF (:230-259) takes rng BY VALUE: its draws do not advance the path's stream,
                    // so the Rand(rng) of :614 returns the same number as the first of them.
                    uint32_t rng_copy = rng;
                    const float3 SNn = normalize3<kFastShade>(surf.SN);
                    const float e0 = rand01(rng_copy), e1 = rand01(rng_copy);
                    const float3 Ld = cosine_hemisphere_aligned<kFastShade>(e0, e1, SNn);
                    const float pdiff = 1.0f - specular_probability<kFastShade>(saturate1(dot3(V, SNn)), specular_f0(surf.albedo, surf.metalness), surf.albedo);
                    const float3 no = hitP + surf.GN * 1e-2f; // :607
                    throughput = throughput * (surf.albedo * (1.0f - surf.metalness)); // :613
                    if (rand01(rng) < pdiff)
                        throughput = f3(fdiv<kFastShade>(throughput.x, pdiff), fdiv<kFastShade>(throughput.y, pdiff), fdiv<kFastShade>(throughput.z, pdiff)); // :614-618
                    a.R.ray_o[i] = make_float4(no.x, no.y, no.z, 0.001f);
                    next_d = make_float4(Ld.x, Ld.y, Ld.z, 1.0f);
                    a.R.path[i] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(rng));
                } // (the last vertex leaves R.path alone: the next sample's ray generation rewrites it)
                if (a.sample + 1 < a.c.samplesPerPixel) // V and the RNG stream carry over to the next sample only
                    a.R.state[i] = make_float4(V.x, V.y, V.z, __uint_as_float(rng));
            }
        }
        if (a.bounce + 1 < a.c.maxPathVertices) // nobody traces or shades a ray after the last vertex
            a.R.ray_d[i] = next_d;
        if (a.sort_keys) { // shadow rays are all (nearly) parallel: grouping them by origin makes a wave's rays walk the same nodes
            uint32_t key = (1u << kSortBits) - 1u; // pixels without a shadow ray sort last
            if (shadow_d.w != 0.0f) {
                key = min(morton30(f3(rec_o.x, rec_o.y, rec_o.z), a.smin, a.sinv) >> (30 - kSortBits), (1u << kSortBits) - 2u);
            }
            a.sort_keys[i] = key;
            a.sort_vals[i] = (uint32_t)i;
        }
        if (a.bsort_keys) { // next bounce ray: direction octant, then origin
            uint32_t key = (1u << kSortBits) - 1u;
            if (next_d.w != 0.0f) {
                const float4 no4 = a.R.ray_o[i];
                key = bounce_sort_key(f3(no4.x, no4.y, no4.z), f3(next_d.x, next_d.y, next_d.z), a.smin, a.sinv);
            }
            a.bsort_keys[i] = key;
            a.bsort_vals[i] = (uint32_t)i;
        }
        if (a.hits && a.bounce == 1) {
            if (a.stats)
                dbg.flags |= trav_iters << 8; // diagnostics: traversal iterations of the bounce ray (tools/gi_divergence.py)
            a.hits[i] = dbg;
        }
    }
    // Store the 64-byte records of the wave's 8x8 tile.  Lane-per-pixel stores would write 16 bytes at a 64-byte stride
    // four times over; transposed through LDS, every store instruction writes two 512-byte runs (one tile row each).
    __shared__ float4 xpose[64 * 4];
    const uint32_t lane = threadIdx.x;
    xpose[lane * 4 + kSrO] = rec_o;
    xpose[lane * 4 + kSrD] = shadow_d;
    xpose[lane * 4 + kSrContrib] = rec_c;
    xpose[lane * 4 + kSrSum] = sum;
    __syncthreads(); // the workgroup is this one wave
    const uint32_t tile_x = blockIdx.x % a.tiles_x, tile_y = blockIdx.x / a.tiles_x;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t row = 2 * k + (lane >> 5), col = (lane & 31u) >> 2, comp = lane & 3u;
        const uint32_t px = tile_x * 8 + col, py = a.row0 + tile_y * 8 + row;
        if (px < a.W && py < a.row1)
            a.R.srec[4 * ((size_t)(py - a.row_begin) * a.W + px) + comp] = xpose[(row * 8 + col) * 4 + comp];
    }
    count_rays(a.shadow_counts, rays);
}

// (A persistent-wave variant with per-lane ray refill was measured and dropped: lanes of a wave finish after
// 23 steps on average and the slowest after ~55, so the refill bookkeeping cost more than the idle lanes it
// recovered: 1.24 ms vs 0.52 ms for the bounce rays at 1080p.)
__global__ __launch_bounds__(64, NEB_TRACE_WAVES) void gi_shadow_trace_kernel(GiArgs a)
{
    __shared__ int stack_mem[kLdsStack * 64];
    uint32_t x, y;
    size_t i;
    if (a.sort_order) { // one lane per entry of the sorted order (every dispatched pixel appears exactly once)
        const uint32_t j = blockIdx.x * 64u + threadIdx.x;
        if (j >= a.n_px)
            return;
        i = a.sort_order[j];
    } else if (!gi_pixel(a, x, y, i)) {
        return;
    }
    const float4* rec = a.R.srec + 4 * i;
    const float4 rd = rec[kSrD];
    float4 sum = rec[kSrSum];
    if (rd.w != 0.0f) {
        const float4 ro = rec[kSrO];
        Hit sh;
        const bool occluded = traverse(a.S, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), ro.w, kTraceMax, true, stack_mem + threadIdx.x, sh, a.stats != 0);
        if (a.stats) { // diagnostics only
            atomicAdd(a.ray_counter + 3, (unsigned long long)sh.node_visits);
            atomicAdd(a.ray_counter + 4, (unsigned long long)sh.tri_tests);
        }
        if (!occluded) { // radiance += BRDF * sunRadiance * throughput (:571-575)
            const float4 c = rec[kSrContrib];
            sum.x += c.x;
            sum.y += c.y;
            sum.z += c.z;
            if (a.hits && a.bounce == 1)
                a.hits[i].flags |= 1u;
        }
    }
    const bool last_vertex = a.bounce + 1 >= a.c.maxPathVertices;
    if (a.sample + 1 == a.c.samplesPerPixel && last_vertex && !a.defer_resolve) { // stands in for NRC Resolve: radiance[cur] += mean over spp
        // (pixels are visited in sorted order here: the read-modify-write is a gather + scatter, so an occluded ray
        // with nothing else to add -- about half of them -- leaves radiance[cur] alone: x + 0 == x)
        if (sum.x != 0.0f || sum.y != 0.0f || sum.z != 0.0f) {
            const float inv_spp = 1.0f / (float)a.c.samplesPerPixel;
            float4 r = a.radiance[i];
            r.x += sum.x * inv_spp;
            r.y += sum.y * inv_spp;
            r.z += sum.z * inv_spp;
            a.radiance[i] = r;
        }
    } else {
        a.R.srec[4 * i + kSrSum] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// G-buffer producer ("next" row f2): primary visibility through the same LBVH
// ------------------------------------------------------------------------------------------------
struct GbufArgs {
    SceneView S;
    float eye[3], xaxis[3], yaxis[3], zaxis[3];
    float tan_half, aspect, m22, m32;
    uint32_t* albedo;
    uint32_t* rough_metal;
    uint2* world_pos;
    uint2* normal;
    uint32_t* depth;
    uint32_t W, H, row_begin, row0, row1, tiles_x;
};

__global__ __launch_bounds__(64) void gbuffer_kernel(GbufArgs a)
{
    __shared__ int stack_mem[kLdsStack * 64];
    const uint32_t lane = threadIdx.x;
    int* stack = stack_mem + lane;
    const uint32_t tile_x = blockIdx.x % a.tiles_x, tile_y = blockIdx.x / a.tiles_x;
    const uint32_t x = tile_x * 8 + (lane & 7), y = a.row0 + tile_y * 8 + (lane >> 3);
    if (x >= a.W || y >= a.row1)
        return;
    const size_t i = (size_t)(y - a.row_begin) * a.W + x;
    const float3 eye = f3(a.eye[0], a.eye[1], a.eye[2]), xa = f3(a.xaxis[0], a.xaxis[1], a.xaxis[2]),
                 ya = f3(a.yaxis[0], a.yaxis[1], a.yaxis[2]), za = f3(a.zaxis[0], a.zaxis[1], a.zaxis[2]);
    const float ndc_x = ((float)x + 0.5f) / (float)a.W * 2.0f - 1.0f;
    const float ndc_y = 1.0f - ((float)y + 0.5f) / (float)a.H * 2.0f;
    const float3 dir = normalize3(xa * (ndc_x * a.aspect * a.tan_half) + ya * (ndc_y * a.tan_half) - za);
    Hit h;
    bool hit = traverse(a.S, eye, dir, 0.0f, 1e30f, false, stack, h);
    float depth = 1.0f;
    float3 hitP = f3(0, 0, 0);
    if (hit) {
        hitP = eye + dir * h.t;
        const float zv = dot3(hitP - eye, za);
        depth = (a.m22 * zv + a.m32) / (-zv);
        if (!(depth >= 0.0f && depth <= 1.0f))
            hit = false;
    }
    float3 alb = f3(0, 0, 0);
    float rm0 = 0.f, rm1 = 0.f;
    float2 egn = make_float2(0.f, 0.f), esn = make_float2(0.f, 0.f);
    uint32_t ds = 0x00ffffffu;
    if (hit) {
        const float4 ids = a.S.tris[3 * h.tri + 2];
        const uint32_t geom = __float_as_uint(ids.y);
        const DevGeom g = a.S.geoms[geom];
        const float b1 = h.u, b2 = h.v, b0 = 1.0f - (b1 + b2);
        rm0 = 1.0f; // deferred_gbuffers.hlsl:91
        float3 GN = f3(0, 0, 1), SN = f3(0, 0, 1);
        if (g.valid) {
            const TriShade ts = load_tri_shade(a.S, h.tri);
            const float3 n0 = ts.n0, n1 = ts.n1, n2 = ts.n2;
            const float3 w0 = normalize3(xform_dir(g.m, n0)), w1 = normalize3(xform_dir(g.m, n1)), w2 = normalize3(xform_dir(g.m, n2));
            GN = normalize3(w0 * b0 + w1 * b1 + w2 * b2);
            SN = GN;
            const float u = ts.uv0.x * b0 + ts.uv1.x * b1 + ts.uv2.x * b2;
            const float v = ts.uv0.y * b0 + ts.uv1.y * b1 + ts.uv2.y * b2;
            if (g.material >= 0) {
                const DevMat m = a.S.mats[g.material];
                if (m.tex[0] >= 0) {
                    const float4 t = sample_texture(a.S, m.tex[0], u, v);
                    alb = f3(t.x, t.y, t.z);
                }
                if (m.tex[1] >= 0) {
                    const float3 tg0 = f3(ts.t0.x, ts.t0.y, ts.t0.z), tg1 = f3(ts.t1.x, ts.t1.y, ts.t1.z), tg2 = f3(ts.t2.x, ts.t2.y, ts.t2.z);
                    const float3 bt0 = normalize3(cross3(normalize3(n0), tg0) * ts.t0.w);
                    const float3 bt1 = normalize3(cross3(normalize3(n1), tg1) * ts.t1.w);
                    const float3 bt2 = normalize3(cross3(normalize3(n2), tg2) * ts.t2.w);
                    const float3 T = normalize3(tg0 * b0 + tg1 * b1 + tg2 * b2);
                    const float3 B = normalize3(bt0 * b0 + bt1 * b1 + bt2 * b2);
                    const float4 t = sample_texture(a.S, m.tex[1], u, v);
                    const float3 N = f3(t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f, t.z * 2.0f - 1.0f);
                    SN = normalize3(T * N.x + B * N.y + GN * N.z);
                }
                if (m.tex[2] >= 0) {
                    const float4 t = sample_texture(a.S, m.tex[2], u, v);
                    rm0 = t.y;
                    rm1 = t.z;
                }
            }
        }
        egn = oct_pack(GN);
        esn = oct_pack(SN);
        ds = (uint32_t)rint((double)depth * 16777215.0) | 0xff000000u;
    }
    a.albedo[i] = small_float_encode(alb.x, 6) | (small_float_encode(alb.y, 6) << 11) | (small_float_encode(alb.z, 5) << 22);
    a.rough_metal[i] = float_to_half_bits(rm0) | (float_to_half_bits(rm1) << 16);
    a.world_pos[i] = make_uint2(float_to_half_bits(hitP.x) | (float_to_half_bits(hitP.y) << 16), float_to_half_bits(hitP.z));
    a.normal[i] = make_uint2(float_to_half_bits(egn.x) | (float_to_half_bits(egn.y) << 16),
                             float_to_half_bits(esn.x) | (float_to_half_bits(esn.y) << 16));
    a.depth[i] = ds;
}

// The reference adds the indirect term into radiance[cur] in a separate step (nrc Resolve, DeferredRenderer.cpp:586).
// neb_gi_resolve is that step when the trace ran with "gi_defer_resolve": it lets a caller overlap the GI stages of
// frame f+1 (which touch only the G-buffer and the GI records) with the SVGF passes of frame f on another stream.
__global__ __launch_bounds__(256) void gi_resolve_kernel(float4* __restrict__ radiance, const float4* __restrict__ srec, size_t first, size_t n,
                                                         float inv_spp)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n)
        return;
    const float4 s = srec[4 * (first + k) + kSrSum];
    float4 r = radiance[first + k];
    r.x += s.x * inv_spp;
    r.y += s.y * inv_spp;
    r.z += s.z * inv_spp;
    radiance[first + k] = r;
}

// ------------------------------------------------------------------------------------------------
// "next" rows f1 (direct sun light, deferred_pbr.hlsl:39-115) and f3 (ACES tonemap, tonemapping.hlsl:3-53)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pbr_direct_kernel(GiArgs a)
{
    __shared__ int stack_mem[kLdsStack * 64];
    uint32_t x, y;
    size_t i;
    const bool active = gi_pixel(a, x, y, i);
    uint32_t rays = 0;
    if (active) {
        const float3 albedo = unpack_r11g11b10(a.albedo[i]);
        const uint2 wp = a.world_pos[i];
        const float3 worldPos = f3(half_bits_to_float(wp.x & 0xffffu), half_bits_to_float(wp.x >> 16), half_bits_to_float(wp.y & 0xffffu));
        const uint32_t nzw = a.normal[i].y;
        const float3 SN = oct_unpack(half_bits_to_float(nzw & 0xffffu), half_bits_to_float(nzw >> 16));
        const uint32_t rm = a.rough_metal[i];
        const float rough = half_bits_to_float(rm & 0xffffu), metal = half_bits_to_float(rm >> 16);
        const float3 eye = f3(a.c.cameraWorldPos[0], a.c.cameraWorldPos[1], a.c.cameraWorldPos[2]);
        const float3 sun_dir = f3(a.c.sunLightDirection[0], a.c.sunLightDirection[1], a.c.sunLightDirection[2]);
        const float3 sun_rad = f3(a.c.sunLightRadiance[0], a.c.sunLightRadiance[1], a.c.sunLightRadiance[2]);
        const float3 V = normalize3(eye - worldPos);
        const float VdotN = fminf(fmaxf(dot3(V, SN), 0.00001f), 1.0f);
        const float3 L = normalize3(-sun_dir);
        const float3 Hv = normalize3(L + V);
        const float LdotN = fminf(fmaxf(dot3(L, SN), 0.00001f), 1.0f);
        const float VdotH = fminf(fmaxf(dot3(V, Hv), 0.00001f), 1.0f);
        const float NdotH = fminf(fmaxf(dot3(SN, Hv), 0.00001f), 1.0f);
        const float3 F0 = specular_f0(albedo, metal);
        const float3 F = fresnel_schlick(F0, VdotH);
        const float3 Kd = f3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z);
        const float denom = 1.0f / (4.0f * VdotN * LdotN); // Brdf_Specular_CookTorrance, brdf.hlsli:100-111
        const float alpha = rough * rough, a2 = alpha * alpha;
        const float dd = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
        const float ndf = a2 / (kPi * dd * dd);
        const float k = alpha * 0.5f;
        const float gsf = (VdotN * (1.0f / (VdotN * (1.0f - k) + k))) * (LdotN * (1.0f / (LdotN * (1.0f - k) + k)));
        const float cs = ndf * gsf;
        const float3 O = Kd * (albedo * kPiInv) + f3(cs * F.x * denom, cs * F.y * denom, cs * F.z * denom);
        // InitRNG(tid.xy, gid.xy, frame): the shader passes the GROUP id as the resolution (:82)
        uint32_t rng = jenkins((x + y * (x / 8u)) ^ jenkins(a.c.frameIndex));
        const float a0 = rand01(rng), a1 = rand01(rng);
        const float angle = a0 * 2.0f * 3.1415926535f, dist = sqrtf(a1);
        const float3 Bv = normalize3(perpendicular(L));
        const float3 T = cross3(Bv, L);
        const float3 inc = normalize3(L + (Bv * sinf(angle) + T * cosf(angle)) * a.c.sunTanHalfAngle * dist);
        Hit h;
        rays = 1;
        const bool occluded = traverse(a.S, worldPos + SN * 1e-2f, inc, 0.0f, 3.402823466e+38f, true, stack_mem + threadIdx.x, h);
        const float vis = occluded ? 0.0f : 1.0f;
        a.radiance[i] = make_float4(O.x * LdotN * sun_rad.x * vis, O.y * LdotN * sun_rad.y * vis, O.z * LdotN * sun_rad.z * vis, 1.0f);
    }
    count_rays(a.shadow_counts, rays);
}

__global__ __launch_bounds__(256) void tonemap_kernel(const float4* __restrict__ radiance, uint32_t* __restrict__ ldr, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const float4 c = radiance[i];
    // ACESInputMat, RRTAndODTFit, ACESOutputMat (tonemapping.hlsl:3-41)
    float v[3] = {0.59719f * c.x + 0.35458f * c.y + 0.04823f * c.z, 0.07600f * c.x + 0.90834f * c.y + 0.01566f * c.z,
                  0.02840f * c.x + 0.13383f * c.y + 0.83777f * c.z};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float aa = v[r] * (v[r] + 0.0245786f) - 0.000090537f;
        const float bb = v[r] * (0.983729f * v[r] + 0.4329510f) + 0.238081f;
        v[r] = aa / bb;
    }
    const float o0 = saturate1(1.60475f * v[0] + -0.53108f * v[1] + -0.07367f * v[2]);
    const float o1 = saturate1(-0.10208f * v[0] + 1.10813f * v[1] + -0.00605f * v[2]);
    const float o2 = saturate1(-0.00327f * v[0] + -0.07276f * v[1] + 1.07602f * v[2]);
    const float luma = saturate1(o0 * 0.2126f + o1 * 0.7152f + o2 * 0.0722f);
    ldr[i] = (uint32_t)(o0 * 255.0f + 0.5f) | ((uint32_t)(o1 * 255.0f + 0.5f) << 8) | ((uint32_t)(o2 * 255.0f + 0.5f) << 16) |
             ((uint32_t)(luma * 255.0f + 0.5f) << 24);
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

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
using namespace neb;

static int gi_fail(neb_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
    else
        snprintf(buf, sizeof(buf), "%s", what);
    ctx->last_error = buf;
    return code;
}

#define GI_HIP(ctx, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess)                               \
            return gi_fail((ctx), NEB_ERR_HIP, #call, e_);  \
    } while (0)

template <typename T>
static hipError_t upload(GiState* g, const std::vector<T>& h, const T** out)
{
    *out = nullptr;
    if (h.empty())
        return hipSuccess;
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, h.size() * sizeof(T));
    if (e != hipSuccess)
        return e;
    g->allocs.push_back(d);
    *out = (const T*)d;
    return hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

extern "C" {

int neb_gi_set_scene(neb_ctx* ctx, const neb_geometry_desc* geoms, uint32_t n_geoms, const neb_material_desc* mats,
                     uint32_t n_mats, const neb_texture_desc* texs, uint32_t n_texs)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if ((n_geoms && !geoms) || (n_mats && !mats) || (n_texs && !texs))
        return gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_set_scene: null table");
    GI_HIP(ctx, hipSetDevice(ctx->device));
    GI_HIP(ctx, hipDeviceSynchronize());
    gi_destroy(ctx->gi);
    ctx->gi = nullptr;
    GiState* g = new GiState();
    std::vector<DevGeom> dgeoms(n_geoms);
    std::vector<DevMat> dmats(n_mats);
    std::vector<DevTex> dtexs(n_texs);
    std::vector<uint32_t> indices, texels;
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
    for (uint32_t i = 0; i < n_texs; ++i) {
        if (!texs[i].rgba8 || !texs[i].width || !texs[i].height) {
            delete g;
            return gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_set_scene: empty texture");
        }
        dtexs[i].offset = (uint32_t)(texels.size() / 4); // in footprint entries (uint4)
        dtexs[i].w = texs[i].width;
        dtexs[i].h = texs[i].height;
        dtexs[i].pad = 0;
        const uint32_t* px = (const uint32_t*)texs[i].rgba8;
        const uint32_t tw = texs[i].width, th = texs[i].height;
        const size_t base = texels.size();
        texels.resize(base + (size_t)tw * th * 4);
        for (uint32_t y = 0; y < th; ++y) {
            const uint32_t y1 = (y + 1) % th;
            for (uint32_t x = 0; x < tw; ++x) {
                const uint32_t x1 = (x + 1) % tw;
                uint32_t* q = &texels[base + ((size_t)y * tw + x) * 4];
                q[0] = px[(size_t)y * tw + x];
                q[1] = px[(size_t)y * tw + x1];
                q[2] = px[(size_t)y1 * tw + x];
                q[3] = px[(size_t)y1 * tw + x1];
            }
        }
    }
    // interleaved footprints of the materials whose three maps share one size (DevMat::bundle); 64 B per texel position
    std::vector<uint4> bundles;
    constexpr size_t kBundleBudget = (size_t)4 << 30; // bytes; beyond it the remaining materials sample their maps separately
    for (uint32_t i = 0; i < n_mats; ++i) {
        DevMat& d = dmats[i];
        d.bundle = d.bundle_w = d.bundle_h = d.pad = 0;
        if (d.tex[0] < 0 || d.tex[1] < 0 || d.tex[2] < 0)
            continue;
        const DevTex &ta = dtexs[d.tex[0]], &tn = dtexs[d.tex[1]], &tr = dtexs[d.tex[2]];
        if (ta.w != tn.w || ta.w != tr.w || ta.h != tn.h || ta.h != tr.h)
            continue;
        const size_t n_pos = (size_t)ta.w * ta.h;
        if ((bundles.size() + 4 * n_pos) * sizeof(uint4) > kBundleBudget || bundles.size() / 4 + n_pos > 0xffffffffull)
            continue;
        d.bundle = (uint32_t)(bundles.size() / 4);
        d.bundle_w = ta.w;
        d.bundle_h = ta.h;
        const size_t base = bundles.size();
        bundles.resize(base + 4 * n_pos);
        const uint4* fa = reinterpret_cast<const uint4*>(texels.data()) + ta.offset;
        const uint4* fn = reinterpret_cast<const uint4*>(texels.data()) + tn.offset;
        const uint4* fr = reinterpret_cast<const uint4*>(texels.data()) + tr.offset;
        for (size_t k = 0; k < n_pos; ++k) {
            bundles[base + 4 * k] = fa[k];
            bundles[base + 4 * k + 1] = fn[k];
            bundles[base + 4 * k + 2] = fr[k];
            bundles[base + 4 * k + 3] = make_uint4(0, 0, 0, 0);
        }
    }
    if (bundles.empty())
        bundles.push_back(make_uint4(0, 0, 0, 0));
    g->n_tris = (uint32_t)(g->h_tris.size() / 12);
    memcpy(g->scene_min, smin, sizeof(smin));
    memcpy(g->scene_max, smax, sizeof(smax));
    hipError_t e = hipSuccess;
    if ((e = upload(g, dgeoms, &g->view.geoms)) != hipSuccess || (e = upload(g, dmats, &g->view.mats)) != hipSuccess ||
        (e = upload(g, dtexs, &g->view.texs)) != hipSuccess || (e = upload(g, indices, &g->view.indices)) != hipSuccess ||
        (e = upload(g, normals, &g->view.normals)) != hipSuccess || (e = upload(g, uvs, &g->view.uvs)) != hipSuccess ||
        (e = upload(g, tangents, &g->view.tangents)) != hipSuccess || (e = upload(g, texels, &g->view.texels)) != hipSuccess ||
        (e = upload(g, bundles, &g->view.bundles)) != hipSuccess) {
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
    GI_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = g->n_tris;
    g->built = true;
    g->n_nodes = 0;
    g->view.root = -1;
    if (n == 0)
        return NEB_OK;
    auto dalloc = [&](size_t bytes, bool keep) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess)
            return nullptr;
        if (keep)
            g->allocs.push_back(p);
        return p;
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
                std::vector<std::pair<int, int>> work{{bin_root, 0}};
                wide.emplace_back();
                while (!work.empty()) {
                    const auto [bi, wi] = work.back();
                    work.pop_back();
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
                                work.push_back({c[k].id, ch[k]});
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
    if (e != hipSuccess)
        return gi_fail(ctx, NEB_ERR_HIP, "neb_gi_build_bvh", e);
    g->view.tris = d_sorted;
    g->view.shade = d_shade;
    g->view.nodes = d_wide;
    g->view.root = root_code;
    g->n_nodes = (uint32_t)wide.size();
    std::vector<float>().swap(g->h_tris);
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

int neb_gi_trace_rows(neb_ctx* ctx, const neb_gi_constants* c, uint32_t row0, uint32_t row1, neb_stream stream)
{
    if (!ctx || !c)
        return ctx ? gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_trace: null constants") : NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    if (!g || !g->built)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace: scene/BVH not ready (neb_gi_set_scene + neb_gi_build_bvh)");
    if (row0 < ctx->row_begin || row1 > ctx->row_end || row0 > row1)
        return gi_fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_gi_trace: rows not resident");
    if (c->samplesPerPixel == 0 || c->maxPathVertices > 8)
        return gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_trace: samplesPerPixel must be >= 1 and maxPathVertices <= 8 (MaxPathtracingRecursionDepth)");
    if (row0 == row1)
        return NEB_OK;
    const size_t npx = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    if (g->debug_hits && !g->d_hits) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, npx * sizeof(neb_gi_hit)));
        GI_HIP(ctx, hipMemset(p, 0, npx * sizeof(neb_gi_hit)));
        g->allocs.push_back(p);
        g->d_hits = (neb_gi_hit*)p;
    }
    if (!g->d_records) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, npx * sizeof(float4) * 9));
        g->allocs.push_back(p);
        g->d_records = (float4*)p;
    }
    GiArgs a;
    a.S = g->view;
    a.c = *c;
    a.R.ray_o = g->d_records;
    a.R.ray_d = g->d_records + npx;
    a.R.hit = g->d_records + 2 * npx;
    a.R.path = g->d_records + 3 * npx;
    a.R.state = g->d_records + 4 * npx;
    a.R.srec = g->d_records + 5 * npx;
    a.albedo = (const uint32_t*)ctx->planes[NEB_PLANE_ALBEDO][0];
    a.rough_metal = (const uint32_t*)ctx->planes[NEB_PLANE_ROUGH_METAL][0];
    a.world_pos = (const uint2*)ctx->planes[NEB_PLANE_WORLDPOS][0];
    a.normal = (const uint2*)ctx->planes[NEB_PLANE_NORMAL][ctx->cur];
    a.radiance = (float4*)ctx->planes[NEB_PLANE_RADIANCE][ctx->cur];
    a.hits = g->debug_hits ? g->d_hits : nullptr;
    a.ray_counter = g->d_ray_counter;
    a.W = ctx->W;
    a.row_begin = ctx->row_begin;
    a.row0 = row0;
    a.row1 = row1;
    a.tiles_x = (ctx->W + 7) / 8;
    a.stats = g->debug_hits ? 1u : 0u;
    a.sort_keys = a.sort_vals = nullptr;
    a.sort_order = nullptr;
    a.first_px = (uint32_t)((size_t)(row0 - ctx->row_begin) * ctx->W);
    a.n_px = (uint32_t)((size_t)(row1 - row0) * ctx->W);
    for (int q = 0; q < 3; ++q) {
        a.smin[q] = g->scene_min[q];
        a.sinv[q] = 1.0f / fmaxf(g->scene_max[q] - g->scene_min[q], 1e-20f);
    }
    a.bsort_keys = a.bsort_vals = nullptr;
    a.raygen_only = 0;
    if (g->sort_shadow || g->sort_bounce) {
        if (!g->d_sort) {
            void* p = nullptr;
            GI_HIP(ctx, hipMalloc(&p, 8 * npx * sizeof(uint32_t))); // {keys, vals, keys_out, vals_out} x {shadow, bounce}
            g->allocs.push_back(p);
            g->d_sort = (uint32_t*)p;
            const size_t bytes = ray_sort_scratch_bytes(npx);
            GI_HIP(ctx, hipMalloc(&p, bytes));
            g->allocs.push_back(p);
            g->d_sort_temp = p;
            g->sort_temp_bytes = bytes;
        }
        if (g->sort_shadow) {
            a.sort_keys = g->d_sort;
            a.sort_vals = g->d_sort + npx;
        }
        if (g->sort_bounce) {
            a.bsort_keys = g->d_sort + 4 * npx;
            a.bsort_vals = g->d_sort + 5 * npx;
            a.raygen_only = 1;
        }
    }
    a.defer_resolve = g->defer_resolve ? 1u : 0u;
    g->pending_spp = c->samplesPerPixel;
    g->pending_row0 = row0;
    g->pending_row1 = row1;
    const uint32_t tiles_y = (row1 - row0 + 7) / 8;
    const dim3 grid(a.tiles_x * tiles_y), block(64);
    const size_t n_blocks = (size_t)a.tiles_x * ((ctx->row_end - ctx->row_begin + 7) / 8 + 1);
    if (!g->d_block_counts) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, 2 * n_blocks * sizeof(uint32_t)));
        GI_HIP(ctx, hipMemset(p, 0, 2 * n_blocks * sizeof(uint32_t)));
        g->allocs.push_back(p);
        g->d_block_counts = (uint32_t*)p;
        g->n_block_counts = n_blocks;
    }
    a.bounce_counts = g->d_block_counts;
    a.shadow_counts = g->d_block_counts + g->n_block_counts;
    const uint32_t n_vertices = c->maxPathVertices > 1 ? c->maxPathVertices - 1 : 1; // path vertices traced per sample
    for (uint32_t s = 0; s < c->samplesPerPixel; ++s) {
        a.sample = s;
        for (uint32_t b = 1; b <= n_vertices; ++b) { // for (bounce = 1; bounce < nrcMaxPathVertices; ++bounce), :495
            a.bounce = b;
            if (b == 1)
                hipLaunchKernelGGL(gi_raygen_trace_kernel, grid, block, 0, (hipStream_t)stream, a);
            if (g->sort_bounce) {
                uint32_t* bs = g->d_sort + 4 * npx; // {keys, vals, keys_tmp, order}
                GI_HIP(ctx, ray_sort_pairs(bs + a.first_px, bs + npx + a.first_px, bs + 2 * npx + a.first_px, bs + 3 * npx + a.first_px,
                                           bs + npx + a.first_px, a.n_px, kSortBits, g->d_sort_temp, (hipStream_t)stream));
                GiArgs b1 = a;
                b1.sort_order = bs + npx + a.first_px;
                hipLaunchKernelGGL(gi_bounce_trace_kernel, dim3((a.n_px + 63) / 64), block, 0, (hipStream_t)stream, b1);
            } else if (b > 1) {
                hipLaunchKernelGGL(gi_bounce_trace_kernel, grid, block, 0, (hipStream_t)stream, a);
            }
            hipLaunchKernelGGL(gi_shade_kernel, grid, block, 0, (hipStream_t)stream, a);
            if (g->sort_shadow) {
                // {keys, vals} are the shade kernel's output and the sort's ping; {keys_tmp, vals_tmp} its pong; the
                // sorted pixel indices land back in vals
                GI_HIP(ctx, ray_sort_pairs(g->d_sort + a.first_px, g->d_sort + npx + a.first_px, g->d_sort + 2 * npx + a.first_px,
                                           g->d_sort + 3 * npx + a.first_px, g->d_sort + npx + a.first_px, a.n_px, kSortBits, g->d_sort_temp,
                                           (hipStream_t)stream));
                GiArgs b2 = a;
                b2.sort_order = g->d_sort + npx + a.first_px;
                hipLaunchKernelGGL(gi_shadow_trace_kernel, dim3((a.n_px + 63) / 64), block, 0, (hipStream_t)stream, b2);
            } else {
                hipLaunchKernelGGL(gi_shadow_trace_kernel, grid, block, 0, (hipStream_t)stream, a);
            }
        }
    }
    GI_HIP(ctx, hipGetLastError());
    return NEB_OK;
}

int neb_gi_resolve(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    if (!g || !g->d_records || !g->defer_resolve)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_resolve: nothing pending (set option gi_defer_resolve=1 and call neb_gi_trace first)");
    if (g->pending_row1 <= g->pending_row0)
        return NEB_OK;
    const size_t npx = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    const size_t first = (size_t)(g->pending_row0 - ctx->row_begin) * ctx->W, n = (size_t)(g->pending_row1 - g->pending_row0) * ctx->W;
    hipLaunchKernelGGL(gi_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float4*)ctx->planes[NEB_PLANE_RADIANCE][ctx->cur], g->d_records + 5 * npx, first, n, 1.0f / (float)g->pending_spp);
    GI_HIP(ctx, hipGetLastError());
    return NEB_OK;
}

int neb_pbr_direct(neb_ctx* ctx, const neb_gi_constants* c, neb_stream stream)
{
    if (!ctx || !c)
        return ctx ? gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_pbr_direct: null constants") : NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    if (!g || !g->built)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_pbr_direct: scene/BVH not ready (neb_gi_set_scene + neb_gi_build_bvh)");
    GiArgs a{};
    a.S = g->view;
    a.c = *c;
    a.albedo = (const uint32_t*)ctx->planes[NEB_PLANE_ALBEDO][0];
    a.rough_metal = (const uint32_t*)ctx->planes[NEB_PLANE_ROUGH_METAL][0];
    a.world_pos = (const uint2*)ctx->planes[NEB_PLANE_WORLDPOS][0];
    a.normal = (const uint2*)ctx->planes[NEB_PLANE_NORMAL][ctx->cur];
    a.radiance = (float4*)ctx->planes[NEB_PLANE_RADIANCE][ctx->cur];
    a.W = ctx->W;
    a.row_begin = ctx->row_begin;
    a.row0 = ctx->row_begin;
    a.row1 = ctx->row_end;
    a.tiles_x = (ctx->W + 7) / 8; // Dispatch((W+7)/8, (H+7)/8): DeferredRenderer.cpp:382
    const uint32_t tiles_y = (a.row1 - a.row0 + 7) / 8;
    const size_t n_blocks = (size_t)a.tiles_x * ((ctx->row_end - ctx->row_begin + 7) / 8 + 1);
    if (!g->d_block_counts) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, 2 * n_blocks * sizeof(uint32_t)));
        GI_HIP(ctx, hipMemset(p, 0, 2 * n_blocks * sizeof(uint32_t)));
        g->allocs.push_back(p);
        g->d_block_counts = (uint32_t*)p;
        g->n_block_counts = n_blocks;
    }
    a.bounce_counts = g->d_block_counts;
    a.shadow_counts = g->d_block_counts + g->n_block_counts;
    hipLaunchKernelGGL(pbr_direct_kernel, dim3(a.tiles_x * tiles_y), dim3(64), 0, (hipStream_t)stream, a);
    GI_HIP(ctx, hipGetLastError());
    return NEB_OK;
}

int neb_tonemap(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    const size_t n = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)ctx->planes[NEB_PLANE_RADIANCE][ctx->cur], (uint32_t*)ctx->planes[NEB_PLANE_LDR][0], n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return gi_fail(ctx, NEB_ERR_HIP, "neb_tonemap", e);
    return NEB_OK;
}

int neb_gi_trace(neb_ctx* ctx, const neb_gi_constants* c, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    return neb_gi_trace_rows(ctx, c, ctx->row_begin, ctx->row_end, stream);
}

int neb_gi_ray_count(neb_ctx* ctx, uint64_t* rays, int reset, neb_stream stream)
{
    if (!ctx || !ctx->gi)
        return ctx ? gi_fail(ctx, NEB_ERR_STATE, "neb_gi_ray_count: no scene") : NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    unsigned long long v[8] = {};
    std::vector<uint32_t> counts(2 * g->n_block_counts);
    GI_HIP(ctx, hipMemcpyAsync(v, g->d_ray_counter, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (g->d_block_counts)
        GI_HIP(ctx, hipMemcpyAsync(counts.data(), g->d_block_counts, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   (hipStream_t)stream));
    GI_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    if (reset) {
        GI_HIP(ctx, hipMemsetAsync(g->d_ray_counter, 0, sizeof(v), (hipStream_t)stream));
        if (g->d_block_counts)
            GI_HIP(ctx, hipMemsetAsync(g->d_block_counts, 0, counts.size() * sizeof(uint32_t), (hipStream_t)stream));
    }
    unsigned long long total = 0;
    for (uint32_t c : counts)
        total += c;
    v[0] = total;
    if (rays)
        *rays = total;
    memcpy(g->last_stats, v, sizeof(v));
    return NEB_OK;
}

int neb_gi_traversal_stats(neb_ctx* ctx, uint64_t out[5])
{
    if (!ctx || !ctx->gi || !out)
        return NEB_ERR_INVALID_ARG;
    for (int k = 0; k < 5; ++k)
        out[k] = ctx->gi->last_stats[k];
    return NEB_OK;
}

int neb_gi_download_hits(neb_ctx* ctx, neb_gi_hit* host, neb_stream stream)
{
    if (!ctx || !host)
        return NEB_ERR_INVALID_ARG;
    if (!ctx->gi || !ctx->gi->d_hits)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_download_hits: set option gi_debug_hits=1 and trace first");
    const size_t npx = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    GI_HIP(ctx, hipMemcpyAsync(host, ctx->gi->d_hits, npx * sizeof(neb_gi_hit), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GI_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return NEB_OK;
}

int neb_gbuffer_raycast(neb_ctx* ctx, const neb_camera* cam, neb_stream stream)
{
    if (!ctx || !cam)
        return NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    if (!g || !g->built)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gbuffer_raycast: scene/BVH not ready");
    auto norm = [](float* v) {
        const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] /= l;
        v[1] /= l;
        v[2] /= l;
    };
    GbufArgs a;
    a.S = g->view;
    float z[3] = {cam->eye[0] - cam->target[0], cam->eye[1] - cam->target[1], cam->eye[2] - cam->target[2]};
    norm(z);
    float x[3] = {cam->up[1] * z[2] - cam->up[2] * z[1], cam->up[2] * z[0] - cam->up[0] * z[2], cam->up[0] * z[1] - cam->up[1] * z[0]};
    norm(x);
    const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    memcpy(a.eye, cam->eye, 12);
    memcpy(a.xaxis, x, 12);
    memcpy(a.yaxis, y, 12);
    memcpy(a.zaxis, z, 12);
    a.tan_half = tanf(cam->vfov_deg * (3.14159265f / 180.0f) * 0.5f);
    a.aspect = (float)ctx->W / (float)ctx->H;
    a.m22 = cam->zfar / (cam->znear - cam->zfar); // XMMatrixPerspectiveFovRH
    a.m32 = cam->znear * cam->zfar / (cam->znear - cam->zfar);
    a.albedo = (uint32_t*)ctx->planes[NEB_PLANE_ALBEDO][0];
    a.rough_metal = (uint32_t*)ctx->planes[NEB_PLANE_ROUGH_METAL][0];
    a.world_pos = (uint2*)ctx->planes[NEB_PLANE_WORLDPOS][0];
    a.normal = (uint2*)ctx->planes[NEB_PLANE_NORMAL][ctx->cur];
    a.depth = (uint32_t*)ctx->planes[NEB_PLANE_DEPTH][ctx->cur];
    a.W = ctx->W;
    a.H = ctx->H;
    a.row_begin = ctx->row_begin;
    a.row0 = ctx->row_begin;
    a.row1 = ctx->row_end;
    a.tiles_x = (ctx->W + 7) / 8;
    const uint32_t tiles_y = (a.row1 - a.row0 + 7) / 8;
    hipLaunchKernelGGL(gbuffer_kernel, dim3(a.tiles_x * tiles_y), dim3(64), 0, (hipStream_t)stream, a);
    GI_HIP(ctx, hipGetLastError());
    return NEB_OK;
}

} // extern "C"

namespace neb {
int gi_set_sort_rays(neb_ctx* ctx, int mask)
{
    if (!ctx->gi || mask < 0 || mask > 3)
        return NEB_ERR_STATE;
    ctx->gi->sort_shadow = (mask & 1) != 0;
    ctx->gi->sort_bounce = (mask & 2) != 0;
    return NEB_OK;
}
int gi_set_defer_resolve(neb_ctx* ctx, int on)
{
    if (!ctx->gi)
        return NEB_ERR_STATE;
    ctx->gi->defer_resolve = on != 0;
    return NEB_OK;
}
int gi_set_debug_hits(neb_ctx* ctx, int on)
{
    if (!ctx->gi)
        return NEB_ERR_STATE;
    ctx->gi->debug_hits = on != 0;
    return NEB_OK;
}
} // namespace neb

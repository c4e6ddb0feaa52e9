// gi_device.h -- device-side arithmetic of the GI path: vector helpers, the RNG and BRDF of the reference's shaders,
// G-buffer encodings, the BVH4 traverser, texture footprints and surface reconstruction.  Included by gi.hip (wavefront
// kernels) and gi_build.hip (record packing).
//
// Floating-point contraction is OFF from here on so that ray setup and the Moeller-Trumbore test round exactly like
// the scalar CPU oracle (hit/miss decisions at triangle edges then agree).
#pragma once
#pragma clang fp contract(off)

#include "gi_internal.h"

namespace neb {

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
// sin / cos of an angle in [0, 2 pi] by a FIXED sequence of IEEE operations (rintf, exact products inside fmaf, two polynomials on
// [-pi/4, pi/4], quadrant selection): HLSL's sin / cos are implementation-defined hardware approximations (pathtracer.hlsl's
// bounce direction, brdf.hlsli:166-185, and sun-disk sample, :533-557), so a restatement has to pick an algorithm -- and when the
// CPU oracle and the device pick THE SAME one, every bounce ray is the same bits on both sides and the only closest hits that can
// still differ are exact ties.  (With libm's sinf on one side and ocml's on the other the directions differed in the last ulp, and at
// 3840 x 2160 some twenty rays per frame grazed an edge differently: hit against miss, one of them 1.5e-3 of a frame's L2 norm.)
// Accuracy ~1 ulp; coefficients: the Cephes single-precision minimax polynomials.  The same text lives in oracle/trace_ref.cpp and
// nebulae_amd/csrc/gi_device.h.
__device__ __forceinline__ void det_sincosf(float x, float& s, float& c)
{
    const float k = rintf(x * 0.636619772367581343f);                         // nearest multiple of pi / 2
    float r = fmaf(-k, 1.57079637050628662109375f, x);                        // x - k * pi/2, the product exact inside the fma
    r = fmaf(-k, -4.37113900018624283e-8f, r);                                // ... the low part of pi / 2
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    s = (q == 0) ? ps : (q == 1) ? pc : (q == 2) ? -ps : -pc;
    c = (q == 0) ? pc : (q == 1) ? -ps : (q == 2) ? -pc : ps;
}
// pow(x, 5) as x^2 * x^2 * x under both policies (the oracle's form too: libm's powf and ocml's differ in the last ulp)
template <bool FAST> __device__ __forceinline__ float fpow5(float x)
{
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
template <bool FAST = false> __device__ __forceinline__ float3 oct_unpack(float ex, float ey)
{
    float3 v = f3(ex, ey, 1.0f - fabsf(ex) - fabsf(ey));
    if (v.z < 0.0f) {
        const float sx = (v.x > 0.f) ? 1.f : -1.f, sy = (v.y > 0.f) ? 1.f : -1.f;
        const float nx = (1.0f - fabsf(v.y)) * sx, ny = (1.0f - fabsf(v.x)) * sy;
        v.x = nx;
        v.y = ny;
    }
    return normalize3<FAST>(v);
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
template <bool FAST = false, bool FAST_TRIG = false> __device__ __forceinline__ float3 cosine_hemisphere_aligned(float u0, float u1, float3 sn) // brdf.hlsli:166-185
{
    const float a = fsqrt<FAST>(u0), b = kPiTwo * u1;
    float sb, cb;
    if (FAST_TRIG) {
        sb = __sinf(b), cb = __cosf(b);
    } else {
        det_sincosf(b, sb, cb);
    }
    const float3 z = f3(a * cb, a * sb, fsqrt<FAST>(1.0f - u0));
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
constexpr int kSpillStack = 64 - kLdsStack; // deeper entries go to a private (scratch) array; rarely touched (TravStackT)
static_assert((kLdsStack + kSpillStack) / 3 == 21, "GiState::max_bvh_depth assumes a 64-entry traversal stack");

struct Hit {
    float t, u, v;
    uint32_t tri;
    uint32_t node_visits, tri_tests; // traversal statistics (neb_gi_traversal_stats)
    // wave stamps (STATS builds of the loop only; the same value in every lane that is still walking): iterations of the loop,
    // iterations that ran a node phase / a leaf phase, and the lanes that were live in them (neb_gi_wave_stats)
    uint32_t w_iters, w_node_iters, w_node_lanes, w_leaf_iters, w_leaf_lanes;
    // where in the (breadth-first) node array the visits fall (STATS builds): visits of nodes with an index below 64 / 256 / 1024 / 4096,
    // and node phases that ended with more than 12 entries on the stack (neb_gi_node_index_stats)
    uint32_t top_visits[4], deep_sp;
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

// Entry distance of the ray into a child box of a BVH4 node, as an ordered uint key (misses = 0xffffffff), from the
// distances to the planes the ray meets first (a*) and last (b*) on each axis.  fminf / fmaxf drop NaNs (inf - inf for
// axis-parallel rays), which only makes the interval more conservative; hits themselves are decided by the triangle test.
// t0 >= tmin >= 0: its bit pattern orders like an unsigned integer; the low 2 bits carry the slot.
__device__ __forceinline__ uint32_t slab_key_t(float ax, float ay, float az, float bx, float by, float bz, float tmin, float tmax, uint32_t slot)
{
    const float t0 = fmaxf(fmaxf(ax, ay), fmaxf(az, tmin));
    const float t1 = fminf(fminf(bx, by), fminf(bz, tmax));
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
typedef __attribute__((address_space(3))) int LdsInt; // an LDS word, typed: a pop through a generic pointer became a flat_load (round 5)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i32x4 LdsI4;
// N = entries of the ray's stack kept in LDS (lane-contiguous columns of an array [N][64]); the deeper ones, 64 - N, live in a private array
template <int N>
struct TravStackT {
    LdsInt* lds; // this lane's column of an LDS array [N][64]
    int* spill;  // private array of 64 - N entries
    int sp;
    __device__ __forceinline__ void push(int v)
    {
        if (sp < N)
            lds[64 * sp] = v;
        else if (sp < 64)
            spill[sp - N] = v;
        else
            return; // unreachable: neb_gi_build_bvh refuses trees with 3 * depth > 64 (GiState::max_bvh_depth)
        sp++;
    }
    __device__ __forceinline__ int pop()
    {
        sp--;
        // the LDS word is read whatever the depth (ds_read_b32, no branch in front of it); the rare deep entry replaces it
        int v = lds[64 * (sp < N ? sp : 0)];
        if (sp >= N)
            v = spill[sp - N];
        return v;
    }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    __device__ __forceinline__ int depth() const { return sp; }
};
// The same stack with the ADDRESS of its next free LDS entry as its only state (round 5, the closest-hit pass): a push is compare +
// ds_write + add, a pop add + compare + ds_read -- no index-to-address arithmetic, no clamp.  `array`: the wave's LDS array [N][64]
// (wave-uniform: a scalar operand of the compares); entry k of lane l at array + 64 k + l.  Entries N .. 63 live in the private array,
// addressed by the depth the address implies.
template <int N>
struct TravStackA {
    LdsInt* top;   // next free entry of this lane's column (keeps counting past the LDS part: an address that is never dereferenced)
    LdsInt* array; // wave-uniform
    int* spill;
    __device__ __forceinline__ TravStackA(LdsInt* column, LdsInt* array_, int* spill_) : top(column), array(array_), spill(spill_) {}
    __device__ __forceinline__ int depth() const { return (int)((uint32_t)(top - array) >> 6); }
    __device__ __forceinline__ bool empty() const { return top < array + 64; }
    __device__ __forceinline__ void push(int v)
    {
        if (top < array + 64 * N)
            *top = v;
        else if (top < array + 64 * 64)
            spill[depth() - N] = v;
        else
            return; // unreachable, see TravStackT
        top += 64;
    }
    __device__ __forceinline__ int pop()
    {
        top -= 64;
        int v;
        if (top < array + 64 * N)
            v = *top;
        else
            v = spill[depth() - N];
        return v;
    }
};

// Closest-hit (ANY_HIT = false) or first-hit (ANY_HIT = true) traversal of the BVH4.
// A step handles an inner node and then, if the lane lands on a leaf, the leaf in the same iteration
// (if-if), so lanes in the node phase and lanes in the leaf phase of a wave do not serialise two memory
// round trips per iteration.  Shadow rays skip the front-to-back ordering of the children.
constexpr int kTravDone = (int)0x80000000;

// The traversal loop proper: starts from (node, st, hit, found) and runs until the ray is done.  Every ray walks the 64-byte
// quantised nodes (Bvh4NodeQ, gi_internal.h): four 16-byte loads per visit.
// LDS_SELECT (closest-hit walks): child_slots = 16 bytes of LDS per lane; the node's four child codes are parked there and the (up to) four that
// the ordered keys name are read back by slot -- `v_and, v_lshl_add, ds_read_b32` per child instead of the seven VALU instructions of a
// two-level v_cndmask selection (round 5).
// Measured and moved out of the product (git tag r05-traversal-arms holds them all, docs/NOTEBOOK.md 10.1 their numbers): the 128-byte
// exact-plane walk and tail suspension (round 2); "pop-ahead" (the next node requested before the leaf's triangle tests: one memory wait
// per iteration), the top of the tree in LDS, branch-free pushes, the four child codes read up front, multi-wave workgroups (round 5).
template <bool ANY_HIT, bool STATS, bool LDS_SELECT, class Stack>
__device__ __forceinline__ void traverse_core(const SceneView& S, float3 o, float3 d, float tmin, Stack& st, int& node, Hit& hit, bool& found,
                                              LdsInt* child_slots = nullptr)
{
    const float3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float3 oinv = f3(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    const bool negx = d.x < 0.0f, negy = d.y < 0.0f, negz = d.z < 0.0f;
    constexpr uint32_t kMiss = 0xffffffffu;
    while (node != kTravDone) {
        if (STATS) {
            const uint32_t nn = (uint32_t)__popcll(__ballot(node >= 0));
            hit.w_iters++;
            hit.w_node_iters += nn ? 1u : 0u;
            hit.w_node_lanes += nn;
        }
        if (node >= 0) {
            if (STATS) {
                hit.node_visits++;
                hit.top_visits[0] += node < 64, hit.top_visits[1] += node < 256, hit.top_visits[2] += node < 1024, hit.top_visits[3] += node < 4096;
            }
            uint32_t k0, k1, k2, k3;
            int4 ch;
            {
                const char* nodes = reinterpret_cast<const char*>(S.qnodes);
                const uint32_t nb = (uint32_t)node << 6;
                const float4 p0 = *reinterpret_cast<const float4*>(nodes + nb);         // {origin.xyz, scale.x}
                const float4 p1 = *reinterpret_cast<const float4*>(nodes + (nb + 16u)); // {scale.yz, qlo.x, qlo.y}
                const uint4 p2 = *reinterpret_cast<const uint4*>(nodes + (nb + 32u));   // {qlo.z, qhi.xyz}
                ch = *reinterpret_cast<const int4*>(nodes + (nb + 48u));
                // plane distance = (origin + q scale - o) / d = q (scale inv) + (origin inv - o inv)
                const float sx = p0.w * inv.x, sy = p1.x * inv.y, sz = p1.y * inv.z;
                const float bx = fmaf(p0.x, inv.x, -oinv.x), by = fmaf(p0.y, inv.y, -oinv.y), bz = fmaf(p0.z, inv.z, -oinv.z);
                const uint32_t lox = __float_as_uint(p1.z), loy = __float_as_uint(p1.w);
                const uint32_t qnx = negx ? p2.y : lox, qfx = negx ? lox : p2.y; // entry plane by the sign of the direction
                const uint32_t qny = negy ? p2.z : loy, qfy = negy ? loy : p2.z;
                const uint32_t qnz = negz ? p2.w : p2.x, qfz = negz ? p2.x : p2.w;
                auto key = [&](uint32_t c) { // (v_cvt_f32_ubyte<c>: one instruction per plane)
                    auto un = [&](uint32_t q) { return (float)((q >> (8u * c)) & 0xffu); };
                    return slab_key_t(fmaf(un(qnx), sx, bx), fmaf(un(qny), sy, by), fmaf(un(qnz), sz, bz), fmaf(un(qfx), sx, bx),
                                      fmaf(un(qfy), sy, by), fmaf(un(qfz), sz, bz), tmin, hit.t, c);
                };
                k0 = key(0u), k1 = key(1u), k2 = key(2u), k3 = key(3u);
            }
            if constexpr (!ANY_HIT && LDS_SELECT)
                *reinterpret_cast<LdsI4*>(child_slots) = i32x4{ch.x, ch.y, ch.z, ch.w};
            // select by the slot bits without branches: from the lane's LDS slots, or by two levels of v_cndmask
            auto child_of = [&](uint32_t key) -> int {
                if constexpr (LDS_SELECT)
                    return child_slots[key & 3u];
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
            if (STATS)
                hit.deep_sp += st.depth() > 12;
            if (node == kTravDone && !st.empty())
                node = st.pop();
        }
        // Any-hit rays batch their leaf steps: a lane that holds a leaf waits until kLeafBatch lanes of the wave do (or
        // none has a node left), so the triangle code runs with fuller waves (shadow pass 213 -> 206 us; the closest-hit
        // pass, whose lanes need the shrunk hit.t at once, measured no gain at 4 / 12 and lost at 24).
        const bool holds_leaf = node < 0 && node != kTravDone;
        bool run_leaves = true;
        if (ANY_HIT && kLeafBatch > 1)
            run_leaves = __popcll(__ballot(holds_leaf)) >= kLeafBatch || __ballot(node >= 0) == 0ull;
        if (STATS) {
            const uint32_t nl = (uint32_t)__popcll(__ballot(holds_leaf && run_leaves));
            hit.w_leaf_iters += nl ? 1u : 0u;
            hit.w_leaf_lanes += nl;
        }
        if (holds_leaf && run_leaves) {
            const uint32_t code = (uint32_t)~node;
            const uint32_t first = code >> 2, count = (code & 3u) + 1u;
            if (STATS)
                hit.tri_tests += count;
            // (closest-hit walks do not keep `found` through the loop -- a lane mask in scalar registers, re-merged every iteration: it is
            // hit.tri != ~0u at the end)
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
                    hit.t = t, hit.u = u, hit.v = v, hit.tri = first;
                    if constexpr (ANY_HIT)
                        found = true;
                }
                if (intersect_tri_regs(a1, b1, c1, o, d, tmin, hit.t, t, u, v)) {
                    hit.t = t, hit.u = u, hit.v = v, hit.tri = second;
                    if constexpr (ANY_HIT)
                        found = true;
                }
            } else {
                for (uint32_t k = 0; k < count; ++k) {
                    float t, u, v;
                    if (intersect_tri(S.tris, first + k, o, d, tmin, hit.t, t, u, v)) {
                        hit.t = t, hit.u = u, hit.v = v, hit.tri = first + k;
                        if constexpr (ANY_HIT)
                            found = true;
                    }
                }
            }
            if (ANY_HIT && found) {
                node = kTravDone;
                return;
            }
            node = st.empty() ? kTravDone : st.pop();
        }
    }
    if constexpr (!ANY_HIT)
        found = hit.tri != ~0u;
}

// lds_stack: this lane's column of the wave's LDS array [N][64].  FAST (the closest-hit pass of the frame, round 5): the stack's state is the
// LDS address of its next free entry (TravStackA; lds_array = the wave's array) and the child codes go through child_slots (16 bytes of LDS
// per lane, see traverse_core).
template <bool ANY_HIT, bool STATS, int N = kLdsStack, bool FAST = false>
__device__ bool traverse_t(const SceneView& S, float3 o, float3 d, float tmin, float tmax, int* lds_stack, Hit& hit, int* lds_array = nullptr,
                           int* child_slots = nullptr)
{
    hit.t = tmax;
    hit.tri = ~0u;
    hit.node_visits = hit.tri_tests = 0;
    hit.w_iters = hit.w_node_iters = hit.w_node_lanes = hit.w_leaf_iters = hit.w_leaf_lanes = 0;
    hit.top_visits[0] = hit.top_visits[1] = hit.top_visits[2] = hit.top_visits[3] = hit.deep_sp = 0;
    if (S.n_tris == 0)
        return false;
    bool found = false;
    int spill_mem[64 - N];
    // A ray whose direction is zero or whose origin / direction is not a number (a G-buffer normal of 0 makes one) would pass every
    // slab test -- min / max drop the NaNs -- and walk the WHOLE tree, 0.1 s per wave on a 262 k-triangle scene, without ever
    // hitting a triangle (det == 0 or NaN).  It hits nothing: it does not start.
    const float dd = dot3(d, d), oo = dot3(o, o);
    const bool walkable = dd > 0.0f && dd < __builtin_inff() && oo < __builtin_inff();
    int node = walkable ? S.root : kTravDone;
    if constexpr (FAST) {
        TravStackA<N> st((LdsInt*)lds_stack, (LdsInt*)lds_array, spill_mem);
        traverse_core<ANY_HIT, STATS, !ANY_HIT>(S, o, d, tmin, st, node, hit, found, (LdsInt*)child_slots);
    } else {
        TravStackT<N> st{(LdsInt*)lds_stack, spill_mem, 0};
        traverse_core<ANY_HIT, STATS, false>(S, o, d, tmin, st, node, hit, found);
    }
    return found;
}

// Any-hit walk of ONE ray by FOUR lanes (a quad: lanes 4 k .. 4 k + 3), for the latency-bound tail (gi_shadow_list_kernel: the
// few shadow rays the sun table leaves have the machine to themselves, and a lone wave is bound by the length of its dependent
// instruction chain -- ~250 instructions per step of the one-lane-per-ray loop).  Lane q of the quad decodes and tests child q of
// the BVH4 node (a quarter of the arithmetic; the node's 64 bytes are read by all four lanes: one line), the four verdicts meet in
// a ballot, every lane of the quad keeps the same stack pointer and reads and writes the same LDS column; in a leaf
// lanes 0 and 1 test its (up to) two triangles.  Same nodes, same box arithmetic (slab_key_t on the same plane distances), same
// triangle test on the same operands, same visiting order as traverse_core<true, .>: the answer is the traversal's.
// `stack`: this quad's LDS column -- entry k at stack[16 * k] -- 64 entries deep (the builder bounds the tree: 3 x 21 pending nodes).
template <bool STATS>
__device__ __forceinline__ bool traverse_any_quad(const SceneView& S, float3 o, float3 d, float tmin, float tmax, int* stack, uint32_t& node_visits,
                                                  uint32_t& tri_tests)
{
    node_visits = tri_tests = 0;
    if (S.n_tris == 0)
        return false;
    const float dd = dot3(d, d), oo = dot3(o, o);
    if (!(dd > 0.0f && dd < __builtin_inff() && oo < __builtin_inff())) // (see traverse_t: such a ray does not start)
        return false;
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u, qbase = lane & 60u;
    const float3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float3 oinv = f3(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    const bool negx = d.x < 0.0f, negy = d.y < 0.0f, negz = d.z < 0.0f;
    constexpr uint32_t kMiss = 0xffffffffu;
    int node = S.root, sp = 0;
    bool found = false;
    auto push = [&](int v) { // (all four lanes store the same word: each lane's later read is ordered behind its OWN store, no cross-lane fence needed)
        stack[16 * sp] = v;
        ++sp;
    };
    auto pop = [&]() -> int {
        --sp;
        return stack[16 * sp];
    };
    while (node != kTravDone) {
        if (node >= 0) {
            if (STATS)
                node_visits++;
            const char* nodes = reinterpret_cast<const char*>(S.qnodes);
            const uint32_t nb = (uint32_t)node << 6;
            const float4 p0 = *reinterpret_cast<const float4*>(nodes + nb);
            const float4 p1 = *reinterpret_cast<const float4*>(nodes + (nb + 16u));
            const uint4 p2 = *reinterpret_cast<const uint4*>(nodes + (nb + 32u));
            const int4 ch = *reinterpret_cast<const int4*>(nodes + (nb + 48u));
            const float sx = p0.w * inv.x, sy = p1.x * inv.y, sz = p1.y * inv.z;
            const float bx = fmaf(p0.x, inv.x, -oinv.x), by = fmaf(p0.y, inv.y, -oinv.y), bz = fmaf(p0.z, inv.z, -oinv.z);
            const uint32_t lox = __float_as_uint(p1.z), loy = __float_as_uint(p1.w);
            const uint32_t qnx = negx ? p2.y : lox, qfx = negx ? lox : p2.y;
            const uint32_t qny = negy ? p2.z : loy, qfy = negy ? loy : p2.z;
            const uint32_t qnz = negz ? p2.w : p2.x, qfz = negz ? p2.x : p2.w;
            auto un = [&](uint32_t w) { return (float)((w >> (8u * q)) & 0xffu); };
            const uint32_t key = slab_key_t(fmaf(un(qnx), sx, bx), fmaf(un(qny), sy, by), fmaf(un(qnz), sz, bz), fmaf(un(qfx), sx, bx),
                                            fmaf(un(qfy), sy, by), fmaf(un(qfz), sz, bz), tmin, tmax, q);
            const uint32_t m = (uint32_t)(__ballot(key != kMiss) >> qbase) & 0xfu; // bit c: child c of this quad's node is hit
            // the order of traverse_core's any-hit branch: go on with the first hit child in slot order, stack the others
            node = kTravDone;
            if (m & 8u)
                node = ch.w;
            if (m & 4u) {
                if (node != kTravDone)
                    push(node);
                node = ch.z;
            }
            if (m & 2u) {
                if (node != kTravDone)
                    push(node);
                node = ch.y;
            }
            if (m & 1u) {
                if (node != kTravDone)
                    push(node);
                node = ch.x;
            }
            if (node == kTravDone && sp)
                node = pop();
        }
        if (node < 0 && node != kTravDone) {
            const uint32_t code = (uint32_t)~node;
            const uint32_t first = code >> 2, count = (code & 3u) + 1u;
            if (STATS)
                tri_tests += count;
            // lanes 0 .. count - 1 take one triangle each (count <= kMaxLeafTris <= 4); the others repeat the last one
            const uint32_t ti = first + min(q, count - 1u);
            float t, u, v;
            const bool h = intersect_tri(S.tris, ti, o, d, tmin, tmax, t, u, v);
            if ((uint32_t)(__ballot(h) >> qbase) & 0xfu) {
                found = true;
                node = kTravDone;
            } else {
                node = sp ? pop() : kTravDone;
            }
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

// The (up to) three filtered map fetches of a material at one uv.  A material whose three maps share one size keeps
// their bilinear footprints interleaved and trimmed to the channels the shaders read, one 32-byte entry per texel
// position (DevMat::bundle): one 32-byte fetch per hit instead of three 16-byte ones in three places.  Maps a material does not have are left untouched (the caller uses the factors).
struct MapSamples {
    float4 albedo = {0.f, 0.f, 0.f, 0.f}, normal = {0.f, 0.f, 0.f, 0.f}, rm = {0.f, 0.f, 0.f, 0.f};
};
template <bool FAST = false> __device__ __forceinline__ void sample_material_maps(const SceneView& S, const DevMat& m, float u, float v, MapSamples& out)
{
    if (m.bundle_w != 0) { // then all three maps exist
        int x0, y0;
        float fx, fy;
        texel_position(m.bundle_w, m.bundle_h, u, v, x0, y0, fx, fy);
        // one 32-byte entry: the four texels of the footprint, 8 bytes each {albedo.rgb, normal.rgb, roughness (rm.g), metalness (rm.b)}
        // -- the eight channels the shaders read (alpha and rm.r are never sampled); same UNORM8 values, same filter arithmetic
        const uint4* e = S.bundles + 2 * ((size_t)m.bundle + (size_t)y0 * m.bundle_w + x0);
        const uint4 top = e[0], bot = e[1]; // {t00.lo, t00.hi, t10.lo, t10.hi}, {t01.lo, t01.hi, t11.lo, t11.hi}
        float ch[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            auto un = [&](uint32_t lo, uint32_t hi) {
                const float q = (float)(((c < 4 ? lo : hi) >> (8 * (c & 3))) & 0xffu);
                return FAST ? q * (1.0f / 255.0f) : q / 255.0f;
            };
            const float a = un(top.x, top.y), b = un(top.z, top.w), cc = un(bot.x, bot.y), dd = un(bot.z, bot.w);
            const float tp = a + fx * (b - a), bt = cc + fx * (dd - cc);
            ch[c] = tp + fy * (bt - tp);
        }
        out.albedo = make_float4(ch[0], ch[1], ch[2], 0.0f);
        out.normal = make_float4(ch[3], ch[4], ch[5], 0.0f);
        out.rm = make_float4(0.0f, ch[6], ch[7], 0.0f);
        return;
    }
    if (m.tex[0] >= 0)
        out.albedo = sample_texture<FAST>(S, m.tex[0], u, v);
    if (m.tex[1] >= 0)
        out.normal = sample_texture<FAST>(S, m.tex[1], u, v);
    if (m.tex[2] >= 0)
        out.rm = sample_texture<FAST>(S, m.tex[2], u, v);
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
    uint32_t geom; // GeometryIndex() of the triangle (r6.w, low 30 bits); PrimitiveIndex() is r7.x (debug records only)
    uint32_t lit;  // sun-visibility table (r6.w, top 2 bits): bit 0 = shadow rays from the +GN side are all unoccluded, bit 1 = -GN side
    uint32_t hint[kHints], hint_side; // occluder hints (r7.yzw): triangles likely to shadow rays that start on side hint_side (0: +GN, 1: -GN)
};
// `r` = the triangle's 8 x float4 record; piece k sits at r[k ^ swz] (swz = 0 in global memory; the copy the shade pass
// stages in LDS is XOR-swizzled per lane to spread the banks)
template <typename Ptr> __device__ __forceinline__ TriShade load_tri_shade_at(Ptr r, uint32_t swz)
{
    const float4 r0 = r[0 ^ swz], r1 = r[1 ^ swz], r2 = r[2 ^ swz], r6 = r[6 ^ swz];
    TriShade t;
    t.n0 = f3(r0.x, r0.y, r0.z);
    t.n1 = f3(r1.x, r1.y, r1.z);
    t.n2 = f3(r2.x, r2.y, r2.z);
    t.uv0 = make_float2(r0.w, r1.w);
    t.uv1 = make_float2(r2.w, r6.x);
    t.uv2 = make_float2(r6.y, r6.z);
    t.geom = __float_as_uint(r6.w) & kGeomMask;
    t.lit = __float_as_uint(r6.w) >> kLitShift;
    t.t0 = r[3 ^ swz];
    t.t1 = r[4 ^ swz];
    t.t2 = r[5 ^ swz];
    const float4 r7 = r[7 ^ swz];
    unpack_hints(r7, t.hint, t.hint_side);
    return t;
}
__device__ __forceinline__ TriShade load_tri_shade(const SceneView& S, uint32_t tri) { return load_tri_shade_at(S.shade + 8 * (size_t)tri, 0u); }

// ReconstructSurfaceData (pathtracer.hlsl:299-395); `tri` is the sorted triangle index of the hit.
template <bool FAST = false> __device__ __forceinline__ bool reconstruct_surface(const SceneView& S, const TriShade& ts, float bu, float bv, Surface& out, uint32_t& geom)
{
    // the record names its geometry itself: one gathered line per hit, and the geometry / material table reads hang off it
    // instead of off a second gather into the triangle array
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
    MapSamples maps;
    sample_material_maps<FAST>(S, m, u, v, maps);
    if (m.tex[0] < 0) {
        out.albedo = f3(m.albedo[0], m.albedo[1], m.albedo[2]);
    } else {
        const float4 t = maps.albedo;
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
        const float4 t = maps.normal;
        const float3 N = f3(t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f, t.z * 2.0f - 1.0f);
        out.SN = normalize3<FAST>(T * N.x + B * N.y + out.GN * N.z); // mul(N, float3x3(T, B, GN))
    }
    if (m.tex[2] < 0) {
        out.roughness = m.rough;
        out.metalness = m.metal;
    } else {
        const float4 t = maps.rm;
        out.roughness = t.y; // .g
        out.metalness = t.z; // .b
    }
    return true;
}

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

} // namespace neb

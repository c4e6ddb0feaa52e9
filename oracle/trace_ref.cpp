/*
 * oracle/trace_ref.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see trace_ref.h).
 * Scalar C++ path tracer following assets/shaders/pathtracer.hlsl with NRC stubbed, over its own
 * binned-SAH BVH (the product builds a different tree -- an on-device LBVH; true closest hits and
 * any-hit occlusion do not depend on the tree).
 */
#include "trace_ref.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svgf_ref.h"

namespace {

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a)
{
    float l = length(a);
    return {a.x / l, a.y / l, a.z / l};
}
inline float saturate(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float h2f(uint16_t h) { return svgf_ref_f16_to_f32(h); }

constexpr float PI = 3.14159265f; // brdf.hlsli:28
constexpr float PI_INV = 1.0f / PI;
constexpr float PI_TWO = 2.0f * PI;
constexpr float TRACING_MAX_DISTANCE = 10000.0f; // pathtracer.hlsl:9

// ---- rand.hlsli:6-55 ----
inline uint32_t jenkins(uint32_t x)
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}
inline uint32_t init_rng(uint32_t px, uint32_t py, uint32_t res_x, uint32_t frame)
{
    uint32_t s = (px * 1u + py * res_x) ^ jenkins(frame); // dot(pixel, uint2(1, resolution.x))
    return jenkins(s);
}
inline float rand01(uint32_t& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    uint32_t b = 0x3f800000u | (s >> 9);
    float f;
    memcpy(&f, &b, 4);
    return f - 1.f;
}

// ---- octahedron_encoding.hlsli:16-34 ----
inline V3 oct_unpack(float ex, float ey)
{
    V3 v = {ex, ey, 1.0f - fabsf(ex) - fabsf(ey)};
    if (v.z < 0.0f) {
        float sx = (v.x > 0.f) ? 1.f : -1.f, sy = (v.y > 0.f) ? 1.f : -1.f;
        float nx = (1.0f - fabsf(v.y)) * sx, ny = (1.0f - fabsf(v.x)) * sy;
        v.x = nx;
        v.y = ny;
    }
    return normalize(v);
}
inline void oct_pack(V3 v, float* e)
{
    float s = 1.0f / (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
    float px = v.x * s, py = v.y * s;
    if (v.z <= 0.0f) {
        float sx = (px > 0.f) ? 1.f : -1.f, sy = (py > 0.f) ? 1.f : -1.f;
        e[0] = (1.0f - fabsf(py)) * sx;
        e[1] = (1.0f - fabsf(px)) * sy;
    } else {
        e[0] = px;
        e[1] = py;
    }
}

// ---- R11G11B10_FLOAT (unsigned small floats: 5-bit exponent bias 15, 6/6/5-bit mantissa) ----
inline float small_float_decode(uint32_t bits, int mbits)
{
    uint32_t e = bits >> mbits, m = bits & ((1u << mbits) - 1u);
    float scale = (float)(1u << mbits);
    if (e == 0)
        return ldexpf((float)m / scale, -14);
    if (e == 31)
        return m ? NAN : INFINITY;
    return ldexpf(1.0f + (float)m / scale, (int)e - 15);
}
inline uint32_t small_float_encode(float f, int mbits)
{
    // Our definition for the (hardware-side, unpinned) RTV conversion: negatives and NaN -> 0,
    // round-to-nearest-even, overflow -> largest finite value.
    if (!(f > 0.0f))
        return 0;
    const uint32_t max_bits = (30u << mbits) | ((1u << mbits) - 1u);
    int e;
    float m = frexpf(f, &e); // f = m * 2^e, m in [0.5,1)
    e -= 1;                  // f = (2m) * 2^e with 2m in [1,2)
    if (e > 15)
        return max_bits;
    if (e < -14) { // denormal: units of 2^(-14 - mbits)
        float q = nearbyintf(ldexpf(f, 14 + mbits));
        return (uint32_t)q; // may carry into the smallest normal: same bit pattern arithmetic
    }
    float q = nearbyintf(ldexpf(2.0f * m - 1.0f, mbits)); // mantissa in [0, 2^mbits]
    uint32_t bits = ((uint32_t)(e + 15) << mbits) + (uint32_t)q;
    return bits > max_bits ? max_bits : bits;
}


// sin / cos of an angle in [0, 2 pi] by a FIXED sequence of IEEE operations (rintf, exact products inside fmaf, two polynomials on
// [-pi/4, pi/4], quadrant selection): HLSL's sin / cos are implementation-defined hardware approximations (pathtracer.hlsl's
// bounce direction, brdf.hlsli:166-185, and sun-disk sample, :533-557), so a restatement has to pick an algorithm -- and when the
// CPU oracle and the device pick THE SAME one, every bounce ray is the same bits on both sides and the only closest hits that can
// still differ are exact ties.  (With libm's sinf on one side and ocml's on the other the directions differed in the last ulp, and at
// 3840 x 2160 some twenty rays per frame grazed an edge differently: hit against miss, one of them 1.5e-3 of a frame's L2 norm.)
// Accuracy ~1 ulp; coefficients: the Cephes single-precision minimax polynomials.  The same text lives in oracle/trace_ref.cpp and
// nebulae_amd/csrc/gi_device.h.
inline void det_sincosf(float x, float& s, float& c)
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

// ---- brdf.hlsli ----
inline float luminance(V3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }
inline V3 specular_f0(V3 albedo, float metal)
{
    return {lerp(0.04f, albedo.x, metal), lerp(0.04f, albedo.y, metal), lerp(0.04f, albedo.z, metal)};
}
inline V3 diffuse_reflectance(V3 albedo, float metal) { return albedo * (1.0f - metal); }
inline V3 fresnel_schlick(V3 f0, float vdoth) // brdf.hlsli:22-25 (as written: 1 - VdotH^5)
{
    // pow(x, 5) as the product x^2 * x^2 * x (the same on the device): libm's powf and ocml's differ in the last ulp, which moved the
    // diffuse probability and -- once in ten million pixels -- the outcome of Rand < pd
    const float v2 = vdoth * vdoth;
    float k = 1.0f - v2 * v2 * vdoth;
    return {f0.x + (1.0f - f0.x) * k, f0.y + (1.0f - f0.y) * k, f0.z + (1.0f - f0.z) * k};
}
inline float specular_probability(float vdotn, V3 f0, V3 albedo) // brdf.hlsli:129-143
{
    float dr = luminance(albedo);
    float fres = saturate(luminance(fresnel_schlick(f0, saturate(vdotn))));
    float diff = dr * (1.0f - fres);
    float p = diff / fmaxf(0.0001f, fres + diff);
    return p < 0.1f ? 0.1f : (p > 0.9f ? 0.9f : p);
}
inline V3 cosine_hemisphere_aligned(float u0, float u1, V3 sn) // brdf.hlsli:166-185
{
    float a = sqrtf(u0), b = PI_TWO * u1, sb, cb;
    det_sincosf(b, sb, cb);
    V3 z = {a * cb, a * sb, sqrtf(1.0f - u0)};
    V3 up = fabsf(sn.z) < 0.999f ? v3(0, 0, 1) : v3(1, 0, 0);
    V3 tx = normalize(cross(up, sn));
    V3 ty = cross(sn, tx);
    return normalize(tx * z.x + ty * z.y + sn * z.z);
}
inline V3 perpendicular(V3 u) // sun_disk_sampling.hlsli:45-52
{
    V3 a = {fabsf(u.x), fabsf(u.y), fabsf(u.z)};
    uint32_t xm = ((a.x - a.y) < 0 && (a.x - a.z) < 0) ? 1 : 0;
    uint32_t ym = (a.y - a.z) < 0 ? (1 ^ xm) : 0;
    uint32_t zm = 1 ^ (xm | ym);
    return cross(u, v3((float)xm, (float)ym, (float)zm));
}

struct Surface {
    V3 GN, SN, albedo;
    float roughness, metalness;
};

// EvaluateDirectBRDF (pathtracer.hlsl:209-228) + brdf.hlsli:35-111
V3 evaluate_direct_brdf(const Surface& s, V3 V, V3 L)
{
    V3 N = s.SN;
    V3 Hv = normalize(V + L);
    float LdotN = dot(L, N), VdotH = saturate(dot(V, Hv)), VdotN = dot(V, N), NdotH = dot(N, Hv);
    V3 F0 = specular_f0(s.albedo, s.metalness);
    V3 F = fresnel_schlick(F0, saturate(VdotH));
    V3 Kd = {1.0f - F.x, 1.0f - F.y, 1.0f - F.z};
    V3 diff = Kd * (s.albedo * PI_INV);
    float vn = saturate(VdotN), ln = saturate(LdotN), nh = saturate(NdotH);
    V3 spec = {0, 0, 0};
    float den = 4.0f * vn * ln;
    if (den > 0.0f) { // reference: 1/0 * 0 = NaN, discarded there by NRC; defined as 0 here (trace_ref.h)
        float alpha = s.roughness * s.roughness;
        float a2 = alpha * alpha;
        float dd = (nh * nh) * (a2 - 1.0f) + 1.0f;
        float ndf = a2 / (PI * dd * dd);
        float k = alpha * 0.5f;
        float gv = vn * (1.0f / (vn * (1.0f - k) + k));
        float gl = ln * (1.0f / (ln * (1.0f - k) + k));
        float c = ndf * (gv * gl);
        float inv = 1.0f / den;
        spec = {c * F.x * inv, c * F.y * inv, c * F.z * inv};
    }
    return diff + spec;
}

// ---- scene ----
struct Tri {
    V3 v0, e1, e2;
    uint32_t geom, prim;
};
struct Node {
    float bmin[3], bmax[3];
    uint32_t left, count; // count > 0: leaf, `left` = first triangle
};
struct Geometry {
    trace_ref_geometry g;
};

} // namespace

struct trace_ref_scene {
    std::vector<Geometry> geoms;
    std::vector<trace_ref_material> mats;
    std::vector<trace_ref_texture> texs;
    std::vector<Tri> tris;
    std::vector<Node> nodes;
};

namespace {

inline uint32_t read_index(const trace_ref_geometry& g, uint32_t i)
{
    const uint8_t* p = (const uint8_t*)g.indices + (size_t)i * g.indexStride;
    if (g.indexStride == 2) {
        uint16_t v;
        memcpy(&v, p, 2);
        return v;
    }
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
inline void read_attr(const trace_ref_geometry& g, int a, uint32_t vtx, float* out, int n)
{
    memcpy(out, (const uint8_t*)g.attributes[a] + (size_t)vtx * g.attributeStrides[a], sizeof(float) * n);
}
inline V3 xform_point(const float* m, V3 p) // (p,1) * M, row-vector convention
{
    return {p.x * m[0] + p.y * m[4] + p.z * m[8] + m[12], p.x * m[1] + p.y * m[5] + p.z * m[9] + m[13],
            p.x * m[2] + p.y * m[6] + p.z * m[10] + m[14]};
}
inline V3 xform_dir(const float* m, V3 p) // (p,0) * M
{
    return {p.x * m[0] + p.y * m[4] + p.z * m[8], p.x * m[1] + p.y * m[5] + p.z * m[9],
            p.x * m[2] + p.y * m[6] + p.z * m[10]};
}

struct Builder {
    std::vector<Node>& nodes;
    std::vector<V3> bmin, bmax, cen;
    std::vector<uint32_t> idx;

    void bounds(uint32_t a, uint32_t b, float* mn, float* mx, float* cmn, float* cmx)
    {
        for (int k = 0; k < 3; ++k) {
            mn[k] = cmn[k] = 3.4e38f;
            mx[k] = cmx[k] = -3.4e38f;
        }
        for (uint32_t i = a; i < b; ++i) {
            const float* lo = &bmin[idx[i]].x;
            const float* hi = &bmax[idx[i]].x;
            const float* c = &cen[idx[i]].x;
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], lo[k]);
                mx[k] = std::max(mx[k], hi[k]);
                cmn[k] = std::min(cmn[k], c[k]);
                cmx[k] = std::max(cmx[k], c[k]);
            }
        }
    }
    static float area(const float* mn, const float* mx)
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
    uint32_t build(uint32_t a, uint32_t b)
    {
        uint32_t ni = (uint32_t)nodes.size();
        nodes.push_back(Node());
        float mn[3], mx[3], cmn[3], cmx[3];
        bounds(a, b, mn, mx, cmn, cmx);
        memcpy(nodes[ni].bmin, mn, 12);
        memcpy(nodes[ni].bmax, mx, 12);
        uint32_t n = b - a;
        int axis = 0;
        float ext = cmx[0] - cmn[0];
        for (int k = 1; k < 3; ++k)
            if (cmx[k] - cmn[k] > ext) {
                ext = cmx[k] - cmn[k];
                axis = k;
            }
        if (n <= 4 || ext <= 0.f) {
            if (n > 8 && ext <= 0.f) { // coincident centroids: split in half
                uint32_t mid = a + n / 2;
                uint32_t l = build(a, mid), r = build(mid, b);
                nodes[ni].left = l;
                nodes[ni].count = 0;
                (void)r;
                return ni;
            }
            nodes[ni].left = a;
            nodes[ni].count = n;
            return ni;
        }
        // binned SAH, 16 bins
        const int NB = 16;
        float bmn[NB][3], bmx[NB][3];
        uint32_t cnt[NB] = {};
        for (int i = 0; i < NB; ++i)
            for (int k = 0; k < 3; ++k) {
                bmn[i][k] = 3.4e38f;
                bmx[i][k] = -3.4e38f;
            }
        float scale = NB / ext;
        auto bin_of = [&](uint32_t t) {
            int bi = (int)(((&cen[t].x)[axis] - cmn[axis]) * scale);
            return bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
        };
        for (uint32_t i = a; i < b; ++i) {
            int bi = bin_of(idx[i]);
            cnt[bi]++;
            for (int k = 0; k < 3; ++k) {
                bmn[bi][k] = std::min(bmn[bi][k], (&bmin[idx[i]].x)[k]);
                bmx[bi][k] = std::max(bmx[bi][k], (&bmax[idx[i]].x)[k]);
            }
        }
        float best = 3.4e38f;
        int best_split = -1;
        float lmn[3], lmx[3], rarea[NB];
        uint32_t rc[NB];
        float rmn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, rmx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        uint32_t c = 0;
        for (int i = NB - 1; i > 0; --i) {
            for (int k = 0; k < 3; ++k) {
                rmn[k] = std::min(rmn[k], bmn[i][k]);
                rmx[k] = std::max(rmx[k], bmx[i][k]);
            }
            c += cnt[i];
            rc[i] = c;
            rarea[i] = c ? area(rmn, rmx) : 0.f;
        }
        for (int k = 0; k < 3; ++k) {
            lmn[k] = 3.4e38f;
            lmx[k] = -3.4e38f;
        }
        c = 0;
        for (int i = 0; i < NB - 1; ++i) {
            for (int k = 0; k < 3; ++k) {
                lmn[k] = std::min(lmn[k], bmn[i][k]);
                lmx[k] = std::max(lmx[k], bmx[i][k]);
            }
            c += cnt[i];
            if (c == 0 || rc[i + 1] == 0)
                continue;
            float cost = area(lmn, lmx) * c + rarea[i + 1] * rc[i + 1];
            if (cost < best) {
                best = cost;
                best_split = i;
            }
        }
        uint32_t mid;
        if (best_split < 0) {
            mid = a + n / 2;
        } else {
            mid = (uint32_t)(std::partition(idx.begin() + a, idx.begin() + b, [&](uint32_t t) { return bin_of(t) <= best_split; }) -
                             idx.begin());
            if (mid == a || mid == b)
                mid = a + n / 2;
        }
        uint32_t l = build(a, mid);
        build(mid, b); // right child is always l's subtree end; store explicitly below
        nodes[ni].left = l;
        nodes[ni].count = 0;
        return ni;
    }
};

// Children of an inner node: left = nodes[n].left, right = the node created right after the whole
// left subtree (pre-order); right_children() below tabulates it.

struct Hit {
    float t, u, v;
    uint32_t tri;
};

inline bool intersect_tri(const Tri& tr, V3 o, V3 d, float tmin, float tmax, float& t, float& u, float& v)
{
    // Moeller-Trumbore; the device kernel uses the same operation order.
    V3 p = cross(d, tr.e2);
    float det = dot(tr.e1, p);
    if (det == 0.0f)
        return false;
    float inv = 1.0f / det;
    V3 tv = o - tr.v0;
    u = dot(tv, p) * inv;
    if (u < 0.0f || u > 1.0f)
        return false;
    V3 q = cross(tv, tr.e1);
    v = dot(d, q) * inv;
    if (v < 0.0f || u + v > 1.0f)
        return false;
    t = dot(tr.e2, q) * inv;
    return t > tmin && t < tmax;
}

inline bool slab(const Node& n, V3 o, V3 inv, float tmin, float tmax, float& tnear)
{
    float t0 = tmin, t1 = tmax;
    const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
    for (int k = 0; k < 3; ++k) {
        float a = (n.bmin[k] - oo[k]) * ii[k], b = (n.bmax[k] - oo[k]) * ii[k];
        if (a > b)
            std::swap(a, b);
        // NaN (0 * inf) compares false and leaves the interval unchanged: conservative
        if (a > t0)
            t0 = a;
        if (b < t1)
            t1 = b;
    }
    tnear = t0;
    return t0 <= t1;
}

} // namespace

// diagnostics: per-thread tallies, folded into the totals when a parallel region ends (a shared counter bumped from the
// traversal loop made 16 threads barely faster than one)
static unsigned long long g_node_visits = 0, g_tri_tests = 0;
static thread_local unsigned long long t_node_visits = 0, t_tri_tests = 0;
static void fold_traversal_stats()
{
#pragma omp atomic
    g_node_visits += t_node_visits;
#pragma omp atomic
    g_tri_tests += t_tri_tests;
    t_node_visits = t_tri_tests = 0;
}
extern "C" void trace_ref_stats(unsigned long long* nodes, unsigned long long* tris, int reset)
{
    *nodes = g_node_visits;
    *tris = g_tri_tests;
    if (reset)
        g_node_visits = g_tri_tests = 0;
}

static bool trace(const trace_ref_scene* s, const std::vector<uint32_t>& right, V3 o, V3 d, float tmin, float tmax,
                  bool any_hit, Hit& hit)
{
    hit.t = tmax;
    hit.tri = ~0u;
    if (s->nodes.empty())
        return false;
    V3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    bool found = false;
    while (sp) {
        uint32_t ni = stack[--sp];
        const Node& n = s->nodes[ni];
        float tn;
        if (!slab(n, o, inv, tmin, hit.t, tn))
            continue;
        t_node_visits++;
        if (n.count) {
            t_tri_tests += n.count;
            for (uint32_t i = 0; i < n.count; ++i) {
                float t, u, v;
                if (intersect_tri(s->tris[n.left + i], o, d, tmin, hit.t, t, u, v)) {
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    hit.tri = n.left + i;
                    found = true;
                    if (any_hit)
                        return true;
                }
            }
        } else {
            uint32_t l = n.left, r = right[ni];
            float tl, tr;
            bool hl = slab(s->nodes[l], o, inv, tmin, hit.t, tl), hr = slab(s->nodes[r], o, inv, tmin, hit.t, tr);
            if (hl && hr) {
                if (tl <= tr) {
                    stack[sp++] = r;
                    stack[sp++] = l;
                } else {
                    stack[sp++] = l;
                    stack[sp++] = r;
                }
            } else if (hl) {
                stack[sp++] = l;
            } else if (hr) {
                stack[sp++] = r;
            }
        }
    }
    return found;
}

// SampleLevel(linear, wrap, mip 0) of an RGBA8 UNORM texture (pathtracer.hlsl:359,377,390).
static void sample_texture(const trace_ref_texture& t, float u, float v, float* rgba)
{
    float x = u * (float)t.width - 0.5f, y = v * (float)t.height - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int W = (int)t.width, H = (int)t.height;
    int x0 = (int)fx0 % W, y0 = (int)fy0 % H;
    if (x0 < 0)
        x0 += W;
    if (y0 < 0)
        y0 += H;
    int x1 = (x0 + 1) % W, y1 = (y0 + 1) % H;
    const uint8_t* p00 = t.rgba8 + 4 * ((size_t)y0 * W + x0);
    const uint8_t* p10 = t.rgba8 + 4 * ((size_t)y0 * W + x1);
    const uint8_t* p01 = t.rgba8 + 4 * ((size_t)y1 * W + x0);
    const uint8_t* p11 = t.rgba8 + 4 * ((size_t)y1 * W + x1);
    for (int c = 0; c < 4; ++c) {
        float a = (float)p00[c] / 255.0f, b = (float)p10[c] / 255.0f, cc = (float)p01[c] / 255.0f, d = (float)p11[c] / 255.0f;
        float top = a + fx * (b - a), bot = cc + fx * (d - cc);
        rgba[c] = top + fy * (bot - top);
    }
}

// ReconstructSurfaceData (pathtracer.hlsl:299-395)
static bool reconstruct_surface(const trace_ref_scene* s, uint32_t prim, uint32_t geom_index, float bu, float bv, Surface& out)
{
    const Geometry& G = s->geoms[geom_index];
    const trace_ref_geometry& g = G.g;
    float b0 = 1.0f - (bu + bv), b1 = bu, b2 = bv;
    if (!g.indices || !g.attributes[0] || !g.attributes[1] || !g.attributes[2] || !g.attributes[3])
        return false;
    uint32_t i0 = read_index(g, prim * 3 + 0), i1 = read_index(g, prim * 3 + 1), i2 = read_index(g, prim * 3 + 2);
    float n0[3], n1[3], n2[3];
    read_attr(g, 1, i0, n0, 3);
    read_attr(g, 1, i1, n1, 3);
    read_attr(g, 1, i2, n2, 3);
    V3 gn = normalize(v3(n0[0] * b0 + n1[0] * b1 + n2[0] * b2, n0[1] * b0 + n1[1] * b1 + n2[1] * b2,
                         n0[2] * b0 + n1[2] * b1 + n2[2] * b2));
    out.GN = normalize(xform_dir(g.surfaceToWorld, gn)); // :340
    float t0[2], t1[2], t2[2];
    read_attr(g, 2, i0, t0, 2);
    read_attr(g, 2, i1, t1, 2);
    read_attr(g, 2, i2, t2, 2);
    float u = t0[0] * b0 + t1[0] * b1 + t2[0] * b2, v = t0[1] * b0 + t1[1] * b1 + t2[1] * b2;
    if (g.materialIndex < 0)
        return false; // :349
    const trace_ref_material& m = s->mats[g.materialIndex];
    float tx[4];
    if (m.textureIndices[0] < 0) {
        out.albedo = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
    } else {
        sample_texture(s->texs[m.textureIndices[0]], u, v, tx);
        out.albedo = v3(tx[0], tx[1], tx[2]);
    }
    if (m.textureIndices[1] < 0) {
        out.SN = out.GN;
    } else {
        float a0[4], a1[4], a2[4], tg[4];
        read_attr(g, 3, i0, a0, 4);
        read_attr(g, 3, i1, a1, 4);
        read_attr(g, 3, i2, a2, 4);
        for (int k = 0; k < 4; ++k)
            tg[k] = a0[k] * b0 + a1[k] * b1 + a2[k] * b2;
        float l4 = sqrtf(tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2] + tg[3] * tg[3]); // normalize(float4) :371
        for (int k = 0; k < 4; ++k)
            tg[k] /= l4;
        V3 T = v3(tg[0], tg[1], tg[2]);
        V3 B = normalize(cross(out.GN, T) * tg[3]);
        sample_texture(s->texs[m.textureIndices[1]], u, v, tx);
        V3 N = v3(tx[0] * 2.0f - 1.0f, tx[1] * 2.0f - 1.0f, tx[2] * 2.0f - 1.0f);
        out.SN = normalize(T * N.x + B * N.y + out.GN * N.z); // mul(N, float3x3(T, B, GN))
    }
    if (m.textureIndices[2] < 0) {
        out.roughness = m.roughnessMetalness[0];
        out.metalness = m.roughnessMetalness[1];
    } else {
        sample_texture(s->texs[m.textureIndices[2]], u, v, tx);
        out.roughness = tx[1]; // .g
        out.metalness = tx[2]; // .b
    }
    return true;
}

extern "C" {

uint32_t trace_ref_pack_r11g11b10(const float* rgb)
{
    return small_float_encode(rgb[0], 6) | (small_float_encode(rgb[1], 6) << 11) | (small_float_encode(rgb[2], 5) << 22);
}

void trace_ref_unpack_r11g11b10(uint32_t v, float* rgb)
{
    rgb[0] = small_float_decode(v & 0x7ffu, 6);
    rgb[1] = small_float_decode((v >> 11) & 0x7ffu, 6);
    rgb[2] = small_float_decode((v >> 22) & 0x3ffu, 5);
}

trace_ref_scene* trace_ref_scene_create(const trace_ref_geometry* geoms, uint32_t n_geoms, const trace_ref_material* mats,
                                        uint32_t n_mats, const trace_ref_texture* texs, uint32_t n_texs)
{
    trace_ref_scene* s = new trace_ref_scene();
    s->mats.assign(mats, mats + n_mats);
    s->texs.assign(texs, texs + n_texs);
    for (uint32_t gi = 0; gi < n_geoms; ++gi) {
        Geometry G;
        G.g = geoms[gi];
        s->geoms.push_back(G);
        const trace_ref_geometry& g = geoms[gi];
        if (!g.indices || !g.attributes[0])
            continue;
        for (uint32_t p = 0; p + 2 < g.numIndices; p += 3) {
            float a[3], b[3], c[3];
            read_attr(g, 0, read_index(g, p), a, 3);
            read_attr(g, 0, read_index(g, p + 1), b, 3);
            read_attr(g, 0, read_index(g, p + 2), c, 3);
            V3 w0 = xform_point(g.surfaceToWorld, v3(a[0], a[1], a[2]));
            V3 w1 = xform_point(g.surfaceToWorld, v3(b[0], b[1], b[2]));
            V3 w2 = xform_point(g.surfaceToWorld, v3(c[0], c[1], c[2]));
            Tri t;
            t.v0 = w0;
            t.e1 = w1 - w0;
            t.e2 = w2 - w0;
            t.geom = gi;
            t.prim = p / 3;
            s->tris.push_back(t);
        }
    }
    // BVH
    std::vector<Tri> sorted;
    if (!s->tris.empty()) {
        Builder B{s->nodes, {}, {}, {}, {}};
        size_t n = s->tris.size();
        B.bmin.resize(n);
        B.bmax.resize(n);
        B.cen.resize(n);
        B.idx.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const Tri& t = s->tris[i];
            V3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
            B.bmin[i] = v3(std::min(a.x, std::min(b.x, c.x)), std::min(a.y, std::min(b.y, c.y)), std::min(a.z, std::min(b.z, c.z)));
            B.bmax[i] = v3(std::max(a.x, std::max(b.x, c.x)), std::max(a.y, std::max(b.y, c.y)), std::max(a.z, std::max(b.z, c.z)));
            B.cen[i] = (B.bmin[i] + B.bmax[i]) * 0.5f;
            // The boxes are PADDED (1e-5 of the largest coordinate, at least 1e-6): the slab test below is plain float arithmetic, an
            // axis-aligned triangle has a box of zero thickness, and a ray that meets such a triangle on its rim could be turned away by
            // the box although the triangle test would accept it -- seen once in ~10 M rays as a hit the device reports (its boxes are
            // rounded outwards) and this tracer missed.  The acceleration structure must never decide a hit: only the triangle test does.
            {
                const float m = std::max(std::max(std::max(fabsf(B.bmin[i].x), fabsf(B.bmax[i].x)), std::max(fabsf(B.bmin[i].y), fabsf(B.bmax[i].y))),
                                         std::max(fabsf(B.bmin[i].z), fabsf(B.bmax[i].z)));
                const float pad = std::max(1e-5f * m, 1e-6f);
                B.bmin[i] = B.bmin[i] - v3(pad, pad, pad);
                B.bmax[i] = B.bmax[i] + v3(pad, pad, pad);
            }
            B.idx[i] = (uint32_t)i;
        }
        s->nodes.reserve(2 * n);
        B.build(0, (uint32_t)n);
        sorted.resize(n);
        for (size_t i = 0; i < n; ++i)
            sorted[i] = s->tris[B.idx[i]];
        s->tris.swap(sorted);
    }
    return s;
}

void trace_ref_scene_destroy(trace_ref_scene* s) { delete s; }
uint32_t trace_ref_scene_triangles(const trace_ref_scene* s) { return (uint32_t)s->tris.size(); }

} // extern "C"

// right-child table: the right child of inner node i is the first node after i's left subtree.
static std::vector<uint32_t> right_children(const trace_ref_scene* s)
{
    std::vector<uint32_t> right(s->nodes.size(), 0);
    // subtree end via recursion-free pass: node order is pre-order, so compute sizes backwards
    std::vector<uint32_t> size(s->nodes.size(), 1);
    for (size_t i = s->nodes.size(); i-- > 0;) {
        const Node& n = s->nodes[i];
        if (!n.count) {
            uint32_t l = n.left;
            uint32_t r = l + size[l];
            right[i] = r;
            size[i] = 1 + size[l] + size[r];
        }
    }
    return right;
}

extern "C" uint64_t trace_ref_gi(const trace_ref_scene* s, uint32_t W, uint32_t H, uint32_t row0, uint32_t row1,
                                 const trace_ref_constants* c, const uint32_t* albedo_p, const uint16_t* rough_metal,
                                 const uint16_t* world_pos, const uint16_t* normal, float* radiance, trace_ref_hit* hits,
                                 int threads)
{
    const std::vector<uint32_t> right = right_children(s);
    uint64_t rays = 0;
    const V3 cam = v3(c->cameraWorldPos[0], c->cameraWorldPos[1], c->cameraWorldPos[2]);
    const V3 sky = v3(c->skyColor[0], c->skyColor[1], c->skyColor[2]);
    const V3 sun_dir = v3(c->sunLightDirection[0], c->sunLightDirection[1], c->sunLightDirection[2]);
    const V3 sun_rad = v3(c->sunLightRadiance[0], c->sunLightRadiance[1], c->sunLightRadiance[2]);
    if (row1 > H)
        row1 = H;
#pragma omp parallel num_threads(threads) reduction(+ : rays)
    {
#pragma omp for schedule(dynamic, 4) nowait
    for (int yy = (int)row0; yy < (int)row1; ++yy) {
        for (uint32_t x = 0; x < W; ++x) {
            const uint32_t y = (uint32_t)yy;
            const size_t i = (size_t)y * W + x;
            uint32_t rng = init_rng(x, y, W, c->frameIndex); // pathtracer.hlsl:402
            float alb[3];
            trace_ref_unpack_r11g11b10(albedo_p[i], alb);
            const V3 albedo = v3(alb[0], alb[1], alb[2]);
            const V3 worldPos = v3(h2f(world_pos[4 * i]), h2f(world_pos[4 * i + 1]), h2f(world_pos[4 * i + 2]));
            const V3 SN = oct_unpack(h2f(normal[4 * i + 2]), h2f(normal[4 * i + 3]));
            const float metalness = h2f(rough_metal[2 * i + 1]);
            V3 V = cam - worldPos; // :431 -- NOT reset per sample (it is overwritten at :522)
            V3 sum = {0, 0, 0};
            trace_ref_hit dbg = {-1.0f, ~0u, ~0u, 0};
            for (uint32_t sidx = 0; sidx < c->samplesPerPixel; ++sidx) {
                (void)rand01(rng); // consumed by NrcCreatePathState (:438)
                V3 throughput = {1, 1, 1};
                V3 rad = {0, 0, 0};
                const V3 F0 = specular_f0(albedo, metalness);
                throughput = throughput * diffuse_reflectance(albedo, metalness); // :474
                float pd = 1.0f - specular_probability(saturate(dot(normalize(V), SN)), F0, albedo);
                if (rand01(rng) < pd)
                    throughput = v3(throughput.x / pd, throughput.y / pd, throughput.z / pd); // :476-479
                float u0 = rand01(rng), u1 = rand01(rng);
                V3 dir = cosine_hemisphere_aligned(u0, u1, SN);
                V3 org = worldPos + SN * 1e-2f;
                dbg = {-1.0f, ~0u, ~0u, 0};
                float tmin = 0.01f; // primary vertex: :140; later bounces 0.001 (:609)
                for (uint32_t bounce = 1; bounce < c->maxPathVertices; ++bounce) { // :495
                    Hit h;
                    rays++;
                    if (!trace(s, right, org, dir, tmin, TRACING_MAX_DISTANCE, false, h)) {
                        rad = rad + sky * throughput; // :508
                        break;
                    }
                    const Tri& tr = s->tris[h.tri];
                    if (bounce == 1) {
                        dbg.t = h.t;
                        dbg.geometry = tr.geom;
                        dbg.primitive = tr.prim;
                    }
                    Surface surf;
                    if (!reconstruct_surface(s, tr.prim, tr.geom, h.u, h.v, surf))
                        break; // :514-518
                    V3 hitP = org + dir * h.t;
                    V = normalize(-dir); // :522
                    float a0 = rand01(rng), a1 = rand01(rng);
                    float angle = a0 * 2.0f * 3.1415926535f, dist = sqrtf(a1);
                    V3 L = normalize(-sun_dir);
                    V3 Bv = normalize(perpendicular(L));
                    V3 T = cross(Bv, L);
                    float sn_a, cs_a;
                    det_sincosf(angle, sn_a, cs_a);
                    V3 inc = normalize(L + (Bv * sn_a + T * cs_a) * c->sunTanHalfAngle * dist);
                    bool transition = dot(surf.GN, inc) <= 0.0f;
                    V3 so = hitP + (transition ? -surf.GN : surf.GN) * 1e-2f;
                    Hit sh;
                    rays++;
                    if (!trace(s, right, so, inc, 0.001f, TRACING_MAX_DISTANCE, true, sh)) {
                        V3 O = evaluate_direct_brdf(surf, V, L) * sun_rad;
                        rad = rad + O * throughput; // :574 (no N.L term, BRDF at the disk centre L)
                        if (bounce == 1)
                            dbg.flags |= 1u;
                    }
                    if (bounce == c->maxPathVertices - 1)
                        break; // :579-583
                    // EvaluateIndirectBRDF (:230-259) takes rng BY VALUE: its Rand2 draws do not advance the path's
                    // stream, so the Rand(rng) of :614 below returns the same number as the first of them.
                    uint32_t rng_copy = rng;
                    V3 SNn = normalize(surf.SN);
                    float e0 = rand01(rng_copy), e1 = rand01(rng_copy);
                    V3 Ld = cosine_hemisphere_aligned(e0, e1, SNn);
                    float pdiff = 1.0f - specular_probability(saturate(dot(V, SNn)), specular_f0(surf.albedo, surf.metalness), surf.albedo);
                    org = hitP + surf.GN * 1e-2f; // :607
                    dir = Ld;
                    tmin = 0.001f;
                    throughput = throughput * diffuse_reflectance(surf.albedo, surf.metalness); // :613
                    if (rand01(rng) < pdiff)
                        throughput = v3(throughput.x / pdiff, throughput.y / pdiff, throughput.z / pdiff); // :614-618
                }
                sum = sum + rad;
            }
            const float inv_spp = 1.0f / (float)c->samplesPerPixel;
            radiance[4 * i + 0] += sum.x * inv_spp;
            radiance[4 * i + 1] += sum.y * inv_spp;
            radiance[4 * i + 2] += sum.z * inv_spp;
            if (hits)
                hits[i] = dbg;
        }
    }
        fold_traversal_stats();
    }
    return rays;
}

extern "C" void trace_ref_gbuffer(const trace_ref_scene* s, uint32_t W, uint32_t H, const trace_ref_camera* cam,
                                  uint32_t* albedo_p, uint16_t* rough_metal, uint16_t* world_pos, uint16_t* normal,
                                  uint32_t* depth_stencil, int threads)
{
    const std::vector<uint32_t> right = right_children(s);
    const V3 eye = v3(cam->eye[0], cam->eye[1], cam->eye[2]);
    const V3 zaxis = normalize(eye - v3(cam->target[0], cam->target[1], cam->target[2])); // LookAtRH
    const V3 xaxis = normalize(cross(v3(cam->up[0], cam->up[1], cam->up[2]), zaxis));
    const V3 yaxis = cross(zaxis, xaxis);
    const float tan_half = tanf(cam->vfov_deg * (PI / 180.0f) * 0.5f);
    const float aspect = (float)W / (float)H;
    const float zn = cam->znear, zf = cam->zfar;
    const float m22 = zf / (zn - zf), m32 = zn * zf / (zn - zf); // XMMatrixPerspectiveFovRH
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(dynamic, 4) nowait
    for (int yy = 0; yy < (int)H; ++yy) {
        for (uint32_t x = 0; x < W; ++x) {
            const size_t i = (size_t)yy * W + x;
            float ndc_x = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f;
            float ndc_y = 1.0f - ((float)yy + 0.5f) / (float)H * 2.0f;
            V3 dir = normalize(xaxis * (ndc_x * aspect * tan_half) + yaxis * (ndc_y * tan_half) - zaxis);
            Hit h;
            bool hit = trace(s, right, eye, dir, 0.0f, 1e30f, false, h);
            float depth = 1.0f;
            V3 hitP = {0, 0, 0};
            if (hit) {
                hitP = eye + dir * h.t;
                float zv = dot(hitP - eye, zaxis); // view-space z (negative in front)
                depth = (m22 * zv + m32) / (-zv);
                if (!(depth >= 0.0f && depth <= 1.0f))
                    hit = false; // clipped by the near/far planes
            }
            float alb[3] = {0, 0, 0}, rm[2] = {0, 0}, en[4] = {0, 0, 0, 0};
            uint32_t ds = 0x00ffffffu; // depth 1.0, stencil 0 (DeferredRenderer.cpp:693-709)
            if (hit) {
                const Tri& tr = s->tris[h.tri];
                const trace_ref_geometry& g = s->geoms[tr.geom].g;
                float b1 = h.u, b2 = h.v, b0 = 1.0f - (b1 + b2);
                uint32_t i0 = read_index(g, tr.prim * 3), i1 = read_index(g, tr.prim * 3 + 1), i2 = read_index(g, tr.prim * 3 + 2);
                float n0[3], n1[3], n2[3];
                read_attr(g, 1, i0, n0, 3);
                read_attr(g, 1, i1, n1, 3);
                read_attr(g, 1, i2, n2, 3);
                // VS: worldNormal = normalize(mul(float4(n,0), InstanceToWorld)); PS: GN = normalize(interp)
                V3 w0 = normalize(xform_dir(g.surfaceToWorld, v3(n0[0], n0[1], n0[2])));
                V3 w1 = normalize(xform_dir(g.surfaceToWorld, v3(n1[0], n1[1], n1[2])));
                V3 w2 = normalize(xform_dir(g.surfaceToWorld, v3(n2[0], n2[1], n2[2])));
                V3 GN = normalize(w0 * b0 + w1 * b1 + w2 * b2);
                V3 SN = GN;
                float t0[2], t1[2], t2[2];
                read_attr(g, 2, i0, t0, 2);
                read_attr(g, 2, i1, t1, 2);
                read_attr(g, 2, i2, t2, 2);
                float u = t0[0] * b0 + t1[0] * b1 + t2[0] * b2, v = t0[1] * b0 + t1[1] * b1 + t2[1] * b2;
                rm[0] = 1.0f; // deferred_gbuffers.hlsl:91: factors are ignored, default (1, 0)
                rm[1] = 0.0f;
                if (g.materialIndex >= 0) {
                    const trace_ref_material& m = s->mats[g.materialIndex];
                    float tx[4];
                    if (m.textureIndices[0] >= 0) { // albedo stays 0 without a map (:74-78)
                        sample_texture(s->texs[m.textureIndices[0]], u, v, tx);
                        alb[0] = tx[0];
                        alb[1] = tx[1];
                        alb[2] = tx[2];
                    }
                    if (m.textureIndices[1] >= 0 && g.attributes[3]) {
                        float a0[4], a1[4], a2[4];
                        read_attr(g, 3, i0, a0, 4);
                        read_attr(g, 3, i1, a1, 4);
                        read_attr(g, 3, i2, a2, 4);
                        // VS: tangent = t.xyz; bitangent = normalize(cross(normalize(n), t.xyz) * t.w)
                        V3 tg0 = v3(a0[0], a0[1], a0[2]), tg1 = v3(a1[0], a1[1], a1[2]), tg2 = v3(a2[0], a2[1], a2[2]);
                        V3 bt0 = normalize(cross(normalize(v3(n0[0], n0[1], n0[2])), tg0) * a0[3]);
                        V3 bt1 = normalize(cross(normalize(v3(n1[0], n1[1], n1[2])), tg1) * a1[3]);
                        V3 bt2 = normalize(cross(normalize(v3(n2[0], n2[1], n2[2])), tg2) * a2[3]);
                        V3 T = normalize(tg0 * b0 + tg1 * b1 + tg2 * b2);
                        V3 B = normalize(bt0 * b0 + bt1 * b1 + bt2 * b2);
                        sample_texture(s->texs[m.textureIndices[1]], u, v, tx);
                        V3 N = v3(tx[0] * 2.0f - 1.0f, tx[1] * 2.0f - 1.0f, tx[2] * 2.0f - 1.0f);
                        SN = normalize(T * N.x + B * N.y + GN * N.z);
                    }
                    if (m.textureIndices[2] >= 0) {
                        sample_texture(s->texs[m.textureIndices[2]], u, v, tx);
                        rm[0] = tx[1];
                        rm[1] = tx[2];
                    }
                }
                oct_pack(GN, en);
                oct_pack(SN, en + 2);
                double q = nearbyint((double)depth * 16777215.0);
                ds = (uint32_t)q | 0xff000000u; // stencil ref 0xFF on draw (DeferredRenderer.cpp:284)
            }
            albedo_p[i] = trace_ref_pack_r11g11b10(alb);
            rough_metal[2 * i] = svgf_ref_f32_to_f16(rm[0]);
            rough_metal[2 * i + 1] = svgf_ref_f32_to_f16(rm[1]);
            world_pos[4 * i] = svgf_ref_f32_to_f16(hitP.x);
            world_pos[4 * i + 1] = svgf_ref_f32_to_f16(hitP.y);
            world_pos[4 * i + 2] = svgf_ref_f32_to_f16(hitP.z);
            world_pos[4 * i + 3] = 0;
            for (int k = 0; k < 4; ++k)
                normal[4 * i + k] = svgf_ref_f32_to_f16(en[k]);
            depth_stencil[i] = ds;
        }
    }
        fold_traversal_stats();
    }
}

// ---- deferred_pbr.hlsl:39-115 ----
extern "C" uint64_t trace_ref_pbr_direct(const trace_ref_scene* s, uint32_t W, uint32_t H, const trace_ref_constants* c,
                                         const uint32_t* albedo_p, const uint16_t* rough_metal, const uint16_t* world_pos,
                                         const uint16_t* normal, float* radiance, int threads)
{
    const std::vector<uint32_t> right = right_children(s);
    const V3 eye = v3(c->cameraWorldPos[0], c->cameraWorldPos[1], c->cameraWorldPos[2]);
    const V3 sun_dir = v3(c->sunLightDirection[0], c->sunLightDirection[1], c->sunLightDirection[2]);
    const V3 sun_rad = v3(c->sunLightRadiance[0], c->sunLightRadiance[1], c->sunLightRadiance[2]);
    uint64_t rays = 0;
    auto clampf = [](float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); };
#pragma omp parallel num_threads(threads) reduction(+ : rays)
    {
#pragma omp for schedule(dynamic, 4) nowait
    for (int yy = 0; yy < (int)H; ++yy) {
        for (uint32_t x = 0; x < W; ++x) {
            const uint32_t y = (uint32_t)yy;
            const size_t i = (size_t)y * W + x;
            float alb[3];
            trace_ref_unpack_r11g11b10(albedo_p[i], alb);
            const V3 albedo = v3(alb[0], alb[1], alb[2]);
            const V3 worldPos = v3(h2f(world_pos[4 * i]), h2f(world_pos[4 * i + 1]), h2f(world_pos[4 * i + 2]));
            const V3 SN = oct_unpack(h2f(normal[4 * i + 2]), h2f(normal[4 * i + 3]));
            const float rough = h2f(rough_metal[2 * i]), metal = h2f(rough_metal[2 * i + 1]);
            const V3 V = normalize(eye - worldPos);
            const float VdotN = clampf(dot(V, SN), 0.00001f, 1.0f);
            const V3 L = normalize(-sun_dir);
            const V3 Hv = normalize(L + V);
            const float LdotN = clampf(dot(L, SN), 0.00001f, 1.0f);
            const float VdotH = clampf(dot(V, Hv), 0.00001f, 1.0f);
            const float NdotH = clampf(dot(SN, Hv), 0.00001f, 1.0f);
            const V3 F0 = specular_f0(albedo, metal);
            const V3 F = fresnel_schlick(F0, VdotH);
            const V3 Kd = {1.0f - F.x, 1.0f - F.y, 1.0f - F.z};
            // Brdf_Specular_CookTorrance (brdf.hlsli:100-111)
            const float denom = 1.0f / (4.0f * VdotN * LdotN);
            const float alpha = rough * rough, a2 = alpha * alpha;
            const float dd = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
            const float ndf = a2 / (PI * dd * dd);
            const float k = alpha * 0.5f;
            const float gsf = (VdotN * (1.0f / (VdotN * (1.0f - k) + k))) * (LdotN * (1.0f / (LdotN * (1.0f - k) + k)));
            const float cs = ndf * gsf;
            const V3 O = Kd * (albedo * PI_INV) + v3(cs * F.x * denom, cs * F.y * denom, cs * F.z * denom);
            // InitRNG(tid.xy, gid.xy, frame): the GROUP id is passed as the resolution (:82)
            uint32_t rng = init_rng(x, y, x / 8u, c->frameIndex);
            const float a0 = rand01(rng), a1 = rand01(rng);
            const float angle = a0 * 2.0f * 3.1415926535f, dist = sqrtf(a1);
            const V3 Bv = normalize(perpendicular(L));
            const V3 T = cross(Bv, L);
            float sn_a, cs_a;
            det_sincosf(angle, sn_a, cs_a);
            const V3 inc = normalize(L + (Bv * sn_a + T * cs_a) * c->sunTanHalfAngle * dist);
            Hit h;
            rays++;
            const bool occluded = trace(s, right, worldPos + SN * 1e-2f, inc, 0.0f, 3.402823466e+38f, true, h);
            const float vis = occluded ? 0.0f : 1.0f;
            radiance[4 * i + 0] = O.x * LdotN * sun_rad.x * vis;
            radiance[4 * i + 1] = O.y * LdotN * sun_rad.y * vis;
            radiance[4 * i + 2] = O.z * LdotN * sun_rad.z * vis;
            radiance[4 * i + 3] = 1.0f;
        }
    }
        fold_traversal_stats();
    }
    return rays;
}

// ---- tonemapping.hlsl:3-53 ----
extern "C" void trace_ref_tonemap(uint32_t W, uint32_t H, const float* radiance, uint8_t* rgba8)
{
    static const float in_m[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83777f}};
    static const float out_m[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const float* c = radiance + 4 * i;
        float v[3], o[3];
        for (int r = 0; r < 3; ++r)
            v[r] = in_m[r][0] * c[0] + in_m[r][1] * c[1] + in_m[r][2] * c[2];
        for (int r = 0; r < 3; ++r) {
            const float a = v[r] * (v[r] + 0.0245786f) - 0.000090537f;
            const float b = v[r] * (0.983729f * v[r] + 0.4329510f) + 0.238081f;
            v[r] = a / b;
        }
        for (int r = 0; r < 3; ++r)
            o[r] = saturate(out_m[r][0] * v[0] + out_m[r][1] * v[1] + out_m[r][2] * v[2]);
        const float luma = o[0] * 0.2126f + o[1] * 0.7152f + o[2] * 0.0722f;
        const float q[4] = {o[0], o[1], o[2], saturate(luma)};
        for (int r = 0; r < 4; ++r)
            rgba8[4 * i + r] = (uint8_t)(q[r] * 255.0f + 0.5f); // UNORM8 round-to-nearest
    }
}

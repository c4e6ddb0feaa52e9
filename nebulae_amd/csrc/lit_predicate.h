// lit_predicate.h -- the geometric certificate behind the sun-visibility table ("gi_sun_table", gi_build.hip / gi.hip).
//
// The shadow rays of the GI path all point at the sun disk (assets/shaders/pathtracer.hlsl:533-575: origin = hitP +- GN * 1e-2,
// direction = normalize(L + (Bv sin + T cos) * sunTanHalfAngle * sqrt(u)), |offset| <= tan(half angle)).  For a triangle R of
// the scene and a side sigma in {+1, -1} this header decides, CONSERVATIVELY, whether any such ray that starts on R (offset to
// side sigma of its interpolated normal) can meet a triangle O of the scene.  If no triangle can -- R itself included -- every
// shadow ray from (R, sigma) is unoccluded and the any-hit traversal can be skipped with the same answer.
// "Cannot" must be certain; "may" costs only speed.  Everything is done in double precision in SUN COORDINATES
// (a, b, h): h = p . L grows towards the sun, (a, b) span the plane perpendicular to it.  In these coordinates a ray from o
// reaches q iff s = h_q - h_o > 0 and |q_ab - o_ab| <= s * tau: lateral drift is at most tau per unit climbed.
//
// Plain C++ (no HIP types) so that the CPU prototype / checker (tools/lit_proto.cpp) compiles the very same code.
#pragma once
#include <cmath>

#ifdef __HIPCC__
#define NEB_LIT_HD __host__ __device__ inline
#else
#define NEB_LIT_HD inline
#endif

namespace neb {
namespace lit {

constexpr double kOffset = 1e-2;      // the shadow ray leaves the surface by GN * 1e-2 (pathtracer.hlsl:560)
// Slack on heights and on lateral reach (Frame::margin): what fp32 rounding can move between the exact geometry this header reasons
// about and what the device computes -- the hit point hitP = org + dir * t of a bounce ray that may have started anywhere in the scene
// (three roundings at the magnitude of the scene's coordinates and of the ray's length), the ray origin hitP +- GN * 1e-2 (one more),
// and the traverser's triangle test on (origin - v0, e1, e2), a dozen operations on differences of scene coordinates.  Every one of
// them is an ulp of a number no larger than the scene: the slack is kMarginUlps ulps of the largest coordinate magnitude of the
// scene box, never below the 2e-4 world units rounds 1-4 validated on scenes of +-15 units (there: 109 ulps).  (Round 4 had the
// absolute 2e-4 alone: at |coordinate| ~ 2000 that is under two ulps.)  The ray offset is 1e-2 whatever the scene's size
// (pathtracer.hlsl:560), so the certificate has a scale beyond which it cannot hold: where the slack exceeds kMaxMarginShare of the
// offset, gi_sun_table_update does not build a table at all (margin_usable) -- every shadow ray of such a scene is traced.
constexpr double kMarginFloor = 2e-4;
constexpr double kMarginUlps = 96.0;
constexpr double kMaxMarginShare = 0.25;
NEB_LIT_HD double margin_for(double scene_abs_max)
{
    const double m = kMarginUlps * 1.1920928955078125e-7 * scene_abs_max; // FLT_EPSILON = the ulp of 1
    return m > kMarginFloor ? m : kMarginFloor;
}
NEB_LIT_HD bool margin_usable(double margin) { return margin <= kMaxMarginShare * kOffset; }
constexpr double kMinFacing = 0.05;   // |n . L| below this: the receiver is (nearly) edge-on to the sun, no certificate
constexpr double kMinNormalDot = 0.25; // vertex normals further apart than this: the interpolated normal is not bounded well enough

struct Frame {
    double A[3], B[3], L[3]; // orthonormal, L towards the sun
    double tau;              // tan(half angle of the sun disk), padded
    double margin;           // slack on heights and lateral reach, from the scene's size (margin_for)
};

// scene_abs_max: the largest |coordinate| of the scene's bounding box (every ray origin, hit point and triangle lies inside it)
NEB_LIT_HD void make_frame(const float sun_direction[3], float tan_half_angle, double scene_abs_max, Frame& F)
{
    F.margin = margin_for(scene_abs_max);
    double l[3] = {-(double)sun_direction[0], -(double)sun_direction[1], -(double)sun_direction[2]};
    const double n = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
    for (int k = 0; k < 3; ++k)
        F.L[k] = l[k] / n;
    // any unit vector not parallel to L
    int m = 0;
    if (std::fabs(F.L[1]) < std::fabs(F.L[m]))
        m = 1;
    if (std::fabs(F.L[2]) < std::fabs(F.L[m]))
        m = 2;
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = e[0] * F.L[0] + e[1] * F.L[1] + e[2] * F.L[2];
    double a[3] = {e[0] - d * F.L[0], e[1] - d * F.L[1], e[2] - d * F.L[2]};
    const double an = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    for (int k = 0; k < 3; ++k)
        F.A[k] = a[k] / an;
    F.B[0] = F.L[1] * F.A[2] - F.L[2] * F.A[1];
    F.B[1] = F.L[2] * F.A[0] - F.L[0] * F.A[2];
    F.B[2] = F.L[0] * F.A[1] - F.L[1] * F.A[0];
    F.tau = (double)tan_half_angle * (1.0 + 1e-3) + 1e-7;
}

struct Tri { // a triangle in sun coordinates
    double a[3], b[3], h[3];
};

NEB_LIT_HD void to_sun(const Frame& F, const double p[3], double& a, double& b, double& h)
{
    a = p[0] * F.A[0] + p[1] * F.A[1] + p[2] * F.A[2];
    b = p[0] * F.B[0] + p[1] * F.B[1] + p[2] * F.B[2];
    h = p[0] * F.L[0] + p[1] * F.L[1] + p[2] * F.L[2];
}

// One side of a receiving triangle: where its shadow rays start.
struct Receiver {
    Tri t;
    bool valid;              // false: no certificate possible (edge-on, degenerate, wild normals) -> never "lit"
    double ga, gb, h0;       // the triangle's plane h = h0 + ga a + gb b
    double grad1;            // |ga| + |gb|
    double c_lo, c_hi;       // origins lie at plane(o_ab) + c_lo <= h_o <= plane(o_ab) + c_hi
    double off_a[2], off_b[2]; // interval of the lateral origin offset
    double en_a[3], en_b[3], en_c[3]; // outward unit edge normals of the projected triangle: n . (a, b) <= c inside
    double en_off[3];        // support of the offset box along each edge normal
    double h_min;            // lowest origin height
    double bb_a[2], bb_b[2]; // lateral box of the origins
};

// v[i]: world-space vertices; gn[i]: world-space UNIT vertex normals as the shade kernel interpolates them
// (normalize(xform_dir(M, n_i)); the interpolated GN is a normalised non-negative combination of them); sigma = +1 / -1.
NEB_LIT_HD void make_receiver(const Frame& F, const double v[3][3], const double gn[3][3], int sigma, Receiver& R)
{
    R.valid = false;
    for (int i = 0; i < 3; ++i)
        to_sun(F, v[i], R.t.a[i], R.t.b[i], R.t.h[i]);
    // plane over (a, b)
    const double ua = R.t.a[1] - R.t.a[0], ub = R.t.b[1] - R.t.b[0], uh = R.t.h[1] - R.t.h[0];
    const double wa = R.t.a[2] - R.t.a[0], wb = R.t.b[2] - R.t.b[0], wh = R.t.h[2] - R.t.h[0];
    const double area2 = ua * wb - ub * wa; // twice the signed projected area
    const double len_u = std::sqrt(ua * ua + ub * ub + uh * uh), len_w = std::sqrt(wa * wa + wb * wb + wh * wh);
    if (!(len_u > 0.0) || !(len_w > 0.0))
        return;
    // face normal (unnormalised) = u x w; its h component is area2
    const double nx = ub * wh - uh * wb, ny = uh * wa - ua * wh, nz = area2;
    const double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
    if (!(nn > 0.0) || std::fabs(nz) < kMinFacing * nn)
        return;
    R.ga = -nx / nz;
    R.gb = -ny / nz;
    R.h0 = R.t.h[0] - R.ga * R.t.a[0] - R.gb * R.t.b[0];
    R.grad1 = std::fabs(R.ga) + std::fabs(R.gb);
    // interval of the unit normal's sun components over the cone of the vertex normals
    double dmin = 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) {
            const double d = gn[i][0] * gn[j][0] + gn[i][1] * gn[j][1] + gn[i][2] * gn[j][2];
            dmin = d < dmin ? d : dmin;
        }
    if (!(dmin >= kMinNormalDot))
        return;
    const double grow = 1.0 / std::sqrt(dmin); // |sum c_i g_i| >= sqrt(dmin) sum c_i
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
    for (int i = 0; i < 3; ++i) {
        double c[3];
        to_sun(F, gn[i], c[0], c[1], c[2]);
        const double len = std::sqrt(gn[i][0] * gn[i][0] + gn[i][1] * gn[i][1] + gn[i][2] * gn[i][2]);
        if (!(len > 0.99 && len < 1.01))
            return;
        for (int k = 0; k < 3; ++k) {
            lo[k] = c[k] < lo[k] ? c[k] : lo[k];
            hi[k] = c[k] > hi[k] ? c[k] : hi[k];
        }
    }
    double off[3][2];
    for (int k = 0; k < 3; ++k) {
        double l = lo[k] < 0 ? lo[k] * grow : lo[k], h = hi[k] > 0 ? hi[k] * grow : hi[k];
        l = l * kOffset * 1.001 - 1e-6;
        h = h * kOffset * 1.001 + 1e-6;
        if (sigma > 0) {
            off[k][0] = l, off[k][1] = h;
        } else {
            off[k][0] = -h, off[k][1] = -l;
        }
    }
    R.off_a[0] = off[0][0], R.off_a[1] = off[0][1];
    R.off_b[0] = off[1][0], R.off_b[1] = off[1][1];
    // h_o = plane(o_ab - off_ab) + off_h = plane(o_ab) + off_h - ga off_a - gb off_b
    const double ga_hi = R.ga * off[0][0] > R.ga * off[0][1] ? R.ga * off[0][0] : R.ga * off[0][1];
    const double gb_hi = R.gb * off[1][0] > R.gb * off[1][1] ? R.gb * off[1][0] : R.gb * off[1][1];
    R.c_lo = off[2][0] - ga_hi - gb_hi;
    const double ga_lo = R.ga * off[0][0] < R.ga * off[0][1] ? R.ga * off[0][0] : R.ga * off[0][1];
    const double gb_lo = R.gb * off[1][0] < R.gb * off[1][1] ? R.gb * off[1][0] : R.gb * off[1][1];
    R.c_hi = off[2][1] - ga_lo - gb_lo;
    // outward edge normals of the projected triangle
    const double sgn = area2 > 0 ? 1.0 : -1.0;
    for (int e = 0; e < 3; ++e) {
        const int i = e, j = (e + 1) % 3;
        const double da = R.t.a[j] - R.t.a[i], db = R.t.b[j] - R.t.b[i];
        const double len = std::sqrt(da * da + db * db);
        if (!(len > 0.0))
            return;
        // for a counter-clockwise triangle the outward normal of edge i -> j is (db, -da)
        R.en_a[e] = sgn * db / len;
        R.en_b[e] = -sgn * da / len;
        R.en_c[e] = R.en_a[e] * R.t.a[i] + R.en_b[e] * R.t.b[i];
        const double sa = R.en_a[e] * R.off_a[0] > R.en_a[e] * R.off_a[1] ? R.en_a[e] * R.off_a[0] : R.en_a[e] * R.off_a[1];
        const double sb = R.en_b[e] * R.off_b[0] > R.en_b[e] * R.off_b[1] ? R.en_b[e] * R.off_b[0] : R.en_b[e] * R.off_b[1];
        R.en_off[e] = sa + sb;
    }
    double hmin = R.t.h[0], amin = R.t.a[0], amax = R.t.a[0], bmin = R.t.b[0], bmax = R.t.b[0];
    for (int i = 1; i < 3; ++i) {
        hmin = R.t.h[i] < hmin ? R.t.h[i] : hmin;
        amin = R.t.a[i] < amin ? R.t.a[i] : amin;
        amax = R.t.a[i] > amax ? R.t.a[i] : amax;
        bmin = R.t.b[i] < bmin ? R.t.b[i] : bmin;
        bmax = R.t.b[i] > bmax ? R.t.b[i] : bmax;
    }
    R.h_min = hmin + off[2][0];
    R.bb_a[0] = amin + R.off_a[0], R.bb_a[1] = amax + R.off_a[1];
    R.bb_b[0] = bmin + R.off_b[0], R.bb_b[1] = bmax + R.off_b[1];
    R.valid = true;
}

// The parts of a Receiver may_occlude / cover_mask read, under the same names (both are templates over the receiver's type): the device build hands these
// from lane to lane through LDS, so that any lane of a wave can test a candidate for any other (gi_sun_table.hip).
struct OccludeView {
    double h_min, c_lo, grad1, h0, ga, gb, bb_a[2], bb_b[2], en_a[3], en_b[3], en_c[3], en_off[3];
};
struct CoverView {
    struct { double a[3], b[3]; } t;
    double bb_a[2], bb_b[2], off_a[2], off_b[2], c_lo, c_hi, h0, ga, gb;
};

// Can a shadow ray from receiver R meet triangle O?  false = certainly not.
template <class RV>
NEB_LIT_HD bool may_occlude(const Frame& F, const RV& R, const Tri& O)
{
    double hmax = O.h[0] > O.h[1] ? O.h[0] : O.h[1];
    hmax = O.h[2] > hmax ? O.h[2] : hmax;
    const double smax = hmax - R.h_min + F.margin; // the most a ray can have climbed when it meets O
    if (smax <= 0.0)
        return false;
    const double rho = smax * F.tau + F.margin;    // ... and drifted sideways
    const double bound = R.c_lo - R.grad1 * rho - F.margin;
    // Two shortcuts that decide most candidates without the clip below (round 5: the build's time is these tests, not the walk).  (1) cannot change the
    // answer: the quantity tested at the end, f = h - plane_R(a, b), is LINEAR over O, and the clipped polygon is a part of O -- if no vertex of O rises
    // above the bound, no vertex of the polygon does (the receiver's coplanar neighbours, everything below it).  (2) is sound and a little sharper than the
    // clip: every ray origin lies in the lateral box bb, a ray drifts at most rho, so O wholly outside bb pushed out by rho cannot be met -- the three
    // edge half-planes alone form a triangle whose corners reach beyond that box (an obtuse receiver's far corner), where the clip says "may".
    {
        bool above = false;
        for (int i = 0; i < 3; ++i)
            above = above || !(O.h[i] - (R.h0 + R.ga * O.a[i] + R.gb * O.b[i]) <= bound);
        if (!above)
            return false;
        double a0 = O.a[0], a1 = O.a[0], b0 = O.b[0], b1 = O.b[0];
        for (int i = 1; i < 3; ++i) {
            a0 = O.a[i] < a0 ? O.a[i] : a0, a1 = O.a[i] > a1 ? O.a[i] : a1;
            b0 = O.b[i] < b0 ? O.b[i] : b0, b1 = O.b[i] > b1 ? O.b[i] : b1;
        }
        if (a1 < R.bb_a[0] - rho || a0 > R.bb_a[1] + rho || b1 < R.bb_b[0] - rho || b0 > R.bb_b[1] + rho)
            return false;
    }
    // clip O (as a 3-D polygon) by the three vertical planes through R's edges, pushed out by the drift and the offset box
    double pa[8], pb[8], ph[8], qa[8], qb[8], qh[8];
    int n = 3;
    for (int i = 0; i < 3; ++i)
        pa[i] = O.a[i], pb[i] = O.b[i], ph[i] = O.h[i];
    for (int e = 0; e < 3 && n > 0; ++e) {
        const double na = R.en_a[e], nb = R.en_b[e];
        const double c = R.en_c[e] + R.en_off[e] + rho * (std::fabs(na) + std::fabs(nb));
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double di = na * pa[i] + nb * pb[i] - c, dj = na * pa[j] + nb * pb[j] - c;
            if (di <= 0.0) {
                qa[m] = pa[i], qb[m] = pb[i], qh[m] = ph[i];
                ++m;
            }
            if ((di <= 0.0) != (dj <= 0.0)) {
                const double t = di / (di - dj);
                qa[m] = pa[i] + t * (pa[j] - pa[i]);
                qb[m] = pb[i] + t * (pb[j] - pb[i]);
                qh[m] = ph[i] + t * (ph[j] - ph[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i)
            pa[i] = qa[i], pb[i] = qb[i], ph[i] = qh[i];
    }
    if (n == 0)
        return false;
    // s = h_q - h_o <= [h_q - plane_R(q_ab)] + grad1 * rho - c_lo; linear over the clipped polygon: the maximum is at a vertex
    for (int i = 0; i < n; ++i) {
        const double f = ph[i] - (R.h0 + R.ga * pa[i] + R.gb * pb[i]);
        // (an intersection point is computed with rounding of ~1e-16 relative: covered by the margin)
        if (!(f <= bound))
            return true;
    }
    return false;
}

// Which of kCoverSamples fixed sample origins of R (a triangular grid of barycentrics, offset to the middle of the origin box)
// a ray towards the centre of the sun disk would leave through O: bit k of the mask.  Only RANKS occluder hints
// (gi_sun_table.hip): a hint is tried with the traverser's own triangle test, so a poor estimate costs speed, never correctness.
constexpr int kCoverSamples = 28;
// cover_mask counts a sample when O's height there exceeds the sample origin's by 1e-3: h_O - (h0 + ga a + gb b) > cover_floor(R) at the sample's (a, b)
template <class RV>
NEB_LIT_HD double cover_floor(const RV& R)
{
    const double ma = 0.5 * (R.off_a[0] + R.off_a[1]), mb = 0.5 * (R.off_b[0] + R.off_b[1]), mc = 0.5 * (R.c_lo + R.c_hi);
    return 1e-3 + mc - R.ga * ma - R.gb * mb;
}
template <class RV>
NEB_LIT_HD unsigned cover_mask(const RV& R, const Tri& O)
{
    // quick reject on the lateral boxes
    double oa0 = O.a[0], oa1 = O.a[0], ob0 = O.b[0], ob1 = O.b[0];
    for (int i = 1; i < 3; ++i) {
        oa0 = O.a[i] < oa0 ? O.a[i] : oa0, oa1 = O.a[i] > oa1 ? O.a[i] : oa1;
        ob0 = O.b[i] < ob0 ? O.b[i] : ob0, ob1 = O.b[i] > ob1 ? O.b[i] : ob1;
    }
    if (oa1 < R.bb_a[0] || oa0 > R.bb_a[1] || ob1 < R.bb_b[0] || ob0 > R.bb_b[1])
        return 0u;
    { // (the height difference is linear over O: nowhere above the sample origins' plane, no sample counted -- the receiver's coplanar neighbours)
        const double fl = cover_floor(R);
        bool above = false;
        for (int i = 0; i < 3; ++i)
            above = above || O.h[i] - (R.h0 + R.ga * O.a[i] + R.gb * O.b[i]) > fl;
        if (!above)
            return 0u;
    }
    const double ua = O.a[1] - O.a[0], ub = O.b[1] - O.b[0], wa = O.a[2] - O.a[0], wb = O.b[2] - O.b[0];
    const double det = ua * wb - ub * wa;
    if (det == 0.0)
        return 0u; // edge-on to the sun: no area to stand in
    const double inv = 1.0 / det;
    // The sample origins are a lattice p(r, c) = P + bu E1 + bv E2 on R's projection; the barycentrics (s, t) of p in O's projection, the third one
    // u = 1 - s - t, and the height of O over the sample origin there are all AFFINE in (bu, bv): three values each, then one fma per row and one per sample.
    const double ma = 0.5 * (R.off_a[0] + R.off_a[1]), mb = 0.5 * (R.off_b[0] + R.off_b[1]);
    const double e1a = R.t.a[1] - R.t.a[0], e1b = R.t.b[1] - R.t.b[0], e2a = R.t.a[2] - R.t.a[0], e2b = R.t.b[2] - R.t.b[0];
    const double da = R.t.a[0] + ma - O.a[0], db = R.t.b[0] + mb - O.b[0];
    const double s0 = (da * wb - db * wa) * inv, su = (e1a * wb - e1b * wa) * inv, sv = (e2a * wb - e2b * wa) * inv;
    const double t0 = (ua * db - ub * da) * inv, tu = (ua * e1b - ub * e1a) * inv, tv = (ua * e2b - ub * e2a) * inv;
    const double u0 = 1.0 - s0 - t0, uu = -su - tu, uv = -sv - tv;
    const double dh1 = O.h[1] - O.h[0], dh2 = O.h[2] - O.h[0];
    // f(p) = h_O(p) - (h0 + ga pa + gb pb) must exceed cover_floor(R) (see there)
    const double d0 = O.h[0] + s0 * dh1 + t0 * dh2 - (R.h0 + R.ga * (R.t.a[0] + ma) + R.gb * (R.t.b[0] + mb)) - cover_floor(R);
    const double du = su * dh1 + tu * dh2 - (R.ga * e1a + R.gb * e1b), dv = sv * dh1 + tv * dh2 - (R.ga * e2a + R.gb * e2b);
    unsigned mask = 0u;
    int k = 0;
    for (int r = 0; r < 7; ++r) {
        const double bu = (r + 1.0 / 3.0) / 7.0;
        const double sr = s0 + bu * su, tr = t0 + bu * tu, ur = u0 + bu * uu, dr = d0 + bu * du;
        for (int c = 0; c < 7 - r; ++c, ++k) {
            const double bv = (c + 1.0 / 3.0) / 7.0;
            const double sk = sr + bv * sv, tk = tr + bv * tv, uk = ur + bv * uv, dk = dr + bv * dv;
            const double in = sk < tk ? (sk < uk ? sk : uk) : (tk < uk ? tk : uk);
            if (in >= 0.0 && dk > 0.0)
                mask |= 1u << k;
        }
    }
    return mask;
}

// How much of R's projected triangle lies in the shadow of O: the area of O's projection inside R's, counted only where that part
// of O is above every ray origin of R.  Only RANKS occluder hints (gi_sun_table.hip): a hint is tried with the traverser's own
// triangle test, so a poor estimate costs speed, never correctness.
NEB_LIT_HD double shadow_cover(const Receiver& R, const Tri& O)
{
    double pa[8], pb[8], ph[8], qa[8], qb[8], qh[8];
    int n = 3;
    for (int i = 0; i < 3; ++i)
        pa[i] = O.a[i], pb[i] = O.b[i], ph[i] = O.h[i];
    for (int e = 0; e < 3 && n > 0; ++e) {
        const double na = R.en_a[e], nb = R.en_b[e], c = R.en_c[e];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double di = na * pa[i] + nb * pb[i] - c, dj = na * pa[j] + nb * pb[j] - c;
            if (di <= 0.0) {
                qa[m] = pa[i], qb[m] = pb[i], qh[m] = ph[i];
                ++m;
            }
            if ((di <= 0.0) != (dj <= 0.0)) {
                const double t = di / (di - dj);
                qa[m] = pa[i] + t * (pa[j] - pa[i]);
                qb[m] = pb[i] + t * (pb[j] - pb[i]);
                qh[m] = ph[i] + t * (ph[j] - ph[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i)
            pa[i] = qa[i], pb[i] = qb[i], ph[i] = qh[i];
    }
    if (n < 3)
        return 0.0;
    double area2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        if (!(ph[i] - (R.h0 + R.ga * pa[i] + R.gb * pb[i]) > R.c_hi + 1e-3))
            return 0.0; // this part of O is not (wholly) above the ray origins
        area2 += pa[i] * pb[j] - pa[j] * pb[i];
    }
    return 0.5 * (area2 < 0 ? -area2 : area2);
}

} // namespace lit
} // namespace neb

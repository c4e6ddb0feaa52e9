// tools/lit_proto.cpp -- CPU prototype / checker of the sun-visibility table: the same certificate as the device build
// (nebulae_amd/csrc/lit_predicate.h), candidates found through a uniform grid in sun coordinates instead of the BVH.
// Build: g++ -O2 -fopenmp -shared -fPIC -I nebulae_amd/csrc tools/lit_proto.cpp -o build_variants/liblit_proto.so
#include <algorithm>
#include <cstdint>
#include <vector>

#include "lit_predicate.h"

using namespace neb::lit;

// the largest |coordinate| of the scene (what the device takes from the scene box): sets the certificate's slack
static double scene_abs_max(int n, const float* verts)
{
    double m = 0.0;
    for (long i = 0; i < 9L * n; ++i)
        m = std::max(m, std::fabs((double)verts[i]));
    return m;
}

// area of polygon O (projected) clipped to the projected triangle R (no dilation): how much of R's footprint O covers
static double covered_area(const Receiver& R, const Tri& O, double shrink)
{
    double pa[8], pb[8], qa[8], qb[8];
    int n = 3;
    // shrink O towards its centroid by `shrink` (absolute), crude but conservative enough for an estimate
    const double ca = (O.a[0] + O.a[1] + O.a[2]) / 3, cb = (O.b[0] + O.b[1] + O.b[2]) / 3;
    for (int i = 0; i < 3; ++i) {
        const double da = O.a[i] - ca, db = O.b[i] - cb, l = std::sqrt(da * da + db * db);
        const double k = l > 0 ? std::max(0.0, 1.0 - 2.0 * shrink / l) : 0.0;
        pa[i] = ca + da * k, pb[i] = cb + db * k;
    }
    for (int e = 0; e < 3 && n > 0; ++e) {
        const double na = R.en_a[e], nb = R.en_b[e], c = R.en_c[e];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double di = na * pa[i] + nb * pb[i] - c, dj = na * pa[j] + nb * pb[j] - c;
            if (di <= 0.0) { qa[m] = pa[i], qb[m] = pb[i]; ++m; }
            if ((di <= 0.0) != (dj <= 0.0)) { const double t = di / (di - dj); qa[m] = pa[i] + t * (pa[j] - pa[i]); qb[m] = pb[i] + t * (pb[j] - pb[i]); ++m; }
        }
        n = m;
        for (int i = 0; i < n; ++i) pa[i] = qa[i], pb[i] = qb[i];
    }
    double A = 0;
    for (int i = 0; i < n; ++i) { const int j = (i + 1) % n; A += pa[i] * pb[j] - pa[j] * pb[i]; }
    return std::fabs(A) * 0.5;
}

// per triangle and side: the share of its footprint covered by the best / the two best single occluders above it
extern "C" void lit_proto_hints(int n, const float* verts, const float* normals, const float* sun_dir, float tan_half, float* cover /* n x 2 x 2 */)
{
    Frame F;
    make_frame(sun_dir, tan_half, scene_abs_max(n, verts), F);
    std::vector<Tri> T(n);
    double amin = 1e30, amax = -1e30, bmin = 1e30, bmax = -1e30;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const double p[3] = {verts[9 * i + 3 * k], verts[9 * i + 3 * k + 1], verts[9 * i + 3 * k + 2]};
            to_sun(F, p, T[i].a[k], T[i].b[k], T[i].h[k]);
            amin = std::min(amin, T[i].a[k]), amax = std::max(amax, T[i].a[k]);
            bmin = std::min(bmin, T[i].b[k]), bmax = std::max(bmax, T[i].b[k]);
        }
    const double cell = 0.25;
    const int gx = (int)((amax - amin) / cell) + 1, gy = (int)((bmax - bmin) / cell) + 1;
    std::vector<std::vector<int>> grid((size_t)gx * gy);
    auto cx = [&](double a) { return std::min(gx - 1, std::max(0, (int)((a - amin) / cell))); };
    auto cy = [&](double b) { return std::min(gy - 1, std::max(0, (int)((b - bmin) / cell))); };
    for (int i = 0; i < n; ++i) {
        const double a0 = std::min({T[i].a[0], T[i].a[1], T[i].a[2]}), a1 = std::max({T[i].a[0], T[i].a[1], T[i].a[2]});
        const double b0 = std::min({T[i].b[0], T[i].b[1], T[i].b[2]}), b1 = std::max({T[i].b[0], T[i].b[1], T[i].b[2]});
        for (int y = cy(b0); y <= cy(b1); ++y)
            for (int x = cx(a0); x <= cx(a1); ++x)
                grid[(size_t)y * gx + x].push_back(i);
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; ++i) {
        double v[3][3], g[3][3];
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c)
                v[k][c] = verts[9 * i + 3 * k + c], g[k][c] = normals[9 * i + 3 * k + c];
        for (int side = 0; side < 2; ++side) {
            cover[4 * i + 2 * side] = cover[4 * i + 2 * side + 1] = 0.f;
            Receiver R;
            make_receiver(F, v, g, side == 0 ? +1 : -1, R);
            if (!R.valid)
                continue;
            const double ua = R.t.a[1] - R.t.a[0], ub = R.t.b[1] - R.t.b[0], wa = R.t.a[2] - R.t.a[0], wb = R.t.b[2] - R.t.b[0];
            const double area = 0.5 * std::fabs(ua * wb - ub * wa);
            double hmaxR = std::max({R.t.h[0], R.t.h[1], R.t.h[2]}) + 0.02;
            double best = 0, second = 0;
            std::vector<int> seen;
            for (int y = cy(R.bb_b[0]); y <= cy(R.bb_b[1]); ++y)
                for (int x = cx(R.bb_a[0]); x <= cx(R.bb_a[1]); ++x)
                    for (int j : grid[(size_t)y * gx + x]) {
                        if (j == i || std::find(seen.begin(), seen.end(), j) != seen.end())
                            continue;
                        seen.push_back(j);
                        if (std::min({T[j].h[0], T[j].h[1], T[j].h[2]}) < hmaxR)
                            continue; // only occluders wholly above the receiver
                        const double c = covered_area(R, T[j], 0.0) / area;
                        if (c > best) { second = best; best = c; } else if (c > second) second = c;
                    }
            cover[4 * i + 2 * side] = (float)best;
            cover[4 * i + 2 * side + 1] = (float)std::min(1.0, best + second);
        }
    }
}

extern "C" void lit_proto_flags(int n, const float* verts, const float* normals, const float* sun_dir, float tan_half, unsigned char* flags,
                                double* stats)
{
    Frame F;
    make_frame(sun_dir, tan_half, scene_abs_max(n, verts), F);
    if (!margin_usable(F.margin)) { // the device builds no table for a scene this large (gi_sun_table_update): nothing is called lit
        std::fill(flags, flags + n, (unsigned char)0);
        if (stats)
            stats[0] = stats[1] = 0;
        return;
    }
    std::vector<Tri> T(n);
    double amin = 1e30, amax = -1e30, bmin = 1e30, bmax = -1e30, hmax = -1e30;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const double p[3] = {verts[9 * i + 3 * k], verts[9 * i + 3 * k + 1], verts[9 * i + 3 * k + 2]};
            to_sun(F, p, T[i].a[k], T[i].b[k], T[i].h[k]);
            amin = std::min(amin, T[i].a[k]), amax = std::max(amax, T[i].a[k]);
            bmin = std::min(bmin, T[i].b[k]), bmax = std::max(bmax, T[i].b[k]);
            hmax = std::max(hmax, T[i].h[k]);
        }
    const double cell = 0.25;
    const int gx = (int)((amax - amin) / cell) + 1, gy = (int)((bmax - bmin) / cell) + 1;
    std::vector<std::vector<int>> grid((size_t)gx * gy);
    auto cx = [&](double a) { return std::min(gx - 1, std::max(0, (int)((a - amin) / cell))); };
    auto cy = [&](double b) { return std::min(gy - 1, std::max(0, (int)((b - bmin) / cell))); };
    for (int i = 0; i < n; ++i) {
        const double a0 = std::min({T[i].a[0], T[i].a[1], T[i].a[2]}), a1 = std::max({T[i].a[0], T[i].a[1], T[i].a[2]});
        const double b0 = std::min({T[i].b[0], T[i].b[1], T[i].b[2]}), b1 = std::max({T[i].b[0], T[i].b[1], T[i].b[2]});
        for (int y = cy(b0); y <= cy(b1); ++y)
            for (int x = cx(a0); x <= cx(a1); ++x)
                grid[(size_t)y * gx + x].push_back(i);
    }
    double tests = 0, invalid = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tests, invalid)
    for (int i = 0; i < n; ++i) {
        double v[3][3], g[3][3];
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c)
                v[k][c] = verts[9 * i + 3 * k + c], g[k][c] = normals[9 * i + 3 * k + c];
        unsigned char f = 0;
        for (int side = 0; side < 2; ++side) {
            Receiver R;
            make_receiver(F, v, g, side == 0 ? +1 : -1, R);
            if (!R.valid) {
                invalid += 1;
                continue;
            }
            const double reach = (hmax - R.h_min + F.margin) * F.tau + F.margin;
            bool lit = true;
            for (int y = cy(R.bb_b[0] - reach); y <= cy(R.bb_b[1] + reach) && lit; ++y)
                for (int x = cx(R.bb_a[0] - reach); x <= cx(R.bb_a[1] + reach) && lit; ++x)
                    for (int j : grid[(size_t)y * gx + x]) {
                        tests += 1;
                        if (may_occlude(F, R, T[j])) {
                            lit = false;
                            break;
                        }
                    }
            if (lit)
                f |= (unsigned char)(1u << side);
        }
        flags[i] = f;
    }
    if (stats) {
        stats[0] = tests;
        stats[1] = invalid;
    }
}

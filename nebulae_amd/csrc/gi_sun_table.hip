// gi_sun_table.hip -- the sun table: what can be known about a triangle's shadow rays before any of them is traced.
//
// Every shadow ray of the GI path points at the sun disk (assets/shaders/pathtracer.hlsl:533-575): it leaves the hit point by
// GN * 1e-2 on the side the disk sample is on and goes along normalize(L + offset), |offset| <= tan(0.29 deg).  Whether such a ray
// can meet anything is to a large extent a property of the TRIANGLE it starts on, the scene and the sun -- so it is decided once
// per sun position, one thread per triangle, which walks the BVH4 (the exact 128-byte nodes) for everything that reaches into the
// column the triangle's rays can sweep (its footprint along the sun direction, widened by tan(half angle) per unit climbed):
//   * LIT BITS (one per side): set when the certificate of lit_predicate.h (double precision, conservative: "cannot occlude" is
//     certain) holds for every triangle found there, the receiver itself included.  On the bench frame 95 % of the unoccluded
//     shadow rays start on such a (triangle, side).
//   * OCCLUDER HINTS (up to four): the triangles that together shadow most of the receiver (greedy cover over 28 sample origins).
//     gi_shade_kernel tries them with the traverser's own triangle test on the very ray -- a hit is the any-hit traversal's hit,
//     exactly, by construction.  90 % of the occluded shadow rays end at a hint.
// Both live in the spare bits / last 12 bytes of the triangle's 128-byte shading record, which gi_shade_kernel has in LDS anyway:
// not one extra byte of traffic for the flags, 48 bytes per hint tried.  The answer is the traversal's own
// (tests/test_sun_table_gpu.py: radiance, hit records and sun-visibility flags bit-identical with the table on and off, every GI scene
// and the bench frame).  Rebuilt (3 ms at 262 k triangles) when sunLightDirection / sunTanHalfAngle have changed AND held for two
// dispatches, or the scene is rebuilt; a sun that moves every frame -- the reference marks such frames dynamic,
// src/DeferredRenderer.cpp:169-171 -- is traced the plain way meanwhile.
//
// This takes the place of the sun-space height map sized in round 3: a height map has to resolve the 1e-2 ray offset against the
// slope of every lit surface (a 4096^2 map for a floor, and a max-plus filter for the cone), costs a scattered fetch per ray, is only
// as exact as its cells, and says nothing about the occluded rays; the per-triangle form uses the exact geometry.
// (What the lit bits alone are worth: 8 us -- the unoccluded rays of an open court leave the tree after 3 node visits on average;
// the hints and the compaction of what is left, gi.hip, are the other 77.)
#include <cstdlib>
#include <type_traits>

#include "gi_device.h"
#include "lit_predicate.h"

namespace neb {

struct SunTableArgs {
    SceneView S;
    lit::Frame F;
    double scene_hmax; // highest point of the scene along L
    double box_pad;    // padding of the node boxes (floats) in the double-precision cull: an ulp of the scene's largest coordinate, at least 1e-5
    float4* shade;     // the shading records (writable view of S.shade)
    unsigned long long* counts; // [0] sides proven lit (+), [1] (-), [2] triangles with an occluder hint, [3] triangles listed for the hint pass
    uint32_t* hint_list;        // the triangles whose primary side is not proven lit, | that side << 31 (pass 1 appends, pass 2 reads)
    unsigned long long* walk_stats; // diagnostics or null: [6 * (PASS - 1) + ...] = {node visits, longest walk, candidates tested, wave time sum, longest wave (10-ns ticks), waves}
};

__device__ inline void node_child_box(const Bvh4Node& nd, int q, double lo[3], double hi[3])
{
    const float l0[4] = {nd.lox.x, nd.lox.y, nd.lox.z, nd.lox.w}, l1[4] = {nd.loy.x, nd.loy.y, nd.loy.z, nd.loy.w},
                l2[4] = {nd.loz.x, nd.loz.y, nd.loz.z, nd.loz.w};
    const float h0[4] = {nd.hix.x, nd.hix.y, nd.hix.z, nd.hix.w}, h1[4] = {nd.hiy.x, nd.hiy.y, nd.hiy.z, nd.hiy.w},
                h2[4] = {nd.hiz.x, nd.hiz.y, nd.hiz.z, nd.hiz.w};
    lo[0] = l0[q], lo[1] = l1[q], lo[2] = l2[q];
    hi[0] = h0[q], hi[1] = h1[q], hi[2] = h2[q];
}

// Two passes (round 5; round 4 did both in one walk of the whole column, 15.5 ms at 262 k triangles):
// PASS 1, the lit bits: per side a walk that ENDS as soon as the side has met something it cannot rule out (a side turned away from the sun meets its own
// triangle before any walk), and the triangles whose primary side is left unproven appended to a list;
// PASS 2, the hints, for those only, in dense waves, over the primary side's column, ending once the four best candidates cover all 28 sample origins.
// One Receiver is live at a time and nothing a lane keeps is indexed at run time except the node stack and may_occlude's clip buffers (round 5: the one-walk
// kernel held both sides' receivers and the candidate table in scratch, 1.5-1.8 KB per lane at 255 registers; the build's time was its candidate tests).
#ifndef NEB_SUN_EDGE_CULL
#define NEB_SUN_EDGE_CULL 1
#endif
#ifndef NEB_SUN_WAVES
#define NEB_SUN_WAVES 1 // waves per SIMD the register allocation of the lit pass aims for (measured: 1 = 256 registers 1.91 ms, 2 = 176 registers + spills 2.14 ms)
#endif

// the kCand candidates that cover most sample origins so far, best first -- registers only: insertion is a chain of compare-and-swap
constexpr int kCand = 8;
struct SunCands {
    uint32_t tri[kCand], mask[kCand];
    int pop[kCand];
    uint32_t seen;
    __device__ void clear()
    {
#pragma unroll
        for (int k = 0; k < kCand; ++k)
            tri[k] = kNoHint, mask[k] = 0u, pop[k] = 0;
        seen = 0u;
    }
    __device__ void insert(uint32_t tj, uint32_t m)
    {
        int pc = tj < kNoHint ? __popc(m) : 0; // (a triangle whose index does not fit a hint's 23 bits is no candidate: it would end the list of hints where it stands)
        seen += pc ? 1u : 0u;
#pragma unroll
        for (int k = 0; k < kCand; ++k) { // (behind the entries that cover as many: first found stays first)
            const bool up = pc > pop[k];
            const uint32_t t_ = tri[k], m_ = mask[k];
            const int p_ = pop[k];
            tri[k] = up ? tj : t_, mask[k] = up ? m : m_, pop[k] = up ? pc : p_;
            tj = up ? t_ : tj, m = up ? m_ : m, pc = up ? p_ : pc;
        }
    }
    __device__ bool top4_cover_all() const
    {
        constexpr uint32_t kAll = (1u << lit::kCoverSamples) - 1u;
        return ((mask[0] | mask[1] | mask[2] | mask[3]) & kAll) == kAll;
    }
    // greedy cover: up to kHints candidates, each the one that adds most samples not yet covered
    __device__ void choose(uint32_t hint[kHints])
    {
        uint32_t covered = 0u;
#pragma unroll
        for (int h = 0; h < kHints; ++h) {
            int best = -1, gain = 0;
#pragma unroll
            for (int k = 0; k < kCand; ++k) {
                const int gk = tri[k] != kNoHint ? __popc(mask[k] & ~covered) : 0;
                if (gk > gain)
                    gain = gk, best = k;
            }
            uint32_t chosen = kNoHint;
#pragma unroll
            for (int k = 0; k < kCand; ++k)
                if (k == best) {
                    chosen = tri[k];
                    covered |= mask[k];
                    tri[k] = kNoHint;
                }
            hint[h] = chosen;
        }
    }
};

// What a lane of the wave needs of ANOTHER lane's receiver to test a candidate for it, through LDS (field f of lane l at view[64 * f + l]).
template <class Fn>
__device__ __forceinline__ void each_field(lit::OccludeView& V, Fn fn)
{
    fn(0, V.h_min), fn(1, V.c_lo), fn(2, V.grad1), fn(3, V.h0), fn(4, V.ga), fn(5, V.gb), fn(6, V.bb_a[0]), fn(7, V.bb_a[1]), fn(8, V.bb_b[0]), fn(9, V.bb_b[1]);
#pragma unroll
    for (int e = 0; e < 3; ++e)
        fn(10 + e, V.en_a[e]), fn(13 + e, V.en_b[e]), fn(16 + e, V.en_c[e]), fn(19 + e, V.en_off[e]);
}
template <class Fn>
__device__ __forceinline__ void each_field(lit::CoverView& V, Fn fn)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
        fn(i, V.t.a[i]), fn(3 + i, V.t.b[i]);
    fn(6, V.bb_a[0]), fn(7, V.bb_a[1]), fn(8, V.bb_b[0]), fn(9, V.bb_b[1]), fn(10, V.off_a[0]), fn(11, V.off_a[1]), fn(12, V.off_b[0]), fn(13, V.off_b[1]);
    fn(14, V.c_lo), fn(15, V.c_hi), fn(16, V.h0), fn(17, V.ga), fn(18, V.gb);
}
constexpr int kOccludeFields = 22, kCoverFields = 19;
__device__ __forceinline__ void view_of(const lit::Receiver& R, lit::OccludeView& V)
{
    V.h_min = R.h_min, V.c_lo = R.c_lo, V.grad1 = R.grad1, V.h0 = R.h0, V.ga = R.ga, V.gb = R.gb;
    V.bb_a[0] = R.bb_a[0], V.bb_a[1] = R.bb_a[1], V.bb_b[0] = R.bb_b[0], V.bb_b[1] = R.bb_b[1];
#pragma unroll
    for (int e = 0; e < 3; ++e)
        V.en_a[e] = R.en_a[e], V.en_b[e] = R.en_b[e], V.en_c[e] = R.en_c[e], V.en_off[e] = R.en_off[e];
}
__device__ __forceinline__ void view_of(const lit::Receiver& R, lit::CoverView& V)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
        V.t.a[i] = R.t.a[i], V.t.b[i] = R.t.b[i];
    V.bb_a[0] = R.bb_a[0], V.bb_a[1] = R.bb_a[1], V.bb_b[0] = R.bb_b[0], V.bb_b[1] = R.bb_b[1];
    V.off_a[0] = R.off_a[0], V.off_a[1] = R.off_a[1], V.off_b[0] = R.off_b[0], V.off_b[1] = R.off_b[1];
    V.c_lo = R.c_lo, V.c_hi = R.c_hi, V.h0 = R.h0, V.ga = R.ga, V.gb = R.gb;
}

// what one wave keeps in LDS during a walk (one wave = one workgroup)
constexpr int kSunQueue = 32;
template <bool HINT>
struct SunWaveLds {
    uint32_t queue[kSunQueue * 64];               // per lane: the candidate triangles its walk has reached and nobody has tested yet
    uint32_t result[HINT ? kSunQueue * 64 : 64];  // hints: the cover mask of queue entry k of lane l at [64 * k + l]; lit bits: [l] != 0 once a candidate of lane l may occlude
    uint32_t start[64];                           // exclusive prefix sum of the lanes' queue lengths
    uint32_t own_tri[64];                         // the receivers' own triangles
    double view[(HINT ? kCoverFields : kOccludeFields) * 64];
};

// Walks everything that reaches into the column the rays of receiver R (triangle ti) can sweep.  HINT = false: returns whether the certificate held for every
// triangle found (ends soon after the first that may occlude).  HINT = true: ranks the triangles that shadow R's sample origins into `cands`.  EVERY lane of the
// wave calls this (active = false: nothing to walk) -- the walk only QUEUES the triangles of the leaves it reaches; when some lane's queue is nearly full or no
// lane walks any more the WAVE tests what its lanes have queued, pair p = (receiver, candidate) on lane p % 64 whoever the receiver belongs to.
// (Tested inside the walk -- up to eight candidates behind any node of any lane -- a wave ran iterations x 8 test slots for the 20 candidates a lane has on
// average: 2.5 % of the lanes busy, 1.2 ms per wave, the longest wave 6.6 ms = the launch.  Each lane testing its own queue: the longest wave 4.2 ms.)
template <bool HINT>
__device__ bool sun_walk_column(const SunTableArgs& a, const lit::Receiver& R, bool active, uint32_t ti, SunCands& cands, SunWaveLds<HINT>& lds, uint32_t& visits, uint32_t& tested)
{
    using View = typename std::conditional<HINT, lit::CoverView, lit::OccludeView>::type;
    const uint32_t lane = threadIdx.x & 63u;
    const double reach = (a.scene_hmax - R.h_min + a.F.margin) * a.F.tau + a.F.margin;
    const double qa0 = R.bb_a[0] - reach, qa1 = R.bb_a[1] + reach, qb0 = R.bb_b[0] - reach, qb1 = R.bb_b[1] + reach, qh = R.h_min - a.F.margin;
    bool alive = active;
    uint32_t qn = 0; // entries in this lane's queue (queue[64 * k + lane] = entry k)
    {
        View mine;
        view_of(R, mine);
        each_field(mine, [&](int f, double& x) { lds.view[64 * f + lane] = x; });
        lds.own_tri[lane] = ti;
        if (!HINT)
            lds.result[lane] = 0u;
    }
    auto queue_leaf = [&](int code) {
        const uint32_t c = (uint32_t)~code, first = c >> 2, count = (c & 3u) + 1u;
        for (uint32_t k = 0; k < count; ++k)
            lds.queue[64 * qn++ + lane] = first + k;
    };
    // the wave empties its lanes' queues
    auto test_queued = [&]() {
        uint32_t incl = qn; // inclusive prefix sum over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            incl += lane >= (uint32_t)off ? up : 0u;
        }
        const uint32_t total = __shfl(incl, 63);
        lds.start[lane] = incl - qn;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t base = 0; base < total; base += 64) {
            const uint32_t p = base + lane;
            if (p < total) {
                uint32_t o = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1)
                    o += lds.start[o + step] <= p ? step : 0u;
                const uint32_t k = p - lds.start[o];
                const uint32_t tj = lds.queue[64 * k + o];
                uint32_t mask = 0u;
                if (tj < a.S.n_tris && (HINT ? tj != lds.own_tri[o] : lds.result[o] == 0u)) { // (a receiver that has lost its certificate needs no more tests)
                    ++tested;
                    View V;
                    each_field(V, [&](int f, double& x) { x = lds.view[64 * f + o]; });
                    const float4 o0 = a.S.tris[3 * tj], o1 = a.S.tris[3 * tj + 1], o2 = a.S.tris[3 * tj + 2];
                    const double w[3][3] = {{o0.x, o0.y, o0.z},
                                            {(double)o0.x + o0.w, (double)o0.y + o1.x, (double)o0.z + o1.y},
                                            {(double)o0.x + o1.z, (double)o0.y + o1.w, (double)o0.z + o2.x}};
                    lit::Tri O;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        lit::to_sun(a.F, w[i], O.a[i], O.b[i], O.h[i]);
                    if constexpr (HINT) {
                        mask = lit::cover_mask(V, O);
                    } else {
                        if (lit::may_occlude(a.F, V, O))
                            lds.result[o] = 1u;
                    }
                }
                if constexpr (HINT)
                    lds.result[64 * k + o] = mask;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (HINT) {
            for (uint32_t k = 0; k < qn; ++k) {
                const uint32_t m = lds.result[64 * k + lane];
                if (m)
                    cands.insert(lds.queue[64 * k + lane], m);
            }
        } else {
            alive = alive && lds.result[lane] == 0u;
        }
        qn = 0;
    };
    int stack[64];
    int sp = 0;
    int node = active ? a.S.root : kTravDone;
    if (node != kTravDone && node < 0) { // a scene of one leaf
        queue_leaf(node);
        node = kTravDone;
    }
    // A receiver the size of a wall strip has a column the size of the scene: its hint walk would be the launch (the long-thin stand-in: 2021 node visits
    // for one triangle, 15.7 of the build's 18 ms in ONE wave, against 47 on average) -- and four hints cannot cover such a receiver anyway.  The hint walk
    // of a receiver ends after kSunHintVisits nodes with the best candidates it has met (hints are only hints).  The bench scene's longest is 219.
    constexpr uint32_t kSunHintVisits = 320u;
    uint32_t walked = 0;
    for (;;) {
        const bool walking = node != kTravDone && alive && !(HINT && (cands.top4_cover_all() || walked >= kSunHintVisits));
        if (walking) {
            ++visits;
            ++walked;
            const Bvh4Node nd = a.S.nodes[node];
            const int ch[4] = {nd.child.x, nd.child.y, nd.child.z, nd.child.w};
            node = kTravDone;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double lo[3], hi[3];
                node_child_box(nd, q, lo, hi);
                if (!(lo[0] <= hi[0])) // unused slot: inverted box
                    continue;
                // sun-space bounds of the world box: centre +- |axis| . half extent (padded: the boxes are floats)
                const double c[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
                const double e[3] = {0.5 * (hi[0] - lo[0]) + a.box_pad, 0.5 * (hi[1] - lo[1]) + a.box_pad, 0.5 * (hi[2] - lo[2]) + a.box_pad};
                const double ca = c[0] * a.F.A[0] + c[1] * a.F.A[1] + c[2] * a.F.A[2];
                const double ea = e[0] * fabs(a.F.A[0]) + e[1] * fabs(a.F.A[1]) + e[2] * fabs(a.F.A[2]);
                if (ca + ea < qa0 || ca - ea > qa1)
                    continue;
                const double cb = c[0] * a.F.B[0] + c[1] * a.F.B[1] + c[2] * a.F.B[2];
                const double eb = e[0] * fabs(a.F.B[0]) + e[1] * fabs(a.F.B[1]) + e[2] * fabs(a.F.B[2]);
                if (cb + eb < qb0 || cb - eb > qb1)
                    continue;
                const double chh = c[0] * a.F.L[0] + c[1] * a.F.L[1] + c[2] * a.F.L[2];
                const double eh = e[0] * fabs(a.F.L[0]) + e[1] * fabs(a.F.L[1]) + e[2] * fabs(a.F.L[2]);
                if (chh + eh < qh) // wholly below the lowest ray origin
                    continue;
                // ... and against the receiver's FOOTPRINT, not only its box: a node wholly beyond one edge of the projected triangle -- pushed out by the
                // offset box and by the drift a ray can have when it has climbed to the node's top, exactly the half-planes lit::may_occlude clips every
                // triangle with -- holds nothing that can matter (may_occlude would clip each of its triangles to nothing; cover_mask finds no sample under
                // them).  For a large receiver the box is twice the triangle.
                if constexpr (NEB_SUN_EDGE_CULL) {
                    const double rho = ((chh + eh) - R.h_min + a.F.margin) * a.F.tau + a.F.margin;
                    bool outside = false;
#pragma unroll
                    for (int ed = 0; ed < 3; ++ed) {
                        const double na = R.en_a[ed], nb = R.en_b[ed];
                        const double lim = R.en_c[ed] + R.en_off[ed] + rho * (fabs(na) + fabs(nb));
                        // least value of na * a + nb * b over the node's box (the box's sun-space extents ea, eb bound every corner)
                        outside = outside || (na * ca + nb * cb - (fabs(na) * ea + fabs(nb) * eb) > lim);
                    }
                    if (outside)
                        continue;
                }
                if (ch[q] < 0) {
                    queue_leaf(ch[q]);
                } else if (node == kTravDone) {
                    node = ch[q];
                } else if (sp < 64) {
                    stack[sp++] = ch[q];
                } else { // cannot happen (neb_gi_build_bvh bounds the depth); if it did, no certificate
                    alive = false;
                }
            }
            if (node == kTravDone && sp > 0)
                node = stack[--sp];
        }
        // a node adds at most 4 leaves x 4 triangles
        const bool wave_walks = __ballot(walking) != 0ull;
        if (__ballot(qn > (uint32_t)kSunQueue - 16u) != 0ull || !wave_walks) {
            test_queued();
            if (!wave_walks)
                break;
        }
    }
    return alive;
}

template <int PASS>
__global__ __launch_bounds__(64, PASS == 1 ? NEB_SUN_WAVES : 2) void sun_table_kernel(SunTableArgs a)
{
    static_assert(PASS == 1 || PASS == 2, "pass 1 = lit bits, pass 2 = hints");
    uint32_t ti = blockIdx.x * blockDim.x + threadIdx.x;
    bool mine = ti < a.S.n_tris;
    int primary = 0;
    if constexpr (PASS == 2) {
        mine = ti < (uint32_t)a.counts[3];
        if (mine) {
            const uint32_t entry = a.hint_list[ti]; // triangle | its primary side << 31
            ti = entry & 0x7fffffffu;
            primary = (int)(entry >> 31);
        }
    }
    uint32_t flags = 0;
    bool hinted = false, listed = false;
    __shared__ SunWaveLds<PASS == 2> lds;
    uint32_t visits = 0, tested = 0;
    const unsigned long long clock0 = a.walk_stats ? wall_clock64() : 0ull;
    // (every lane of the wave goes through the walks below, with or without a triangle of its own: the wave tests its lanes' candidates together)
    double v[3][3], gn[3][3];
    float4 r6 = {0.f, 0.f, 0.f, 0.f};
    uint32_t geom = 0;
    bool ok = false;
    if (mine) {
        const float4 t0 = a.S.tris[3 * ti], t1 = a.S.tris[3 * ti + 1], t2 = a.S.tris[3 * ti + 2];
        // the triangle the traverser tests: (v0, v0 + e1, v0 + e2) with e1, e2 as stored
        v[0][0] = t0.x, v[0][1] = t0.y, v[0][2] = t0.z;
        v[1][0] = (double)t0.x + t0.w, v[1][1] = (double)t0.y + t1.x, v[1][2] = (double)t0.z + t1.y;
        v[2][0] = (double)t0.x + t1.z, v[2][1] = (double)t0.y + t1.w, v[2][2] = (double)t0.z + t2.x;
        const float4* rec = a.S.shade + 8 * (size_t)ti;
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        r6 = rec[6];
        geom = __float_as_uint(r6.w) & kGeomMask;
        const DevGeom g = a.S.geoms[geom];
        ok = g.valid; // (a submesh without its attribute streams ends the path at the hit: no shadow ray ever starts there)
        if (ok) {
            const float n[3][3] = {{r0.x, r0.y, r0.z}, {r1.x, r1.y, r1.z}, {r2.x, r2.y, r2.z}};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                // GN = normalize(xform_dir(M, normalize(sum b_i n_i))) lies in the cone of xform_dir(M, n_i): (p, 0) * M, row vectors
                const double x = (double)n[i][0] * g.m[0] + (double)n[i][1] * g.m[3] + (double)n[i][2] * g.m[6];
                const double y = (double)n[i][0] * g.m[1] + (double)n[i][1] * g.m[4] + (double)n[i][2] * g.m[7];
                const double z = (double)n[i][0] * g.m[2] + (double)n[i][1] * g.m[5] + (double)n[i][2] * g.m[8];
                const double l = sqrt(x * x + y * y + z * z);
                ok = ok && l > 0.0 && l < 1e30;
                gn[i][0] = x / l, gn[i][1] = y / l, gn[i][2] = z / l;
            }
        }
    }
    if (!ok) { // (something finite for the arithmetic of a lane that only helps)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            v[i][0] = v[i][1] = v[i][2] = 0.0, gn[i][0] = gn[i][1] = 0.0, gn[i][2] = 1.0;
    }
    SunCands cands;
    cands.clear();
    uint32_t hint[kHints];
#pragma unroll
    for (int h = 0; h < kHints; ++h)
        hint[h] = kNoHint;
    if constexpr (PASS == 1) {
        // The side whose rays the hints are for (the one turned to the sun: the shader picks the side of GN the disk sample is on) is the one whose
        // origins sit higher over the plane.
        bool valid0 = false, valid1 = false, alive0 = false, alive1 = false;
        double clo0 = 0.0, clo1 = 0.0;
#pragma unroll 1
        for (int sd = 0; sd < 2; ++sd) {
            lit::Receiver R;
            lit::make_receiver(a.F, v, gn, sd ? -1 : +1, R);
            const bool valid = ok && R.valid;
            // its own triangle first (the walk would meet it: the receiver lies in its own column) -- the side turned away from the sun ends here
            bool alive = valid && !lit::may_occlude(a.F, R, R.t);
            alive = sun_walk_column<false>(a, R, alive, ti, cands, lds, visits, tested);
            if (sd == 0)
                valid0 = valid, alive0 = alive, clo0 = R.c_lo;
            else
                valid1 = valid, alive1 = alive, clo1 = R.c_lo;
        }
        primary = (valid0 && valid1) ? (clo1 > clo0 ? 1 : 0) : (valid1 ? 1 : 0);
        flags = (alive0 ? 1u : 0u) | (alive1 ? 2u : 0u);
        if (mine) {
            float4 w6 = r6;
            w6.w = __uint_as_float(geom | (flags << kLitShift));
            a.shade[8 * (size_t)ti + 6] = w6;
        }
        listed = mine && (primary ? (valid1 && !alive1) : (valid0 && !alive0));
    } else { // (a lit side needs no hints: every ray of it is answered by the lit bit)
        lit::Receiver R;
        lit::make_receiver(a.F, v, gn, primary ? -1 : +1, R);
        const bool valid = ok && R.valid;
        sun_walk_column<true>(a, R, valid, ti, cands, lds, visits, tested);
        if (valid)
            cands.choose(hint);
    }
    if (mine) {
        // r7 = {PrimitiveIndex, then kHints x 23-bit triangle indices and the side they are for: see pack_hints}
        float4 w7 = a.S.shade[8 * (size_t)ti + 7];
        pack_hints(hint, (uint32_t)primary, w7);
        a.shade[8 * (size_t)ti + 7] = w7;
        hinted = hint[0] != kNoHint;
    }
    const unsigned long long m0 = __ballot(PASS == 1 && (flags & 1u) != 0u), m1 = __ballot(PASS == 1 && (flags & 2u) != 0u), m2 = __ballot(hinted);
    if (a.walk_stats) { // diagnostics (tools/sun_table_walks.py): per pass {node visits, longest walk, candidates tested, shadowing candidates, wave time sum / max in 10-ns ticks}
        unsigned long long vs = visits, mx = visits, ts = tested;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            vs += __shfl_down(vs, off);
            mx = max(mx, (unsigned long long)__shfl_down(mx, off));
            ts += __shfl_down(ts, off);
        }
        if ((threadIdx.x & 63u) == 0) {
            const unsigned long long dt = wall_clock64() - clock0;
            unsigned long long* w = a.walk_stats + 6 * (PASS - 1);
            atomicAdd(w + 0, vs);
            atomicMax(w + 1, mx);
            atomicAdd(w + 2, ts);
            atomicAdd(w + 3, dt);
            atomicMax(w + 4, dt);
            atomicAdd(w + 5, 1ull);
        }
    }
    const unsigned long long m3 = __ballot(listed);
    unsigned long long base = 0;
    if ((threadIdx.x & 63u) == 0) {
        if (m0)
            atomicAdd(a.counts + 0, (unsigned long long)__popcll(m0));
        if (m1)
            atomicAdd(a.counts + 1, (unsigned long long)__popcll(m1));
        if (m2)
            atomicAdd(a.counts + 2, (unsigned long long)__popcll(m2));
        if (m3)
            base = atomicAdd(a.counts + 3, (unsigned long long)__popcll(m3));
    }
    if (PASS == 1 && m3) { // one atomic per wave: the wave's listed triangles go to consecutive slots
        base = (unsigned long long)__shfl((int)(uint32_t)base, 0);
        if (listed)
            a.hint_list[(uint32_t)base + (uint32_t)__popcll(m3 & ((1ull << (threadIdx.x & 63u)) - 1ull))] = ti | ((uint32_t)primary << 31);
    }
}

__global__ void sun_table_clear_kernel(float4* shade, uint32_t n)
{
    const uint32_t ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti < n) {
        float4* p = shade + 8 * (size_t)ti + 6;
        p->w = __uint_as_float(__float_as_uint(p->w) & kGeomMask);
        const uint32_t none[kHints] = {kNoHint, kNoHint, kNoHint, kNoHint};
        pack_hints(none, 0u, p[1]);
    }
}

// behind a launch that rewrites the flags: dispatches on other streams order themselves after it (gi_sun_table_order)
static hipError_t mark_rewrite(GiState* g, hipStream_t stream)
{
    if (!g->sun_table_event)
        if (hipError_t e = hipEventCreateWithFlags(&g->sun_table_event, hipEventDisableTiming); e != hipSuccess)
            return e;
    g->sun_table_stream = stream;
    g->sun_table_event_pending = true;
    return hipEventRecord(g->sun_table_event, stream);
}

// Every dispatch calls this after gi_sun_table_update: work on `stream` that reads the shading records runs after the last rewrite of
// their flags, whichever stream that was enqueued on.
hipError_t gi_sun_table_order(GiState* g, hipStream_t stream)
{
    if (!g->sun_table_event_pending || stream == g->sun_table_stream)
        return hipSuccess; // (same stream: ordered by the stream itself)
    const hipError_t q = hipEventQuery(g->sun_table_event);
    if (q == hipSuccess) {
        g->sun_table_event_pending = false;
        return hipSuccess;
    }
    (void)hipGetLastError(); // (hipErrorNotReady is an answer, not a failure)
    return hipStreamWaitEvent(stream, g->sun_table_event, 0);
}

// Brings the table in the shading records up to date with (scene, sun) -- or clears it when the option is off.  Enqueue only.
hipError_t gi_sun_table_update(GiState* g, const neb_gi_constants& c, hipStream_t stream)
{
    if (!g->built || g->view.n_tris == 0 || !g->view.shade)
        return hipSuccess;
    const float key[4] = {c.sunLightDirection[0], c.sunLightDirection[1], c.sunLightDirection[2], c.sunTanHalfAngle};
    const bool want = g->sun_table;
    // The flags live in the shading records every dispatch reads.  With two dispatches in flight ("gi_defer_resolve" = 2) the other one may still
    // be running on another stream: whatever rewrites the flags first waits for the device (a change of sun or of the option -- never a steady frame).
    // The same for a host that moved to another stream since its last dispatch: that one's shade pass may still be reading the flags.
    auto quiesce = [&]() { return (g->defer_resolve == 2 || (g->last_dispatch_stream_set && g->last_dispatch_stream != stream)) ? hipDeviceSynchronize() : hipSuccess; };
    if (!want) {
        if (g->sun_table_state != 0) {
            if (hipError_t e = quiesce(); e != hipSuccess)
                return e;
            hipLaunchKernelGGL(sun_table_clear_kernel, dim3((g->view.n_tris + 255) / 256), dim3(256), 0, stream, const_cast<float4*>(g->view.shade), g->view.n_tris);
            if (hipError_t em = mark_rewrite(g, stream); em != hipSuccess)
                return em;
            g->sun_table_state = 0;
        }
        return hipGetLastError();
    }
    if (g->sun_table_state != 0 && !memcmp(key, g->sun_table_key, sizeof(key))) {
        g->sun_table_state = 1; // (back to the sun the flags were built for)
        g->sun_table_age++;
        return hipSuccess;
    }
    // A new sun.  The build takes milliseconds -- twenty frames' worth -- so a sun that is being dragged (a new direction every
    // frame; the reference marks those frames dynamic, src/DeferredRenderer.cpp:169-171) is not chased: the flags in the records
    // stay those of the old sun and are IGNORED (state 2: every shadow ray is traced, as without the table) until the same new sun
    // has been seen on sun_hold consecutive dispatches: two -- or kSunHoldAfterShortLife when the table it replaces lived fewer than kSunTableLife dispatches
    // (a sun that moves in steps of a few frames: a build per step costs more than no table at all; see GiState::sun_hold).  The very first build has nothing
    // to wait for.
    constexpr uint32_t kSunTableLife = 32u, kSunHoldAfterShortLife = 32u;
    if (g->sun_table_state != 0) {
        if (memcmp(key, g->sun_table_pending, sizeof(key)) || g->sun_table_state == 1) { // (state 1: the table's own sun was the last one seen)
            memcpy(g->sun_table_pending, key, sizeof(key));
            g->sun_seen = 0;
            if (g->sun_table_state == 1) // the table has just lost its sun: how long did it serve?
                g->sun_hold = g->sun_hold_option > 0 ? (uint32_t)g->sun_hold_option : (g->sun_table_age < kSunTableLife ? kSunHoldAfterShortLife : 2u);
        }
        g->sun_table_state = 2;
        if (++g->sun_seen < g->sun_hold)
            return hipSuccess;
    }
    const float dd = key[0] * key[0] + key[1] * key[1] + key[2] * key[2];
    // The certificate's slack grows with the scene's coordinates (lit_predicate.h: an ulp there is what fp32 hit points and triangle tests
    // can be off by) while the ray offset stays 1e-2: past +-218 units the slack would eat a quarter of the offset and nothing could be proven
    // with a margin worth the name -- such a scene gets no table (and no 16-ms build), every shadow ray is traced.
    double scene_abs_max = 0.0;
    for (int k = 0; k < 3; ++k)
        scene_abs_max = fmax(scene_abs_max, fmax(fabs((double)g->scene_min[k]), fabs((double)g->scene_max[k])));
    const bool too_large = !lit::margin_usable(lit::margin_for(scene_abs_max));
    // ... and a disk beyond kMaxSunTanHalfAngle (3.4 degrees across; the reference's default is 0.58, src/DeferredRenderer.h:111-125) gets no table either:
    // its columns widen with the disk -- measured on the bench scene, 5 degrees: 8 ms of build for 32 % of the queries and no gain, 30 degrees: 70 ms for 7 %.
    constexpr float kMaxSunTanHalfAngle = 0.03f;
    if (too_large || !(dd > 0.0f) || !(dd < 1e30f) || !(key[3] >= 0.0f) || !(key[3] <= kMaxSunTanHalfAngle)) { // no usable sun / scale: no certificate, every ray is traced
        if (g->sun_table_state != 0) {
            if (hipError_t e = quiesce(); e != hipSuccess)
                return e;
            hipLaunchKernelGGL(sun_table_clear_kernel, dim3((g->view.n_tris + 255) / 256), dim3(256), 0, stream, const_cast<float4*>(g->view.shade), g->view.n_tris);
            if (hipError_t em = mark_rewrite(g, stream); em != hipSuccess)
                return em;
        }
        g->sun_table_state = 0;
        return hipGetLastError();
    }
    if (!g->d_sun_counts) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, 24 * sizeof(unsigned long long));
        if (e != hipSuccess)
            return e;
        g->allocs.push_back(p);
        g->d_sun_counts = (unsigned long long*)p;
    }
    hipError_t e = quiesce();
    if (e != hipSuccess)
        return e;
    e = hipMemsetAsync(g->d_sun_counts, 0, 24 * sizeof(unsigned long long), stream);
    if (e != hipSuccess)
        return e;
    SunTableArgs a;
    a.S = g->view;
    lit::make_frame(key, key[3], scene_abs_max, a.F);
    a.box_pad = fmax(1e-5, 1.1920928955078125e-7 * scene_abs_max);
    double hmax = -1e300;
    for (int k = 0; k < 8; ++k) {
        const double p[3] = {(k & 1) ? g->scene_max[0] : g->scene_min[0], (k & 2) ? g->scene_max[1] : g->scene_min[1],
                             (k & 4) ? g->scene_max[2] : g->scene_min[2]};
        hmax = fmax(hmax, p[0] * a.F.L[0] + p[1] * a.F.L[1] + p[2] * a.F.L[2]);
    }
    a.scene_hmax = hmax + fmax(1e-3, a.F.margin);
    a.shade = const_cast<float4*>(g->view.shade);
    a.counts = g->d_sun_counts;
    a.walk_stats = getenv("NEB_SUN_WALK_STATS") ? g->d_sun_counts + 8 : nullptr;
    if (g->sun_hint_list_cap < g->view.n_tris) { // (a scene rebuild may have changed the reference count)
        void* p = nullptr;
        if (hipError_t em = hipMalloc(&p, (size_t)g->view.n_tris * sizeof(uint32_t)); em != hipSuccess)
            return em;
        g->allocs.push_back(p);
        g->d_sun_hint_list = (uint32_t*)p;
        g->sun_hint_list_cap = g->view.n_tris;
    }
    a.hint_list = g->d_sun_hint_list;
    for (hipEvent_t& ev : g->sun_build_ev)
        if (!ev)
            if (hipError_t em = hipEventCreate(&ev); em != hipSuccess)
                return em;
    if (hipError_t em = hipEventRecord(g->sun_build_ev[0], stream); em != hipSuccess)
        return em;
    hipLaunchKernelGGL(sun_table_kernel<1>, dim3((g->view.n_tris + 63) / 64), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(sun_table_kernel<2>, dim3((g->view.n_tris + 63) / 64), dim3(64), 0, stream, a); // (waves past the list's end leave at once)
    if (hipError_t em = hipEventRecord(g->sun_build_ev[1], stream); em != hipSuccess)
        return em;
    if (hipError_t em = mark_rewrite(g, stream); em != hipSuccess)
        return em;
    memcpy(g->sun_table_key, key, sizeof(key));
    g->sun_table_state = 1;
    g->sun_table_age = 0;
    g->sun_seen = 0;
    g->sun_table_builds++;
    g->tail_phase = g->tail_tune ? 1 : 0; // (a new table: which pass takes what it leaves is measured again)
    g->tail_sorted = false;
    return hipGetLastError();
}

} // namespace neb

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
// and the bench frame).  Rebuilt (15 ms at 262 k triangles) when sunLightDirection / sunTanHalfAngle have changed AND held for two
// dispatches, or the scene is rebuilt; a sun that moves every frame -- the reference marks such frames dynamic,
// src/DeferredRenderer.cpp:169-171 -- is traced the plain way meanwhile.
//
// This takes the place of the sun-space height map sized in round 3: a height map has to resolve the 1e-2 ray offset against the
// slope of every lit surface (a 4096^2 map for a floor, and a max-plus filter for the cone), costs a scattered fetch per ray, is only
// as exact as its cells, and says nothing about the occluded rays; the per-triangle form uses the exact geometry.
// (What the lit bits alone are worth: 8 us -- the unoccluded rays of an open court leave the tree after 3 node visits on average;
// the hints and the compaction of what is left, gi.hip, are the other 77.)
#include <cstdlib>

#include "gi_device.h"
#include "lit_predicate.h"

namespace neb {

struct SunTableArgs {
    SceneView S;
    lit::Frame F;
    double scene_hmax; // highest point of the scene along L
    double box_pad;    // padding of the node boxes (floats) in the double-precision cull: an ulp of the scene's largest coordinate, at least 1e-5
    float4* shade;     // the shading records (writable view of S.shade)
    unsigned long long* counts; // [0] sides proven lit (+), [1] (-), [2] triangles with an occluder hint, [3] triangles listed for the hint pass, [4] ... for pass 3
    uint32_t* hint_list;        // two-pass build: the triangles whose primary side is not proven lit (pass 1 appends, pass 2 reads)
    unsigned long long* walk_stats; // diagnostics or null: [3 * PASS + {0, 1, 2}] = {node visits, longest walk, walks over 1000 visits}
    uint32_t* retry_list;       // pass 1: the triangles whose budget of node visits ran out with a side still unproven (pass 3 walks them to the end)
    uint32_t lit_budget;        // pass 1: node visits per triangle, 0 = unlimited (then no pass 3)
    uint32_t hint_budget;       // pass 2: candidate triangles (those that shadow at least one sample origin) looked at per receiver, 0 = all
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

// PASS 0: lit bits and hints in one walk of the whole column (rounds 4: 15.5 ms at 262 k triangles; kept as the A/B arm, NEB_SUN_TABLE_PASSES=1).
// PASS 1 + PASS 2 (round 5, the default): the lit bits first -- a walk that ENDS as soon as every valid side has met something it cannot rule out
// (most unlit triangles do within a few leaves) -- and the triangles whose primary side is left unproven appended to a list; then the hint walk
// for those only, in dense waves, over the primary side's own column, ending once the four best candidates cover all 28 sample origins.
// Same certificate, same lit bits; the hints may differ from the one-pass choice (they are only ever hints: tried with the traverser's own test).
#ifndef NEB_SUN_EDGE_CULL
#define NEB_SUN_EDGE_CULL 1
#endif
constexpr uint32_t kSunLitBudget = 0u; // (see SunTableArgs::lit_budget; NEB_SUN_LIT_BUDGET overrides)
constexpr uint32_t kSunHintBudget = 0u; // (see SunTableArgs::hint_budget; NEB_SUN_HINT_BUDGET overrides)
template <int PASS>
__global__ __launch_bounds__(64) void sun_table_kernel(SunTableArgs a)
{
    uint32_t ti = blockIdx.x * blockDim.x + threadIdx.x;
    bool mine = ti < a.S.n_tris;
    if constexpr (PASS == 2) {
        mine = ti < (uint32_t)a.counts[3];
        if (mine)
            ti = a.hint_list[ti];
    }
    if constexpr (PASS == 3) { // the lit pass of the triangles pass 1 ran out of budget on: the same walk to its end, in dense waves
        mine = ti < (uint32_t)a.counts[4];
        if (mine)
            ti = a.retry_list[ti];
    }
    constexpr bool kLit = PASS == 0 || PASS == 1 || PASS == 3, kHint = PASS == 0 || PASS == 2;
    bool undecided = false;
    uint32_t flags = 0;
    uint32_t hint[kHints];
    for (int h = 0; h < kHints; ++h)
        hint[h] = kNoHint;
    bool hinted = false, listed = false;
    uint32_t visits_total = 0;
    if (mine) {
        const float4 t0 = a.S.tris[3 * ti], t1 = a.S.tris[3 * ti + 1], t2 = a.S.tris[3 * ti + 2];
        // the triangle the traverser tests: (v0, v0 + e1, v0 + e2) with e1, e2 as stored
        const double v[3][3] = {{t0.x, t0.y, t0.z},
                                {(double)t0.x + t0.w, (double)t0.y + t1.x, (double)t0.z + t1.y},
                                {(double)t0.x + t1.z, (double)t0.y + t1.w, (double)t0.z + t2.x}};
        const float4* rec = a.S.shade + 8 * (size_t)ti;
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r6 = rec[6];
        const uint32_t geom = __float_as_uint(r6.w) & kGeomMask;
        const DevGeom g = a.S.geoms[geom];
        lit::Receiver R[2];
        R[0].valid = R[1].valid = false;
        if (g.valid) { // (a submesh without its attribute streams ends the path at the hit: no shadow ray ever starts there)
            const float n[3][3] = {{r0.x, r0.y, r0.z}, {r1.x, r1.y, r1.z}, {r2.x, r2.y, r2.z}};
            double gn[3][3];
            bool ok = true;
            for (int i = 0; i < 3; ++i) {
                // GN = normalize(xform_dir(M, normalize(sum b_i n_i))) lies in the cone of xform_dir(M, n_i): (p, 0) * M, row vectors
                const double x = (double)n[i][0] * g.m[0] + (double)n[i][1] * g.m[3] + (double)n[i][2] * g.m[6];
                const double y = (double)n[i][0] * g.m[1] + (double)n[i][1] * g.m[4] + (double)n[i][2] * g.m[7];
                const double z = (double)n[i][0] * g.m[2] + (double)n[i][1] * g.m[5] + (double)n[i][2] * g.m[8];
                const double l = sqrt(x * x + y * y + z * z);
                ok = ok && l > 0.0 && l < 1e30;
                gn[i][0] = x / l, gn[i][1] = y / l, gn[i][2] = z / l;
            }
            if (ok) {
                lit::make_receiver(a.F, v, gn, +1, R[0]);
                lit::make_receiver(a.F, v, gn, -1, R[1]);
            }
        }
        bool alive[2] = {R[0].valid, R[1].valid};
        // Occluder hints for the side the rays of this triangle start on (the one turned to the sun: the shader picks the side of
        // GN the disk sample is on): the two triangles whose shadow covers most of the footprint.  gi_shade_kernel tries them
        // with the traverser's own triangle test before it emits a shadow ray -- any hit answers "occluded" exactly.
        const int primary = (R[0].valid && R[1].valid) ? (R[1].c_lo > R[0].c_lo ? 1 : 0) : (R[1].valid ? 1 : 0);
        // the kCand candidates that cover most sample origins so far (cover_mask), best first
        constexpr int kCand = 8;
        uint32_t cand_tri[kCand], cand_mask[kCand];
        int cand_pop[kCand];
        for (int k = 0; k < kCand; ++k)
            cand_tri[k] = kNoHint, cand_mask[k] = 0u, cand_pop[k] = 0;
        if (R[0].valid || R[1].valid) {
            // the column both sides' rays can sweep, in sun coordinates
            double qa[2] = {1e300, -1e300}, qb[2] = {1e300, -1e300}, qh = 1e300;
            for (int s = 0; s < 2; ++s)
                if (R[s].valid) {
                    if (PASS == 2 && s != primary)
                        continue; // (the hint pass: the primary side's own column)
                    const double reach = (a.scene_hmax - R[s].h_min + a.F.margin) * a.F.tau + a.F.margin;
                    qa[0] = fmin(qa[0], R[s].bb_a[0] - reach), qa[1] = fmax(qa[1], R[s].bb_a[1] + reach);
                    qb[0] = fmin(qb[0], R[s].bb_b[0] - reach), qb[1] = fmax(qb[1], R[s].bb_b[1] + reach);
                    qh = fmin(qh, R[s].h_min - a.F.margin);
                }
            int stack[64];
            int sp = 0;
            int node = a.S.root;
            uint32_t cands_seen = 0;
            auto test_leaf = [&](int code) {
                const uint32_t c = (uint32_t)~code, first = c >> 2, count = (c & 3u) + 1u;
                for (uint32_t k = 0; k < count; ++k) {
                    const uint32_t tj = first + k;
                    const float4 o0 = a.S.tris[3 * tj], o1 = a.S.tris[3 * tj + 1], o2 = a.S.tris[3 * tj + 2];
                    const double w[3][3] = {{o0.x, o0.y, o0.z},
                                            {(double)o0.x + o0.w, (double)o0.y + o1.x, (double)o0.z + o1.y},
                                            {(double)o0.x + o1.z, (double)o0.y + o1.w, (double)o0.z + o2.x}};
                    lit::Tri O;
                    for (int i = 0; i < 3; ++i)
                        lit::to_sun(a.F, w[i], O.a[i], O.b[i], O.h[i]);
                    if (kLit)
                        for (int s = 0; s < 2; ++s)
                            if (alive[s] && lit::may_occlude(a.F, R[s], O))
                                alive[s] = false;
                    if (kHint && R[primary].valid && tj != ti) {
                        const uint32_t m = lit::cover_mask(R[primary], O);
                        const int pc = __popc(m);
                        cands_seen += pc ? 1u : 0u;
                        if (pc > cand_pop[kCand - 1]) { // insertion into the sorted candidate list
                            int at = kCand - 1;
                            while (at > 0 && cand_pop[at - 1] < pc) {
                                cand_tri[at] = cand_tri[at - 1], cand_mask[at] = cand_mask[at - 1], cand_pop[at] = cand_pop[at - 1];
                                --at;
                            }
                            cand_tri[at] = tj, cand_mask[at] = m, cand_pop[at] = pc;
                        }
                    }
                }
            };
            if (node < 0) { // a scene of one leaf
                test_leaf(node);
                node = kTravDone;
            }
            constexpr uint32_t kAllSamples = (1u << lit::kCoverSamples) - 1u;
            uint32_t visits = 0;
            auto walk_done = [&]() {
                if (PASS == 1 && a.lit_budget && visits >= a.lit_budget && (alive[0] || alive[1])) {
                    undecided = true; // out of budget with something still unproven: pass 3 walks this triangle to the end
                    return true;
                }
                if (PASS == 1 || PASS == 3)
                    return !alive[0] && !alive[1]; // nothing left to prove
                if (PASS == 2) // the four best cover every sample origin -- or the receiver has seen its budget of shadowing triangles
                    return ((cand_mask[0] | cand_mask[1] | cand_mask[2] | cand_mask[3]) & kAllSamples) == kAllSamples || (a.hint_budget && cands_seen >= a.hint_budget);
                return false; // (one pass: the whole column -- a side that is already known to be shadowed still wants its hints)
            };
            while (node != kTravDone && !walk_done()) {
                ++visits;
                ++visits_total;
                const Bvh4Node nd = a.S.nodes[node];
                const int ch[4] = {nd.child.x, nd.child.y, nd.child.z, nd.child.w};
                node = kTravDone;
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
                    if (ca + ea < qa[0] || ca - ea > qa[1])
                        continue;
                    const double cb = c[0] * a.F.B[0] + c[1] * a.F.B[1] + c[2] * a.F.B[2];
                    const double eb = e[0] * fabs(a.F.B[0]) + e[1] * fabs(a.F.B[1]) + e[2] * fabs(a.F.B[2]);
                    if (cb + eb < qb[0] || cb - eb > qb[1])
                        continue;
                    const double chh = c[0] * a.F.L[0] + c[1] * a.F.L[1] + c[2] * a.F.L[2];
                    const double eh = e[0] * fabs(a.F.L[0]) + e[1] * fabs(a.F.L[1]) + e[2] * fabs(a.F.L[2]);
                    if (chh + eh < qh) // wholly below the lowest ray origin
                        continue;
                    // (round 5) ... and against the receiver's FOOTPRINT, not only its box: a node wholly beyond one edge of the projected triangle --
                    // pushed out by the offset box and by the drift a ray can have when it has climbed to the node's top, exactly the half-planes
                    // lit::may_occlude clips every triangle with -- holds nothing that can matter to that side (may_occlude would clip each of its
                    // triangles to nothing; cover_mask finds no sample under them).  For a large receiver the box is twice the triangle.
                    if constexpr (NEB_SUN_EDGE_CULL) {
                        bool keep = false;
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd) {
                            if (!R[sd].valid || (PASS == 2 && sd != primary) || (kLit && !kHint && !alive[sd]))
                                continue;
                            const double rho = ((chh + eh) - R[sd].h_min + a.F.margin) * a.F.tau + a.F.margin;
                            bool outside = false;
#pragma unroll
                            for (int ed = 0; ed < 3; ++ed) {
                                const double na = R[sd].en_a[ed], nb = R[sd].en_b[ed];
                                const double lim = R[sd].en_c[ed] + R[sd].en_off[ed] + rho * (fabs(na) + fabs(nb));
                                // least value of na * a + nb * b over the node's box (the box's sun-space extents ea, eb bound every corner)
                                outside = outside || (na * ca + nb * cb - (fabs(na) * ea + fabs(nb) * eb) > lim);
                            }
                            keep = keep || !outside;
                        }
                        if (!keep)
                            continue;
                    }
                    if (ch[q] < 0) {
                        test_leaf(ch[q]);
                    } else if (node == kTravDone) {
                        node = ch[q];
                    } else if (sp < 64) {
                        stack[sp++] = ch[q];
                    } else { // cannot happen (neb_gi_build_bvh bounds the depth); if it did, no certificate
                        alive[0] = alive[1] = false;
                    }
                }
                if (node == kTravDone && sp > 0)
                    node = stack[--sp];
            }
        }
        flags = undecided ? 0u : (alive[0] ? 1u : 0u) | (alive[1] ? 2u : 0u);
        if constexpr (PASS != 2) {
            float4 w6 = r6;
            w6.w = __uint_as_float(geom | (flags << kLitShift));
            a.shade[8 * (size_t)ti + 6] = w6;
        }
        listed = (PASS == 1 || PASS == 3) && !undecided && R[primary].valid && !alive[primary];
        // greedy cover: up to kHints candidates, each the one that adds most samples not yet covered
        if (kHint && (PASS == 2 || !alive[primary])) { // (a lit side needs no hints: every ray of it is answered by the lit bit)
            uint32_t covered = 0u;
            for (int h = 0; h < kHints; ++h) {
                int best = -1, gain = 0;
                for (int k = 0; k < kCand; ++k) {
                    const int gk = cand_tri[k] != kNoHint ? __popc(cand_mask[k] & ~covered) : 0;
                    if (gk > gain)
                        gain = gk, best = k;
                }
                if (best < 0)
                    break;
                hint[h] = cand_tri[best];
                covered |= cand_mask[best];
                cand_tri[best] = kNoHint;
            }
        }
        // r7 = {PrimitiveIndex, then kHints x 21-bit triangle indices and the side they are for: see pack_hints}
        float4 w7 = rec[7];
        pack_hints(hint, (uint32_t)primary, w7);
        a.shade[8 * (size_t)ti + 7] = w7;
        hinted = hint[0] != kNoHint;
    }
    const unsigned long long m0 = __ballot(kLit && (flags & 1u) != 0u), m1 = __ballot(kLit && (flags & 2u) != 0u), m2 = __ballot(hinted);
    if (a.walk_stats) { // diagnostics (tools/sun_table_walks.py): how long the walks are -- per pass {sum of node visits, longest walk, walks over 1000 visits}
        unsigned long long v = visits_total, mx = visits_total, big = visits_total > 1000u ? 1ull : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_down(v, off);
            mx = max(mx, (unsigned long long)__shfl_down(mx, off));
            big += __shfl_down(big, off);
        }
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(a.walk_stats + 3 * PASS + 0, v);
            atomicMax(a.walk_stats + 3 * PASS + 1, mx);
            atomicAdd(a.walk_stats + 3 * PASS + 2, big);
        }
    }
    const unsigned long long m3 = __ballot(listed), m4 = __ballot(undecided);
    unsigned long long base = 0, base4 = 0;
    if ((threadIdx.x & 63u) == 0) {
        if (m0)
            atomicAdd(a.counts + 0, (unsigned long long)__popcll(m0));
        if (m1)
            atomicAdd(a.counts + 1, (unsigned long long)__popcll(m1));
        if (m2)
            atomicAdd(a.counts + 2, (unsigned long long)__popcll(m2));
        if (m3)
            base = atomicAdd(a.counts + 3, (unsigned long long)__popcll(m3));
        if (m4)
            base4 = atomicAdd(a.counts + 4, (unsigned long long)__popcll(m4));
    }
    if (PASS == 1 && m4) {
        base4 = (unsigned long long)__shfl((int)(uint32_t)base4, 0);
        if (undecided)
            a.retry_list[(uint32_t)base4 + (uint32_t)__popcll(m4 & ((1ull << (threadIdx.x & 63u)) - 1ull))] = ti;
    }
    if ((PASS == 1 || PASS == 3) && m3) { // one atomic per wave: the wave's listed triangles go to consecutive slots
        base = (unsigned long long)__shfl((int)(uint32_t)base, 0);
        if (listed)
            a.hint_list[(uint32_t)base + (uint32_t)__popcll(m3 & ((1ull << (threadIdx.x & 63u)) - 1ull))] = ti;
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
        return hipSuccess;
    }
    // A new sun.  The build takes milliseconds -- twenty frames' worth -- so a sun that is being dragged (a new direction every
    // frame; the reference marks those frames dynamic, src/DeferredRenderer.cpp:169-171) is not chased: the flags in the records
    // stay those of the old sun and are IGNORED (state 2: every shadow ray is traced, as without the table) until the same new sun
    // has been seen on two consecutive dispatches.  The very first build has nothing to wait for.
    if (g->sun_table_state != 0 && memcmp(key, g->sun_table_pending, sizeof(key))) {
        memcpy(g->sun_table_pending, key, sizeof(key));
        g->sun_table_state = 2;
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
    if (too_large || !(dd > 0.0f) || !(dd < 1e30f) || !(key[3] >= 0.0f) || !(key[3] < 1.0f)) { // no usable sun / scale: no certificate, every ray is traced
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
    static const bool one_pass = getenv("NEB_SUN_TABLE_PASSES") && atoi(getenv("NEB_SUN_TABLE_PASSES")) == 1; // (A/B arm: round 4's single walk)
    if (one_pass) {
        a.hint_list = a.retry_list = nullptr;
        a.hint_budget = a.lit_budget = 0;
        hipLaunchKernelGGL(sun_table_kernel<0>, dim3((g->view.n_tris + 63) / 64), dim3(64), 0, stream, a);
    } else {
        if (g->sun_hint_list_cap < g->view.n_tris) { // (a scene rebuild may have changed the reference count)
            void* p = nullptr;
            if (hipError_t em = hipMalloc(&p, 2 * (size_t)g->view.n_tris * sizeof(uint32_t)); em != hipSuccess)
                return em;
            g->allocs.push_back(p);
            g->d_sun_hint_list = (uint32_t*)p;
            g->sun_hint_list_cap = g->view.n_tris;
        }
        a.hint_list = g->d_sun_hint_list;
        a.retry_list = g->d_sun_hint_list + g->view.n_tris; // (the second half of the same allocation)
        static const uint32_t lit_budget = getenv("NEB_SUN_LIT_BUDGET") ? (uint32_t)atoi(getenv("NEB_SUN_LIT_BUDGET")) : kSunLitBudget;
        a.lit_budget = lit_budget;
        static const uint32_t budget = getenv("NEB_SUN_HINT_BUDGET") ? (uint32_t)atoi(getenv("NEB_SUN_HINT_BUDGET")) : kSunHintBudget;
        a.hint_budget = budget;
        hipLaunchKernelGGL(sun_table_kernel<1>, dim3((g->view.n_tris + 63) / 64), dim3(64), 0, stream, a);
        if (a.lit_budget)
            hipLaunchKernelGGL(sun_table_kernel<3>, dim3((g->view.n_tris + 63) / 64), dim3(64), 0, stream, a);
        hipLaunchKernelGGL(sun_table_kernel<2>, dim3((g->view.n_tris + 63) / 64), dim3(64), 0, stream, a); // (waves past the list's end leave at once)
    }
    if (hipError_t em = mark_rewrite(g, stream); em != hipSuccess)
        return em;
    memcpy(g->sun_table_key, key, sizeof(key));
    g->sun_table_state = 1;
    g->sun_table_builds++;
    return hipGetLastError();
}

} // namespace neb

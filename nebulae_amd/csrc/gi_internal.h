// gi_internal.h -- scene tables, BVH node layouts and the GI state shared by gi_build.hip (scene upload, BVH build)
// and gi.hip (wavefront kernels, C ABI of the trace).  Tuning knobs are -D macros with the measured defaults.
#pragma once
#include <cmath>
#include <cstdio>
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
    // When the material has all three maps and they share one size, their bilinear footprints are stored interleaved, one
    // 32-byte entry per texel position: the 4 texels of the footprint x {albedo.rgb, normal.rgb, roughness, metalness}
    // -- the three filtered fetches of a hit (same uv) are then ONE 32-byte access.  bundle_w == 0: not bundled.
    uint32_t bundle;   // first entry, in 32-byte units, into SceneView::bundles
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

// 128-byte BVH4 node produced by collapsing the binary SAH tree (SoA per axis: one float4 per bound and axis).
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
// The same tree in 64 bytes per node, for the rays that only ask "is anything in the way" (any-hit: sun visibility).
// The child boxes are 8-bit offsets from the node's own (padded) lower corner in units of scale = extent / 255, rounded
// OUTWARDS with margin (quantise_nodes_kernel, gi_build.hip, says exactly what is proven): a decoded box contains the exact
// one, so a ray visits the nodes an exact walk would plus possibly more, and the triangle tests behind them are the exact
// ones -- up to rays that graze a box face at the rounding level of the folded slab arithmetic, where any two fp32
// traversals may differ.  Child c of axis a: byte c of qlo[a] / qhi[a]; an unused
// slot is the inverted box 255 / 0 (scale > 0 always).  Four 16-byte loads per node visit instead of seven, and half
// the cache footprint: the traversal passes are bound by the texture-address unit.
struct Bvh4NodeQ {
    float ox, oy, oz, sx; //  0
    float sy, sz;         // 16
    uint32_t qlox, qloy;
    uint32_t qloz, qhix, qhiy, qhiz; // 32
    int4 child;           // 48: as Bvh4Node::child
};
static_assert(sizeof(Bvh4NodeQ) == 64, "quantised node layout");
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
#ifndef NEB_FAST_SHADE
#define NEB_FAST_SHADE 1 // gi_shade_kernel uses the 1-ulp hardware rcp / rsq / sqrt (see fdiv)
#endif
constexpr bool kFastShade = NEB_FAST_SHADE != 0;
#ifndef NEB_MAX_LEAF_TRIS
#define NEB_MAX_LEAF_TRIS 2 // 1..4 (the leaf code keeps count - 1 in two bits); measured 1/2/3/4: 1407 / 1390 / 1403 / 1500 us of GI per 1080p frame
#endif
constexpr int kMaxLeafTris = NEB_MAX_LEAF_TRIS;
// The geometry word of a triangle's shading record (r6.w) carries the sun-visibility table in its two top bits
// (gi_sun_table.hip): bit kLitShift = every shadow ray leaving the triangle on its +GN side is unoccluded, bit kLitShift + 1 = -GN side.
constexpr uint32_t kLitShift = 30u, kGeomMask = (1u << kLitShift) - 1u;
// ... and r7.yzw = occluder hints: the (up to) kHints triangles most likely to shadow rays that start on this one, as 23-bit
// indices (scenes of up to 8.4 M triangles; a triangle whose index does not fit is simply no candidate: gi_sun_table.hip) + the side
// (0: +GN, 1: -GN) they were chosen for.  Bits 0-22 hint 0, 23-45 hint 1, 46-68 hint 2, 69-91 hint 3, bit 95 the side.
// (Round 4 had 21 bits: in leaf order the top of a scene -- what shadows a sun ray -- sorts LAST, so at 2.6 M triangles nearly every
// occluder was out of range and the hints answered nothing.)
constexpr int kHints = 4;
constexpr uint32_t kNoHint = 0x7fffffu;
__host__ __device__ inline void pack_hints(const uint32_t hint[kHints], uint32_t side, float4& r7)
{
    uint32_t h[kHints];
    for (int k = 0; k < kHints; ++k)
        h[k] = hint[k] < kNoHint ? hint[k] : kNoHint;
    const unsigned long long lo = (unsigned long long)h[0] | ((unsigned long long)h[1] << 23) | ((unsigned long long)(h[2] & 0x3ffffu) << 46);
    union { uint32_t u; float f; } y, z, w;
    y.u = (uint32_t)lo;
    z.u = (uint32_t)(lo >> 32);
    w.u = (h[2] >> 18) | (h[3] << 5) | (side << 31);
    r7.y = y.f, r7.z = z.f, r7.w = w.f;
}
__host__ __device__ inline void unpack_hints(const float4& r7, uint32_t hint[kHints], uint32_t& side)
{
    union { float f; uint32_t u; } y, z, w;
    y.f = r7.y, z.f = r7.z, w.f = r7.w;
    const unsigned long long lo = (unsigned long long)y.u | ((unsigned long long)z.u << 32);
    hint[0] = (uint32_t)lo & kNoHint;
    hint[1] = (uint32_t)(lo >> 23) & kNoHint;
    hint[2] = ((uint32_t)(lo >> 46) | (w.u << 18)) & kNoHint;
    hint[3] = (w.u >> 5) & kNoHint;
    side = w.u >> 31;
}

struct SceneView {
    const float4* tris;      // 3 x float4 per triangle: {v0.xyz, e1.x} {e1.yz, e2.xy} {e2.z, geom, prim, -}
    const Bvh4Node* nodes;
    const Bvh4NodeQ* qnodes; // the 64-byte copy the any-hit rays walk
    const DevGeom* geoms;
    const DevMat* mats;
    const DevTex* texs;
    const uint32_t* indices;
    const float* normals;    // 3 per vertex
    const float* uvs;        // 2 per vertex
    const float* tangents;   // 4 per vertex
    const uint32_t* texels;  // RGBA8 bilinear footprint table: 4 texels (16 B) per texel position, see sample_texture
    const uint4* bundles;    // per-material interleaved footprints (2 x uint4 per texel position), see DevMat::bundle
    const float4* shade;     // 8 x float4 (one 128-B line) per sorted triangle: see pack_shade_records_kernel
    uint32_t n_tris;
    int32_t root;            // root node index, or a leaf code (< 0) for a single-triangle scene
    uint32_t n_qnodes;       // nodes in qnodes (breadth-first: the first ones are the top of the tree)
};

struct GiState {
    SceneView view{};
    std::vector<void*> allocs;
    // host copies kept for the build
    std::vector<float> h_tris; // 12 floats per triangle
    float scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};
    uint32_t n_tris = 0, n_nodes = 0, bvh_depth = 0, build_passes = 0;
    float build_ms = 0.f; // wall time of the last successful neb_gi_build_bvh
    size_t texture_table_bytes = 0; // footprint tables + material bundles on the device
    uint32_t max_bvh_depth = 21; // (kLdsStack + kSpillStack) / 3: deeper trees are refused by neb_gi_build_bvh ("gi_max_bvh_depth" lowers it)
    bool built = false;
    unsigned long long* d_ray_counter = nullptr;
    neb_gi_hit* d_hits = nullptr;
    bool debug_hits = false;
    // What ONE dispatch of neb_gi_trace writes besides radiance[cur].  Two sets ("gi_defer_resolve" = 2): the GI stages of two frames may be in flight
    // on two streams, each on its own set; neb_gi_resolve retires them in the order they were traced.
    struct DispatchSet {
        float4* d_records = nullptr; // 5 float4 planes + one 4 x float4 record plane over the resident pixels (GiRecords)
        uint32_t* d_sort = nullptr;  // 4 x npx uint32: keys, vals, keys_out, vals_out
        void* d_sort_temp = nullptr;
        uint32_t* d_block_counts = nullptr; // [3][n_block_counts]: bounce rays / shadow rays / shadow rays answered by the sun table, per workgroup
        uint32_t* d_list = nullptr;  // [2][kListSegments] counters, then [kListSegments][cap] ray records
        uint32_t list_epoch = 0;     // shade launches so far: picks the counter set
        uint32_t pending_spp = 1, pending_row0 = 0, pending_row1 = 0;
        bool awaiting_resolve = false; // "gi_defer_resolve" = 1: this set's sums have not been added into radiance[cur] yet
        uint32_t resolve_seq = 0;      // ... and the order in which the sets were filled (neb_gi_resolve takes the oldest)
        neb_gi_constants split_c{}; // neb_gi_trace_begin's arguments, for its neb_gi_trace_finish
        uint32_t split_row0 = 0, split_row1 = 0;
    } sets[2];
    uint32_t resolve_seq = 0;
    uint32_t* d_tile_order = nullptr; // tuning: an explicit tile order for the closest-hit pass (neb_gi_debug_set_tile_order)
    uint32_t tile_order_n = 0, tile_order_cap = 0;
    uint32_t traces = 0, resolves = 0; // deferred dispatches issued / retired (set = count & 1 with two sets)
    uint32_t begun = 0, finished = 0;  // neb_gi_trace_begin / _finish calls (set = count & 1)
    uint32_t* d_direct_counts = nullptr; // neb_pbr_direct's own per-workgroup ray counts (it may run beside a GI dispatch)
    unsigned long long last_stats[16] = {};
    int defer_resolve = 0; // "gi_defer_resolve": 0 fused; 1 the indirect term waits in the records for neb_gi_resolve; 2 the same on two record sets
    bool exact_shade = false; // "gi_exact_shade": gi_shade_kernel<false>, the oracle's C arithmetic
    bool sort_shadow = true;  // "gi_sort_rays" bit 0
    bool sort_shadow_auto = true; // until "gi_sort_rays" is set: sort the shadow rays of dispatches of 1.5 M pixels and more only
    bool sort_bounce = false; // "gi_sort_rays" bit 1
    size_t sort_temp_bytes = 0;
    size_t n_block_counts = 0;
    // the sun-visibility table in the shading records (gi_sun_table.hip)
    bool sun_table = true;            // option "gi_sun_table"
    int sun_table_state = 0;          // 0: the records carry no flags; 1: flags of sun_table_key = this frame's sun; 2: flags of another sun (ignored)
    float sun_table_pending[4] = {0, 0, 0, 0}; // a new sun seen once: the table follows when it has been seen sun_hold times in a row
    // A build costs five frames' time and earns a tenth of a frame per dispatch: a table pays for itself after ~30 dispatches.  A new sun is built for when it
    // has held for TWO dispatches -- unless the table it replaces lived fewer than kSunTableLife dispatches (a sun that moves in steps: a build per step made
    // the frames twice as slow as no table at all), then for kSunHoldAfterShortLife.  "gi_sun_hold" = N pins the count (0 = this rule).
    uint32_t sun_hold = 2, sun_seen = 0, sun_table_age = 0;
    int sun_hold_option = 0;
    float sun_table_key[4] = {0, 0, 0, 0}; // {sunLightDirection, sunTanHalfAngle} the flags were built for
    unsigned long long* d_sun_counts = nullptr; // sides proven lit {+, -} by the last build
    uint32_t* d_sun_hint_list = nullptr;        // two-pass build: triangles whose primary side the first pass left unproven
    uint32_t sun_hint_list_cap = 0;
    uint32_t sun_table_builds = 0;
    // The flags are rewritten IN PLACE in the shading records, on the stream of the dispatch that noticed the new sun, while the host
    // state says "table valid" from the moment of the enqueue: a dispatch on ANOTHER stream ("gi_defer_resolve" = 2, or a host that
    // alternates streams) waits for this event first, or its shade pass could read the previous sun's lit bits of records the
    // 15-ms build has not reached yet.
    hipEvent_t sun_table_event = nullptr;   // recorded behind every launch that rewrites the flags
    hipEvent_t sun_build_ev[2] = {nullptr, nullptr}; // around the last build's two launches (neb_gi_sun_table_build_ms)
    hipStream_t sun_table_stream = nullptr; // ... on this stream
    bool sun_table_event_pending = false;   // not yet seen complete
    hipStream_t last_dispatch_stream = nullptr; // stream of the last dispatch that read the flags (a rewrite on another stream waits for the device)
    bool last_dispatch_stream_set = false;
    unsigned long long table_rays = 0; // shadow rays answered by the table as of the last neb_gi_ray_count
    int sun_hints = 4;                // option "gi_sun_hints": occluder hints the shade pass tries per hit (0, 2 or 4)
    bool compact_shadow = true;       // with the table on: the rays it leaves are compacted into lists by the shade pass ("gi_sun_table" = 2: off)
    // Which pass takes the rays the table leaves is MEASURED once per table build ("gi_sun_table" = 1, the default): the first dispatch with a new table
    // runs the compacted lists between two events, the second the sorted pass (same bits), a later one reads both times and keeps the faster -- the lists
    // win for most suns (0.51 against 0.58 ms of GI on the bench frame), the sorted pass where the table leaves many long rays (a sun straight overhead:
    // 0.79 against 0.92).  3 = always lists, 2 = always the sorted pass.
    bool tail_tune = true;
    int tail_phase = 0;               // 0 decided / nothing to decide, 1 time the lists next, 2 time the sorted pass next, 3 both enqueued: waiting for their events
    bool tail_sorted = false;         // the decision for this table
    hipEvent_t tail_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float tail_us[2] = {0.f, 0.f};    // the two times of the last measurement {lists, sorted pass}
};
hipError_t gi_sun_table_update(GiState* g, const neb_gi_constants& c, hipStream_t stream);
hipError_t gi_sun_table_order(GiState* g, hipStream_t stream);

void gi_on_resize(GiState* g);
void gi_destroy(GiState* g);

} // namespace neb

// ---- host helpers of the C ABI entry points ----
static inline int gi_fail(neb_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess)
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

// scoped: run the rest of the entry point on the context's device, restore the caller's device on return
#define GI_GUARD(ctx)                                                        \
    neb::DeviceGuard gi_guard_((ctx)->device);                               \
    if (gi_guard_.err != hipSuccess)                                         \
    return gi_fail((ctx), NEB_ERR_HIP, "hipSetDevice", gi_guard_.err)

template <typename T>
static inline hipError_t upload(neb::GiState* g, const std::vector<T>& h, const T** out)
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

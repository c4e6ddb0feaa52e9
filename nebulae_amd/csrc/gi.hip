// gi.hip -- one-bounce indirect-diffuse GI on gfx950: the wavefront kernels and the C ABI of the trace
// (scene upload and BVH build: gi_build.hip; device-side arithmetic and the traverser: gi_device.h).
//
// Reference behaviour (paths relative to the reference checkout):
//   assets/shaders/pathtracer.hlsl:397-625 (PathtracerRG, query variant, NRC stubbed per
//   rtxgi/Nrc.hlsli:579-621, nrcMaxPathVertices = 2), :299-395 ReconstructSurfaceData, :209-228
//   EvaluateDirectBRDF; brdf.hlsli; rand.hlsli; sun_disk_sampling.hlsli:45-52;
//   src/nri/GIProcessedScene.cpp:16-137 (scene tables); RTAccelerationStructureBuilder.cpp:14-130
//   (driver BVH -> replaced by a binned-SAH tree built on the device and collapsed to 4-wide nodes, gi_build.hip); deferred_gbuffers.hlsl:36-104 (G-buffer encodings).
// Not a DXR transliteration: there is no ray-gen/miss/closest-hit pipeline and no driver BVH.  The path runs as
// wavefront stages over per-pixel records: a wave owns an 8x8 pixel tile, each lane generates its ray from the
// G-buffer and walks the BVH4 with a per-lane stack kept in LDS (lane-contiguous, conflict-free); a second kernel
// shades the hits from one 128-byte record per triangle; the sun shadow rays are sorted by origin (raysort.hip) and
// go through the same traverser in any-hit mode.
//
#include "gi_device.h" // (turns floating-point contraction off for this file: see there)

namespace neb {

// ------------------------------------------------------------------------------------------------
// GI as three wavefront stages per sample (one wave per 8x8 pixel tile, records indexed by pixel):
//   gi_raygen_trace_kernel : G-buffer -> RNG -> throughput -> cosine ray -> closest-hit traversal
//   gi_shade_kernel        : miss -> sky; hit -> ReconstructSurfaceData, sun-disk shadow ray + BRDF contribution
//   gi_shadow_trace_kernel : any-hit traversal of the shadow ray, accumulate; last sample adds into radiance[cur]
// Splitting keeps the two traversal kernels at <= 64 VGPRs (8 waves/SIMD) and the register-hungry shading
// away from them.
// ------------------------------------------------------------------------------------------------
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
    uint32_t* shadow_counts;     // per-workgroup shadow-ray counts (every sun-visibility query, traced or answered by the table)
    uint32_t* table_counts;      // per-workgroup count of the shadow rays the sun table answered
    uint32_t sun_table;          // 1: the shading records carry the sun-visibility table of THIS frame's sun (gi_sun_table.hip)
    uint32_t hint_pairs;         // pairs of occluder hints the shade pass tries per hit (option "gi_sun_hints" / 2: 0, 1 or 2)
    uint32_t W, row_begin, row0, row1, tiles_x;
    uint32_t sample;             // index of the sample this launch handles
    uint32_t bounce;             // path vertex this launch handles: 1 .. maxPathVertices - 1
    uint32_t stats;              // 1: also count shadow-ray traversal steps (slow path, diagnostics)
    uint32_t defer_resolve;      // 1: leave the frame's sum in the record plane; neb_gi_resolve adds it into radiance[cur] later
    const uint32_t* tile_order;  // closest-hit pass: workgroup b takes tile tile_order[b] (tuning: neb_gi_debug_set_tile_order), or null: the XCD-aware default
    uint32_t* sort_keys;         // shadow-ray sorting ("gi_sort_shadow_rays"): Morton key of the ray origin per pixel, or null
    uint32_t* sort_vals;         // pixel index per key
    const uint32_t* sort_order;  // pixel indices in key order (after the radix sort), or null: pixel order
    uint32_t* bsort_keys;        // same for the bounce rays ("gi_sort_rays" bit 1): key = direction octant | origin Morton code
    uint32_t* bsort_vals;
    uint32_t raygen_only;        // 1: gi_raygen_trace_kernel only writes the ray record and its key (a sorted trace follows)
    // Shadow rays that still need a walk once the sun table has answered the rest (~14 % on the bench frame), compacted by the
    // shade pass: kListSegments lists, workgroup b appends to list b % kListSegments with ONE atomic per wave (spread over the
    // segments: a single hot word takes only ~90 atomics/us), each list sized for every pixel of its workgroups, so none can overflow.
    float4* list;                // [kListSegments][list_cap] self-contained 64-byte ray records {origin, pixel}{direction, 1}{contribution, -}{sum, -}, or null: the uncompacted paths
    uint32_t* list_counts;       // [2][kListSegments]: the set this launch fills (list_set) and the one gi_shadow_trace_kernel clears for the next
    uint32_t list_cap, list_set;
    float smin[3], sinv[3];      // scene box for the Morton keys
    uint32_t first_px, n_px;     // dispatched pixel range [first_px, first_px + n_px) of the resident planes
};
// Position of this workgroup in the dispatch (tile number, or run of 64 sorted rays).  Workgroups are dealt round-robin
// over the 8 XCDs, each with its own L2.  With RUNS > 0 the dispatch is cut into RUNS segments and inside a segment each
// XCD takes one contiguous eighth, so the workgroups resident on an XCD cover a compact piece of the frame.  Speed only,
// and measured per kernel (1080p atrium): the shade pass gains a little (123 -> 120 us at 16 segments: its tile stores and
// G-buffer reads), the closest-hit pass is indifferent (378 -> 375 us), the any-hit pass over the sorted rays LOSES (177 -> 186 us,
// 222 us with one segment, where the XCDs also finish unevenly) and keeps the round-robin order.
template <uint32_t RUNS>
__device__ __forceinline__ uint32_t gi_block()
{
    if constexpr (RUNS == 0u)
        return blockIdx.x;
    const uint32_t seg_len = ((gridDim.x + RUNS - 1u) / RUNS + 7u) & ~7u; // (segments start on XCD 0)
    const uint32_t seg = blockIdx.x / seg_len, b = blockIdx.x - seg * seg_len;
    const uint32_t n = min(seg_len, gridDim.x - seg * seg_len);
    const uint32_t q = n >> 3, r = n & 7u, xcd = b & 7u, k = b >> 3;
    return seg * seg_len + (xcd < r ? xcd * (q + 1u) + k : r * (q + 1u) + (xcd - r) * q + k);
}
constexpr uint32_t kShadeRuns = 16u, kRaygenRuns = 16u;
constexpr uint32_t kListSegments = 128u;
#ifndef NEB_TAIL_STAMPS
#define NEB_TAIL_STAMPS 0
#endif
#ifndef NEB_LIST_QUAD_BELOW
#define NEB_LIST_QUAD_BELOW 1500000u // dispatches of fewer pixels walk their ray lists with four lanes per ray (row strips; a 1080p frame does not)
#endif
#ifndef NEB_LIST_WAVES
#define NEB_LIST_WAVES 6 // waves per SIMD it is register-budgeted for (it is bound by latency, not by occupancy)
#endif

template <uint32_t RUNS = 0u>
__device__ __forceinline__ bool gi_pixel(const GiArgs& a, uint32_t& x, uint32_t& y, size_t& i)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = gi_block<RUNS>();
    const uint32_t tile_x = blk % a.tiles_x, tile_y = blk / a.tiles_x;
    x = tile_x * 8 + (lane & 7);
    y = a.row0 + tile_y * 8 + (lane >> 3);
    i = (size_t)(y - a.row_begin) * a.W + x;
    return x < a.W && y < a.row1;
}

// Ray accounting without same-address atomics (one hot word saturates at ~90 atomics/us on MI355X, which cost
// more than the traversal itself): every workgroup owns one slot of a per-kernel count array; the host sums them.
__device__ __forceinline__ void count_rays(uint32_t* block_counts, uint32_t mine, uint32_t slot)
{
    uint32_t total = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        total += __shfl_down(total, off);
    if ((threadIdx.x & 63u) == 0 && total)
        block_counts[slot] += total; // slot owned by this wave; launches on one stream are ordered
}
__device__ __forceinline__ void count_rays(uint32_t* block_counts, uint32_t mine) { count_rays(block_counts, mine, blockIdx.x); }

#ifndef NEB_FAST_RAYGEN
#define NEB_FAST_RAYGEN 0 // A/B arm: 1 = 1-ulp hardware rcp / rsq / sqrt / x^5 in ray generation, 2 = also v_sin / v_cos.  Measured: the
                          // kernel takes 380 us either way (377 us exact) although ray generation is ~15 % of its instructions, and
                          // six parity tests leave their 2e-5 band (a direction that moves by an ulp lands on another texel footprint)
#endif
// One wave per 8x8 pixel tile.  LDS per wave: the closest-hit walk's stack (kRgStack entries per lane: no bounce ray of the bench frame ever holds
// more than 12, tools/gi_wave_stamps.py; deeper entries go to the private array) + the child slots of traverse_core (16 bytes per lane) = 4 KB,
// eight waves per SIMD.
#ifndef NEB_RG_STACK
#define NEB_RG_STACK 12
#endif
constexpr int kRgStack = NEB_RG_STACK;
static_assert(32 * (kRgStack * 256 + 1024) <= 160 * 1024 || NEB_TRACE_WAVES < 8, "LDS budget of the closest-hit pass: 32 waves per CU of stacks and child slots in 160 KB");
template <bool FAST, bool FAST_TRIG>
__global__ __launch_bounds__(64, NEB_TRACE_WAVES) void gi_raygen_trace_kernel(GiArgs a)
{
    __shared__ int stack_mem[kRgStack * 64];
    __shared__ int child_slot_mem[256];
    uint32_t x, y;
    size_t i64;
    bool active;
    if (a.tile_order) { // (tuning arm: an explicit tile order, e.g. the most expensive tiles of the last frame first)
        const uint32_t blk = a.tile_order[blockIdx.x], tile_x = blk % a.tiles_x, tile_y = blk / a.tiles_x;
        x = tile_x * 8 + (threadIdx.x & 7);
        y = a.row0 + tile_y * 8 + (threadIdx.x >> 3);
        i64 = (size_t)(y - a.row_begin) * a.W + x;
        active = x < a.W && y < a.row1;
    } else {
        active = gi_pixel<kRaygenRuns>(a, x, y, i64);
    }
    const uint32_t i = (uint32_t)i64, lane = threadIdx.x; // (32 bits: one register across the walk)
    uint32_t rays = 0;
    uint32_t wave_stamp[5] = {0u, 0u, 0u, 0u, 0u}; // diagnostics (a.stats): the closest-hit loop's wave stamps, see Hit
    if (active) {
        const float3 albedo = unpack_r11g11b10(a.albedo[i]);
        const uint2 wp = a.world_pos[i];
        const float3 worldPos = f3(half_bits_to_float(wp.x & 0xffffu), half_bits_to_float(wp.x >> 16), half_bits_to_float(wp.y & 0xffffu));
        const uint32_t nzw = a.normal[i].y;
        const float3 SN = oct_unpack<FAST>(half_bits_to_float(nzw & 0xffffu), half_bits_to_float(nzw >> 16));
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
        const float pd = 1.0f - specular_probability<FAST>(saturate1(dot3(normalize3<FAST>(V), SN)), F0, albedo);
        if (rand01(rng) < pd)
            throughput = f3(fdiv<FAST>(throughput.x, pd), fdiv<FAST>(throughput.y, pd), fdiv<FAST>(throughput.z, pd)); // :476-479
        const float u0 = rand01(rng), u1 = rand01(rng);
        const float3 dir = cosine_hemisphere_aligned<FAST, FAST_TRIG>(u0, u1, SN);
        const float3 org = worldPos + SN * 1e-2f; // :138
        const bool bounce = a.c.maxPathVertices > 1; // for (bounce = 1; bounce < nrcMaxPathVertices; ...)
        a.R.path[i] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(rng));
        a.R.ray_o[i] = make_float4(org.x, org.y, org.z, 0.01f);
        a.R.ray_d[i] = make_float4(dir.x, dir.y, dir.z, bounce ? 1.0f : 0.0f);
        if (a.sample + 1 < a.c.samplesPerPixel) // only the next sample of this pixel reads it
            a.R.state[i] = make_float4(V.x, V.y, V.z, __uint_as_float(rng));
        float4 h = make_float4(bounce ? -1.0f : -2.0f, 0.f, 0.f, 0.f); // -2: no bounce at all, nothing is added
        rays = bounce ? 1u : 0u;
        wave_stamp[0] = wave_stamp[1] = wave_stamp[2] = wave_stamp[3] = wave_stamp[4] = 0u;
        if (a.bsort_keys) {
            a.bsort_keys[i] = bounce ? bounce_sort_key(org, dir, a.smin, a.sinv) : (1u << kSortBits) - 1u;
            a.bsort_vals[i] = (uint32_t)i;
        }
        if (bounce && !a.raygen_only && a.S.n_tris) {
            Hit hit;
            const bool found = a.stats ? traverse_t<false, true, kRgStack, true>(a.S, org, dir, 0.01f, kTraceMax, stack_mem + lane, hit, stack_mem, child_slot_mem + 4u * lane)
                                       : traverse_t<false, false, kRgStack, true>(a.S, org, dir, 0.01f, kTraceMax, stack_mem + lane, hit, stack_mem, child_slot_mem + 4u * lane);
            if (found)
                h = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.tri));
            if (a.stats) { // diagnostics only
                wave_stamp[0] = hit.w_iters, wave_stamp[1] = hit.w_node_iters, wave_stamp[2] = hit.w_node_lanes, wave_stamp[3] = hit.w_leaf_iters,
                wave_stamp[4] = hit.w_leaf_lanes;
                atomicAdd(a.ray_counter + 1, (unsigned long long)hit.node_visits);
                atomicAdd(a.ray_counter + 2, (unsigned long long)hit.tri_tests);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    atomicAdd(a.ray_counter + 11 + k, (unsigned long long)hit.top_visits[k]);
                atomicAdd(a.ray_counter + 15, (unsigned long long)hit.deep_sp);
                a.R.srec[4 * i + kSrContrib].w = __uint_as_float(hit.node_visits + ((hit.tri_tests + 3u) >> 2)); // loop iterations of this ray
            }
        }
        a.R.hit[i] = h;
    }
    count_rays(a.bounce_counts, rays);
    if (a.stats) { // the stamps are wave-uniform among the lanes that walked longest: the wave's totals are the maxima over its lanes
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            uint32_t v = wave_stamp[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                v = max(v, (uint32_t)__shfl_xor((int)v, off));
            if (lane == 0)
                atomicAdd(a.ray_counter + 6 + k, (unsigned long long)v);
        }
        if (lane == 0)
            atomicAdd(a.ray_counter + 5, 1ull);
    }
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

// What gi_shade_kernel leaves for a pixel's shadow pass (GiRecords::srec) and how many shadow rays it made.
struct ShadeOut {
    float4 rec_o = {0.f, 0.f, 0.f, 0.f}, shadow_d = {0.f, 0.f, 0.f, 0.f} /* valid = 0: no shadow ray */, rec_c = {0.f, 0.f, 0.f, 0.f}, sum = {0.f, 0.f, 0.f, 0.f};
    uint32_t rays = 0;  // sun-visibility queries of this pixel (0 / 1)
    uint32_t table = 0; // ... answered by the sun table instead of a traversal
};

// Shading of path vertex a.bounce of pixel i, whose bounce ray ended in `h` {t (< 0 miss), u, v, triangle}: miss -> sky; hit ->
// ReconstructSurfaceData, sun-disk shadow ray + BRDF contribution, and for maxPathVertices > 2 the next bounce ray.
// FAST: the arithmetic policy of gi_device.h (1-ulp hardware rcp / rsq / sqrt / sin / cos, the forms an HLSL compiler emits);
// FAST = false is the C arithmetic of the oracle (option "gi_exact_shade", see neb_set_option).
// STAGED: the hit triangle's shading record is already in LDS at `staged` (gi_shade_kernel); otherwise it is fetched here.
template <bool FAST, bool STAGED>
__device__ __forceinline__ void shade_pixel(const GiArgs& a, size_t i, const float4 h, ShadeOut& o, const float4* staged, uint32_t staged_swz)
{
    const float4 pth = a.R.path[i];
    const float4 rd = a.R.ray_d[i];
    const uint32_t trav_iters = a.stats ? __float_as_uint(a.R.srec[4 * i + kSrContrib].w) : 0u; // diagnostics (written by the tracer)
    float3 throughput = f3(pth.x, pth.y, pth.z);
    if (!(a.sample == 0 && a.bounce == 1))
        o.sum = a.R.srec[4 * i + kSrSum];
    float4 next_d = make_float4(0.f, 0.f, 0.f, 0.f);   // alive = 0: the path ends here
    neb_gi_hit dbg = {-1.0f, ~0u, ~0u, 0u};
    const bool alive = rd.w != 0.0f; // (vertex 1 with maxPathVertices <= 1: hit.x == -2, nothing is added)
    if (alive && h.x == -1.0f) { // miss: radiance += skyColor * throughput (:508)
        o.sum.x += a.c.skyColor[0] * throughput.x;
        o.sum.y += a.c.skyColor[1] * throughput.y;
        o.sum.z += a.c.skyColor[2] * throughput.z;
    } else if (alive && h.x >= 0.0f) {
        const uint32_t tri = __float_as_uint(h.w);
        Surface surf;
        uint32_t geom;
        TriShade ts;
        if constexpr (STAGED)
            ts = load_tri_shade_at(staged, staged_swz);
        else
            ts = load_tri_shade(a.S, tri);
        const bool shaded = reconstruct_surface<FAST>(a.S, ts, h.y, h.z, surf, geom);
        dbg.t = h.x;
        dbg.geometry = geom;
        if (a.hits)
            dbg.primitive = __float_as_uint(a.S.shade[8 * (size_t)tri + 7].x);
        if (shaded) {
            const float4 ro = a.R.ray_o[i];
            const float3 org = f3(ro.x, ro.y, ro.z), dir = f3(rd.x, rd.y, rd.z);
            const float3 hitP = org + dir * h.x;
            const float3 V = normalize3<FAST>(-dir); // :522
            uint32_t rng = __float_as_uint(pth.w);
            const float a0 = rand01(rng), a1 = rand01(rng);
            // The GEOMETRY of the shadow ray -- the sun's frame, the disk sample, the direction -- is computed in the oracle's
            // arithmetic under both policies (IEEE square root and division, det_sincosf): a visibility query is a discrete
            // outcome, and a direction that differs in its last ulp (hardware rsq / sin / cos) flips about one query in eight
            // million at a silhouette -- 5e-4 of a frame's L2 norm each time.  ~150 instructions per hit in a pass bound by
            // memory; the BRDF and the surface reconstruction keep the 1-ulp hardware forms (no discrete outcome hangs on them).
            const float angle = a0 * 2.0f * 3.1415926535f, dist = fsqrt<false>(a1);
            const float3 sun_dir = f3(a.c.sunLightDirection[0], a.c.sunLightDirection[1], a.c.sunLightDirection[2]);
            const float3 sun_rad = f3(a.c.sunLightRadiance[0], a.c.sunLightRadiance[1], a.c.sunLightRadiance[2]);
            const float3 L = normalize3<false>(-sun_dir);
            const float3 Bv = normalize3<false>(perpendicular(L));
            const float3 T = cross3(Bv, L);
            float sn_a, cs_a;
            det_sincosf(angle, sn_a, cs_a);
            const float3 inc = normalize3<false>(L + (Bv * sn_a + T * cs_a) * a.c.sunTanHalfAngle * dist);
            const bool transition = dot3(surf.GN, inc) <= 0.0f;
            const float3 so = hitP + (transition ? -surf.GN : surf.GN) * 1e-2f;
            const float3 O = evaluate_direct_brdf<FAST>(surf, V, L) * sun_rad * throughput; // :573-574
            o.rays = 1;
            const uint32_t side = transition ? 1u : 0u;
            bool hinted_hit = false;
            if (a.sun_table && !((ts.lit >> side) & 1u) && ts.hint_side == side && ts.hint[0] != kNoHint) {
                // Occluder hints: the (up to) four triangles that together shadow most of this triangle (chosen once per sun
                // position).  They are tried with the traverser's own test on the very ray the shadow pass would walk -- same
                // operands, same arithmetic -- so a hit is a hit of the any-hit traversal too: "occluded", exactly, without the
                // walk.  Two at a time: the second pair is only fetched by the lanes the first pair did not stop.
                float tt, uu, vv;
#pragma unroll
                for (int pair = 0; pair < kHints && !hinted_hit; pair += 2) {
                    const uint32_t h0 = ts.hint[pair];
                    if (h0 == kNoHint || (uint32_t)pair >= 2u * a.hint_pairs)
                        break;
                    const uint32_t h1 = ts.hint[pair + 1] != kNoHint ? ts.hint[pair + 1] : h0;
                    const float4 a0 = a.S.tris[3 * h0], b0 = a.S.tris[3 * h0 + 1], c0 = a.S.tris[3 * h0 + 2];
                    const float4 a1 = a.S.tris[3 * h1], b1 = a.S.tris[3 * h1 + 1], c1 = a.S.tris[3 * h1 + 2];
                    hinted_hit = intersect_tri_regs(a0, b0, c0, so, inc, 0.001f, kTraceMax, tt, uu, vv) ||
                                 intersect_tri_regs(a1, b1, c1, so, inc, 0.001f, kTraceMax, tt, uu, vv);
                }
            }
            if (hinted_hit) {
                o.table = 1; // occluded: nothing is added, no ray
            } else if (a.sun_table && ((ts.lit >> side) & 1u)) {
                // every shadow ray that leaves this triangle on this side is unoccluded (proven once per sun position,
                // gi_sun_table.hip): what gi_shadow_trace_kernel would do after its walk -- radiance += BRDF * sunRadiance * throughput
                // (:571-575) -- is done here, same operands, same order; no ray, no sort key, no record to gather
                o.sum.x += O.x;
                o.sum.y += O.y;
                o.sum.z += O.z;
                o.table = 1;
                dbg.flags |= 1u;
            } else {
                o.rec_o = make_float4(so.x, so.y, so.z, 0.001f);
                o.shadow_d = make_float4(inc.x, inc.y, inc.z, 1.0f);
                o.rec_c = make_float4(O.x, O.y, O.z, 0.f);
            }
            if (a.bounce + 1 < a.c.maxPathVertices) { // not the last vertex (:579-583): sample the next bounce
                // EvaluateIndirectBRDF (:230-259) takes rng BY VALUE: its draws do not advance the path's stream,
                // so the Rand(rng) of :614 returns the same number as the first of them.
                uint32_t rng_copy = rng;
                const float3 SNn = normalize3<FAST>(surf.SN);
                const float e0 = rand01(rng_copy), e1 = rand01(rng_copy);
                const float3 Ld = cosine_hemisphere_aligned<FAST>(e0, e1, SNn);
                const float pdiff = 1.0f - specular_probability<FAST>(saturate1(dot3(V, SNn)), specular_f0(surf.albedo, surf.metalness), surf.albedo);
                const float3 no = hitP + surf.GN * 1e-2f; // :607
                throughput = throughput * (surf.albedo * (1.0f - surf.metalness)); // :613
                if (rand01(rng) < pdiff)
                    throughput = f3(fdiv<FAST>(throughput.x, pdiff), fdiv<FAST>(throughput.y, pdiff), fdiv<FAST>(throughput.z, pdiff)); // :614-618
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
        if (o.shadow_d.w != 0.0f) {
            key = min(morton30(f3(o.rec_o.x, o.rec_o.y, o.rec_o.z), a.smin, a.sinv) >> (30 - kSortBits), (1u << kSortBits) - 2u);
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
            dbg.flags |= min(trav_iters, 4095u) << 8; // diagnostics: traversal iterations of the bounce ray (tools/gi_divergence.py); bits 20..31: shadow-ray node visits
        a.hits[i] = dbg;
    }
}

// One wave per 8x8 pixel tile.
template <bool FAST>
__global__ __launch_bounds__(64, NEB_SHADE_WAVES) void gi_shade_kernel(GiArgs a)
{
    uint32_t x, y;
    size_t i;
    const bool active = gi_pixel<kShadeRuns>(a, x, y, i);
    ShadeOut o;
    float4 h = make_float4(-2.0f, 0.f, 0.f, 0.f);
    if (active)
        h = a.R.hit[i];
    // The 128-byte shading records of the wave's (up to) 64 hit triangles are gathered COOPERATIVELY: eight lanes fetch
    // the eight 16-byte pieces of one record -- one cache line per eight lanes instead of one per lane and load -- straight
    // into LDS (LDS-DMA), eight records per instruction.  The texture-address units, which the scattered per-lane gathers
    // kept 80 % busy, see an eighth of the line lookups; each lane then reads its own record from LDS.  Piece j of the
    // record of lane s lands at [s * 8 + j] and holds global piece j ^ (s & 7): an XOR swizzle that spreads the banks.
    __shared__ float4 smem[64 * 8];
    const uint32_t lane = threadIdx.x;
    const bool has_hit = active && h.x >= 0.0f;
    const uint32_t my_tri = has_hit ? __float_as_uint(h.w) : ~0u; // ~0: nothing to fetch for this lane (there may be no record array at all)
#pragma unroll
    for (uint32_t it = 0; it < 8; ++it) {
        const uint32_t src = it * 8u + (lane >> 3);
        const uint32_t t = (uint32_t)__shfl((int)my_tri, (int)src);
        if (t != ~0u) { // (lanes masked off leave their LDS slot as it is: nobody reads it)
            const float4* g = a.S.shade + 8 * (size_t)t + ((lane & 7u) ^ (src & 7u));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + it * 64u), 16, 0, 0);
        }
    }
    __syncthreads(); // (waits for the DMA: an LDS-DMA is a pending LDS write on the VM counter)
    if (active)
        shade_pixel<FAST, true>(a, i, h, o, smem + lane * 8u, lane & 7u);
    if (a.list) {
        // Compacted tail (the sun table is on: most pixels have no ray left to trace).  A pixel WITHOUT a ray is finished here, in
        // tile order: what gi_shadow_trace_kernel does for it after gathering its record in sorted order -- radiance[cur] += sum / spp
        // on the last vertex of the last sample (the stand-in for NRC Resolve), otherwise carry the sum -- same operands, same
        // order.  A pixel WITH a ray writes its 64-byte record and appends itself to one of the ray lists.
        const bool trace = active && o.shadow_d.w != 0.0f;
        const bool final_vertex = a.sample + 1 == a.c.samplesPerPixel && a.bounce + 1 >= a.c.maxPathVertices && !a.defer_resolve;
        if (active && !trace) {
            if (final_vertex) {
                if (o.sum.x != 0.0f || o.sum.y != 0.0f || o.sum.z != 0.0f) {
                    const float inv_spp = 1.0f / (float)a.c.samplesPerPixel;
                    float4 r = a.radiance[i];
                    r.x += o.sum.x * inv_spp;
                    r.y += o.sum.y * inv_spp;
                    r.z += o.sum.z * inv_spp;
                    a.radiance[i] = r;
                }
            } else {
                a.R.srec[4 * i + kSrSum] = o.sum;
            }
        }
        const unsigned long long m = __ballot(trace);
        if (m) {
            const uint32_t seg = blockIdx.x % kListSegments;
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(a.list_counts + a.list_set * kListSegments + seg, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, 0);
            if (trace) { // the ray goes into its list slot whole (tmin is the constant 1e-3: its word carries the pixel)
                const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                float4* rec = a.list + 4 * ((size_t)seg * a.list_cap + slot);
                rec[kSrO] = make_float4(o.rec_o.x, o.rec_o.y, o.rec_o.z, __uint_as_float((uint32_t)i));
                rec[kSrD] = o.shadow_d;
                rec[kSrContrib] = o.rec_c;
                rec[kSrSum] = o.sum;
            }
        }
        count_rays(a.shadow_counts, o.rays);
        count_rays(a.table_counts, o.table);
        return;
    }
    // Store the 64-byte records of the wave's 8x8 tile.  Lane-per-pixel stores would write 16 bytes at a 64-byte stride
    // four times over; transposed through LDS, every store instruction writes two 512-byte runs (one tile row each).
    __syncthreads(); // every lane is done with its staged record: the buffer is reused
    float4* xpose = smem;
    xpose[lane * 4 + kSrO] = o.rec_o;
    xpose[lane * 4 + kSrD] = o.shadow_d;
    xpose[lane * 4 + kSrContrib] = o.rec_c;
    xpose[lane * 4 + kSrSum] = o.sum;
    __syncthreads(); // the workgroup is this one wave
    const uint32_t blk = gi_block<kShadeRuns>();
    const uint32_t tile_x = blk % a.tiles_x, tile_y = blk / a.tiles_x;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t row = 2 * k + (lane >> 5), col = (lane & 31u) >> 2, comp = lane & 3u;
        const uint32_t px = tile_x * 8 + col, py = a.row0 + tile_y * 8 + row;
        if (px < a.W && py < a.row1)
            a.R.srec[4 * ((size_t)(py - a.row_begin) * a.W + px) + comp] = xpose[(row * 8 + col) * 4 + comp];
    }
    count_rays(a.shadow_counts, o.rays);
    count_rays(a.table_counts, o.table);
}

// (A persistent-wave variant with per-lane ray refill was measured and dropped: lanes of a wave finish after
// 23 steps on average and the slowest after ~55, so the refill bookkeeping cost more than the idle lanes it
// recovered: 1.24 ms vs 0.52 ms for the bounce rays at 1080p.)
// The shadow pass of one wave: lane l traces the shadow ray of pixel i (valid lanes) and finishes the pixel.
__device__ __forceinline__ void shadow_wave(const GiArgs& a, size_t i, bool valid, int* stack_mem)
{
    // The wave's 64 shadow records (64 bytes each, scattered: the pixels come in sorted order) are gathered cooperatively --
    // four lanes fetch the four 16-byte pieces of one record, sixteen records per instruction, by LDS-DMA into the memory
    // that serves as the traversal stack afterwards -- so the texture-address units see a quarter of the line lookups
    // (see gi_shade_kernel).  Piece j of the record of lane s lands at [s * 4 + j] and holds piece j ^ (s & 3).
    float4* stage = reinterpret_cast<float4*>(stack_mem);
    const uint32_t lane = threadIdx.x;
    const uint32_t my_px = valid ? (uint32_t)i : ~0u;
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
        const uint32_t src = it * 16u + (lane >> 2);
        const uint32_t p = (uint32_t)__shfl((int)my_px, (int)src);
        if (p != ~0u) {
            const float4* g = a.R.srec + 4 * (size_t)p + ((lane & 3u) ^ (src & 3u));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(stage + it * 64u), 16, 0, 0);
        }
    }
    __syncthreads(); // (waits for the DMA)
    const uint32_t swz = lane & 3u;
    const float4 rd = stage[lane * 4u + (kSrD ^ swz)], ro = stage[lane * 4u + (kSrO ^ swz)], contrib = stage[lane * 4u + (kSrContrib ^ swz)];
    float4 sum = stage[lane * 4u + (kSrSum ^ swz)];
    __syncthreads(); // every lane holds its record: the memory becomes the stack
    if (!valid)
        return;
    if (rd.w != 0.0f) {
        Hit sh;
        const bool occluded = traverse(a.S, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), ro.w, kTraceMax, true, stack_mem + threadIdx.x, sh, a.stats != 0);
        if (a.stats) { // diagnostics only
            if (a.hits && a.bounce == 1)
                a.hits[i].flags |= min(sh.node_visits, 4095u) << 20; // node visits of the shadow ray (tools/shadow_tail_stats.py)
            atomicAdd(a.ray_counter + 3, (unsigned long long)sh.node_visits);
            atomicAdd(a.ray_counter + 4, (unsigned long long)sh.tri_tests);
        }
        if (!occluded) { // radiance += BRDF * sunRadiance * throughput (:571-575)
            sum.x += contrib.x;
            sum.y += contrib.y;
            sum.z += contrib.z;
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

// Every dispatched pixel once: in sorted order (a.sort_order) or in 8x8 tiles.
__global__ __launch_bounds__(64, NEB_TRACE_WAVES) void gi_shadow_trace_kernel(GiArgs a)
{
    __shared__ int stack_mem[kLdsStack * 64];
    static_assert(kLdsStack * 64 * sizeof(int) >= 64 * 4 * sizeof(float4), "the shadow records are staged in the stack's LDS");
    uint32_t x, y;
    size_t i = 0;
    bool valid;
    if (a.sort_order) { // one lane per entry of the sorted order (every dispatched pixel appears exactly once)
        const uint32_t j = blockIdx.x * 64u + threadIdx.x;
        valid = j < a.n_px;
        if (valid)
            i = a.sort_order[j];
    } else {
        valid = gi_pixel(a, x, y, i);
    }
    shadow_wave(a, i, valid, stack_mem);
}

// The compacted rays of the shade pass (a.list).  The lists are sized for every pixel (32 k waves' worth at 1080p) but hold ~8 % of
// them, and what is left is bound by LATENCY: 160 k rays of <= 37 node visits each take 50 us because a lone wave needs ~0.8 us per
// dependent step (profiles/r04*_tail*.txt: a launch of one wave per slot 82 us -- most of it dispatching waves that read a counter and
// leave; with the tree warm in cache 52 instead of 56; two nodes per step 68).  So: a fixed grid, workgroup b takes chunks
// b / kListSegments, + kListChunks, ... of list b % kListSegments -- one counter read, no scan -- and a ray comes out of its list slot
// whole (64 bytes, consecutive slots: no gather through a pixel index).
#if NEB_TAIL_STAMPS // diagnostics build: per wave {start, end (s_memrealtime, 100 MHz), rays, max node visits}
__device__ unsigned long long g_tail_stamps[8192 * 4];
#endif
constexpr uint32_t kListChunks = 64u; // chunks of 64 rays per list taken in parallel: kListSegments x kListChunks = 8192 waves = the chip's wave slots
// QUAD: four lanes per ray (traverse_any_quad, gi_device.h), 16 rays per wave at a time.  Which one runs is a matter of how many rays there
// are (measured, profiles/r04_tail_stamps.txt and tools/strip_host_cost.py): the 160 k rays a whole 1080p frame leaves put 2.5 one-lane-per-ray
// waves on every SIMD, which is then bound by instruction issue -- four lanes per ray are four times the waves for the same ~23 M
// wave-instructions: 57.6 us against 53.7.  The 20-80 k rays of a row strip leave most SIMDs with one wave or none, a wave is bound by
// the length of its own dependent chain, and a quarter of the arithmetic per lane shortens it: a 135 / 270 / 540-row strip's frame
// 212 / 278 / 404 us against 227 / 294 / 419.  Same results either way.
template <bool QUAD>
__global__ __launch_bounds__(64, NEB_LIST_WAVES) void gi_shadow_list_kernel(GiArgs a)
{
    __shared__ int stack_mem[kLdsStack * 64];
    const uint32_t lane = threadIdx.x;
#if NEB_TAIL_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    uint32_t st_rays = 0, st_visits = 0;
#endif
    const uint32_t seg = blockIdx.x % kListSegments, chunks = gridDim.x / kListSegments;
    const uint32_t count = a.list_counts[a.list_set * kListSegments + seg];
    if (blockIdx.x == 0) // the other set of counters is idle until the next shade launch fills it: clear it for that launch
        for (uint32_t k = lane; k < kListSegments; k += 64u)
            a.list_counts[(a.list_set ^ 1u) * kListSegments + k] = 0u;
#ifdef NEB_TAIL_SKIP // timing-only build (wrong results): trace every NEB_TAIL_SKIP-th list -- is the tail bound by its rays or by its longest wave?
    if (seg % NEB_TAIL_SKIP)
        return;
#endif
    static_assert(kLdsStack * 64 >= 16 * 64, "a quad's LDS column holds the whole 64-entry stack");
    constexpr uint32_t kRaysPerWave = QUAD ? 16u : 64u;
    const uint32_t slot = QUAD ? lane >> 2 : lane; // the ray of the wave's chunk this lane works on
    for (uint32_t first = (blockIdx.x / kListSegments) * kRaysPerWave; first < count; first += chunks * kRaysPerWave) {
        if (first + slot >= count)
            continue; // (the workgroup is this one wave: no barrier below)
        const float4* rec = a.list + 4 * ((size_t)seg * a.list_cap + first + slot);
        const float4 ro = rec[kSrO], rd = rec[kSrD], contrib = rec[kSrContrib];
        float4 sum = rec[kSrSum];
        const size_t i = __float_as_uint(ro.w);
        Hit sh;
        bool occluded;
        if constexpr (QUAD) {
            occluded = a.stats ? traverse_any_quad<true>(a.S, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), 0.001f, kTraceMax, stack_mem + slot, sh.node_visits, sh.tri_tests)
                               : traverse_any_quad<false>(a.S, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), 0.001f, kTraceMax, stack_mem + slot, sh.node_visits, sh.tri_tests);
            if ((lane & 3u) != 0u)
                continue; // the quad's first lane finishes the pixel
        } else {
#if NEB_TAIL_STAMPS
            occluded = traverse(a.S, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), 0.001f, kTraceMax, true, stack_mem + lane, sh, true);
#else
            occluded = traverse(a.S, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), 0.001f, kTraceMax, true, stack_mem + lane, sh, a.stats != 0);
#endif
        }
#if NEB_TAIL_STAMPS
        st_rays += 1;
        st_visits = max(st_visits, sh.node_visits);
#endif
        if (a.stats) { // diagnostics only
            if (a.hits && a.bounce == 1)
                a.hits[i].flags |= min(sh.node_visits, 4095u) << 20;
            atomicAdd(a.ray_counter + 3, (unsigned long long)sh.node_visits);
            atomicAdd(a.ray_counter + 4, (unsigned long long)sh.tri_tests);
        }
        if (!occluded) { // radiance += BRDF * sunRadiance * throughput (:571-575)
            sum.x += contrib.x;
            sum.y += contrib.y;
            sum.z += contrib.z;
            if (a.hits && a.bounce == 1)
                a.hits[i].flags |= 1u;
        }
        const bool last_vertex = a.bounce + 1 >= a.c.maxPathVertices;
        if (a.sample + 1 == a.c.samplesPerPixel && last_vertex && !a.defer_resolve) { // stands in for NRC Resolve (as shadow_wave)
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
#if NEB_TAIL_STAMPS
    {
        uint32_t r = st_rays, v = st_visits;
        for (int off = 32; off > 0; off >>= 1) {
            r += (uint32_t)__shfl_xor((int)r, off);
            v = max(v, (uint32_t)__shfl_xor((int)v, off));
        }
        if (lane == 0 && blockIdx.x < 8192u) {
            g_tail_stamps[4 * blockIdx.x + 0] = t_start;
            g_tail_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
            g_tail_stamps[4 * blockIdx.x + 2] = r;
            g_tail_stamps[4 * blockIdx.x + 3] = v;
        }
    }
#endif
}

#if NEB_TAIL_STAMPS
} // namespace neb
extern "C" int neb_debug_tail_stamps(unsigned long long* host)
{
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(neb::g_tail_stamps), sizeof(neb::g_tail_stamps)) == hipSuccess ? 0 : -1;
}
namespace neb {
#endif

// ------------------------------------------------------------------------------------------------
// G-buffer producer ("next" row f2): primary visibility through the same BVH
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
                MapSamples maps;
                sample_material_maps(a.S, m, u, v, maps);
                if (m.tex[0] >= 0) {
                    const float4 t = maps.albedo;
                    alb = f3(t.x, t.y, t.z);
                }
                if (m.tex[1] >= 0) {
                    const float3 tg0 = f3(ts.t0.x, ts.t0.y, ts.t0.z), tg1 = f3(ts.t1.x, ts.t1.y, ts.t1.z), tg2 = f3(ts.t2.x, ts.t2.y, ts.t2.z);
                    const float3 bt0 = normalize3(cross3(normalize3(n0), tg0) * ts.t0.w);
                    const float3 bt1 = normalize3(cross3(normalize3(n1), tg1) * ts.t1.w);
                    const float3 bt2 = normalize3(cross3(normalize3(n2), tg2) * ts.t2.w);
                    const float3 T = normalize3(tg0 * b0 + tg1 * b1 + tg2 * b2);
                    const float3 B = normalize3(bt0 * b0 + bt1 * b1 + bt2 * b2);
                    const float4 t = maps.normal;
                    const float3 N = f3(t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f, t.z * 2.0f - 1.0f);
                    SN = normalize3(T * N.x + B * N.y + GN * N.z);
                }
                if (m.tex[2] >= 0) {
                    const float4 t = maps.rm;
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
        float sn_a, cs_a;
        det_sincosf(angle, sn_a, cs_a);
        const float3 inc = normalize3(L + (Bv * sn_a + T * cs_a) * a.c.sunTanHalfAngle * dist);
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

} // namespace neb

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
using namespace neb;

extern "C" {

// phase 0: the whole dispatch (neb_gi_trace).  1: ray generation + closest-hit walk only (neb_gi_trace_begin); 2: the shading and shadow passes of
// the dispatch begun longest ago (neb_gi_trace_finish; `after_shade`: an event to record between the two).
static int gi_dispatch(neb_ctx* ctx, const neb_gi_constants* c, uint32_t row0, uint32_t row1, neb_stream stream, int phase, hipEvent_t after_shade)
{
    if (!ctx || !c)
        return ctx ? gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_trace: null constants") : NEB_ERR_INVALID_ARG;
    if (int rc = neb::svgf_flush_pending(ctx)) // (a held-back temporal pass reads / writes the planes this call touches)
        return rc;
    GiState* g = ctx->gi;
    if (!g || !g->built)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace: scene/BVH not ready (neb_gi_set_scene + neb_gi_build_bvh)");
    if (row0 < ctx->row_begin || row1 > ctx->row_end || row0 > row1)
        return gi_fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_gi_trace: rows not resident");
    if (c->samplesPerPixel == 0 || c->maxPathVertices > 8)
        return gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_gi_trace: samplesPerPixel must be >= 1 and maxPathVertices <= 8 (MaxPathtracingRecursionDepth)");
    if (row0 == row1) { // nothing to trace; a begun dispatch still pairs with its finish
        if (phase == 1 && g->begun - g->finished < 2u) {
            GiState::DispatchSet& e = g->sets[g->begun & 1u];
            e.split_c = *c;
            e.split_row0 = e.split_row1 = row0;
            g->begun++;
        } else if (phase == 2) {
            g->finished++;
        }
        return NEB_OK;
    }
    GI_GUARD(ctx);
    ScopedRange range("GI: Query Pass"); // "GI: NRC Query Pass", DeferredRenderer.cpp:434 (the NRC calls are stubbed here)
    const size_t npx = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    // the set of per-dispatch buffers: with "gi_defer_resolve" = 2 two deferred dispatches may be in flight (on two streams), on alternating sets
    if (g->defer_resolve == 2 && g->traces - g->resolves >= 2u)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace: both record sets hold a dispatch that neb_gi_resolve has not retired yet");
    if (phase == 0 && g->begun != g->finished)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace: a dispatch begun with neb_gi_trace_begin still waits for its neb_gi_trace_finish");
    if (phase != 0) {
        if (g->defer_resolve == 2)
            return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace_begin / _finish: not together with gi_defer_resolve = 2 (the two-call form alternates the record sets itself; 1 is fine)");
        if (c->samplesPerPixel != 1 || c->maxPathVertices > 2 || g->sort_bounce)
            return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace_begin / _finish: one sample and one bounce per pixel only, bounce-ray sorting off (use neb_gi_trace)");
        if (phase == 1 && g->begun - g->finished >= 2u)
            return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace_begin: both record sets hold a dispatch that neb_gi_trace_finish has not completed yet");
    }
    GiState::DispatchSet& ds = g->sets[phase == 1 ? (g->begun & 1u) : phase == 2 ? (g->finished & 1u) : g->defer_resolve == 2 ? (g->traces & 1u) : 0u];
    if (g->debug_hits && !g->d_hits) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, npx * sizeof(neb_gi_hit)));
        GI_HIP(ctx, hipMemset(p, 0, npx * sizeof(neb_gi_hit)));
        g->allocs.push_back(p);
        g->d_hits = (neb_gi_hit*)p;
    }
    if (!ds.d_records) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, npx * sizeof(float4) * 9));
        g->allocs.push_back(p);
        ds.d_records = (float4*)p;
    }
    GiArgs a;
    a.S = g->view;
    a.c = *c;
    a.R.ray_o = ds.d_records;
    a.R.ray_d = ds.d_records + npx;
    a.R.hit = ds.d_records + 2 * npx;
    a.R.path = ds.d_records + 3 * npx;
    a.R.state = ds.d_records + 4 * npx;
    a.R.srec = ds.d_records + 5 * npx;
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
    a.tile_order = (g->d_tile_order && g->tile_order_n == a.tiles_x * ((row1 - row0 + 7) / 8)) ? g->d_tile_order : nullptr;
    a.first_px = (uint32_t)((size_t)(row0 - ctx->row_begin) * ctx->W);
    a.n_px = (uint32_t)((size_t)(row1 - row0) * ctx->W);
    for (int q = 0; q < 3; ++q) {
        a.smin[q] = g->scene_min[q];
        a.sinv[q] = 1.0f / fmaxf(g->scene_max[q] - g->scene_min[q], 1e-20f);
    }
    a.bsort_keys = a.bsort_vals = nullptr;
    a.raygen_only = 0;
    // The shadow-ray sort pays for its six launches only on a dispatch large enough (measured, 1920 pixels wide: 136 rows GI
    // 169 us with it against 146 without, 272 rows 225 / 203, 544 rows 343 / 341, 1080 rows 590 / 605): unless the option was set
    // explicitly it is on from 1.5 M pixels.  Results do not depend on it (every pixel is written once, whatever the order).
    // With the sun table most shadow rays are answered in the shade pass; the rest are compacted into ray lists there and traced
    // unsorted (the sort's six launches cost more than coherence is worth to the ~14 % that are left).
    // the sun-visibility table: brought up to date with this frame's sun (a rebuild only when the sun or the scene changed -- and
    // then only once the new sun has held for a second frame: a sun that is being dragged is traced the plain way meanwhile)
    if (phase != 1) { // (the flags are read by the shade pass)
        GI_HIP(ctx, gi_sun_table_update(g, *c, (hipStream_t)stream));
        GI_HIP(ctx, gi_sun_table_order(g, (hipStream_t)stream)); // a rewrite of the flags enqueued on another stream comes first
        g->last_dispatch_stream = (hipStream_t)stream;
        g->last_dispatch_stream_set = true;
    }
    a.sun_table = g->sun_table_state == 1 ? 1u : 0u;
    a.hint_pairs = (uint32_t)g->sun_hints / 2u;
    // the pass that takes the rays the table leaves: the compacted lists, unless the measurement of this table said the sorted pass (GiState::tail_tune)
    int tail_slot = -1; // 0 / 2: this dispatch is one of the two timed ones (events tail_ev[slot], [slot + 1] around its shade + shadow launches)
    bool tail_warm = false;
    if (a.sun_table && g->compact_shadow && g->tail_tune && phase != 1) {
        if (g->tail_phase == 3 && hipEventQuery(g->tail_ev[1]) == hipSuccess && hipEventQuery(g->tail_ev[3]) == hipSuccess) {
            float ms_lists = 0.f, ms_sorted = 0.f;
            if (hipEventElapsedTime(&ms_lists, g->tail_ev[0], g->tail_ev[1]) == hipSuccess && hipEventElapsedTime(&ms_sorted, g->tail_ev[2], g->tail_ev[3]) == hipSuccess) {
                g->tail_us[0] = ms_lists * 1e3f, g->tail_us[1] = ms_sorted * 1e3f;
                g->tail_sorted = ms_sorted < 0.95f * ms_lists; // (the lists stay unless the sorted pass is clearly faster)
            }
            g->tail_phase = 0;
        } else if (g->tail_phase == 1 && !ds.d_list) {
            // (the lists are allocated by this very dispatch: its first run pays for touching them -- it runs untimed, the next one counts)
        } else if (g->tail_phase == 2 && !ds.d_sort && (g->sort_shadow_auto ? a.n_px >= 1500000u : g->sort_shadow)) {
            tail_warm = true; // the same for the sorted pass's buffers
        } else if (g->tail_phase == 1 || g->tail_phase == 2) {
            tail_slot = g->tail_phase == 1 ? 0 : 2;
            for (int k = 0; k < 2; ++k)
                if (!g->tail_ev[tail_slot + k])
                    GI_HIP(ctx, hipEventCreate(&g->tail_ev[tail_slot + k]));
            g->tail_phase++;
        }
        (void)hipGetLastError(); // (hipErrorNotReady from a query is an answer, not a failure)
    }
    const bool compact = a.sun_table && g->compact_shadow && !tail_warm && (tail_slot >= 0 ? tail_slot == 0 : !g->tail_sorted);
    const bool sort_shadow = !compact && (g->sort_shadow_auto ? a.n_px >= 1500000u : g->sort_shadow);
    if (sort_shadow || g->sort_bounce) {
        if (!ds.d_sort) {
            void* p = nullptr;
            GI_HIP(ctx, hipMalloc(&p, 8 * npx * sizeof(uint32_t))); // {keys, vals, keys_out, vals_out} x {shadow, bounce}
            g->allocs.push_back(p);
            ds.d_sort = (uint32_t*)p;
            const size_t bytes = ray_sort_scratch_bytes(npx);
            GI_HIP(ctx, hipMalloc(&p, bytes));
            g->allocs.push_back(p);
            ds.d_sort_temp = p;
            g->sort_temp_bytes = bytes;
        }
        if (sort_shadow) {
            a.sort_keys = ds.d_sort;
            a.sort_vals = ds.d_sort + npx;
        }
        if (g->sort_bounce) {
            a.bsort_keys = ds.d_sort + 4 * npx;
            a.bsort_vals = ds.d_sort + 5 * npx;
            a.raygen_only = 1;
        }
    }
    a.defer_resolve = g->defer_resolve ? 1u : 0u;
    ds.pending_spp = c->samplesPerPixel;
    ds.pending_row0 = row0;
    ds.pending_row1 = row1;
    if (g->defer_resolve == 2)
        g->traces++;
    if (g->defer_resolve == 1 && phase != 1) { // this set's sums wait for neb_gi_resolve, oldest first (the two-call form alternates the sets)
        ds.awaiting_resolve = true;
        ds.resolve_seq = ++g->resolve_seq;
    }
    const uint32_t tiles_y = (row1 - row0 + 7) / 8;
    const dim3 grid(a.tiles_x * tiles_y), block(64);
    const size_t n_blocks = (size_t)a.tiles_x * ((ctx->row_end - ctx->row_begin + 7) / 8 + 1);
    if (!ds.d_block_counts) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, 3 * n_blocks * sizeof(uint32_t))); // {bounce, shadow, shadow answered by the sun table} rays per workgroup
        GI_HIP(ctx, hipMemset(p, 0, 3 * n_blocks * sizeof(uint32_t)));
        g->allocs.push_back(p);
        ds.d_block_counts = (uint32_t*)p;
        g->n_block_counts = n_blocks;
    }
    a.bounce_counts = ds.d_block_counts;
    a.shadow_counts = ds.d_block_counts + g->n_block_counts;
    a.table_counts = ds.d_block_counts + 2 * g->n_block_counts;
    a.list = nullptr;
    a.list_counts = nullptr;
    a.list_cap = a.list_set = 0;
    uint32_t list_waves = 0;
    if (compact) {
        const uint32_t wg_per_list = (uint32_t)((g->n_block_counts + kListSegments - 1) / kListSegments);
        if (!ds.d_list) {
            void* p = nullptr;
            // (64 bytes per slot, a slot per pixel: what the per-pixel shadow-record plane takes; the two counter sets sit in front)
            GI_HIP(ctx, hipMalloc(&p, (size_t)kListSegments * wg_per_list * 64u * 64u + 4096u));
            GI_HIP(ctx, hipMemset(p, 0, 4096u));
            g->allocs.push_back(p);
            ds.d_list = (uint32_t*)p;
        }
        a.list_counts = ds.d_list;
        a.list = (float4*)((char*)ds.d_list + 4096);
        a.list_cap = wg_per_list * 64u;
        list_waves = kListSegments * (wg_per_list < kListChunks ? wg_per_list : kListChunks);
    }
    const uint32_t n_vertices = c->maxPathVertices > 1 ? c->maxPathVertices - 1 : 1; // path vertices traced per sample
    for (uint32_t s = 0; s < c->samplesPerPixel; ++s) {
        a.sample = s;
        for (uint32_t b = 1; b <= n_vertices; ++b) { // for (bounce = 1; bounce < nrcMaxPathVertices; ++bounce), :495
            a.bounce = b;
            if (b == 1 && phase != 2) {
                hipLaunchKernelGGL((gi_raygen_trace_kernel<(NEB_FAST_RAYGEN >= 1), (NEB_FAST_RAYGEN >= 2)>), grid, block, 0, (hipStream_t)stream, a);
            }
            if (phase == 1)
                continue;
            if (g->sort_bounce) {
                uint32_t* bs = ds.d_sort + 4 * npx; // {keys, vals, keys_tmp, order}
                GI_HIP(ctx, ray_sort_pairs(bs + a.first_px, bs + npx + a.first_px, bs + 2 * npx + a.first_px, bs + 3 * npx + a.first_px,
                                           bs + npx + a.first_px, a.n_px, kSortBits, ds.d_sort_temp, (hipStream_t)stream));
                GiArgs b1 = a;
                b1.sort_order = bs + npx + a.first_px;
                hipLaunchKernelGGL(gi_bounce_trace_kernel, dim3((a.n_px + 63) / 64), block, 0, (hipStream_t)stream, b1);
            } else if (b > 1) {
                hipLaunchKernelGGL(gi_bounce_trace_kernel, grid, block, 0, (hipStream_t)stream, a);
            }
            a.list_set = ds.list_epoch & 1u;
            if (tail_slot >= 0 && s == 0 && b == 1)
                GI_HIP(ctx, hipEventRecord(g->tail_ev[tail_slot], (hipStream_t)stream));
            if (kFastShade && !g->exact_shade)
                hipLaunchKernelGGL(gi_shade_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
            else
                hipLaunchKernelGGL(gi_shade_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
            if (after_shade)
                GI_HIP(ctx, hipEventRecord(after_shade, (hipStream_t)stream));
            if (a.list) {
                if (a.n_px < NEB_LIST_QUAD_BELOW)
                    hipLaunchKernelGGL(gi_shadow_list_kernel<true>, dim3(list_waves), block, 0, (hipStream_t)stream, a);
                else
                    hipLaunchKernelGGL(gi_shadow_list_kernel<false>, dim3(list_waves), block, 0, (hipStream_t)stream, a);
                ds.list_epoch++;
            } else if (sort_shadow) {
                // {keys, vals} are the shade kernel's output and the sort's ping; {keys_tmp, vals_tmp} its pong; the
                // sorted pixel indices land back in vals
                GI_HIP(ctx, ray_sort_pairs(ds.d_sort + a.first_px, ds.d_sort + npx + a.first_px, ds.d_sort + 2 * npx + a.first_px,
                                           ds.d_sort + 3 * npx + a.first_px, ds.d_sort + npx + a.first_px, a.n_px, kSortBits, ds.d_sort_temp,
                                           (hipStream_t)stream));
                GiArgs b2 = a;
                b2.sort_order = ds.d_sort + npx + a.first_px;
                hipLaunchKernelGGL(gi_shadow_trace_kernel, dim3((a.n_px + 63) / 64), block, 0, (hipStream_t)stream, b2);
            } else {
                hipLaunchKernelGGL(gi_shadow_trace_kernel, grid, block, 0, (hipStream_t)stream, a);
            }
        }
    }
    if (tail_slot >= 0)
        GI_HIP(ctx, hipEventRecord(g->tail_ev[tail_slot + 1], (hipStream_t)stream));
    GI_HIP(ctx, hipGetLastError());
    if (phase == 1) {
        ds.split_c = *c;
        ds.split_row0 = row0;
        ds.split_row1 = row1;
        g->begun++;
    } else if (phase == 2) {
        g->finished++;
    }
    return NEB_OK;
}

int neb_gi_trace_rows(neb_ctx* ctx, const neb_gi_constants* c, uint32_t row0, uint32_t row1, neb_stream stream)
{
    return gi_dispatch(ctx, c, row0, row1, stream, 0, nullptr);
}

int neb_gi_trace_begin(neb_ctx* ctx, const neb_gi_constants* c, uint32_t row0, uint32_t row1, neb_stream stream)
{
    return gi_dispatch(ctx, c, row0, row1, stream, 1, nullptr);
}

int neb_gi_trace_finish(neb_ctx* ctx, neb_stream stream, void* after_shade_event)
{
    if (!ctx || !ctx->gi)
        return ctx ? gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace_finish: no scene") : NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    if (g->begun == g->finished)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_trace_finish: nothing begun (neb_gi_trace_begin first)");
    const GiState::DispatchSet& ds = g->sets[g->finished & 1u];
    const neb_gi_constants c = ds.split_c;
    return gi_dispatch(ctx, &c, ds.split_row0, ds.split_row1, stream, 2, (hipEvent_t)after_shade_event);
}

int neb_gi_resolve(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (int rc = neb::svgf_flush_pending(ctx))
        return rc;
    GiState* g = ctx->gi;
    if (!g || !g->defer_resolve || (g->defer_resolve == 2 && g->resolves == g->traces))
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_resolve: nothing pending (set option gi_defer_resolve=1 and call neb_gi_trace first)");
    uint32_t which = g->defer_resolve == 2 ? (g->resolves & 1u) : 0u; // (two sets: the oldest dispatch first)
    if (g->defer_resolve == 1) {
        const bool a0 = g->sets[0].awaiting_resolve, a1 = g->sets[1].awaiting_resolve;
        if (!a0 && !a1)
            return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_resolve: nothing pending (set option gi_defer_resolve=1 and call neb_gi_trace first)");
        which = (a0 && a1) ? (g->sets[1].resolve_seq < g->sets[0].resolve_seq ? 1u : 0u) : (a1 ? 1u : 0u);
        g->sets[which].awaiting_resolve = false;
    }
    GiState::DispatchSet& ds = g->sets[which];
    if (!ds.d_records)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_resolve: nothing pending (set option gi_defer_resolve=1 and call neb_gi_trace first)");
    if (g->defer_resolve == 2)
        g->resolves++;
    if (ds.pending_row1 <= ds.pending_row0)
        return NEB_OK;
    GI_GUARD(ctx);
    ScopedRange range("GI: Resolve query data"); // "GI: Resolve NRC query data", DeferredRenderer.cpp:567
    const size_t npx = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    const size_t first = (size_t)(ds.pending_row0 - ctx->row_begin) * ctx->W, n = (size_t)(ds.pending_row1 - ds.pending_row0) * ctx->W;
    hipLaunchKernelGGL(gi_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float4*)ctx->planes[NEB_PLANE_RADIANCE][ctx->cur], ds.d_records + 5 * npx, first, n, 1.0f / (float)ds.pending_spp);
    GI_HIP(ctx, hipGetLastError());
    return NEB_OK;
}

int neb_pbr_direct(neb_ctx* ctx, const neb_gi_constants* c, neb_stream stream)
{
    if (!ctx || !c)
        return ctx ? gi_fail(ctx, NEB_ERR_INVALID_ARG, "neb_pbr_direct: null constants") : NEB_ERR_INVALID_ARG;
    if (int rc = neb::svgf_flush_pending(ctx)) // (a held-back temporal pass reads / writes the planes this call touches)
        return rc;
    GiState* g = ctx->gi;
    if (!g || !g->built)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_pbr_direct: scene/BVH not ready (neb_gi_set_scene + neb_gi_build_bvh)");
    GI_GUARD(ctx);
    ScopedRange range("PBR Direct Lighting + Shadows"); // DeferredRenderer.cpp:338
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
    if (!g->d_direct_counts) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, 3 * n_blocks * sizeof(uint32_t))); // {bounce, shadow, shadow answered by the sun table} rays per workgroup
        GI_HIP(ctx, hipMemset(p, 0, 3 * n_blocks * sizeof(uint32_t)));
        g->allocs.push_back(p);
        g->d_direct_counts = (uint32_t*)p;
        g->n_block_counts = n_blocks;
    }
    a.bounce_counts = g->d_direct_counts;
    a.shadow_counts = g->d_direct_counts + g->n_block_counts;
    a.table_counts = g->d_direct_counts + 2 * g->n_block_counts;
    hipLaunchKernelGGL(pbr_direct_kernel, dim3(a.tiles_x * tiles_y), dim3(64), 0, (hipStream_t)stream, a);
    GI_HIP(ctx, hipGetLastError());
    return NEB_OK;
}

int neb_tonemap(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (int rc = neb::svgf_flush_pending(ctx))
        return rc;
    GI_GUARD(ctx);
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
    GI_GUARD(ctx);
    unsigned long long v[16] = {};
    uint32_t* arrays[3] = {g->sets[0].d_block_counts, g->sets[1].d_block_counts, g->d_direct_counts}; // (every dispatch set, and the direct pass)
    const size_t n3 = 3 * g->n_block_counts;
    std::vector<uint32_t> counts(3 * n3);
    GI_HIP(ctx, hipMemcpyAsync(v, g->d_ray_counter, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    for (int q = 0; q < 3; ++q)
        if (arrays[q])
            GI_HIP(ctx, hipMemcpyAsync(counts.data() + q * n3, arrays[q], n3 * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GI_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    if (reset) {
        GI_HIP(ctx, hipMemsetAsync(g->d_ray_counter, 0, sizeof(v), (hipStream_t)stream));
        for (int q = 0; q < 3; ++q)
            if (arrays[q])
                GI_HIP(ctx, hipMemsetAsync(arrays[q], 0, n3 * sizeof(uint32_t), (hipStream_t)stream));
    }
    unsigned long long total = 0, table = 0;
    for (int q = 0; q < 3; ++q) {
        for (size_t k = 0; k < 2 * g->n_block_counts; ++k)
            total += counts[q * n3 + k];
        for (size_t k = 2 * g->n_block_counts; k < n3; ++k)
            table += counts[q * n3 + k];
    }
    g->table_rays = table;
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

int neb_gi_wave_stats(neb_ctx* ctx, uint64_t out[6])
{
    if (!ctx || !ctx->gi || !out)
        return NEB_ERR_INVALID_ARG;
    for (int k = 0; k < 6; ++k)
        out[k] = ctx->gi->last_stats[5 + k];
    return NEB_OK;
}

int neb_gi_debug_set_tile_order(neb_ctx* ctx, const uint32_t* order, uint32_t n)
{
    if (!ctx || !ctx->gi)
        return NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    GI_GUARD(ctx);
    GI_HIP(ctx, hipDeviceSynchronize());
    if (!order || n == 0) {
        g->tile_order_n = 0;
        return NEB_OK;
    }
    if (g->tile_order_cap < n) {
        void* p = nullptr;
        GI_HIP(ctx, hipMalloc(&p, (size_t)n * sizeof(uint32_t)));
        g->allocs.push_back(p);
        g->d_tile_order = (uint32_t*)p;
        g->tile_order_cap = n;
    }
    GI_HIP(ctx, hipMemcpy(g->d_tile_order, order, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    g->tile_order_n = n;
    return NEB_OK;
}

int neb_gi_debug_sun_walk_stats(neb_ctx* ctx, uint64_t out[12])
{
    if (!ctx || !ctx->gi || !out || !ctx->gi->d_sun_counts)
        return NEB_ERR_INVALID_ARG;
    GI_GUARD(ctx);
    GI_HIP(ctx, hipDeviceSynchronize());
    GI_HIP(ctx, hipMemcpy(out, ctx->gi->d_sun_counts + 8, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return NEB_OK;
}

int neb_gi_node_index_stats(neb_ctx* ctx, uint64_t out[5])
{
    if (!ctx || !ctx->gi || !out)
        return NEB_ERR_INVALID_ARG;
    for (int k = 0; k < 5; ++k)
        out[k] = ctx->gi->last_stats[11 + k];
    return NEB_OK;
}

int neb_gi_sun_table_stats(neb_ctx* ctx, uint64_t out[4], neb_stream stream)
{
    if (!ctx || !ctx->gi || !out)
        return NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    GI_GUARD(ctx);
    unsigned long long sides[2] = {0, 0};
    if (g->d_sun_counts && g->sun_table_state == 1) {
        GI_HIP(ctx, hipMemcpyAsync(sides, g->d_sun_counts, sizeof(sides), hipMemcpyDeviceToHost, (hipStream_t)stream));
        GI_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    }
    out[0] = sides[0];
    out[1] = sides[1];
    out[2] = g->table_rays;
    out[3] = g->sun_table_builds;
    return NEB_OK;
}

int neb_gi_sun_table_build_ms(neb_ctx* ctx, float* ms)
{
    if (!ctx || !ctx->gi || !ms)
        return NEB_ERR_INVALID_ARG;
    GiState* g = ctx->gi;
    if (!g->sun_table_builds || !g->sun_build_ev[1])
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_sun_table_build_ms: no table has been built");
    GI_GUARD(ctx);
    GI_HIP(ctx, hipEventSynchronize(g->sun_build_ev[1]));
    GI_HIP(ctx, hipEventElapsedTime(ms, g->sun_build_ev[0], g->sun_build_ev[1]));
    return NEB_OK;
}

int neb_gi_shadow_tail_mode(neb_ctx* ctx, int* mode, float us[2])
{
    if (!ctx || !ctx->gi || !mode)
        return NEB_ERR_INVALID_ARG;
    const GiState* g = ctx->gi;
    *mode = g->tail_phase != 0 ? -1 : (g->tail_sorted ? 1 : 0);
    if (us)
        us[0] = g->tail_us[0], us[1] = g->tail_us[1];
    return NEB_OK;
}

int neb_gi_download_hits(neb_ctx* ctx, neb_gi_hit* host, neb_stream stream)
{
    if (!ctx || !host)
        return NEB_ERR_INVALID_ARG;
    if (!ctx->gi || !ctx->gi->d_hits)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gi_download_hits: set option gi_debug_hits=1 and trace first");
    GI_GUARD(ctx);
    const size_t npx = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    GI_HIP(ctx, hipMemcpyAsync(host, ctx->gi->d_hits, npx * sizeof(neb_gi_hit), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GI_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return NEB_OK;
}

int neb_gbuffer_raycast(neb_ctx* ctx, const neb_camera* cam, neb_stream stream)
{
    if (!ctx || !cam)
        return NEB_ERR_INVALID_ARG;
    if (int rc = neb::svgf_flush_pending(ctx))
        return rc;
    GiState* g = ctx->gi;
    if (!g || !g->built)
        return gi_fail(ctx, NEB_ERR_STATE, "neb_gbuffer_raycast: scene/BVH not ready");
    GI_GUARD(ctx);
    ScopedRange range("Deferred G-Buffers (geometry)"); // DeferredRenderer.cpp:267
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
    ctx->geom_lo = ctx->geom_hi = 0; // normal[cur] / depth[cur] change: the decoded geometry plane is stale
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
    ctx->gi->sort_shadow_auto = false; // set explicitly: no longer decided by the size of the dispatch
    ctx->gi->sort_bounce = (mask & 2) != 0;
    return NEB_OK;
}
int gi_set_max_bvh_depth(neb_ctx* ctx, int depth)
{
    if (!ctx->gi || depth < 1 || depth > (kLdsStack + kSpillStack) / 3)
        return NEB_ERR_STATE;
    ctx->gi->max_bvh_depth = (uint32_t)depth;
    return NEB_OK;
}
int gi_set_exact_shade(neb_ctx* ctx, int on)
{
    if (!ctx->gi)
        return NEB_ERR_STATE;
    ctx->gi->exact_shade = on != 0;
    return NEB_OK;
}
int gi_set_defer_resolve(neb_ctx* ctx, int on)
{
    if (!ctx->gi)
        return NEB_ERR_STATE;
    if (on < 0 || on > 2)
        return NEB_ERR_STATE;
    GiState* g = ctx->gi;
    if (g->defer_resolve == 2 && g->traces != g->resolves)
        return NEB_ERR_STATE; // (a traced dispatch still waits for its neb_gi_resolve)
    g->defer_resolve = on;
    g->traces = g->resolves = 0;
    return NEB_OK;
}
int gi_set_sun_table(neb_ctx* ctx, int on)
{
    if (!ctx->gi || on < 0 || on > 3)
        return NEB_ERR_STATE;
    GiState* g = ctx->gi;
    g->sun_table = on != 0;
    g->compact_shadow = on != 2; // 2: the table answers, but the remaining rays always keep the sorted / tiled shadow pass
    g->tail_tune = on == 1;      // 1: lists or sorted pass, whichever the measurement of this table says; 3: always the lists
    g->tail_sorted = false;
    g->tail_phase = (g->tail_tune && g->sun_table_state == 1) ? 1 : 0;
    return NEB_OK;
}
int gi_set_sun_hold(neb_ctx* ctx, int n)
{
    if (!ctx->gi || n < 0 || n == 1 || n > 100000)
        return NEB_ERR_STATE;
    ctx->gi->sun_hold_option = n;
    ctx->gi->sun_hold = n > 0 ? (uint32_t)n : 2u;
    return NEB_OK;
}
int gi_set_sun_hints(neb_ctx* ctx, int n)
{
    if (!ctx->gi || (n != 0 && n != 2 && n != 4))
        return NEB_ERR_STATE;
    ctx->gi->sun_hints = n;
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

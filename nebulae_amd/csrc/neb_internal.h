// neb_internal.h -- context layout and kernel launchers behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/nebulae_hip.h"

namespace neb {

struct PlaneInfo {
    uint32_t bytes_per_px;
    uint32_t slots;
};
extern const PlaneInfo kPlaneInfo[NEB_PLANE_COUNT];

// Every entry point that launches or copies runs on its context's device whatever device the calling thread had
// current, and leaves the thread's current device as it found it (a host may hold strip contexts on several GPUs).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        if (prev != device)
            err = hipSetDevice(device);
        else
            prev = -1; // nothing to restore
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Per-pass profiler ranges with the reference's PIX event names (NEB_PIX_SCOPED_EVENT, src/nri/PIXRuntime.h:115-117;
// uses at src/DeferredRenderer.cpp:267,338,434,567,599 and src/SVGFDenoiser.cpp:69,136,155): roctx ranges, resolved at
// run time from the ROCm profiler's marker library (no-ops when it is absent); rocprofv3 --marker-trace shows them.
void marker_push(const char* name);
void marker_pop();
struct ScopedRange {
    explicit ScopedRange(const char* name) { marker_push(name); }
    ~ScopedRange() { marker_pop(); }
    ScopedRange(const ScopedRange&) = delete;
    ScopedRange& operator=(const ScopedRange&) = delete;
};

struct SvgfLaunch {
    int device;            // HIP device ordinal of the context (per-device kernel attributes)
    uint32_t W, H;         // full image size (global clamp uses these)
    uint32_t row_begin;    // first resident image row: plane address of (x, y) is (y - row_begin) * W + x
    uint32_t row_end;      // one past the last resident row
    uint32_t row0, row1;   // image rows to process
    int num_cus;           // persistent-grid sizing
    neb_svgf_params p;
};

// Kernel launchers (enqueue only).  All pointers are device pointers to resident row `row_begin`.
hipError_t launch_temporal(const SvgfLaunch& L, float4* rad_cur, const float4* rad_hist, const uint32_t* depth_cur,
                           const uint32_t* depth_hist, const uint2* normal_cur, const uint2* normal_hist,
                           const uint32_t* mom_hist, uint32_t* mom_cur, uint16_t* variance, float4* geometry, hipStream_t s);

// `geometry`: {decoded shading normal.xyz, depth} per pixel (NEB_PLANE_GEOMETRY), valid for every tap row.
// variant 0 = direct-load kernel, 1 = LDS row-lattice kernel (steps <= 32; wider steps take the direct kernel)
hipError_t launch_atrous(const SvgfLaunch& L, int variant, uint32_t step, const float4* src, float4* dst,
                         const uint16_t* variance, const float4* geometry, hipStream_t s);
// true when launch_atrous(L, variant, step, ...) runs the LDS kernel (what the whole-frame fast path below requires of every level)
bool atrous_lds_serves(const SvgfLaunch& L, int variant, uint32_t step);
// Whole-frame denoise, levels after the first: src holds {r, g, b, lum} (written by the level before); dst gets {r, g, b, lum},
// or -- last -- {r, g, b, alpha} with the alpha dst itself holds (radiance[cur] = the frame's input until that store)
hipError_t launch_atrous_lum(const SvgfLaunch& L, uint32_t step, bool last, const float4* src, float4* dst, const uint16_t* variance,
                             const float4* geometry, hipStream_t s);
// Whole-frame denoise, level 0 with the temporal pass as its staging phase: reads the frame's planes, writes moments[cur],
// variance, the geometry plane and the filtered level-0 radiance ({r, g, b, lum}; {r, g, b, alpha} when it is the only level) to dst
hipError_t launch_atrous_fused_temporal(const SvgfLaunch& L, bool only_level, const float4* rad_cur, const float4* rad_hist, const uint32_t* depth_cur,
                                        const uint32_t* depth_hist, const uint2* normal_cur, const uint2* normal_hist, const uint32_t* mom_hist,
                                        uint32_t* mom_cur, uint16_t* variance, float4* geometry, float4* dst, hipStream_t s);
// decodes depth / normal rows [row0, row1) (all W columns) into the geometry plane
hipError_t launch_decode_geometry(uint32_t W, uint32_t row_begin, uint32_t row0, uint32_t row1, const uint32_t* depth, const uint2* normal,
                                  float4* geometry, hipStream_t s);
hipError_t launch_decode_geometry2(uint32_t W, uint32_t row_begin, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, const uint32_t* depth,
                                   const uint2* normal, float4* geometry, hipStream_t s);

// raysort.hip: stable LSD radix sort of (key, value) pairs on key bits [0, bits), bits <= 16 (enqueue only)
size_t ray_sort_scratch_bytes(size_t n);
hipError_t ray_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t* keys_tmp, uint32_t* vals_tmp, uint32_t* vals_out, size_t n, int bits,
                          void* scratch, hipStream_t stream);

// A neb_svgf_temporal call on a whole-frame context is held back until the neb_svgf_atrous call that normally follows it
// (they then run as one fused chain); every other entry point that reads, writes or orders work on the planes calls this
// first, which submits the held-back pass on its own (the stand-alone kernel) -- NEB_OK or the failing status.
int svgf_flush_pending(neb_ctx* ctx);

struct GiState;               // gi.hip / gi_build.hip: scene tables, BVH, counters
void gi_destroy(GiState* g);
void gi_on_resize(GiState* g);
int gi_set_debug_hits(neb_ctx* ctx, int on);
int gi_set_defer_resolve(neb_ctx* ctx, int on);
int gi_set_sort_rays(neb_ctx* ctx, int mask);
int gi_set_max_bvh_depth(neb_ctx* ctx, int depth);
int gi_set_exact_shade(neb_ctx* ctx, int on);
int gi_set_sun_table(neb_ctx* ctx, int on);
int gi_set_sun_hints(neb_ctx* ctx, int n);
int gi_set_sun_hold(neb_ctx* ctx, int n);

} // namespace neb

struct neb_ctx {
    int device = 0;
    int num_cus = 256; // hipDeviceProp_t::multiProcessorCount (persistent-grid sizing)
    uint32_t W = 0, H = 0, row_begin = 0, row_end = 0, levels = 4;
    void* planes[NEB_PLANE_COUNT][2] = {};
    int cur = 0, hist = 1;
    uint32_t geom_lo = 0, geom_hi = 0; // image rows [geom_lo, geom_hi) of the geometry plane hold this frame's decoded normal / depth
    neb_svgf_params params{};
    int atrous_variant = 1; // 0 = direct-load kernel, 1 = LDS row-lattice kernel
    int fuse = 0;           // option svgf_fuse (opt-in): hold neb_svgf_temporal of a whole frame back so that neb_svgf_atrous can run both as the fused chain
    bool pending_temporal = false; // a neb_svgf_temporal held back for the fused chain
    hipStream_t pending_stream = nullptr;
    int profile = 0;                   // option svgf_profile: neb_svgf_atrous brackets its kernels with events (neb_svgf_level_times)
    std::vector<hipEvent_t> prof_events; // levels + 1 of them once profiling has run
    uint32_t prof_recorded = 0;
    neb::GiState* gi = nullptr;
    // neb_strip_frame* (strips.hip): a side stream for the halo exchange beside level 0, and the events that order it -- created on first use
    struct StripSync {
        hipStream_t xstream = nullptr;
        hipEvent_t ready = nullptr, done = nullptr; // launch stream -> side stream, side stream -> launch stream
        hipEvent_t pushed = nullptr;                // local transport: this strip's boundary rows are in its neighbours' halos
        hipEvent_t frame_done = nullptr;            // ... this strip's frame has been enqueued to its end (its halo rows may be overwritten after it)
        bool frame_done_recorded = false;
    } strip;
    std::string last_error;
};

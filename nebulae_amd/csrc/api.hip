// api.hip -- the C ABI of libnebulae_hip.so (include/nebulae_hip.h): context, planes, SVGF frame logic.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "neb_internal.h"

namespace neb {
const PlaneInfo kPlaneInfo[NEB_PLANE_COUNT] = {
    {16, 2}, // RADIANCE     R32G32B32A32_FLOAT
    {8, 2},  // NORMAL       R16G16B16A16_FLOAT
    {4, 2},  // DEPTH        R24G8
    {4, 2},  // MOMENTS      R16G16_FLOAT
    {2, 1},  // VARIANCE     R16_FLOAT
    {16, 1}, // SCRATCH      R32G32B32A32_FLOAT
    {4, 1},  // ALBEDO       R11G11B10_FLOAT
    {4, 1},  // ROUGH_METAL  R16G16_FLOAT
    {8, 1},  // WORLDPOS     R16G16B16A16_FLOAT
    {4, 1},  // LDR          R8G8B8A8_UNORM
    {16, 1}, // GEOMETRY     R32G32B32A32_FLOAT (decoded shading normal + depth of the current frame)
};
} // namespace neb

namespace neb {
namespace {
using PushFn = int (*)(const char*);
using PopFn = int (*)();
PushFn g_range_push = nullptr;
PopFn g_range_pop = nullptr;
std::once_flag g_range_once;
void resolve_markers()
{
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            continue;
        g_range_push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
        g_range_pop = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
        if (g_range_push && g_range_pop)
            return;
        g_range_push = nullptr;
        g_range_pop = nullptr;
    }
}
} // namespace
void marker_push(const char* name)
{
    std::call_once(g_range_once, resolve_markers);
    if (g_range_push)
        (void)g_range_push(name);
}
void marker_pop()
{
    if (g_range_pop)
        (void)g_range_pop();
}
} // namespace neb

using namespace neb;

static thread_local std::string g_create_error;
static void geometry_invalidate(neb_ctx* ctx);

static int fail(neb_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
    else
        snprintf(buf, sizeof(buf), "%s", what);
    if (ctx)
        ctx->last_error = buf;
    else
        g_create_error = buf;
    return code;
}

#define NEB_HIP(ctx, call)                                  \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess)                               \
            return fail((ctx), NEB_ERR_HIP, #call, e_);     \
    } while (0)

// scoped: run the rest of the entry point on the context's device, restore the caller's device on return
#define NEB_GUARD(ctx)                                                       \
    DeviceGuard neb_guard_((ctx)->device);                                   \
    if (neb_guard_.err != hipSuccess)                                        \
    return fail((ctx), NEB_ERR_HIP, "hipSetDevice", neb_guard_.err)

static void free_planes(neb_ctx* ctx)
{
    for (int p = 0; p < NEB_PLANE_COUNT; ++p)
        for (int s = 0; s < 2; ++s)
            if (ctx->planes[p][s]) {
                (void)hipFree(ctx->planes[p][s]);
                ctx->planes[p][s] = nullptr;
            }
}

static int alloc_planes(neb_ctx* ctx)
{
    // InitSVGFResources (SVGFDenoiser.cpp:283-359).  The reference never clears radiance/moments
    // (SURVEY.md quirk 8); this build defines zero-initialised planes.
    const size_t npx = (size_t)ctx->W * (ctx->row_end - ctx->row_begin);
    for (int p = 0; p < NEB_PLANE_COUNT; ++p)
        for (uint32_t s = 0; s < kPlaneInfo[p].slots; ++s) {
            const size_t bytes = npx * kPlaneInfo[p].bytes_per_px;
            NEB_HIP(ctx, hipMalloc(&ctx->planes[p][s], bytes));
            NEB_HIP(ctx, hipMemset(ctx->planes[p][s], 0, bytes));
        }
    NEB_HIP(ctx, hipDeviceSynchronize());
    return NEB_OK;
}

extern "C" {

const char* neb_version(void) { return "nebulae_hip 0.1 (gfx950)"; }

const char* neb_last_error(const neb_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int neb_create(const neb_create_info* info, neb_ctx** out_ctx)
{
    if (!info || !out_ctx)
        return fail(nullptr, NEB_ERR_INVALID_ARG, "neb_create: null argument");
    *out_ctx = nullptr;
    if (info->width == 0 || info->height == 0)
        return fail(nullptr, NEB_ERR_INVALID_ARG, "neb_create: zero-sized image");
    const uint32_t row_end = info->row_end ? info->row_end : info->height;
    if (info->row_begin >= row_end || row_end > info->height)
        return fail(nullptr, NEB_ERR_INVALID_ARG, "neb_create: bad resident row range");
    if (info->atrous_levels > 16)
        return fail(nullptr, NEB_ERR_INVALID_ARG, "neb_create: atrous_levels > 16");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, NEB_ERR_NO_DEVICE, "neb_create: no HIP device (this library has no CPU fallback)", e);
    if (info->device < 0 || info->device >= ndev)
        return fail(nullptr, NEB_ERR_INVALID_ARG, "neb_create: device ordinal out of range");
    DeviceGuard guard(info->device);
    if (guard.err != hipSuccess)
        return fail(nullptr, NEB_ERR_HIP, "hipSetDevice", guard.err);
    neb_ctx* ctx = new (std::nothrow) neb_ctx();
    if (!ctx)
        return fail(nullptr, NEB_ERR_HIP, "neb_create: out of host memory");
    ctx->device = info->device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, info->device) == hipSuccess && cus > 0)
            ctx->num_cus = cus;
    }
    ctx->W = info->width;
    ctx->H = info->height;
    ctx->row_begin = info->row_begin;
    ctx->row_end = row_end;
    ctx->levels = info->atrous_levels;
    neb_svgf_default_params(&ctx->params);
    int rc = alloc_planes(ctx);
    if (rc != NEB_OK) {
        g_create_error = ctx->last_error;
        free_planes(ctx);
        delete ctx;
        return rc;
    }
    *out_ctx = ctx;
    return NEB_OK;
}

int neb_resize(neb_ctx* ctx, uint32_t width, uint32_t height)
{
    if (!ctx || width == 0 || height == 0)
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_resize: bad argument");
    if (ctx->row_begin != 0 || ctx->row_end != ctx->H)
        return fail(ctx, NEB_ERR_STATE, "neb_resize: only a full-image context can be resized");
    ctx->pending_temporal = false; // (the planes it would have written are about to be freed)
    NEB_GUARD(ctx);
    NEB_HIP(ctx, hipDeviceSynchronize());
    free_planes(ctx);
    gi_on_resize(ctx->gi);
    ctx->W = width;
    ctx->H = height;
    ctx->row_begin = 0;
    ctx->row_end = height;
    return alloc_planes(ctx);
}

int neb_destroy(neb_ctx* ctx)
{
    if (!ctx)
        return NEB_OK;
    DeviceGuard guard(ctx->device);
    (void)hipDeviceSynchronize();
    free_planes(ctx);
    for (hipEvent_t e : ctx->prof_events)
        (void)hipEventDestroy(e);
    for (hipEvent_t e : {ctx->strip.ready, ctx->strip.done, ctx->strip.pushed, ctx->strip.frame_done})
        if (e)
            (void)hipEventDestroy(e);
    if (ctx->strip.xstream)
        (void)hipStreamDestroy(ctx->strip.xstream);
    gi_destroy(ctx->gi);
    delete ctx;
    return NEB_OK;
}

int neb_marker_push(const char* name)
{
    if (!name)
        return NEB_ERR_INVALID_ARG;
    marker_push(name);
    return NEB_OK;
}

int neb_marker_pop(void)
{
    marker_pop();
    return NEB_OK;
}

int neb_begin_frame(neb_ctx* ctx, uint32_t frame_index)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    ctx->cur = (int)(frame_index & 1u); // SVGFDenoiser.cpp:41-42
    ctx->hist = ctx->cur ^ 1;
    geometry_invalidate(ctx); // a new frame has a new G-buffer
    return NEB_OK;
}

int neb_end_frame(neb_ctx* ctx) { return ctx ? svgf_flush_pending(ctx) : NEB_ERR_INVALID_ARG; } // SVGFDenoiser.cpp:45-47

int neb_current_index(const neb_ctx* ctx) { return ctx ? ctx->cur : NEB_ERR_INVALID_ARG; }
int neb_history_index(const neb_ctx* ctx) { return ctx ? ctx->hist : NEB_ERR_INVALID_ARG; }

int neb_svgf_default_params(neb_svgf_params* out)
{
    if (!out)
        return NEB_ERR_INVALID_ARG;
    out->depthSigma = 0.002f; // SVGFDenoiser.h:79-81
    out->alpha = 0.9f;
    out->varianceEps = 1e-4f;
    out->phiColor = 4.0f / 255.0f; // SVGFDenoiser.h:89-91
    out->phiNormal = 128.0f;
    out->phiDepth = 0.002f;
    return NEB_OK;
}

int neb_svgf_set_params(neb_ctx* ctx, const neb_svgf_params* p)
{
    if (!ctx || !p)
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_svgf_set_params: null argument");
    if (!(p->depthSigma > 0.f) || !(p->phiDepth > 0.f) || !(p->phiNormal > 0.f))
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_svgf_set_params: depthSigma, phiDepth, phiNormal must be > 0");
    if (int rc = svgf_flush_pending(ctx)) // (a held-back temporal pass was submitted with the constants of its own call)
        return rc;
    ctx->params = *p;
    return NEB_OK;
}

int neb_svgf_get_params(const neb_ctx* ctx, neb_svgf_params* out)
{
    if (!ctx || !out)
        return NEB_ERR_INVALID_ARG;
    *out = ctx->params;
    return NEB_OK;
}

int neb_set_option(neb_ctx* ctx, const char* key, int value)
{
    if (!ctx || !key)
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_set_option: null argument");
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    if (!strcmp(key, "atrous_variant")) {
        if (value < 0 || value > 1)
            return fail(ctx, NEB_ERR_INVALID_ARG, "neb_set_option: atrous_variant must be 0 (direct kernel) or 1 (LDS kernel)");
        ctx->atrous_variant = value;
        return NEB_OK;
    }
    if (!strcmp(key, "svgf_profile")) {
        if (value < 0 || value > 2)
            return fail(ctx, NEB_ERR_INVALID_ARG, "neb_set_option: svgf_profile must be 0, 1 (an event in front of every kernel) or 2 (first kernel / the rest)");
        ctx->profile = value;
        ctx->prof_recorded = 0;
        return NEB_OK;
    }
    if (!strcmp(key, "svgf_fuse")) {
        if (value < 0 || value > 1)
            return fail(ctx, NEB_ERR_INVALID_ARG, "neb_set_option: svgf_fuse must be 0 or 1");
        ctx->fuse = value;
        return NEB_OK;
    }
    if (!strcmp(key, "gi_sort_rays")) {
        if (gi_set_sort_rays(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_sort_rays needs a scene and a mask 0..3 (bit 0 shadow rays, bit 1 bounce rays)");
        return NEB_OK;
    }
    if (!strcmp(key, "gi_defer_resolve")) {
        if (gi_set_defer_resolve(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_defer_resolve needs a scene (neb_gi_set_scene)");
        return NEB_OK;
    }
    if (!strcmp(key, "gi_exact_shade")) {
        if (gi_set_exact_shade(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_exact_shade needs a scene (neb_gi_set_scene)");
        return NEB_OK;
    }
    if (!strcmp(key, "gi_sun_table")) {
        if (gi_set_sun_table(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_sun_table needs a scene (neb_gi_set_scene) and 0 .. 3");
        return NEB_OK;
    }
    if (!strcmp(key, "gi_sun_hold")) {
        if (gi_set_sun_hold(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_sun_hold needs a scene (neb_gi_set_scene) and 0 or 2 .. 100000");
        return NEB_OK;
    }
    if (!strcmp(key, "gi_sun_hints")) {
        if (gi_set_sun_hints(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_sun_hints needs a scene (neb_gi_set_scene) and 0, 2 or 4");
        return NEB_OK;
    }
    if (!strcmp(key, "gi_max_bvh_depth")) {
        if (gi_set_max_bvh_depth(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_max_bvh_depth needs a scene and a depth in 1..21");
        return NEB_OK;
    }
    if (!strcmp(key, "gi_debug_hits")) {
        if (gi_set_debug_hits(ctx, value) != NEB_OK)
            return fail(ctx, NEB_ERR_STATE, "neb_set_option: gi_debug_hits needs a scene (neb_gi_set_scene)");
        return NEB_OK;
    }
    return fail(ctx, NEB_ERR_INVALID_ARG, "neb_set_option: unknown key");
}

static int resolve_slot(const neb_ctx* ctx, int plane, int slot)
{
    if (plane < 0 || plane >= NEB_PLANE_COUNT)
        return -1;
    if (kPlaneInfo[plane].slots == 1)
        return (slot == 0 || slot == NEB_SLOT_CURRENT) ? 0 : -1;
    if (slot == NEB_SLOT_CURRENT)
        return ctx->cur;
    if (slot == NEB_SLOT_HISTORY)
        return ctx->hist;
    return (slot == 0 || slot == 1) ? slot : -1;
}

int neb_get_plane(neb_ctx* ctx, int plane, int slot, void** dptr, size_t* pitch_bytes, uint32_t* rows)
{
    if (!ctx || !dptr)
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_get_plane: null argument");
    const int s = resolve_slot(ctx, plane, slot);
    if (s < 0)
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_get_plane: bad plane/slot");
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    if (plane == NEB_PLANE_NORMAL || plane == NEB_PLANE_DEPTH)
        geometry_invalidate(ctx); // the caller may write through the pointer: decode again before the next a-trous level
    *dptr = ctx->planes[plane][s];
    if (pitch_bytes)
        *pitch_bytes = (size_t)ctx->W * kPlaneInfo[plane].bytes_per_px;
    if (rows)
        *rows = ctx->row_end - ctx->row_begin;
    return NEB_OK;
}

static int copy_rows(neb_ctx* ctx, int plane, int slot, uint32_t row0, uint32_t nrows, void* host, bool upload,
                     neb_stream stream)
{
    if (!ctx || !host)
        return fail(ctx, NEB_ERR_INVALID_ARG, "copy rows: null argument");
    const int s = resolve_slot(ctx, plane, slot);
    if (s < 0)
        return fail(ctx, NEB_ERR_INVALID_ARG, "copy rows: bad plane/slot");
    if (row0 < ctx->row_begin || row0 + nrows > ctx->row_end)
        return fail(ctx, NEB_ERR_OUT_OF_RANGE, "copy rows: rows not resident in this context");
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    const size_t pitch = (size_t)ctx->W * kPlaneInfo[plane].bytes_per_px;
    char* d = (char*)ctx->planes[plane][s] + (size_t)(row0 - ctx->row_begin) * pitch;
    NEB_GUARD(ctx);
    if (upload && (plane == NEB_PLANE_NORMAL || plane == NEB_PLANE_DEPTH))
        geometry_invalidate(ctx);
    if (upload)
        NEB_HIP(ctx, hipMemcpyAsync(d, host, pitch * nrows, hipMemcpyHostToDevice, (hipStream_t)stream));
    else
        NEB_HIP(ctx, hipMemcpyAsync(host, d, pitch * nrows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return NEB_OK;
}

int neb_upload_rows(neb_ctx* ctx, int plane, int slot, uint32_t row0, uint32_t nrows, const void* host, neb_stream stream)
{
    return copy_rows(ctx, plane, slot, row0, nrows, const_cast<void*>(host), true, stream);
}

int neb_download_rows(neb_ctx* ctx, int plane, int slot, uint32_t row0, uint32_t nrows, void* host, neb_stream stream)
{
    return copy_rows(ctx, plane, slot, row0, nrows, host, false, stream);
}

int neb_stream_synchronize(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    NEB_GUARD(ctx);
    NEB_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return NEB_OK;
}

// ---- SVGF ----

int neb_svgf_reset_history(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    // CopyResource(history <- current); moments/variance are NOT reset (SVGFDenoiser.cpp:57).
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    NEB_GUARD(ctx);
    const size_t bytes = (size_t)ctx->W * (ctx->row_end - ctx->row_begin) * 16;
    NEB_HIP(ctx, hipMemcpyAsync(ctx->planes[NEB_PLANE_RADIANCE][ctx->hist], ctx->planes[NEB_PLANE_RADIANCE][ctx->cur],
                                bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return NEB_OK;
}

// ---- decoded geometry plane: which image rows of it match normal[cur] / depth[cur] of this frame ----
static void geometry_invalidate(neb_ctx* ctx) { ctx->geom_lo = ctx->geom_hi = 0; }

static void geometry_mark(neb_ctx* ctx, uint32_t row0, uint32_t row1)
{
    if (row0 >= row1)
        return;
    if (ctx->geom_lo < ctx->geom_hi && row0 <= ctx->geom_hi && row1 >= ctx->geom_lo) { // touches the valid run: extend it
        ctx->geom_lo = row0 < ctx->geom_lo ? row0 : ctx->geom_lo;
        ctx->geom_hi = row1 > ctx->geom_hi ? row1 : ctx->geom_hi;
    } else {
        ctx->geom_lo = row0;
        ctx->geom_hi = row1;
    }
}

static hipError_t geometry_ensure(neb_ctx* ctx, uint32_t row0, uint32_t row1, hipStream_t stream)
{
    auto decode = [&](uint32_t a, uint32_t b) {
        return a < b ? launch_decode_geometry(ctx->W, ctx->row_begin, a, b, (const uint32_t*)ctx->planes[NEB_PLANE_DEPTH][ctx->cur],
                                              (const uint2*)ctx->planes[NEB_PLANE_NORMAL][ctx->cur], (float4*)ctx->planes[NEB_PLANE_GEOMETRY][0], stream)
                     : hipSuccess;
    };
    hipError_t e = hipSuccess;
    if (ctx->geom_lo >= ctx->geom_hi || row1 < ctx->geom_lo || row0 > ctx->geom_hi) { // nothing valid nearby: decode the request
        e = decode(row0, row1);
        ctx->geom_lo = row0;
        ctx->geom_hi = row1;
        return e;
    }
    if (row0 < ctx->geom_lo || row1 > ctx->geom_hi) {
        // rows beside the valid run are missing: a strip's halo rows, which every level from here on taps.  ALL resident rows that are not
        // valid yet are decoded now, both sides in one launch (a 135-row strip's frame: two 4.6-us launches less than decoding per request)
        e = launch_decode_geometry2(ctx->W, ctx->row_begin, ctx->row_begin, ctx->geom_lo, ctx->geom_hi, ctx->row_end,
                                    (const uint32_t*)ctx->planes[NEB_PLANE_DEPTH][ctx->cur], (const uint2*)ctx->planes[NEB_PLANE_NORMAL][ctx->cur],
                                    (float4*)ctx->planes[NEB_PLANE_GEOMETRY][0], stream);
        ctx->geom_lo = ctx->row_begin;
        ctx->geom_hi = ctx->row_end;
    }
    return e;
}

// option svgf_profile: event k is recorded in front of the k-th kernel of neb_svgf_atrous's chain (k = levels: behind the last)
// (svgf_profile = 2: only in front of the first kernel, in front of the second and behind the last -- every event is a packet of its
// own between two launches, ~2 us; the levels after the first are then timed as one interval with none inside)
static void profile_mark(neb_ctx* ctx, uint32_t k, hipStream_t stream)
{
    if (!ctx->profile)
        return;
    if (ctx->profile == 2) {
        if (k > 1 && k != ctx->levels)
            return;
        if (k > 1)
            k = 2;
    }
    while (ctx->prof_events.size() <= k) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess)
            return;
        ctx->prof_events.push_back(e);
    }
    if (hipEventRecord(ctx->prof_events[k], stream) == hipSuccess && ctx->prof_recorded < k + 1)
        ctx->prof_recorded = k + 1;
}

static SvgfLaunch make_launch(const neb_ctx* ctx, uint32_t row0, uint32_t row1)
{
    SvgfLaunch L;
    L.device = ctx->device;
    L.W = ctx->W;
    L.H = ctx->H;
    L.row_begin = ctx->row_begin;
    L.row_end = ctx->row_end;
    L.row0 = row0;
    L.row1 = row1;
    L.num_cus = ctx->num_cus;
    L.p = ctx->params;
    return L;
}

static int svgf_fused_chain(neb_ctx* ctx, hipStream_t stream);

int neb_svgf_temporal_rows(neb_ctx* ctx, uint32_t row0, uint32_t row1, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (row0 < ctx->row_begin || row1 > ctx->row_end || row0 > row1)
        return fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_svgf_temporal_rows: rows not resident");
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    const int c = ctx->cur, h = ctx->hist;
    NEB_GUARD(ctx);
    ScopedRange range("SVGF: Temporal Accumulation"); // SVGFDenoiser.cpp:69
    // the pass also leaves normal[cur] / depth[cur] decoded in the geometry plane for the a-trous levels -- for the pixels it
    // covers: with a ragged right / bottom remainder (Dispatch(W/8, H/8) floors) the levels' taps clamp into pixels it skips
    const bool fused_geometry = (ctx->W % 8u) == 0 && (ctx->H % 8u) == 0;
    hipError_t e = launch_temporal(make_launch(ctx, row0, row1), (float4*)ctx->planes[NEB_PLANE_RADIANCE][c],
                                   (const float4*)ctx->planes[NEB_PLANE_RADIANCE][h],
                                   (const uint32_t*)ctx->planes[NEB_PLANE_DEPTH][c],
                                   (const uint32_t*)ctx->planes[NEB_PLANE_DEPTH][h],
                                   (const uint2*)ctx->planes[NEB_PLANE_NORMAL][c],
                                   (const uint2*)ctx->planes[NEB_PLANE_NORMAL][h],
                                   (const uint32_t*)ctx->planes[NEB_PLANE_MOMENTS][h],
                                   (uint32_t*)ctx->planes[NEB_PLANE_MOMENTS][c],
                                   (uint16_t*)ctx->planes[NEB_PLANE_VARIANCE][0],
                                   fused_geometry ? (float4*)ctx->planes[NEB_PLANE_GEOMETRY][0] : nullptr, (hipStream_t)stream);
    if (e != hipSuccess)
        return fail(ctx, NEB_ERR_HIP, "svgf_temporal launch", e);
    if (fused_geometry)
        geometry_mark(ctx, row0, row1);
    return NEB_OK;
}

// Can this context's next frame run as the fused chain (temporal pass inside level 0, intermediate planes carrying the
// luminance)?  A whole frame whose every pixel both passes cover (Dispatch(W/8, H/8) floors), every level on the LDS kernel.
static bool fused_chain_possible(const neb_ctx* ctx)
{
    if (ctx->row_begin != 0 || ctx->row_end != ctx->H || (ctx->W % 8u) || (ctx->H % 8u) || ctx->levels == 0)
        return false;
    const SvgfLaunch L = make_launch(ctx, 0, ctx->H);
    for (uint32_t i = 0; i < ctx->levels; ++i)
        if (!atrous_lds_serves(L, ctx->atrous_variant, 1u << i))
            return false;
    return true;
}

int neb_svgf_temporal(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    // Stream-ordered by default, as the reference records at the call (SVGFDenoiser.cpp:116).  Only a host that opted in
    // ("svgf_fuse" = 1: it promises to order work on the planes through neb_* calls) gets the pass held back for the fused chain:
    // neb_svgf_atrous then runs it inside level 0; anything else submits it first.
    if (ctx->fuse && fused_chain_possible(ctx)) {
        ctx->pending_temporal = true;
        ctx->pending_stream = (hipStream_t)stream;
        return NEB_OK;
    }
    return neb_svgf_temporal_rows(ctx, ctx->row_begin, ctx->row_end, stream);
}

// Ping-pong chain of SubmitATrousComputeWavelet (SVGFDenoiser.cpp:146-196): src = cur, dst = hist,
// swapped every level.  The reference asserts an even level count (:197); for odd counts the
// chain runs cur -> hist -> scratch -> hist -> ... -> scratch -> cur through the third radiance
// plane (the reference's unused m_denoisedOutput) so the final image still lands in
// radiance[cur] and becomes next frame's history.  L == 1 filters into
// scratch and copies back.
static void chain_link(const neb_ctx* ctx, uint32_t level, int* sp, int* ss, int* dp, int* ds)
{
    const uint32_t L = ctx->levels;
    auto node = [&](uint32_t i, int* plane, int* slot) { // buffer holding the input of level i (i == L: result)
        *plane = NEB_PLANE_RADIANCE;
        if (i == 0 || (i == L && L != 1)) {
            *slot = ctx->cur;
        } else if (L == 1) {
            *plane = NEB_PLANE_SCRATCH;
            *slot = 0;
        } else if ((L & 1u) == 0) {
            *slot = (i & 1u) ? ctx->hist : ctx->cur;
        } else if (i & 1u) {
            *slot = ctx->hist;
        } else {
            *plane = NEB_PLANE_SCRATCH;
            *slot = 0;
        }
    };
    node(level, sp, ss);
    node(level + 1, dp, ds);
}

int neb_svgf_atrous_level_planes(const neb_ctx* ctx, uint32_t level, int* src_plane, int* src_slot, int* dst_plane,
                                 int* dst_slot)
{
    if (!ctx || !src_plane || !src_slot || !dst_plane || !dst_slot || level >= ctx->levels)
        return NEB_ERR_INVALID_ARG;
    chain_link(ctx, level, src_plane, src_slot, dst_plane, dst_slot);
    return NEB_OK;
}

int neb_svgf_atrous_level_rows(neb_ctx* ctx, uint32_t level, uint32_t row0, uint32_t row1, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (level >= ctx->levels)
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_svgf_atrous_level_rows: level >= atrous_levels");
    if (row0 < ctx->row_begin || row1 > ctx->row_end || row0 > row1)
        return fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_svgf_atrous_level_rows: rows not resident");
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    const uint32_t step = 1u << level;
    // every (globally clamped) tap row must be resident: rows [row0 - 2*step, row1 - 1 + 2*step] clamped to the image
    const uint32_t Hd = (ctx->H / 8u) * 8u;
    const uint32_t r1 = row1 < Hd ? row1 : Hd;
    NEB_GUARD(ctx);
    if (row0 < r1) {
        const uint32_t lo = row0 >= 2 * step ? row0 - 2 * step : 0;
        uint32_t hi = r1 - 1 + 2 * step;
        if (hi > ctx->H - 1)
            hi = ctx->H - 1;
        if (lo < ctx->row_begin || hi >= ctx->row_end)
            return fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_svgf_atrous_level_rows: tap rows (halo) not resident");
        // the tap rows the temporal pass did not decode (halo rows of a strip, or a level run on its own) are decoded now
        hipError_t ge = geometry_ensure(ctx, lo, hi + 1, (hipStream_t)stream);
        if (ge != hipSuccess)
            return fail(ctx, NEB_ERR_HIP, "svgf geometry decode launch", ge);
    }
    int sp, ss, dp, ds;
    chain_link(ctx, level, &sp, &ss, &dp, &ds);
    char range_name[64];
    snprintf(range_name, sizeof(range_name), "SVGF: A-Trous compute %u (step %u)", level, step); // SVGFDenoiser.cpp:155
    ScopedRange range(range_name);
    hipError_t e = launch_atrous(make_launch(ctx, row0, row1), ctx->atrous_variant, step,
                                 (const float4*)ctx->planes[sp][ss], (float4*)ctx->planes[dp][ds],
                                 (const uint16_t*)ctx->planes[NEB_PLANE_VARIANCE][0],
                                 (const float4*)ctx->planes[NEB_PLANE_GEOMETRY][0], (hipStream_t)stream);
    if (e != hipSuccess)
        return fail(ctx, NEB_ERR_HIP, "svgf_atrous launch", e);
    return NEB_OK;
}

int neb_svgf_atrous(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (ctx->row_begin != 0 || ctx->row_end != ctx->H)
        return fail(ctx, NEB_ERR_STATE, "neb_svgf_atrous: context holds a row strip; use neb_svgf_atrous_level_rows");
    if (ctx->pending_temporal && ctx->pending_stream == (hipStream_t)stream && fused_chain_possible(ctx))
        return svgf_fused_chain(ctx, (hipStream_t)stream);
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    ScopedRange range("SVGF: A-Trous Wavelet"); // SVGFDenoiser.cpp:136
    for (uint32_t i = 0; i < ctx->levels; ++i) {
        profile_mark(ctx, i, (hipStream_t)stream);
        int rc = neb_svgf_atrous_level_rows(ctx, i, 0, ctx->H, stream);
        if (rc != NEB_OK)
            return rc;
    }
    profile_mark(ctx, ctx->levels, (hipStream_t)stream);
    if (ctx->levels == 1) {
        const size_t bytes = (size_t)ctx->W * ctx->H * 16;
        NEB_GUARD(ctx);
        NEB_HIP(ctx, hipMemcpyAsync(ctx->planes[NEB_PLANE_RADIANCE][ctx->cur], ctx->planes[NEB_PLANE_SCRATCH][0], bytes,
                                    hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return NEB_OK;
}

// DeferredRenderer::SubmitCommandsSVGFDenoising's two calls (src/DeferredRenderer.cpp:610-611) as ONE entry point: the fused
// chain where the context allows it, the separate kernels otherwise -- the same bits either way, nothing held back.
int neb_svgf_denoise(neb_ctx* ctx, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (ctx->row_begin != 0 || ctx->row_end != ctx->H)
        return fail(ctx, NEB_ERR_STATE, "neb_svgf_denoise: context holds a row strip; use the row-range forms");
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    if (fused_chain_possible(ctx))
        return svgf_fused_chain(ctx, (hipStream_t)stream);
    if (int rc = neb_svgf_temporal_rows(ctx, ctx->row_begin, ctx->row_end, stream))
        return rc;
    return neb_svgf_atrous(ctx, stream);
}

int neb_svgf_level_times(neb_ctx* ctx, float* out_us, uint32_t capacity, uint32_t* n_out)
{
    if (!ctx || !out_us || !n_out)
        return fail(ctx, NEB_ERR_INVALID_ARG, "neb_svgf_level_times: null argument");
    *n_out = 0;
    if (int rc = svgf_flush_pending(ctx))
        return rc;
    if (!ctx->profile || ctx->prof_recorded < 2)
        return fail(ctx, NEB_ERR_STATE, "neb_svgf_level_times: set option svgf_profile = 1 and run neb_svgf_atrous first");
    NEB_GUARD(ctx);
    const uint32_t n = ctx->prof_recorded - 1;
    NEB_HIP(ctx, hipEventSynchronize(ctx->prof_events[n]));
    for (uint32_t k = 0; k < n && k < capacity; ++k) {
        float ms = 0.f;
        NEB_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[k], ctx->prof_events[k + 1]));
        out_us[k] = ms * 1e3f;
    }
    *n_out = n < capacity ? n : capacity;
    return NEB_OK;
}

} // extern "C"

int neb::svgf_flush_pending(neb_ctx* ctx)
{
    if (!ctx || !ctx->pending_temporal)
        return NEB_OK;
    ctx->pending_temporal = false;
    return neb_svgf_temporal_rows(ctx, ctx->row_begin, ctx->row_end, (neb_stream)ctx->pending_stream);
}

// The whole-frame denoise as one chain (same results, bit for bit, as the stand-alone temporal kernel followed by the
// levels): level 0 with the temporal pass as its staging phase -- the accumulated radiance is never written --
//   (radiance[cur], radiance[hist], ...) -> scratch {r, g, b, lum}
// then levels 1 .. L-2 between radiance[hist] and scratch, luminance carried in .w so that a level's staging is two
// verbatim LDS-DMAs, and the last level into radiance[cur], which until that store holds the frame's input and supplies the
// alpha the output carries.  radiance[cur] ends up as SubmitATrousComputeWavelet leaves it (SVGFDenoiser.cpp:146-196: the
// filtered image = next frame's history), moments[cur] / variance as SubmitTemporalAccumulation does; radiance[hist] and
// scratch hold intermediate levels (as radiance[hist] does in the reference), here with the luminance in .w.
static int svgf_fused_chain(neb_ctx* ctx, hipStream_t stream)
{
    ctx->pending_temporal = false;
    NEB_GUARD(ctx);
    const int c = ctx->cur, h = ctx->hist;
    const uint32_t L = ctx->levels;
    const SvgfLaunch launch = make_launch(ctx, 0, ctx->H);
    float4* const rad_cur = (float4*)ctx->planes[NEB_PLANE_RADIANCE][c];
    float4* const rad_hist = (float4*)ctx->planes[NEB_PLANE_RADIANCE][h];
    float4* const scratch = (float4*)ctx->planes[NEB_PLANE_SCRATCH][0];
    const uint16_t* variance = (const uint16_t*)ctx->planes[NEB_PLANE_VARIANCE][0];
    float4* const geometry = (float4*)ctx->planes[NEB_PLANE_GEOMETRY][0];
    {
        ScopedRange range("SVGF: Temporal Accumulation + A-Trous compute 0 (step 1)"); // SVGFDenoiser.cpp:69,155
        profile_mark(ctx, 0, stream);
        hipError_t e = launch_atrous_fused_temporal(launch, L == 1, rad_cur, rad_hist, (const uint32_t*)ctx->planes[NEB_PLANE_DEPTH][c],
                                                    (const uint32_t*)ctx->planes[NEB_PLANE_DEPTH][h], (const uint2*)ctx->planes[NEB_PLANE_NORMAL][c],
                                                    (const uint2*)ctx->planes[NEB_PLANE_NORMAL][h], (const uint32_t*)ctx->planes[NEB_PLANE_MOMENTS][h],
                                                    (uint32_t*)ctx->planes[NEB_PLANE_MOMENTS][c], (uint16_t*)ctx->planes[NEB_PLANE_VARIANCE][0], geometry,
                                                    scratch, stream);
        if (e != hipSuccess)
            return fail(ctx, NEB_ERR_HIP, "svgf fused temporal + a-trous launch", e);
        geometry_mark(ctx, 0, ctx->H);
    }
    ScopedRange range("SVGF: A-Trous Wavelet"); // SVGFDenoiser.cpp:136
    const float4* src = scratch;
    for (uint32_t i = 1; i < L; ++i) {
        const bool last = i + 1 == L;
        float4* dst = last ? rad_cur : (src == scratch ? rad_hist : scratch);
        char range_name[64];
        snprintf(range_name, sizeof(range_name), "SVGF: A-Trous compute %u (step %u)", i, 1u << i); // SVGFDenoiser.cpp:155
        ScopedRange level_range(range_name);
        profile_mark(ctx, i, stream);
        hipError_t e = launch_atrous_lum(launch, 1u << i, last, src, dst, variance, geometry, stream);
        if (e != hipSuccess)
            return fail(ctx, NEB_ERR_HIP, "svgf_atrous launch", e);
        src = dst;
    }
    profile_mark(ctx, L, stream);
    if (L == 1) {
        const size_t bytes = (size_t)ctx->W * ctx->H * 16;
        NEB_HIP(ctx, hipMemcpyAsync(rad_cur, scratch, bytes, hipMemcpyDeviceToDevice, stream));
    }
    return NEB_OK;
}

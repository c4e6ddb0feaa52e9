/*
 * nebulae_hip.h -- C ABI of libnebulae_hip.so: the MI355X (gfx950) back end for the
 * ray-traced GI + SVGF hot path of KatanaMajesty/Nebulae.
 *
 * The reference has no plugin/FFI interface; the seam is the set of calls that
 * Renderer/DeferredRenderer make on SVGFDenoiser and on DeferredRenderer's GI
 * methods.  Each entry point below names the reference member it replaces
 * (paths relative to the reference checkout).  D3D12 command lists become HIP
 * streams: every neb_* "submit" call only ENQUEUES work on the given stream and
 * never synchronises, exactly as the reference records into a command list.
 *
 * Conventions: plain pointers and sizes only; every function returns
 * NEB_OK (0) or a negative neb_status, never throws; neb_last_error() gives
 * the message for the last failure on that context (the reference throws
 * HrException / asserts instead, src/nri/stdafx.h:31-98).  A context is not
 * thread-safe: the caller serialises, as the reference's single render thread does.
 */
#ifndef NEBULAE_HIP_H
#define NEBULAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct neb_ctx neb_ctx;
typedef void* neb_stream; /* hipStream_t (NULL = the null stream) */

typedef enum neb_status {
    NEB_OK = 0,
    NEB_ERR_INVALID_ARG = -1,
    NEB_ERR_HIP = -2,
    NEB_ERR_NO_DEVICE = -3,
    NEB_ERR_STATE = -4,
    NEB_ERR_OUT_OF_RANGE = -5
} neb_status;

/* Device-resident planes.  Formats are the reference's DXGI formats, stored
 * linear, row-major, pitch == width (src/SVGFDenoiser.h:160-168,
 * src/DeferredRenderer.cpp:758-770). */
typedef enum neb_plane {
    NEB_PLANE_RADIANCE = 0,    /* R32G32B32A32_FLOAT 16 B/px, 2 slots (SVGFDenoiser.h:143)       */
    NEB_PLANE_NORMAL = 1,      /* R16G16B16A16_FLOAT  8 B/px, 2 slots; .xy oct GN, .zw oct SN     */
    NEB_PLANE_DEPTH = 2,       /* R24G8               4 B/px, 2 slots; D24_UNORM | stencil << 24  */
    NEB_PLANE_MOMENTS = 3,     /* R16G16_FLOAT        4 B/px, 2 slots; (<Y>, <Y^2>)               */
    NEB_PLANE_VARIANCE = 4,    /* R16_FLOAT           2 B/px, 1 slot                              */
    NEB_PLANE_SCRATCH = 5,     /* R32G32B32A32_FLOAT 16 B/px, 1 slot (reference: m_denoisedOutput) */
    NEB_PLANE_ALBEDO = 6,      /* R11G11B10_FLOAT     4 B/px, 1 slot (G-buffer, GI input)         */
    NEB_PLANE_ROUGH_METAL = 7, /* R16G16_FLOAT        4 B/px, 1 slot                              */
    NEB_PLANE_WORLDPOS = 8,    /* R16G16B16A16_FLOAT  8 B/px, 1 slot                              */
    NEB_PLANE_COUNT = 9
} neb_plane;

/* Slot selectors for the 2-slot (ping-pong) planes. */
#define NEB_SLOT_CURRENT (-1) /* GetCurrentResourceIndex(), SVGFDenoiser.h:24 */
#define NEB_SLOT_HISTORY (-2) /* GetHistoryResourceIndex(), SVGFDenoiser.h:25 */

typedef struct neb_create_info {
    int32_t device;        /* HIP device ordinal */
    uint32_t width;        /* full image width  (SVGFDenoiser::Init, SVGFDenoiser.h:16) */
    uint32_t height;       /* full image height */
    uint32_t row_begin;    /* first image row resident in this context (0 on one GPU) */
    uint32_t row_end;      /* one past the last resident row (0 = height).  Multi-GPU row strips
                              keep [row_begin,row_end) = owned rows + halo rows; see DESIGN.md */
    uint32_t atrous_levels; /* NumAtrousPasses, SVGFDenoiser.h:199 (reference: 4) */
} neb_create_info;

/* SVGFTemporalConstants + SVGFAtrousConstants tunables (SVGFDenoiser.h:76-92);
 * resolution/step are filled in per dispatch as in SVGFDenoiser.cpp:74-75,158-160. */
typedef struct neb_svgf_params {
    float depthSigma;  /* 0.002  */
    float alpha;       /* 0.9    */
    float varianceEps; /* 1e-4   */
    float phiColor;    /* 4/255  */
    float phiNormal;   /* 128    */
    float phiDepth;    /* 0.002  */
} neb_svgf_params;

/* ---- lifecycle: SVGFDenoiser::Init / Resize (SVGFDenoiser.cpp:14-37), DeferredRenderer::Init ---- */
int neb_create(const neb_create_info* info, neb_ctx** out_ctx);
int neb_resize(neb_ctx* ctx, uint32_t width, uint32_t height); /* re-creates (zeroes) all planes */
int neb_destroy(neb_ctx* ctx);
const char* neb_last_error(const neb_ctx* ctx); /* ctx may be NULL: last creation error */
const char* neb_version(void);

/* ---- per-frame bracket: SVGFDenoiser::BeginFrame/EndFrame (SVGFDenoiser.cpp:39-47) ---- */
int neb_begin_frame(neb_ctx* ctx, uint32_t frame_index); /* cur = f & 1, hist = cur ^ 1 */
int neb_end_frame(neb_ctx* ctx);
int neb_current_index(const neb_ctx* ctx);
int neb_history_index(const neb_ctx* ctx);

/* ---- tunables: GetTemporalConstants()/GetATrousConstants() (SVGFDenoiser.h:83,93) ---- */
int neb_svgf_default_params(neb_svgf_params* out);
int neb_svgf_set_params(neb_ctx* ctx, const neb_svgf_params* p);
int neb_svgf_get_params(const neb_ctx* ctx, neb_svgf_params* out);
/* Implementation knobs with no reference counterpart (A/B arms for profiling):
 *   "atrous_variant": 1 = LDS row-lattice kernel (default), 0 = direct-load kernel. */
int neb_set_option(neb_ctx* ctx, const char* key, int value);

/* ---- resource sharing: the ~25 getters of SVGFDenoiser.h:24-70 collapse into one call.
 * Returns a borrowed device pointer to image row `row_begin` of the plane, valid until
 * resize/destroy; pitch in bytes; rows = row_end - row_begin. ---- */
int neb_get_plane(neb_ctx* ctx, int plane, int slot, void** dptr, size_t* pitch_bytes, uint32_t* rows);
/* Host <-> device convenience for harnesses (rows are image rows, must be resident).
 * Asynchronous on `stream`; the host buffer must stay valid until the stream is synchronised. */
int neb_upload_rows(neb_ctx* ctx, int plane, int slot, uint32_t row0, uint32_t nrows, const void* host, neb_stream stream);
int neb_download_rows(neb_ctx* ctx, int plane, int slot, uint32_t row0, uint32_t nrows, void* host, neb_stream stream);
int neb_stream_synchronize(neb_ctx* ctx, neb_stream stream);

/* ---- SVGF entry points (enqueue only) ---- */
/* SVGFDenoiser::ResetHistory (SVGFDenoiser.cpp:49-64): radiance[hist] <- radiance[cur]. */
int neb_svgf_reset_history(neb_ctx* ctx, neb_stream stream);
/* SVGFDenoiser::SubmitTemporalAccumulation (SVGFDenoiser.cpp:66-131) over all resident rows. */
int neb_svgf_temporal(neb_ctx* ctx, neb_stream stream);
/* SVGFDenoiser::SubmitATrousComputeWavelet (SVGFDenoiser.cpp:133-203): all levels, all rows.
 * Only valid when the context holds the full image (row_begin == 0, row_end == height). */
int neb_svgf_atrous(neb_ctx* ctx, neb_stream stream);
/* Row-range forms for multi-GPU row strips (no reference counterpart):
 * image rows [row0,row1) must be resident, and for the a-trous level so must every
 * (globally clamped) tap row.  `level` picks step = 1 << level and the source/destination
 * planes of that level's link in the ping-pong chain. */
int neb_svgf_temporal_rows(neb_ctx* ctx, uint32_t row0, uint32_t row1, neb_stream stream);
int neb_svgf_atrous_level_rows(neb_ctx* ctx, uint32_t level, uint32_t row0, uint32_t row1, neb_stream stream);
/* Which plane/slot level `level` reads and writes (for halo exchange between levels). */
int neb_svgf_atrous_level_planes(const neb_ctx* ctx, uint32_t level, int* src_plane, int* src_slot, int* dst_plane, int* dst_slot);

#ifdef __cplusplus
}
#endif
#endif /* NEBULAE_HIP_H */

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
 * thread-safe: the caller serialises, as the reference's single render thread does;
 * different contexts may be driven from different threads at the same time (tests/test_soak_gpu.py).
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
    NEB_PLANE_LDR = 9,         /* R8G8B8A8_UNORM      4 B/px, 1 slot (tonemapped back buffer, row f3) */
    NEB_PLANE_GEOMETRY = 10,   /* R32G32B32A32_FLOAT 16 B/px, 1 slot: {decoded shading normal.xyz, depth in [0,1]} of the current frame.
                                  No reference counterpart: the temporal pass decodes normal[cur] / depth[cur] once per frame and the
                                  a-trous levels read this instead of decoding them five times (read-only for callers). */
    NEB_PLANE_COUNT = 11
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

/* ---- profiler ranges: NEB_PIX_SCOPED_EVENT (src/nri/PIXRuntime.h:115-117) becomes a roctx range.  Every submit call below
 * brackets itself with the reference's event name ("SVGF: Temporal Accumulation", "SVGF: A-Trous compute i (step s)", ...);
 * these two let the host add its own enclosing ranges ("SVGF Denoising", DeferredRenderer.cpp:599).  No-ops without a
 * ROCm marker library; rocprofv3 --marker-trace records them. ---- */
int neb_marker_push(const char* name);
int neb_marker_pop(void);

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
 *   "atrous_variant": 1 = LDS row-lattice kernel (default; 4 rows per lane for steps <= 4, 2 for steps 8..32; wider steps take the
 *                     direct kernel), 0 = direct-load kernel;
 *   "gi_debug_hits":  1 = neb_gi_trace also records a neb_gi_hit per pixel (needs a scene);
 *   "gi_sort_rays":   mask, bit 0 = radix-sort the shadow rays by origin Morton code before tracing them (until the option is set:
 *                     on for dispatches of 1.5 M pixels and more, off for smaller ones, where the sort costs more than it saves),
 *                     bit 1 = sort the bounce rays by direction octant + origin (default off);
 *   "gi_defer_resolve": 1 = neb_gi_trace leaves the frame's indirect term in its records; neb_gi_resolve adds it;
 *                       2 = the same on TWO sets of records, used alternately: a second neb_gi_trace (the next frame's, on another stream)
 *                       may be issued and run before the first one's neb_gi_resolve has executed; neb_gi_resolve retires the dispatches in
 *                       the order they were traced.  The caller orders the streams: a trace must not start before the resolve that read
 *                       its set (two traces earlier) has finished, a resolve not before its own trace has;
 *   "gi_exact_shade":   1 = hit shading in the C arithmetic of the CPU oracle (IEEE division, sqrt, powf, sinf / cosf) instead of
 *                       the 1-ulp hardware forms an HLSL compiler emits (default).  The two differ by ~1e-7 relative, except where the
 *                       BRDF itself is ill-conditioned (mirror-like roughness: the GGX denominator cancels), where it can be percents;
 *   "gi_sun_table":     1 (default) / 0: answer the sun-visibility query of a hit from the per-triangle table where it is proven
 *                       (neb_gi_sun_table_stats) instead of tracing the shadow ray, and trace the rest -- from compacted ray lists or
 *                       through the sorted pass, whichever two timed dispatches after each table build say is faster
 *                       (neb_gi_shadow_tail_mode); 3 = always the lists, 2 = always the sorted / tiled pass; results are bit-identical
 *                       in all of them.  No table is built for a sun disk wider than 3.4 degrees (sunTanHalfAngle > 0.03);
 *   "gi_sun_hold":      how many consecutive dispatches a NEW sun must be seen on before a table is built for it (a build costs five frames' time):
 *                       0 (default) = two -- or 32 when the table it replaces served fewer than 32 dispatches (a sun that moves in steps of a few
 *                       frames: a build per step would cost more than no table at all); N >= 2 pins the count.  A sun that changes every
 *                       dispatch is never built for; meanwhile every shadow ray is traced.  Results never depend on it;
 *   "gi_sun_hints":     4 (default), 2 or 0: how many of a triangle's occluder hints the shade pass tries (with the traverser's own
 *                       triangle test) before it leaves the shadow ray to the list pass; results are bit-identical in all three;
 *   "gi_max_bvh_depth": 1..21, the deepest BVH4 neb_gi_build_bvh accepts (default 21 = traversal stack / 3);
 *   "svgf_fuse":        0 (default) / 1 (opt-in), see neb_svgf_atrous;   "svgf_profile": 0 (default) / 1 / 2, see neb_svgf_level_times. */
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
/* SVGFDenoiser::SubmitTemporalAccumulation (SVGFDenoiser.cpp:66-131) over all resident rows.
 * The G-buffer of the frame (normal[cur], depth[cur]) must be complete before this call, as the G-buffer pass precedes SVGF in the
 * reference: the a-trous levels of the frame read the decoded copy this pass (or, for rows it did not cover, a lazy decode) made. */
int neb_svgf_temporal(neb_ctx* ctx, neb_stream stream);
/* SVGFDenoiser::SubmitATrousComputeWavelet (SVGFDenoiser.cpp:133-203): all levels, all rows.
 * Only valid when the context holds the full image (row_begin == 0, row_end == height).
 *
 * By default both calls are stream-ordered at the call, as the reference records into its command list at the call
 * (SVGFDenoiser.cpp:116,185): after neb_svgf_temporal returns, the pass is enqueued on `stream`.
 *
 * Option "svgf_fuse" = 1 (OPT-IN; the binding of INTEGRATION.md sets it in SVGFDenoiser::Init): the two calls, made in this order
 * on the same stream with no other neb_* call between them -- what DeferredRenderer::SubmitCommandsSVGFDenoising does
 * (src/DeferredRenderer.cpp:610-611) -- run as ONE chain on a whole-frame context whose width and height are multiples of 8:
 * neb_svgf_temporal only NOTES the request (nothing is enqueued yet), and neb_svgf_atrous runs the temporal pass inside the
 * staging phase of level 0 (the accumulated radiance is never written to memory: quirk 5 makes the FILTERED image the next
 * frame's history), carries the luminance between the levels and writes radiance[cur], moments[cur] and variance exactly as the
 * separate passes do -- the same bits.  Any other neb_* call in between (a plane pointer, an upload or download, a row-range
 * form, neb_end_frame, neb_stream_synchronize ...) first submits the noted pass as its own kernel.
 * WHO CAN SEE THE DIFFERENCE, and so must not opt in (or must call neb_end_frame / neb_stream_synchronize first): a host that kept
 * plane pointers from an earlier neb_get_plane and, between the two calls, orders work on them WITHOUT going through the library
 * -- a raw hipStreamSynchronize / hipEventRecord on `stream`, or a kernel of its own reading moments, variance or the accumulated
 * radiance: until neb_svgf_atrous (or any other neb_* call) it finds them un-accumulated, and a launch error of the noted pass is
 * reported by that later call.  After the chain, radiance[hist] and the scratch plane hold intermediate levels (as radiance[hist]
 * does in the reference), here with the luminance in .w.  neb_svgf_denoise below is the same chain as one explicit call. */
int neb_svgf_atrous(neb_ctx* ctx, neb_stream stream);
/* DeferredRenderer::SubmitCommandsSVGFDenoising's pair of calls (src/DeferredRenderer.cpp:610-611: SubmitTemporalAccumulation, then
 * SubmitATrousComputeWavelet) as ONE entry point, whatever "svgf_fuse" says: the fused chain on a whole-frame context whose width
 * and height are multiples of 8, the two separate passes otherwise; the same bits as neb_svgf_temporal + neb_svgf_atrous, everything
 * enqueued on `stream` when the call returns.  Only valid when the context holds the full image. */
int neb_svgf_denoise(neb_ctx* ctx, neb_stream stream);
/* With option "svgf_profile" = 1, neb_svgf_atrous brackets each of its kernels with events on `stream`; this call waits for the
 * last chain submitted and returns the kernels' durations in microseconds (entry 0 = level 0, fused with the temporal pass when
 * the chain ran fused), *n_out = how many.  With "svgf_profile" = 2 only three events are recorded and two durations returned:
 * the first kernel, and all the others together (an event between two launches costs about 2 us of its own). */
int neb_svgf_level_times(neb_ctx* ctx, float* out_us, uint32_t capacity, uint32_t* n_out);
/* Row-range forms for multi-GPU row strips (no reference counterpart):
 * image rows [row0,row1) must be resident, and for the a-trous level so must every
 * (globally clamped) tap row.  `level` picks step = 1 << level and the source/destination
 * planes of that level's link in the ping-pong chain. */
int neb_svgf_temporal_rows(neb_ctx* ctx, uint32_t row0, uint32_t row1, neb_stream stream);
int neb_svgf_atrous_level_rows(neb_ctx* ctx, uint32_t level, uint32_t row0, uint32_t row1, neb_stream stream);
/* Which plane/slot level `level` reads and writes (for halo exchange between levels). */
int neb_svgf_atrous_level_planes(const neb_ctx* ctx, uint32_t level, int* src_plane, int* src_slot, int* dst_plane, int* dst_slot);

/* ======================= Multi-GPU row strips: halo exchange over RCCL (no reference counterpart) ================
 * One context per GPU holds image rows [row_begin, row_end) = the strip it owns plus halo rows (neb_create_info).  Every stage is
 * per-pixel except the a-trous wavelet, so the only data that crosses GPUs are halo rows of radiance (and variance), swapped with
 * the up / down neighbour by grouped ncclSend / ncclRecv straight out of / into the planes (rows are contiguous: no packing), on the
 * caller's stream.  RCCL is resolved at run time (dlopen of librccl): hosts link nothing extra, and without RCCL these calls return
 * NEB_ERR_STATE while everything single-GPU keeps working.  nebulae_amd/strips.py holds the partition arithmetic (which rows, when). */
typedef struct neb_halo_plane {
    int32_t plane; /* neb_plane */
    int32_t slot;  /* 0, 1, NEB_SLOT_CURRENT or NEB_SLOT_HISTORY */
} neb_halo_plane;
typedef struct neb_halo_swap {
    int32_t peer;                  /* rank in the communicator */
    uint32_t send_row0, send_row1; /* image rows sent to the peer (owned by this strip) */
    uint32_t recv_row0, recv_row1; /* image rows received from it (halo rows of this strip) */
} neb_halo_swap;
/* ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy through the library: one rank makes the 128-byte id, the host carries it to
 * the others however it likes (MPI, a socket, torch.distributed), every rank creates its communicator with it. */
int neb_strips_unique_id(void* id128);
int neb_strips_comm_create(int device, int n_ranks, int rank, const void* id128, void** out_comm);
int neb_strips_comm_destroy(void* comm);
/* ncclGroupStart / ncclGroupEnd.  A host with ONE process (or thread) per GPU never needs them.  A host that drives several GPUs
 * from one thread must bracket the neb_strips_comm_create calls of all its ranks in one group (ncclCommInitRank blocks until every
 * rank has joined: RCCL's rule for one thread with several devices; *out_comm is then valid once the group has ended), and likewise
 * the neb_strips_exchange calls of all its contexts for one exchange (the group each call opens nests inside the caller's). */
int neb_strips_group_begin(void);
int neb_strips_group_end(void);
/* For every listed plane and every swap: send rows [send_row0, send_row1), receive rows [recv_row0, recv_row1), all in ONE RCCL group,
 * enqueued on `stream` (ordered after the kernels that produced the rows, before the ones that read the halo).  Every rank of the
 * communicator must make the matching call.  comm = the ncclComm_t from neb_strips_comm_create (or the host's own). */
int neb_strips_exchange(neb_ctx* ctx, void* comm, const neb_halo_plane* planes, uint32_t n_planes, const neb_halo_swap* swaps, uint32_t n_swaps,
                        neb_stream stream);
const char* neb_strips_last_error(void); /* message of the last failed neb_strips_* call that had no context */

/* ONE call per strip frame (round 5): everything a strip context does between "radiance[cur] holds the direct term" and "radiance[cur]
 * holds this strip's rows of the denoised frame" -- the GI dispatch on its rows, the temporal pass, the halo exchange(s) and the a-trous
 * levels on the row ranges the exchange scheme prescribes -- enqueued by the library in the order and on the streams nebulae_amd/strips.py
 * used to spell out call by call (105 us of host time per 135-row strip frame from Python against 150 us of device time).  The partition
 * arithmetic is the library's (the same as strips.StripPartition: strips of H / n_strips rows; resident rows = strip +- halo, which the
 * context must have been created with: row_begin / row_end of neb_create_info).
 *   scheme NEB_STRIPS_ONCE:      one exchange per frame -- the h = sum_l 2 * 2^l boundary rows of the temporally accumulated radiance and of the
 *                                variance plane, beside the interior rows of level 0 (side stream); level l filters strip +- sum_{m>l} 2 * 2^m rows;
 *   scheme NEB_STRIPS_PER_LEVEL: an exchange of 2 * 2^l rows of the level's source plane in front of every level;
 *   scheme NEB_STRIPS_OVERLAP:   SURVEY 8e's: GI, temporal pass and levels 0 .. L-2 also on the band beyond the strip, one exchange in front
 *                                of the widest level, one of the final image's band rows (next frame's history) after it.
 * Transports: `comm` = an RCCL communicator (neb_strips_comm_create; one process or thread per GPU: every rank makes the matching call),
 * or `peers` = the neighbouring strips' contexts of a host that drives all strips from ONE thread (any devices: rows are pushed with
 * hipMemcpyPeerAsync; scheme ONCE only).  Such a host calls neb_strip_frame_begin for every strip, then neb_strip_frame_finish for every
 * strip, frame after frame; a host with a communicator calls neb_strip_frame (= begin + finish).
 * flags: NEB_STRIP_RESET_HISTORY = neb_svgf_reset_history first (the frame policy's reset, src/DeferredRenderer.cpp:601-609).
 * constants == NULL: no GI dispatch (radiance[cur] is complete as it is). */
enum { NEB_STRIPS_ONCE = 0, NEB_STRIPS_PER_LEVEL = 1, NEB_STRIPS_OVERLAP = 2 };
enum { NEB_STRIP_RESET_HISTORY = 1 };
typedef struct neb_strip_plan {
    uint32_t n_strips, strip; /* this context holds strip `strip` of `n_strips` */
    uint32_t scheme;          /* NEB_STRIPS_* */
    uint32_t flags;           /* NEB_STRIP_* */
} neb_strip_plan;
typedef struct neb_strip_peers {
    neb_ctx* up;   /* the context of strip - 1 (NULL for the first strip) */
    neb_ctx* down; /* the context of strip + 1 (NULL for the last) */
} neb_strip_peers;
struct neb_gi_constants; /* (defined below) */
int neb_strip_frame(neb_ctx* ctx, const struct neb_gi_constants* constants, void* comm, const neb_strip_plan* plan, neb_stream stream);
int neb_strip_frame_begin(neb_ctx* ctx, const struct neb_gi_constants* constants, const neb_strip_plan* plan, const neb_strip_peers* peers, neb_stream stream);
int neb_strip_frame_finish(neb_ctx* ctx, void* comm, const neb_strip_plan* plan, const neb_strip_peers* peers, neb_stream stream);
/* The row ranges the plan implies for this strip (what strips.StripPartition computes; for hosts and tests):
 * out = {owned row0, row1, resident row0, row1, GI / temporal row0, row1, halo rows, band rows}; levels = the context's a-trous levels. */
int neb_strip_rows(const neb_ctx* ctx, const neb_strip_plan* plan, uint32_t out[8]);

/* ======================= GI: one-bounce indirect diffuse =========================================
 * Replaces DeferredRenderer::SubmitCommandsGIPathtrace (src/DeferredRenderer.cpp:396-591) driving
 * assets/shaders/pathtracer.hlsl with the NRC calls stubbed (rtxgi/Nrc.hlsli:579-621), and the DXR
 * driver BVH (src/nri/raytracing/RTAccelerationStructureBuilder.cpp:14-130) with a BVH built on the device (binned SAH, 4-wide). */

/* One submesh: StaticMeshGeometryData (src/nri/GIProcessedScene.h:17-31; shader mirror
 * pathtracer.hlsl:73-87).  The bindless (bufferIndex, offset) pairs become host pointers to the
 * first element; strides are in bytes.  Attribute order (src/nri/StaticMesh.h:14-21): position float3,
 * normal float3, texcoord float2, tangent float4.  neb_gi_set_scene copies everything. */
typedef struct neb_geometry_desc {
    float surfaceToWorld[16]; /* row-major, row-vector convention: world = (p,1) * M (SimpleMath Mat4) */
    int32_t materialIndex;    /* PathtracerInvalidBindlessIndex (-1) = none */
    uint32_t indexStride;     /* 2 or 4 */
    uint32_t numIndices;
    uint32_t numVertices;
    const void* indices;
    const void* attributes[4]; /* NULL = missing: hits on this submesh terminate the path (pathtracer.hlsl:313-318) */
    uint32_t attributeStrides[4];
    uint32_t _pad;
} neb_geometry_desc;

/* StaticMeshMaterialData (src/nri/GIProcessedScene.h:33-39; pathtracer.hlsl:99-105) */
typedef struct neb_material_desc {
    int32_t textureIndices[3]; /* albedo, normal, roughnessMetalness; -1 = use the factors */
    float albedo[4];
    float roughnessMetalness[2];
    uint32_t _pad;
} neb_material_desc;

/* R8G8B8A8_UNORM, one mip, no sRGB decode (src/core/GLTFSceneImporter.cpp:156); sampled linear/wrap at mip 0. */
typedef struct neb_texture_desc {
    const void* rgba8;
    uint32_t width, height;
} neb_texture_desc;

/* GlobalConstants (src/DeferredRenderer.h:219-238, pathtracer.hlsl:11-30) minus the NRC-only members. */
typedef struct neb_gi_constants {
    uint32_t frameIndex;
    uint32_t samplesPerPixel;
    uint32_t maxPathVertices; /* nrcMaxPathVertices, <= 8 (MaxPathtracingRecursionDepth, DeferredRenderer.h:118); 2 = one bounce, the
                                 north-star configuration; > 2 follows the shader's bounce loop with the NRC stubs (row f4) */
    float cameraWorldPos[3];
    float skyColor[3];
    float sunLightDirection[3];
    float sunLightRadiance[3];
    float sunTanHalfAngle;    /* tan(radians(diameter / 2)), DeferredRenderer.cpp:418 */
    float throughputThreshold;
} neb_gi_constants;

/* Per-pixel record of the last sample's bounce ray (the reference's debug UAVs, pathtracer.hlsl:41-42). */
typedef struct neb_gi_hit {
    float t;            /* < 0: miss */
    uint32_t geometry;  /* GeometryIndex() */
    uint32_t primitive; /* PrimitiveIndex() */
    uint32_t flags;     /* bit 0: sun shadow ray unoccluded; bits 8..31: traversal iterations of the bounce ray (diagnostics) */
} neb_gi_hit;

typedef struct neb_camera {
    float eye[3], target[3], up[3]; /* Mat4::CreateLookAt (RH), src/core/InspectCamera.h:44-48 */
    float vfov_deg, znear, zfar;    /* CreatePerspectiveFieldOfView(60 deg, aspect, 0.1, 100), DeferredRenderer.cpp:147-148 */
} neb_camera;

/* DeferredRenderer::InitPathtracerScene -> GIProcessedScene::InitScene (src/nri/GIProcessedScene.cpp:16-137):
 * uploads geometry/material tables and textures and bakes world-space triangles. */
int neb_gi_set_scene(neb_ctx* ctx, const neb_geometry_desc* geoms, uint32_t n_geoms, const neb_material_desc* mats,
                     uint32_t n_mats, const neb_texture_desc* texs, uint32_t n_texs);
/* DeferredRenderer::InitRTAccelerationStructures (src/DeferredRenderer.cpp:978-1030): builds the acceleration structure --
 * on the device, as the reference's driver does (RTAccelerationStructureBuilder.cpp:73-130): Morton sort, binned-SAH splits
 * level by level, collapse to 4-wide nodes; the host reads back one counter per pass, never a node.
 * One-time setup: enqueues on `stream` and SYNCHRONISES it before returning.  On failure the scene keeps its previous
 * state (unbuilt, or the previous valid tree); calling it again rebuilds. */
int neb_gi_build_bvh(neb_ctx* ctx, neb_stream stream);
int neb_gi_scene_info(const neb_ctx* ctx, uint32_t* n_triangles, uint32_t* n_nodes);
/* Device bytes of the scene: {texture footprint tables + material bundles, triangles + shading records, BVH nodes: the
 * builder's 128-byte nodes + the 64-byte quantised nodes the rays walk}. */
int neb_gi_scene_bytes(const neb_ctx* ctx, uint64_t out[3]);
/* Inner-node levels of the BVH4 of the last successful build.  neb_gi_build_bvh returns NEB_ERR_OUT_OF_RANGE (and keeps the
 * previous tree, if any) when the depth exceeds what the traversal stack covers: 21, or "gi_max_bvh_depth". */
int neb_gi_bvh_depth(const neb_ctx* ctx, uint32_t* depth);
/* Passes (levels of the binary SAH tree) the last successful build took: diagnostics. */
int neb_gi_build_passes(const neb_ctx* ctx, uint32_t* passes);
/* Wall time in milliseconds of the last successful neb_gi_build_bvh (the build ends with a stream synchronisation). */
int neb_gi_build_ms(const neb_ctx* ctx, float* ms);
/* DeferredRenderer::SubmitCommandsGIPathtrace: radiance[cur].rgb += mean over spp of the path radiance
 * (stands in for NRC Resolve, DeferredRenderer.cpp:586).  Reads the ALBEDO / ROUGH_METAL / WORLDPOS planes and
 * normal[cur].  _rows: image rows [row0,row1) only (multi-GPU strips). */
int neb_gi_trace(neb_ctx* ctx, const neb_gi_constants* constants, neb_stream stream);
int neb_gi_trace_rows(neb_ctx* ctx, const neb_gi_constants* constants, uint32_t row0, uint32_t row1, neb_stream stream);
/* The same dispatch in two calls, for a host that keeps two frames in flight (the reference's swapchain keeps three, src/nri/Swapchain.h:15):
 * _begin enqueues ray generation + the closest-hit walk -- they read the G-buffer and write only GI records --, _finish the shading and shadow
 * passes of the dispatch begun longest ago, which add into the radiance[cur] of the frame current AT THE _finish CALL.  Two sets of GI records:
 * the _begin of frame f+1 may be enqueued (on another stream) while the _finish of frame f is still executing -- the short, latency-bound
 * shadow pass of frame f then runs beside the closest-hit walk of frame f+1 instead of alone on the chip.  `after_shade_event`: NULL, or a
 * hipEvent_t that _finish records between its two passes (what the next _begin's stream may wait for).  The caller orders the streams: a
 * _finish must not start before its own _begin has completed, a _begin not before the _finish that read its set (two dispatches earlier).
 * One sample and one bounce per pixel only (samplesPerPixel == 1, maxPathVertices <= 2); not together with "gi_defer_resolve". */
int neb_gi_trace_begin(neb_ctx* ctx, const neb_gi_constants* constants, uint32_t row0, uint32_t row1, neb_stream stream);
int neb_gi_trace_finish(neb_ctx* ctx, neb_stream stream, void* after_shade_event);
/* The reference's separate resolve step (nrc Resolve(commandList, GetRadianceOutput()), DeferredRenderer.cpp:586):
 * radiance[cur].rgb += the indirect term of the last neb_gi_trace that ran with "gi_defer_resolve" = 1.  Lets the GI
 * stages of frame f+1 overlap the SVGF passes of frame f on another stream (they share no plane until this call). */
int neb_gi_resolve(neb_ctx* ctx, neb_stream stream);
/* Rays traced (bounce + shadow) by all GI dispatches since the last reset.  Synchronises `stream`. */
int neb_gi_ray_count(neb_ctx* ctx, uint64_t* rays, int reset, neb_stream stream);
/* Diagnostics: counters as of the last neb_gi_ray_count call: {rays, bounce node visits, bounce triangle tests,
 * shadow node visits, shadow triangle tests}; the shadow entries are only collected while "gi_debug_hits" is 1. */
int neb_gi_traversal_stats(neb_ctx* ctx, uint64_t out[5]);
/* The sun-visibility table (no reference counterpart: the reference's shadow rays go to the driver's TraceRay,
 * assets/shaders/pathtracer.hlsl:567-569).  Every shadow ray of the path points at the sun disk; for each (triangle, side) the
 * library decides once per sun position -- exactly, from the scene's geometry -- whether ANY such ray leaving it can meet ANY
 * triangle; where none can, the hit's sun-visibility query is answered without a traversal, with the traversal's own answer.
 * out = {triangle sides proven lit on the +normal side, on the -normal side (last build), shadow rays answered by the table as of
 * the last neb_gi_ray_count call (they are part of its count: a ray = a visibility query), table builds so far}. */
int neb_gi_sun_table_stats(neb_ctx* ctx, uint64_t out[4], neb_stream stream);
/* Device time of the last build of the table (its two launches, between events on the stream it was enqueued on), in milliseconds.  Waits for that build.
 * NEB_ERR_STATE when no table has been built. */
int neb_gi_sun_table_build_ms(neb_ctx* ctx, float* ms);
/* Which pass takes the shadow rays the sun table leaves ("gi_sun_table" = 1): *mode = 0 the compacted lists, 1 the sorted pass, -1 not decided yet
 * for the current table (the first two dispatches after a build are the timed ones); us (may be NULL) = the two times of the last measurement
 * {lists, sorted pass}, shade + shadow launches between events, in microseconds.  Does not synchronise. */
int neb_gi_shadow_tail_mode(neb_ctx* ctx, int* mode, float us[2]);
/* Diagnostics (collected while "gi_debug_hits" is 1, as of the last neb_gi_ray_count call): where the closest-hit pass's waves spend
 * their loop iterations -- {waves, loop iterations, iterations that ran a node phase, lanes live in those, iterations that ran a leaf
 * phase, lanes live in those}; a wave executes every phase some lane needs, so lanes / (64 x iterations) is the lane utilisation. */
int neb_gi_wave_stats(neb_ctx* ctx, uint64_t out[6]);
/* Diagnostics (same collection): where in the breadth-first node array the closest-hit pass's node visits fall -- {visits of nodes with
 * an index below 64, below 256, below 1024, below 4096, node phases that ended with more than 12 entries on the ray's stack}; the
 * total is neb_gi_traversal_stats' bounce node visits.  (What a copy of the top of the tree in LDS would serve.) */
int neb_gi_node_index_stats(neb_ctx* ctx, uint64_t out[5]);
/* Tuning: the order in which the closest-hit pass takes its 8x8 tiles -- workgroup b walks tile order[b] (a permutation of 0 .. n - 1, n = the tiles of the
 * dispatches it applies to; any other dispatch keeps the default, XCD-aware order); NULL / 0 restores the default.  Results do not depend on it.  Synchronises. */
int neb_gi_debug_set_tile_order(neb_ctx* ctx, const uint32_t* order, uint32_t n);
/* Diagnostics of the sun-table build (collected when the environment variable NEB_SUN_WALK_STATS is set at build time of the table): for the lit pass
 * (out[0 .. 5]) and the hint pass (out[6 .. 11]) {node visits of all walks, the longest walk, candidate triangles tested, sum of the waves' run times and
 * the longest wave in 10-ns ticks, waves}.  Synchronises. */
int neb_gi_debug_sun_walk_stats(neb_ctx* ctx, uint64_t out[12]);
/* Debug: when option "gi_debug_hits" is 1, every trace also records neb_gi_hit per resident pixel. */
int neb_gi_download_hits(neb_ctx* ctx, neb_gi_hit* host, neb_stream stream);
/* Diagnostics / tests: runs the library's ray-reordering sort (raysort.hip: stable LSD radix sort on key bits
 * [0, bits), bits <= 16) on n host (key, value) pairs and returns the values in sorted order.  Synchronises. */
int neb_debug_sort_pairs(neb_ctx* ctx, const uint32_t* keys, const uint32_t* values, uint32_t n, int bits, uint32_t* sorted_values);
/* "next" row f1: DeferredRenderer::SubmitCommandsPBRLighting (src/DeferredRenderer.cpp:326-394) driving
 * assets/shaders/deferred_pbr.hlsl:39-115: Cook-Torrance sun light x one any-hit shadow ray per pixel; OVERWRITES
 * radiance[cur] (alpha = 1).  Uses cameraWorldPos, sunLight*, sunTanHalfAngle and frameIndex of the constants. */
int neb_pbr_direct(neb_ctx* ctx, const neb_gi_constants* constants, neb_stream stream);
/* "next" row f3: DeferredRenderer::SubmitCommandsHDRTonemapping (src/DeferredRenderer.cpp:616-660),
 * assets/shaders/tonemapping.hlsl:3-53: ACES fit of radiance[cur] into the R8G8B8A8_UNORM LDR plane (alpha = luma). */
int neb_tonemap(neb_ctx* ctx, neb_stream stream);
/* "next" row f2: primary-visibility G-buffer producer with the encodings of
 * assets/shaders/deferred_gbuffers.hlsl:71-103 (writes ALBEDO, ROUGH_METAL, WORLDPOS, normal[cur], depth[cur]). */
int neb_gbuffer_raycast(neb_ctx* ctx, const neb_camera* camera, neb_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* NEBULAE_HIP_H */

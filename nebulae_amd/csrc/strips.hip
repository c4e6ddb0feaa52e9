// strips.hip -- the halo exchange between neighbouring row strips as a C entry point: grouped ncclSend / ncclRecv
// over RCCL (xGMI: one direct link per neighbour), enqueued on the caller's HIP stream.
//
// No reference counterpart (the reference is single-GPU).  A C++ host that holds one context per
// GPU -- the drop-in INTEGRATION.md describes -- exchanges halo rows with these calls and needs neither PyTorch nor a
// link-time dependency on RCCL: librccl is resolved at run time (dlopen), so the library still loads, and the
// single-GPU path still runs, on a box without it; the strip calls then return NEB_ERR_STATE with a clear message.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "neb_internal.h"

namespace {

// the slice of rccl.h this file needs (ABI of RCCL 2.x / NCCL 2.x)
struct RcclUniqueId {
    char internal[128];
};
using RcclComm = void*;
constexpr int kRcclSuccess = 0;
constexpr int kRcclUint8 = 1; // ncclUint8

struct RcclApi {
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    void* h = nullptr;
    std::string tried;
    // NEB_RCCL_LIBRARY names the library to open instead (a site's own RCCL build; the tests use it to reach the not-found path)
    const char* override_name = getenv("NEB_RCCL_LIBRARY");
    const char* defaults[] = {"librccl.so.1", "librccl.so"};
    for (const char* name : defaults) {
        if (override_name)
            name = override_name;
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h)
            break;
        const char* e = dlerror(); // (read once: the call clears the message)
        tried += std::string(tried.empty() ? "" : "; ") + (e ? e : name);
        if (override_name)
            break;
    }
    if (!h) {
        g_rccl.why = "librccl not found (" + tried + ")";
        return;
    }
    auto sym = [&](const char* n) {
        void* p = dlsym(h, n);
        if (!p)
            g_rccl.why = std::string("librccl lacks ") + n;
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.ok = g_rccl.why.empty();
}

const RcclApi& rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl;
}

thread_local std::string g_strip_error;

int strip_fail(neb_ctx* ctx, int code, const std::string& what)
{
    if (ctx)
        ctx->last_error = what;
    else
        g_strip_error = what;
    return code;
}

int rccl_fail(neb_ctx* ctx, const char* call, int rc)
{
    const RcclApi& r = rccl();
    return strip_fail(ctx, NEB_ERR_HIP, std::string(call) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

} // namespace

extern "C" {

const char* neb_strips_last_error(void) { return g_strip_error.c_str(); }

int neb_strips_unique_id(void* id128)
{
    if (!id128)
        return strip_fail(nullptr, NEB_ERR_INVALID_ARG, "neb_strips_unique_id: null buffer");
    const RcclApi& r = rccl();
    if (!r.ok)
        return strip_fail(nullptr, NEB_ERR_STATE, "neb_strips_unique_id: " + r.why);
    RcclUniqueId id;
    const int rc = r.GetUniqueId(&id);
    if (rc != kRcclSuccess)
        return rccl_fail(nullptr, "ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return NEB_OK;
}

int neb_strips_comm_create(int device, int n_ranks, int rank, const void* id128, void** out_comm)
{
    if (!id128 || !out_comm || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return strip_fail(nullptr, NEB_ERR_INVALID_ARG, "neb_strips_comm_create: bad argument");
    *out_comm = nullptr;
    const RcclApi& r = rccl();
    if (!r.ok)
        return strip_fail(nullptr, NEB_ERR_STATE, "neb_strips_comm_create: " + r.why);
    neb::DeviceGuard guard(device);
    if (guard.err != hipSuccess)
        return strip_fail(nullptr, NEB_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    RcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    // RCCL writes the handle straight into the caller's variable: inside neb_strips_group_begin / _end it may do so only when the group closes
    const int rc = r.CommInitRank(reinterpret_cast<RcclComm*>(out_comm), n_ranks, id, rank);
    if (rc != kRcclSuccess)
        return rccl_fail(nullptr, "ncclCommInitRank", rc);
    return NEB_OK;
}

int neb_strips_group_begin(void)
{
    const RcclApi& r = rccl();
    if (!r.ok)
        return strip_fail(nullptr, NEB_ERR_STATE, "neb_strips_group_begin: " + r.why);
    const int rc = r.GroupStart();
    return rc == kRcclSuccess ? NEB_OK : rccl_fail(nullptr, "ncclGroupStart", rc);
}

int neb_strips_group_end(void)
{
    const RcclApi& r = rccl();
    if (!r.ok)
        return strip_fail(nullptr, NEB_ERR_STATE, "neb_strips_group_end: " + r.why);
    const int rc = r.GroupEnd();
    return rc == kRcclSuccess ? NEB_OK : rccl_fail(nullptr, "ncclGroupEnd", rc);
}

int neb_strips_comm_destroy(void* comm)
{
    if (!comm)
        return NEB_OK;
    const RcclApi& r = rccl();
    if (!r.ok)
        return strip_fail(nullptr, NEB_ERR_STATE, "neb_strips_comm_destroy: " + r.why);
    const int rc = r.CommDestroy(comm);
    return rc == kRcclSuccess ? NEB_OK : rccl_fail(nullptr, "ncclCommDestroy", rc);
}

int neb_strips_exchange(neb_ctx* ctx, void* comm, const neb_halo_plane* planes, uint32_t n_planes, const neb_halo_swap* swaps, uint32_t n_swaps,
                        neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (!comm || (n_planes && !planes) || (n_swaps && !swaps))
        return strip_fail(ctx, NEB_ERR_INVALID_ARG, "neb_strips_exchange: null argument");
    const RcclApi& r = rccl();
    if (!r.ok)
        return strip_fail(ctx, NEB_ERR_STATE, "neb_strips_exchange: " + r.why);
    // validate everything before the first RCCL call: a rank that bails out in the middle of a group leaves its peers hanging
    for (uint32_t p = 0; p < n_planes; ++p) {
        const int pl = planes[p].plane;
        if (pl < 0 || pl >= NEB_PLANE_COUNT)
            return strip_fail(ctx, NEB_ERR_INVALID_ARG, "neb_strips_exchange: bad plane");
        int slot = planes[p].slot;
        if (neb::kPlaneInfo[pl].slots == 1)
            slot = (slot == 0 || slot == NEB_SLOT_CURRENT) ? 0 : -1;
        else if (slot == NEB_SLOT_CURRENT)
            slot = ctx->cur;
        else if (slot == NEB_SLOT_HISTORY)
            slot = ctx->hist;
        if (slot != 0 && slot != 1)
            return strip_fail(ctx, NEB_ERR_INVALID_ARG, "neb_strips_exchange: bad slot");
    }
    for (uint32_t k = 0; k < n_swaps; ++k) {
        const neb_halo_swap& s = swaps[k];
        if (s.send_row0 > s.send_row1 || s.recv_row0 > s.recv_row1 || s.send_row0 < ctx->row_begin || s.send_row1 > ctx->row_end ||
            s.recv_row0 < ctx->row_begin || s.recv_row1 > ctx->row_end || s.peer < 0)
            return strip_fail(ctx, NEB_ERR_OUT_OF_RANGE, "neb_strips_exchange: rows not resident in this context");
    }
    if (int frc = neb::svgf_flush_pending(ctx))
        return frc;
    neb::DeviceGuard guard(ctx->device);
    if (guard.err != hipSuccess)
        return strip_fail(ctx, NEB_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    neb::ScopedRange range("Strips: halo exchange (RCCL)");
    for (uint32_t p = 0; p < n_planes; ++p) // rows of normal[cur] / depth[cur] are about to change: their decoded copy is stale
        if (planes[p].plane == NEB_PLANE_NORMAL || planes[p].plane == NEB_PLANE_DEPTH)
            ctx->geom_lo = ctx->geom_hi = 0;
    int rc = r.GroupStart(); // (nests inside a caller's neb_strips_group_begin / _end: RCCL merges nested groups)
    if (rc != kRcclSuccess)
        return rccl_fail(ctx, "ncclGroupStart", rc);
    int first_bad = kRcclSuccess;
    const char* bad_call = nullptr;
    for (uint32_t p = 0; p < n_planes; ++p) {
        const int pl = planes[p].plane;
        int slot = planes[p].slot;
        if (neb::kPlaneInfo[pl].slots == 1)
            slot = 0;
        else if (slot == NEB_SLOT_CURRENT)
            slot = ctx->cur;
        else if (slot == NEB_SLOT_HISTORY)
            slot = ctx->hist;
        const size_t pitch = (size_t)ctx->W * neb::kPlaneInfo[pl].bytes_per_px;
        char* base = (char*)ctx->planes[pl][slot];
        for (uint32_t k = 0; k < n_swaps; ++k) { // rows are contiguous: no packing, straight out of / into the plane
            const neb_halo_swap& s = swaps[k];
            if (s.send_row1 > s.send_row0) {
                rc = r.Send(base + (size_t)(s.send_row0 - ctx->row_begin) * pitch, (size_t)(s.send_row1 - s.send_row0) * pitch, kRcclUint8, s.peer, comm,
                            (hipStream_t)stream);
                if (rc != kRcclSuccess && first_bad == kRcclSuccess)
                    first_bad = rc, bad_call = "ncclSend";
            }
            if (s.recv_row1 > s.recv_row0) {
                rc = r.Recv(base + (size_t)(s.recv_row0 - ctx->row_begin) * pitch, (size_t)(s.recv_row1 - s.recv_row0) * pitch, kRcclUint8, s.peer, comm,
                            (hipStream_t)stream);
                if (rc != kRcclSuccess && first_bad == kRcclSuccess)
                    first_bad = rc, bad_call = "ncclRecv";
            }
        }
    }
    rc = r.GroupEnd(); // always closed, whatever happened inside
    if (first_bad != kRcclSuccess)
        return rccl_fail(ctx, bad_call, first_bad);
    if (rc != kRcclSuccess)
        return rccl_fail(ctx, "ncclGroupEnd", rc);
    return NEB_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// One call per strip frame (round 5).  The partition arithmetic of nebulae_amd/strips.py (StripPartition) in C, and the enqueue
// order of StripRenderer.submit_commands_svgf_denoising: the Python side keeps both for the CPU (gloo) tests and as the mirror.
// ------------------------------------------------------------------------------------------------
namespace {

struct StripRows {
    uint32_t N, r, L, scheme;
    uint32_t H, h;             // image rows, rows per strip
    uint32_t own0, own1;       // owned rows
    uint32_t halo, band;       // resident rows beyond the strip; rows beyond it that GI / temporal / levels 0 .. L-2 recompute ("overlap")
    uint32_t res0, res1, gi0, gi1;
    uint32_t lo(uint32_t a, uint32_t n) const { return a > n ? a - n : 0u; }
    uint32_t hi(uint32_t b, uint32_t n) const { return b + n < H ? b + n : H; }
    uint32_t extension(uint32_t level) const // rows beyond the strip `level` must also filter: what the later levels still reach into
    {
        if (N == 1 || scheme == NEB_STRIPS_PER_LEVEL)
            return 0;
        uint32_t e = 0;
        const uint32_t last = scheme == NEB_STRIPS_OVERLAP ? (L > 0 ? L - 1 : 0) : L; // overlap: levels l+1 .. L-2; once: l+1 .. L-1
        for (uint32_t m = level + 1; m < last; ++m)
            e += 2u << m;
        return e;
    }
};

int strip_rows(const neb_ctx* ctx, const neb_strip_plan* plan, StripRows& R, std::string& why)
{
    if (!plan || plan->n_strips < 1 || plan->strip >= plan->n_strips || plan->scheme > NEB_STRIPS_OVERLAP) {
        why = "bad plan (n_strips >= 1, strip < n_strips, scheme NEB_STRIPS_ONCE / PER_LEVEL / OVERLAP)";
        return NEB_ERR_INVALID_ARG;
    }
    R.N = plan->n_strips, R.r = plan->strip, R.L = ctx->levels, R.scheme = plan->scheme, R.H = ctx->H;
    if (R.H % R.N) {
        why = "the image height is not divisible by the number of strips";
        return NEB_ERR_INVALID_ARG;
    }
    R.h = R.H / R.N;
    R.own0 = R.r * R.h, R.own1 = R.own0 + R.h;
    R.halo = R.band = 0;
    if (R.N > 1 && R.L > 0) {
        R.halo = R.scheme == NEB_STRIPS_ONCE ? 2u * ((1u << R.L) - 1u) : 2u * (1u << (R.L - 1));
        if (R.scheme == NEB_STRIPS_OVERLAP)
            R.band = 2u * ((1u << (R.L - 1)) - 1u); // sum_{l <= L-2} 2 * 2^l
        if (R.h < R.halo) {
            why = "strips are shorter than the a-trous reach of this scheme";
            return NEB_ERR_OUT_OF_RANGE;
        }
    }
    R.res0 = R.lo(R.own0, R.halo), R.res1 = R.hi(R.own1, R.halo);
    R.gi0 = R.lo(R.own0, R.band), R.gi1 = R.hi(R.own1, R.band);
    return NEB_OK;
}

// the swaps of n boundary rows with the neighbouring strips: what StripPartition._swap lists
uint32_t strip_swaps(const StripRows& R, uint32_t n, neb_halo_swap out[2])
{
    uint32_t k = 0;
    if (n == 0)
        return 0;
    if (R.r > 0)
        out[k++] = neb_halo_swap{(int32_t)(R.r - 1), R.own0, R.own0 + n, R.own0 - n, R.own0};
    if (R.r + 1 < R.N)
        out[k++] = neb_halo_swap{(int32_t)(R.r + 1), R.own1 - n, R.own1, R.own1, R.own1 + n};
    return k;
}

int strip_sync(neb_ctx* ctx)
{
    auto& s = ctx->strip;
    if (s.xstream && s.ready && s.done && s.pushed && s.frame_done)
        return NEB_OK;
    hipError_t e = s.xstream ? hipSuccess : hipStreamCreateWithFlags(&s.xstream, hipStreamNonBlocking);
    for (hipEvent_t* ev : {&s.ready, &s.done, &s.pushed, &s.frame_done})
        if (e == hipSuccess && !*ev) // (a call that failed half way is completed by the next one)
            e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e != hipSuccess)
        return strip_fail(ctx, NEB_ERR_HIP, std::string("neb_strip_frame: stream / event creation: ") + hipGetErrorString(e));
    return NEB_OK;
}

#define STRIP_HIP(ctx, call)                                                                                        \
    do {                                                                                                            \
        hipError_t e_ = (call);                                                                                     \
        if (e_ != hipSuccess)                                                                                       \
            return strip_fail(ctx, NEB_ERR_HIP, std::string("neb_strip_frame: " #call ": ") + hipGetErrorString(e_)); \
    } while (0)
#define STRIP_OK(call)          \
    do {                        \
        const int rc_ = (call); \
        if (rc_ != NEB_OK)      \
            return rc_;         \
    } while (0)

int check_context(neb_ctx* ctx, const StripRows& R, const char* who)
{
    const uint32_t want0 = R.N > 1 ? R.res0 : 0u, want1 = R.N > 1 ? R.res1 : ctx->H;
    if (ctx->row_begin != want0 || ctx->row_end != want1) {
        char msg[200];
        snprintf(msg, sizeof(msg), "%s: the context holds rows [%u, %u), strip %u of %u under this scheme needs [%u, %u)", who, ctx->row_begin, ctx->row_end, R.r, R.N,
                 want0, want1);
        return strip_fail(ctx, NEB_ERR_STATE, msg);
    }
    return NEB_OK;
}

// pushes rows [row0, row1) of (plane, slot) of `src` into the same rows of `dst` (another context: a neighbouring strip, any device)
int push_rows(neb_ctx* src, neb_ctx* dst, int plane, int slot, uint32_t row0, uint32_t row1, hipStream_t stream)
{
    if (row0 < src->row_begin || row1 > src->row_end || row0 < dst->row_begin || row1 > dst->row_end || dst->W != src->W)
        return strip_fail(src, NEB_ERR_OUT_OF_RANGE, "neb_strip_frame_begin: a neighbour's context does not hold the halo rows of this plan");
    const size_t pitch = (size_t)src->W * neb::kPlaneInfo[plane].bytes_per_px;
    const char* from = (const char*)src->planes[plane][slot] + (size_t)(row0 - src->row_begin) * pitch;
    char* to = (char*)dst->planes[plane][slot] + (size_t)(row0 - dst->row_begin) * pitch;
    if (dst->device == src->device)
        STRIP_HIP(src, hipMemcpyAsync(to, from, (size_t)(row1 - row0) * pitch, hipMemcpyDeviceToDevice, stream));
    else
        STRIP_HIP(src, hipMemcpyPeerAsync(to, dst->device, from, src->device, (size_t)(row1 - row0) * pitch, stream));
    return NEB_OK;
}

} // namespace

extern "C" {

int neb_strip_rows(const neb_ctx* ctx, const neb_strip_plan* plan, uint32_t out[8])
{
    if (!ctx || !out)
        return NEB_ERR_INVALID_ARG;
    StripRows R;
    std::string why;
    const int rc = strip_rows(ctx, plan, R, why);
    if (rc != NEB_OK)
        return rc;
    const uint32_t v[8] = {R.own0, R.own1, R.res0, R.res1, R.gi0, R.gi1, R.halo, R.band};
    memcpy(out, v, sizeof(v));
    return NEB_OK;
}

int neb_strip_frame_begin(neb_ctx* ctx, const neb_gi_constants* constants, const neb_strip_plan* plan, const neb_strip_peers* peers, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    StripRows R;
    std::string why;
    if (int rc = strip_rows(ctx, plan, R, why))
        return strip_fail(ctx, rc, "neb_strip_frame_begin: " + why);
    STRIP_OK(check_context(ctx, R, "neb_strip_frame_begin"));
    if (peers && R.N > 1 && R.scheme != NEB_STRIPS_ONCE)
        return strip_fail(ctx, NEB_ERR_INVALID_ARG, "neb_strip_frame_begin: the local transport (peers) serves scheme NEB_STRIPS_ONCE only");
    if (peers && R.N > 1 && ((R.r > 0 && !peers->up) || (R.r + 1 < R.N && !peers->down)))
        return strip_fail(ctx, NEB_ERR_INVALID_ARG, "neb_strip_frame_begin: a neighbouring strip's context is missing");
    if (plan->flags & NEB_STRIP_RESET_HISTORY)
        STRIP_OK(neb_svgf_reset_history(ctx, stream));
    if (constants)
        STRIP_OK(neb_gi_trace_rows(ctx, constants, R.gi0, R.gi1, stream));
    if (R.N == 1) // one strip = the whole frame: the whole-frame calls (the fused chain when the context opted in)
        return neb_svgf_temporal(ctx, stream);
    STRIP_OK(neb_svgf_temporal_rows(ctx, R.gi0, R.gi1, stream));
    if (peers && R.L > 0) {
        // local transport: this strip's boundary rows of the accumulated radiance and of the variance go straight into the neighbours' halos.
        // A neighbour's halo rows of radiance[cur] were a ping-pong buffer of ITS previous frame's levels: the push waits for that frame's end.
        neb::DeviceGuard guard(ctx->device);
        STRIP_OK(strip_sync(ctx));
        neb_halo_swap sw[2];
        const uint32_t n = strip_swaps(R, R.halo, sw);
        for (uint32_t k = 0; k < n; ++k) {
            neb_ctx* peer = sw[k].peer < (int32_t)R.r ? peers->up : peers->down;
            if (peer->strip.frame_done_recorded)
                STRIP_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, peer->strip.frame_done, 0));
            STRIP_OK(push_rows(ctx, peer, NEB_PLANE_RADIANCE, ctx->cur, sw[k].send_row0, sw[k].send_row1, (hipStream_t)stream));
            STRIP_OK(push_rows(ctx, peer, NEB_PLANE_VARIANCE, 0, sw[k].send_row0, sw[k].send_row1, (hipStream_t)stream));
        }
        STRIP_HIP(ctx, hipEventRecord(ctx->strip.pushed, (hipStream_t)stream));
    }
    return NEB_OK;
}

int neb_strip_frame_finish(neb_ctx* ctx, void* comm, const neb_strip_plan* plan, const neb_strip_peers* peers, neb_stream stream)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    StripRows R;
    std::string why;
    if (int rc = strip_rows(ctx, plan, R, why))
        return strip_fail(ctx, rc, "neb_strip_frame_finish: " + why);
    STRIP_OK(check_context(ctx, R, "neb_strip_frame_finish"));
    if (R.N == 1)
        return neb_svgf_atrous(ctx, stream);
    if (!comm && !peers)
        return strip_fail(ctx, NEB_ERR_INVALID_ARG, "neb_strip_frame_finish: more than one strip needs a transport (an RCCL communicator or the neighbours' contexts)");
    if (peers && R.scheme != NEB_STRIPS_ONCE)
        return strip_fail(ctx, NEB_ERR_INVALID_ARG, "neb_strip_frame_finish: the local transport (peers) serves scheme NEB_STRIPS_ONCE only");
    neb::DeviceGuard guard(ctx->device);
    STRIP_OK(strip_sync(ctx));
    hipStream_t st = (hipStream_t)stream;
    auto& sy = ctx->strip;
    neb_halo_swap sw[2];
    const uint32_t L = R.L;
    for (uint32_t level = 0; level < L; ++level) {
        const uint32_t e = R.extension(level), a0 = R.lo(R.own0, e), a1 = R.hi(R.own1, e);
        if (R.scheme == NEB_STRIPS_PER_LEVEL || (R.scheme == NEB_STRIPS_OVERLAP && level == L - 1)) { // the level's source rows, in front of it
            int sp, ss, dp, ds;
            STRIP_OK(neb_svgf_atrous_level_planes(ctx, level, &sp, &ss, &dp, &ds));
            const neb_halo_plane pl{sp, ss};
            const uint32_t n = strip_swaps(R, 2u << level, sw);
            STRIP_OK(neb_strips_exchange(ctx, comm, &pl, 1, sw, n, stream));
        }
        if (R.scheme == NEB_STRIPS_ONCE && level == 0) {
            // the one exchange of the frame runs beside level 0 on the rows that need none of the incoming halo (taps reach 2 rows)
            const uint32_t top = R.r > 0 ? R.own0 + 2 : a0, bot = R.r + 1 < R.N ? R.own1 - 2 : a1;
            if (comm) {
                const neb_halo_plane pl[2] = {{NEB_PLANE_RADIANCE, NEB_SLOT_CURRENT}, {NEB_PLANE_VARIANCE, 0}};
                const uint32_t n = strip_swaps(R, R.halo, sw);
                STRIP_HIP(ctx, hipEventRecord(sy.ready, st));
                STRIP_HIP(ctx, hipStreamWaitEvent(sy.xstream, sy.ready, 0));
                STRIP_OK(neb_strips_exchange(ctx, comm, pl, 2, sw, n, sy.xstream));
                STRIP_HIP(ctx, hipEventRecord(sy.done, sy.xstream));
            }
            if (top < bot)
                STRIP_OK(neb_svgf_atrous_level_rows(ctx, 0, top, bot, stream));
            if (comm) {
                STRIP_HIP(ctx, hipStreamWaitEvent(st, sy.done, 0));
            } else { // local transport: the neighbours pushed their rows in their neb_strip_frame_begin
                if (R.r > 0)
                    STRIP_HIP(ctx, hipStreamWaitEvent(st, peers->up->strip.pushed, 0));
                if (R.r + 1 < R.N)
                    STRIP_HIP(ctx, hipStreamWaitEvent(st, peers->down->strip.pushed, 0));
            }
            if (a0 < top)
                STRIP_OK(neb_svgf_atrous_level_rows(ctx, 0, a0, top, stream));
            if (bot < a1)
                STRIP_OK(neb_svgf_atrous_level_rows(ctx, 0, bot, a1, stream));
        } else {
            STRIP_OK(neb_svgf_atrous_level_rows(ctx, level, a0, a1, stream));
        }
    }
    if (R.scheme == NEB_STRIPS_OVERLAP && R.band) { // both copies of next frame's history must hold the FINAL rows
        int sp, ss, dp = NEB_PLANE_RADIANCE, ds = ctx->cur;
        if (L > 1)
            STRIP_OK(neb_svgf_atrous_level_planes(ctx, L - 1, &sp, &ss, &dp, &ds));
        const neb_halo_plane pl{dp, ds};
        const uint32_t n = strip_swaps(R, R.band, sw);
        STRIP_OK(neb_strips_exchange(ctx, comm, &pl, 1, sw, n, stream));
    }
    if (L == 1) { // a single level filters into the scratch plane: the owned rows go back (as neb_svgf_atrous does for a whole frame)
        const size_t pitch = (size_t)ctx->W * 16u;
        STRIP_HIP(ctx, hipMemcpyAsync((char*)ctx->planes[NEB_PLANE_RADIANCE][ctx->cur] + (size_t)(R.own0 - ctx->row_begin) * pitch,
                                      (const char*)ctx->planes[NEB_PLANE_SCRATCH][0] + (size_t)(R.own0 - ctx->row_begin) * pitch, (size_t)R.h * pitch,
                                      hipMemcpyDeviceToDevice, st));
    }
    if (peers) {
        STRIP_HIP(ctx, hipEventRecord(sy.frame_done, st));
        sy.frame_done_recorded = true;
    }
    return NEB_OK;
}

int neb_strip_frame(neb_ctx* ctx, const neb_gi_constants* constants, void* comm, const neb_strip_plan* plan, neb_stream stream)
{
    const int rc = neb_strip_frame_begin(ctx, constants, plan, nullptr, stream);
    return rc != NEB_OK ? rc : neb_strip_frame_finish(ctx, comm, plan, nullptr, stream);
}

} // extern "C"

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

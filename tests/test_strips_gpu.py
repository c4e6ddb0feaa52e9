"""GPU test of the row-strip path with the HIP backend: two strip contexts on ONE GPU, the RCCL exchange
replaced by a direct device-to-device row copy between them (tests/strip_harness.py; the rendezvous itself is covered
by the gloo tests in test_strips_cpu.py).  GI + SVGF on strips must equal the full-image run bit for bit."""
import numpy as np
import pytest
import torch

from nebulae_amd import scene as S
from nebulae_amd import strips
from nebulae_amd.renderer import RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE
from strip_harness import LockstepStrips

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scheme,W,H,N", [("once", 256, 192, 2), ("per_level", 256, 192, 2),
                                          # the bench's N = 8 shape: ONE 1920x1080 frame in eight 135-row strips (halo 62 rows)
                                          ("once", 1920, 1080, 8), ("per_level", 1920, 1080, 8),
                                          ("overlap", 256, 192, 2), ("overlap", 1920, 1080, 8)])
def test_strips_equal_full_image_gi_plus_svgf(scheme, W, H, N):
    """(the full image runs the fused SVGF chain, the strips the separate kernels: the comparison is also fused == separate)"""
    L = 5
    sc = S.atrium_standin(target_triangles=20000, n_submeshes=40, tex_size=32)
    cam = S.sponza_camera()
    full = strips.StripRenderer(strips.StripPartition(W, H, 1, L), 0)
    ls = LockstepStrips(W, H, N, L, scheme)
    for f in range(1, 6 if N == 2 else 5):
        info = RenderInfo(scene=sc, camera=cam, frame_index=f)
        for r in [full] + ls.rs:
            r.begin_frame(info)
            r.submit_commands_gbuffer()
            cur = r.svgf.get_current_resource_index()
            r.svgf.plane_tensor(PLANE_RADIANCE, cur).zero_()
            r.submit_commands_gi_pathtrace()
        torch.cuda.synchronize()
        ran_full = full.submit_commands_svgf_denoising()
        assert all(x == ran_full for x in ls.denoise())
        torch.cuda.synchronize()
    want = full.svgf.download(PLANE_RADIANCE)
    got = ls.image()
    assert np.isfinite(want).all() and float(np.abs(want[..., :3]).max()) > 0.0
    assert np.array_equal(got, want)
    full.destroy()
    ls.destroy()


@pytest.mark.parametrize("W,H,N,depth", [(256, 192, 2, 1), (256, 192, 2, 2), (1920, 1080, 8, 2)])
def test_strips_with_the_gi_stages_on_a_side_stream_equal_full_image(W, H, N, depth):
    """bench.py --overlap (auto for strips of at most 0.6 M pixels): every strip traces with "gi_defer_resolve" on a side stream and
    adds its indirect term with neb_gi_resolve on the main stream, after the direct term was written there -- the strips must still
    equal the full image (fused dispatch, one stream) bit for bit."""
    L = 5
    sc = S.atrium_standin(target_triangles=20000, n_submeshes=40, tex_size=32)
    cam = S.sponza_camera()
    full = strips.StripRenderer(strips.StripPartition(W, H, 1, L), 0)
    ls = LockstepStrips(W, H, N, L, "once")
    main, sides = torch.cuda.current_stream(), [torch.cuda.Stream() for _ in range(depth)]
    resolved = [None] * depth  # depth 2: "gi_defer_resolve" = 2, two record sets used alternately, the GI stages of two frames in flight
    for f in range(1, 7):
        side, slot = sides[f % depth], f % depth
        info = RenderInfo(scene=sc, camera=cam, frame_index=f)
        full.begin_frame(info)
        full.submit_commands_gbuffer()
        full.svgf.plane_tensor(PLANE_RADIANCE, full.svgf.get_current_resource_index()).fill_(0.125)
        full.submit_commands_gi_pathtrace()
        for r in ls.rs:
            r.begin_frame(info)
            if f == 1:
                r.set_defer_resolve(depth)  # (an option of the GI state: it exists once the first frame has set the scene)
            r.submit_commands_gbuffer()
        drawn = torch.cuda.Event()
        drawn.record(main)
        side.wait_event(drawn)
        if resolved[slot] is not None:
            side.wait_event(resolved[slot])  # the resolve that consumed this record set
        for r in ls.rs:
            r.submit_commands_gi_pathtrace(stream=side.cuda_stream)
            r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).fill_(0.125)  # the direct term, on the main stream
        traced = torch.cuda.Event()
        traced.record(side)
        main.wait_event(traced)
        for r in ls.rs:
            r.submit_commands_gi_resolve()
        resolved[slot] = torch.cuda.Event()
        resolved[slot].record(main)
        ran_full = full.submit_commands_svgf_denoising()
        assert all(x == ran_full for x in ls.denoise())
    torch.cuda.synchronize()
    want = full.svgf.download(PLANE_RADIANCE)
    got = ls.image()
    assert np.isfinite(want).all() and float(np.abs(want[..., :3]).max()) > 0.2
    assert np.array_equal(got, want)
    full.destroy()
    ls.destroy()


@pytest.mark.parametrize("W,H,N", [(256, 192, 2), (1920, 1080, 4)])
def test_strips_with_the_dispatch_in_two_calls_equal_full_image(W, H, N):
    """bench.py's "split" form on strips: every strip walks its rows with neb_gi_trace_begin on a side stream and shades them with
    neb_gi_trace_finish on the main stream once the direct term is there -- still the full image (one call, one stream) bit for bit."""
    L = 5
    sc = S.atrium_standin(target_triangles=20000, n_submeshes=40, tex_size=32)
    cam = S.sponza_camera()
    full = strips.StripRenderer(strips.StripPartition(W, H, 1, L), 0)
    ls = LockstepStrips(W, H, N, L, "once")
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    shaded = None
    for f in range(1, 6):
        info = RenderInfo(scene=sc, camera=cam, frame_index=f)
        full.begin_frame(info)
        full.submit_commands_gbuffer()
        full.svgf.plane_tensor(PLANE_RADIANCE, full.svgf.get_current_resource_index()).fill_(0.125)
        full.submit_commands_gi_pathtrace()
        for r in ls.rs:
            r.begin_frame(info)
            r.submit_commands_gbuffer()
        drawn = torch.cuda.Event()
        drawn.record(main)
        side.wait_event(drawn)
        if shaded is not None:
            side.wait_event(shaded)
        for r in ls.rs:
            r.submit_commands_gi_pathtrace_begin(rows=r.part.gi_rows(r.rank), stream=side.cuda_stream)
            r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).fill_(0.125)  # the direct term, on the main stream
        walked = torch.cuda.Event()
        walked.record(side)
        main.wait_event(walked)
        for r in ls.rs:
            r.submit_commands_gi_pathtrace_finish()
        shaded = torch.cuda.Event()
        shaded.record(main)
        ran_full = full.submit_commands_svgf_denoising()
        assert all(x == ran_full for x in ls.denoise())
    torch.cuda.synchronize()
    want = full.svgf.download(PLANE_RADIANCE)
    got = ls.image()
    assert np.isfinite(want).all() and float(np.abs(want[..., :3]).max()) > 0.2
    assert np.array_equal(got, want)
    full.destroy()
    ls.destroy()


def test_c_entry_point_exchanges_rows_over_rccl():
    """neb_strips_exchange: grouped ncclSend / ncclRecv straight out of / into the planes, on the caller's stream, through an
    RCCL communicator the library creates itself (librccl resolved at run time).  One GPU: a communicator of one rank that
    sends its rows to itself -- the same call a strip makes towards its neighbours (the multi-rank rendezvous runs on the
    driver's multi-GPU node; the partition arithmetic is covered by the gloo tests)."""
    import ctypes as C

    from nebulae_amd import _lib
    from nebulae_amd.svgf import PLANE_VARIANCE, SLOT_CURRENT, NebError, SVGFDenoiser
    W, H = 64, 48
    d = SVGFDenoiser()
    d.init(W, H, atrous_levels=2, row_begin=8, row_end=40)
    lib = d._lib
    ident = (C.c_char * 128)()
    assert lib.neb_strips_unique_id(ident) == 0, lib.neb_strips_last_error()
    comm = C.c_void_p()
    assert lib.neb_strips_comm_create(0, 1, 0, ident, C.byref(comm)) == 0, lib.neb_strips_last_error()
    d.begin_frame(3)
    rng = np.random.default_rng(5)
    rad = rng.normal(size=(32, W, 4)).astype(np.float32)
    var = rng.uniform(size=(32, W)).astype(np.float16)
    d.upload(PLANE_RADIANCE, SLOT_CURRENT, rad)
    d.upload(PLANE_VARIANCE, 0, var)
    planes = (_lib.HaloPlane * 2)(_lib.HaloPlane(PLANE_RADIANCE, SLOT_CURRENT), _lib.HaloPlane(PLANE_VARIANCE, 0))
    swaps = (_lib.HaloSwap * 2)(_lib.HaloSwap(0, 12, 16, 8, 12), _lib.HaloSwap(0, 28, 34, 34, 40))  # (peer, send rows, recv rows: disjoint)
    stream = torch.cuda.current_stream().cuda_stream
    # (inside the caller's own group bracket, as a one-thread multi-GPU host makes the call: the inner group nests)
    assert lib.neb_strips_group_begin() == 0
    d._check(lib.neb_strips_exchange(d._ctx, comm, planes, 2, swaps, 2, C.c_void_p(stream)), "neb_strips_exchange")
    assert lib.neb_strips_group_end() == 0
    torch.cuda.synchronize()
    got_r, got_v = d.download(PLANE_RADIANCE), d.download(PLANE_VARIANCE)
    want_r, want_v = rad.copy(), var.copy()
    want_r[0:4], want_v[0:4] = rad[4:8], var[4:8]          # image rows 8..12 <- 12..16
    want_r[26:32], want_v[26:32] = rad[20:26], var[20:26]  # image rows 34..40 <- 28..34
    assert np.array_equal(got_r, want_r) and np.array_equal(got_v, want_v)
    # rows outside the resident range are refused before any RCCL call is made
    bad = (_lib.HaloSwap * 1)(_lib.HaloSwap(0, 0, 4, 8, 12))
    assert lib.neb_strips_exchange(d._ctx, comm, planes, 2, bad, 1, C.c_void_p(stream)) == -5
    assert lib.neb_strips_comm_destroy(comm) == 0
    d.destroy()


@pytest.mark.parametrize("scheme", ["once", "per_level", "overlap"])
@pytest.mark.parametrize("H,N,L", [(192, 2, 5), (1080, 8, 5), (2160, 4, 5), (240, 2, 3), (96, 1, 4)])
def test_the_librarys_strip_rows_are_the_partitions(scheme, H, N, L):
    """neb_strip_rows (the partition arithmetic of the one-call strip frame, strips.hip) == strips.StripPartition, every strip of every scheme"""
    import ctypes as C

    from nebulae_amd import _lib
    from nebulae_amd.svgf import SVGFDenoiser
    W = 64
    part = strips.StripPartition(W, H, N, L, scheme=scheme)
    for r in range(N):
        res = part.resident(r)
        d = SVGFDenoiser()
        d.init(W, H, atrous_levels=L, row_begin=res[0], row_end=res[1] if N > 1 else 0)
        out = (C.c_uint32 * 8)()
        plan = _lib.StripPlan(N, r, _lib.STRIP_SCHEMES[scheme], 0)
        d._check(d._lib.neb_strip_rows(d._ctx, C.byref(plan), out), "neb_strip_rows")
        assert tuple(out) == (*part.owned(r), *part.resident(r), *part.gi_rows(r), part.halo, part.band), (scheme, r, tuple(out))
        d.destroy()


@pytest.mark.parametrize("W,H,N", [(256, 192, 2), (1920, 1080, 8)])
def test_one_call_strip_frames_with_the_local_transport_equal_full_image(W, H, N):
    """neb_strip_frame_begin / _finish (ONE library call per strip and phase: GI rows -> temporal rows -> halo rows pushed into the neighbours'
    contexts -> level 0's interior beside them -> its border rows -> the other levels) on N strip contexts of one GPU, all on one stream and
    without a host synchronisation inside the frame: the strips must equal the full image (fused chain, one context) bit for bit -- frame
    policy included (frame 1 moves: SVGF skipped; frame 2 resets the history)."""
    L = 5
    sc = S.atrium_standin(target_triangles=20000, n_submeshes=40, tex_size=32)
    cam = S.sponza_camera()
    full = strips.StripRenderer(strips.StripPartition(W, H, 1, L), 0)
    part = strips.StripPartition(W, H, N, L, scheme="once")
    rs = [strips.StripRenderer(part, k) for k in range(N)]
    for f in range(1, 7):
        info = RenderInfo(scene=sc, camera=cam, frame_index=f)
        for r in [full] + rs:
            r.begin_frame(info)
            r.submit_commands_gbuffer()
            r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).fill_(0.125)
        ran_full = full.submit_strip_frame()  # one strip = the whole frame, through the same entry point
        ran = [r.submit_strip_frame_local("begin", rs[k - 1] if k > 0 else None, rs[k + 1] if k + 1 < N else None) for k, r in enumerate(rs)]
        ran2 = [r.submit_strip_frame_local("finish", rs[k - 1] if k > 0 else None, rs[k + 1] if k + 1 < N else None) for k, r in enumerate(rs)]
        assert ran == ran2 == [ran_full] * N
        for r in [full] + rs:
            r.end_frame()
    torch.cuda.synchronize()
    want = full.svgf.download(PLANE_RADIANCE)
    got = np.concatenate([r.svgf.download(PLANE_RADIANCE, row0=part.owned(r.rank)[0], nrows=part.owned(r.rank)[1] - part.owned(r.rank)[0]) for r in rs], axis=0)
    assert np.isfinite(want).all() and float(np.abs(want[..., :3]).max()) > 0.2
    assert np.array_equal(got, want)
    # a plan that does not fit the context is refused with a message, before anything is enqueued
    import ctypes as C

    from nebulae_amd import _lib
    bad = _lib.StripPlan(N, 0, _lib.STRIP_SCHEMES["per_level"], 0)
    assert rs[0]._lib.neb_strip_frame(rs[0]._ctx, None, None, C.byref(bad), None) != 0
    full.destroy()
    for r in rs:
        r.destroy()

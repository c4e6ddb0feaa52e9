"""Multi-GPU screen-space sharding of the hot path: one process per GPU, row strips, RCCL halo rows.

The reference is single-GPU; this is new functionality whose oracle is
"N-GPU output == 1-GPU output, bit for bit".  Every stage of the path is per-pixel except the
a-trous wavelet, whose level l reads rows up to 2 * 2^l away (svgf_atrous.hlsl:51-65), so:

* rank r owns image rows [r*H/N, (r+1)*H/N) and keeps `halo` more rows resident on either side, clipped to the
  image; the G-buffer producer fills depth/normal for all resident rows (no exchange for those);
* GI and temporal accumulation run on the OWNED rows only; history needs no exchange (next frame's temporal
  pass reads history at owned rows only).

Two exchange schemes (``StripPartition(scheme=...)``, both bit-identical to one GPU):

``"once"`` (default) -- ONE exchange per frame.  After the temporal pass each rank swaps its h = sum_l 2*2^l
  (62 for L = 5) boundary rows of the temporally accumulated radiance (16 B/px) and of the variance plane
  (2 B/px, read at a level's centre texel) with its up/down neighbour, then runs level l on
  ``owned +- e_l`` rows, e_l = sum_{m>l} 2*2^m: the rows later levels still reach into.  The redundant a-trous
  work is sum_l 2 e_l rows (392 row-levels against 5 * 540 at 4K/4, +14 % of a stage that is a fifth of the frame)
  and buys 4 fewer exchange / stream-sync points per frame.  This is SURVEY.md 8e's "fully redundant tiling"
  applied to the cheap stage only: GI and temporal are never recomputed.

``"per_level"`` -- L exchanges per frame: before level l the 2 * 2^l boundary rows of that level's source plane
  are swapped; every level runs on owned rows only; halo = 2 * 2^(L-1).

``"overlap"`` -- SURVEY.md 8e's scheme from north_star: every rank ALSO traces, accumulates and filters levels 0 .. L-2 on
  B = sum_{l<=L-2} 2*2^l rows (30 for L = 5) beyond its strip -- the GI dispatch is a pure function of the pixel and the
  frame, so the copies agree bit for bit -- then swaps the 2*2^(L-1) (32) boundary rows of level L-2's output, runs the widest
  level on its own rows, and finally swaps B rows of the FINAL image so that both copies of next frame's radiance history agree
  (quirk 5: the filtered output is the history).  Two exchanges of 32 + 30 rows; the redundancy is paid in the GI stage, three
  quarters of the frame, which is why the cost table never picks it -- built so that the three can be measured.

All move 62 radiance rows per direction and frame at L = 5; rows are contiguous row blocks sent straight out of
the plane (``torch.distributed`` P2P, backend "nccl" = RCCL over xGMI; one direct link per neighbour).

The compute backend is injected (``denoiser_factory``): the product uses the HIP ``SVGFDenoiser``;
the world_size-2 gloo tests inject an oracle-backed stand-in to exercise exactly this file on CPU.
"""
import math

from .renderer import DeferredRenderer
from .svgf import PLANE_RADIANCE, PLANE_SCRATCH, PLANE_VARIANCE, NebError, SVGFDenoiser


def frame_factors(n):
    """(a, b) with a*b == n, as square as possible, b >= a: the weak-scaling frame is (W*a) x (H*b)."""
    a = int(math.isqrt(n))
    while n % a:
        a -= 1
    return a, n // a


# Cost table behind the choice of the exchange scheme (microseconds).  The kernel side is measured on one MI355X; the xGMI side
# (EXCHANGE_LATENCY_US, LINK_GBPS) are GUESSES until a run with N > 1 ranks measures them: `measure_link` does that in a few
# milliseconds at start-up (bench.py calls it), and only a measured link lets "auto" leave the default scheme.
# Override with NEB_STRIPS_EXCHANGE_LATENCY_US / NEB_STRIPS_LINK_GBPS.
ATROUS_US_PER_MPX_LEVEL = 12.8   # one a-trous level over a megapixel (26.6 us per 2.07 Mpx level at 1080p, profiles/r04g_kernel_stats.csv)
GI_TEMPORAL_US_PER_MPX = 250.0   # the GI dispatch + the temporal pass over a megapixel (486 + 32 us per 2.07 Mpx, profiles/r05*_kernel_stats.csv)
# What ONE strip of a 1920x1080 frame takes alone on a GPU, frames in flight as bench.py runs them (profiles/r04_strip_overlap.txt, one MI355X,
# exchange stubbed): the ceiling of the metric's strong-scaling curve before a byte is exchanged -- N strips finish a frame in this time.
STRIP_FRAME_US_1080P = {1: 645.0, 2: 379.0, 4: 237.0, 8: 150.0}


def strong_scaling_ceilings(frame_us_one_gpu=None):
    """{N: (frames/s, speed-up over one GPU)} a 1920x1080 frame in N row strips cannot exceed: a strip's own frame time on one MI355X with the
    exchange stubbed (STRIP_FRAME_US_1080P) -- a 135-row strip is a latency-bound launch sequence, not an eighth of the work.  bench.py prints it in
    `config` so that a SCALE_rNN.json can be read against it."""
    t1 = float(frame_us_one_gpu or STRIP_FRAME_US_1080P[1])
    return {n: {"frames_per_s": round(1e6 / (t if n > 1 else t1), 1), "speedup": round(t1 / (t if n > 1 else t1), 2)} for n, t in STRIP_FRAME_US_1080P.items()}
EXCHANGE_LATENCY_US = 12.0       # GUESS: one grouped send + receive with a neighbour, launch to completion, message size aside
LINK_GBPS = 60.0                 # GUESS: sustained one-direction rate of one xGMI link for row blocks of a few hundred KB


class Link:
    """What one halo exchange with a neighbour costs: latency (us, launch to completion of a grouped send + receive of a few KB)
    and the one-direction rate (GB/s) for row blocks of a few MB.  `measured` says whether the numbers come from measure_link."""

    def __init__(self, latency_us=None, gbps=None, measured=False, note=""):
        import os
        self.latency_us = float(os.environ.get("NEB_STRIPS_EXCHANGE_LATENCY_US", EXCHANGE_LATENCY_US if latency_us is None else latency_us))
        self.gbps = float(os.environ.get("NEB_STRIPS_LINK_GBPS", LINK_GBPS if gbps is None else gbps))
        self.measured, self.note = bool(measured), note

    def label(self):
        return (f"link {'measured' if self.measured else 'UNMEASURED (guessed constants)'}: {self.latency_us:.1f} us per exchange, "
                f"{self.gbps:.1f} GB/s per direction" + (f" ({self.note})" if self.note else ""))


def measure_link(rank, world, group, make_buffer, synchronize, nbytes=2 << 20, iters=20, small_bytes=4096):
    """Times the product's own exchange pattern -- one batch_isend_irecv of a send + a receive with a neighbour, both directions
    at once -- between the pairs (0,1), (2,3), ...: `iters` exchanges of `small_bytes` (the latency) and of `nbytes` (2 MB: the
    size of a 1080p frame's 62 halo rows), outside any timed region.  The result is the MAX over ranks (one all-reduce), so every
    rank feeds the SAME constants to choose_scheme and picks the same scheme.  make_buffer(n) -> a uint8 tensor of n bytes where
    the planes live (device memory under RCCL); synchronize() drains the device."""
    import time

    import torch
    import torch.distributed as dist
    if world < 2:
        return Link()
    peer = rank ^ 1
    active = peer < world
    out = []
    for n in (small_bytes, nbytes):
        tx, rx = make_buffer(n), make_buffer(n)

        def once():
            if active:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, tx, peer, group=group), dist.P2POp(dist.irecv, rx, peer, group=group)]):
                    w.wait()
            synchronize()
        for _ in range(3):
            once()
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(iters):
            once()
        out.append((time.perf_counter() - t0) / iters * 1e6)
    v = torch.tensor(out, dtype=torch.float64)
    v = v.to(tx.device) if (tx.is_cuda and dist.get_backend(group) != "gloo") else v
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    t_small, t_big = float(v[0]), float(v[1])
    # bytes / us / 1e3 = GB/s; when the two sizes take about as long (a latency-dominated or noisy transport) the difference says
    # nothing: fall back to the conservative whole-message rate
    gbps = ((nbytes - small_bytes) / (t_big - t_small) if t_big > 1.25 * t_small else nbytes / max(t_big, 1e-3)) / 1e3
    return Link(t_small, gbps, measured=True,
                note=f"{iters} exchanges of {small_bytes} B: {t_small:.1f} us each; of {nbytes} B: {t_big:.1f} us each; max over ranks; host-synchronised")


def scheme_costs(width, height, world, levels, link=None):
    """-> {"once": us, "per_level": us, "overlap": us}: what each exchange scheme ADDS to a frame of a middle strip.
    once: sum_l 2 e_l redundant row-levels of a-trous + ONE exchange of h = sum 2*2^l rows x 18 B/px (the exchange runs beside
    the interior of level 0: only what exceeds that level's own time is charged); per_level: L exchanges of 2*2^l rows x 16 B/px,
    each a sync point in front of its level (nothing to overlap with)."""
    link = link or Link()
    lat, gbps = link.latency_us, link.gbps
    rows = height // world
    halo = sum(2 * (1 << l) for l in range(levels))
    redundant_rows = sum(2 * sum(2 * (1 << m) for m in range(l + 1, levels)) for l in range(levels))
    once_x = lat + halo * width * 18 / (gbps * 1e3)                       # us (bytes / (GB/s * 1e3) = us)
    level0 = rows * width * 1e-6 * ATROUS_US_PER_MPX_LEVEL
    once = redundant_rows * width * 1e-6 * ATROUS_US_PER_MPX_LEVEL + max(0.0, once_x - level0)
    per_level = sum(lat + 2 * (1 << l) * width * 16 / (gbps * 1e3) for l in range(levels))
    band = sum(2 * (1 << l) for l in range(levels - 1))
    ext = [sum(2 * (1 << m) for m in range(l + 1, levels - 1)) for l in range(levels - 1)]
    overlap = (2 * band * width * 1e-6 * (GI_TEMPORAL_US_PER_MPX + 0.0) + sum(2 * e for e in ext) * width * 1e-6 * ATROUS_US_PER_MPX_LEVEL
               + 2 * lat + (2 * (1 << (levels - 1)) + band) * width * 16 / (gbps * 1e3))
    return {"once": once, "per_level": per_level, "overlap": overlap}


def choose_scheme(width, height, world, levels, link=None):
    """-> (scheme, reason).  "once" needs strips at least as tall as its halo.  Without a MEASURED link the answer is the default,
    "once" (one sync point per frame; the cost table is printed but does not decide: its xGMI constants are guesses); with one
    (measure_link) the cheaper scheme by scheme_costs."""
    if world == 1 or levels == 0:
        return "once", "single strip"
    link = link or Link()
    c = scheme_costs(width, height, world, levels, link)
    halo_once = 2 * ((1 << levels) - 1)
    if height // world < halo_once:
        return "per_level", f"strips of {height // world} rows are shorter than the {halo_once}-row halo of 'once'"
    table = (f"cost table: once +{c['once']:.0f} us, per_level +{c['per_level']:.0f} us, overlap +{c['overlap']:.0f} us per frame at "
             f"{height // world}-row strips; {link.label()}")
    if not link.measured:
        return "once", "default (the link constants are unmeasured, so the table does not decide); " + table
    return min(("once", "per_level", "overlap"), key=lambda k: c[k]), table


class StripPartition:
    def __init__(self, width, height, world, levels, scheme=None, link=None):
        import os
        if height % world:
            raise ValueError(f"image height {height} is not divisible by {world} strips")
        scheme = scheme or os.environ.get("NEB_STRIPS_SCHEME") or "auto"
        if scheme == "auto":
            self.scheme, self.scheme_reason = choose_scheme(width, height, world, levels, link)
        else:
            self.scheme, self.scheme_reason = scheme, "requested"
        if self.scheme not in ("once", "per_level", "overlap"):
            raise ValueError(f"unknown exchange scheme {self.scheme!r}")
        self.W, self.H, self.N, self.L = width, height, world, levels
        # "overlap": rows beyond the strip on which GI, the temporal pass and levels 0 .. L-2 are recomputed
        self.band = sum(2 * (1 << l) for l in range(levels - 1)) if (self.scheme == "overlap" and world > 1) else 0
        if world > 1 and levels > 0:
            self.halo = 2 * ((1 << levels) - 1) if self.scheme == "once" else 2 * (1 << (levels - 1))  # (overlap: max(band, widest reach) = the reach)
        else:
            self.halo = 0
        if world > 1 and height // world < self.halo:
            raise ValueError("strips are shorter than the a-trous reach; use fewer GPUs or fewer levels")

    def owned(self, r):
        h = self.H // self.N
        return r * h, (r + 1) * h

    def resident(self, r):
        a, b = self.owned(r)
        return max(0, a - self.halo), min(self.H, b + self.halo)

    def rows_temporal(self, r):
        a, b = self.gi_rows(r)
        return b - a

    def gi_rows(self, r):
        """image rows [row0, row1) on which rank r runs the GI dispatch and the temporal pass: its strip (+- the band of "overlap")"""
        a, b = self.owned(r)
        return max(0, a - self.band), min(self.H, b + self.band)

    def level_extension(self, level):
        """rows beyond the owned strip that `level` must also filter: what the later levels still reach into"""
        if self.N == 1 or self.scheme == "per_level":
            return 0
        if self.scheme == "overlap":  # levels 0 .. L-2 are local: what levels l+1 .. L-2 still reach into; the widest level: nothing
            return sum(2 * (1 << m) for m in range(level + 1, self.L - 1))
        return sum(2 * (1 << m) for m in range(level + 1, self.L))

    def atrous_rows(self, r, level):
        """image rows [row0, row1) rank r filters at `level`"""
        a, b = self.owned(r)
        e = self.level_extension(level)
        return max(0, a - e), min(self.H, b + e)

    def _swap(self, r, n):
        a, b = self.owned(r)
        out = []
        if r > 0:
            out.append((r - 1, (a, a + n), (a - n, a)))
        if r < self.N - 1:
            out.append((r + 1, (b - n, b), (b, b + n)))
        return out

    def frame_exchange(self, r):
        """scheme "once": [(peer, (send_row0, send_row1), (recv_row0, recv_row1))] after the temporal pass"""
        return self._swap(r, self.halo) if (self.scheme == "once" and self.halo) else []

    def level_exchange(self, r, level):
        """scheme "per_level": the same for the source plane of `level`; scheme "overlap": for the widest level only"""
        if self.N > 1 and (self.scheme == "per_level" or (self.scheme == "overlap" and level == self.L - 1)):
            return self._swap(r, 2 * (1 << level))
        return []

    def history_exchange(self, r):
        """scheme "overlap": after the last level, the rows of the FINAL image that lie in the neighbours' bands (their copies of
        next frame's radiance history)"""
        return self._swap(r, self.band) if (self.scheme == "overlap" and self.band) else []

    def exchanged_bytes_per_frame(self):
        """bytes a middle rank sends per frame (both neighbours)"""
        if self.N == 1:
            return 0
        rows = sum(2 * (1 << l) for l in range(self.L))
        if self.scheme == "overlap":
            rows = 2 * (1 << (self.L - 1)) + self.band
        return 2 * rows * self.W * (16 + 2 if self.scheme == "once" else 16)


class StripRenderer(DeferredRenderer):
    """DeferredRenderer over one row strip.  With world == 1 it is exactly the single-GPU renderer."""

    def __init__(self, part, rank, device=0, group=None, denoiser_factory=SVGFDenoiser, exchange=None):
        """exchange: "torch" (default) = torch.distributed P2P on the planes; "rccl" (or NEB_STRIPS_EXCHANGE=rccl) = the
        library's own C entry point neb_strips_exchange (grouped ncclSend / ncclRecv on its own communicator and a side
        stream): what a C++ host without PyTorch would call.  `group` is then only used once, to carry the RCCL unique id."""
        super().__init__()
        self.svgf = denoiser_factory()
        self.part, self.rank, self.group = part, rank, group
        res0, res1 = part.resident(rank)
        self.init(part.W, part.H, atrous_levels=part.L, device=device, row_begin=res0, row_end=res1 if part.N > 1 else 0)
        self._views = {}
        import os
        self._staging = os.environ.get("NEB_STRIPS_STAGING") or None  # None | "host" | "device"
        self._staging_decided = False
        requested = exchange or os.environ.get("NEB_STRIPS_EXCHANGE")
        if requested not in (None, "torch", "rccl"):
            raise ValueError(f"unknown halo exchange backend {requested!r}")
        self._comm = None
        self._xstream = None
        self._device = device
        self.exchange = requested or "torch"
        if part.N > 1 and self._rccl_wanted(requested):
            # round 5: the library's own transport is the default wherever it can run (ONE transport to debug on a multi-GPU node); when the
            # caller did not ask for it, a librccl that cannot be loaded sends every rank back to torch's P2P together
            self._comm = self._create_comm(required=requested == "rccl")
            self.exchange = "rccl" if self._comm is not None else "torch"
        self._plan = None  # neb_strip_plan of this strip (the one-call frame of the C ABI), made on first use

    def _rccl_wanted(self, requested):
        if requested is not None:
            return requested == "rccl"
        try:  # default: the HIP denoiser on a GPU, ranks talking over RCCL already (not the gloo stand-ins of the CPU tests)
            import torch
            import torch.distributed as dist
            return (isinstance(self.svgf, SVGFDenoiser) and torch.cuda.is_available() and dist.is_initialized()
                    and dist.get_backend(self.group) == "nccl")
        except Exception:
            return False

    def _create_comm(self, required=True):
        """ncclCommInitRank through the library: rank 0 draws the unique id, the torch group (any backend) carries it -- with a flag that
        says whether rank 0 could (librccl present): without it every rank falls back together (or raises, when RCCL was asked for)."""
        import ctypes as C

        import torch
        import torch.distributed as dist
        lib = self._lib
        buf = (C.c_char * 128)()
        ok = 1
        if self.rank == 0:
            ok = 1 if lib.neb_strips_unique_id(buf) == 0 else 0
        on_cuda = dist.get_backend(self.group) != "gloo"
        t = torch.frombuffer(bytearray(bytes(buf) + bytes([ok])), dtype=torch.uint8).clone()
        t = t.cuda() if on_cuda else t
        dist.broadcast(t, src=0, group=self.group)
        raw = bytes(t.cpu().numpy().tobytes())
        if not raw[128]:
            if required:
                raise NebError(f"neb_strips_unique_id failed on rank 0: {lib.neb_strips_last_error().decode() if self.rank == 0 else 'see rank 0'}")
            return None
        comm = C.c_void_p()
        rc = lib.neb_strips_comm_create(self._device, self.part.N, self.rank, raw[:128], C.byref(comm))
        # every rank learns whether EVERY rank has its communicator: one that failed must not leave the others waiting in an exchange
        flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32)
        flag = flag.cuda() if on_cuda else flag
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.cpu()[0]) == 0:
            if rc == 0:
                lib.neb_strips_comm_destroy(comm)
            if required:
                raise NebError(f"neb_strips_comm_create failed on {'this rank' if rc != 0 else 'another rank'} ({rc}): {lib.neb_strips_last_error().decode()}")
            return None
        return comm

    # ---- the one-call strip frame of the C ABI (neb_strip_frame*, strips.hip): the partition arithmetic and the enqueue order below, in C ----
    def strip_plan(self, reset_history=False):
        from . import _lib
        return _lib.StripPlan(self.part.N, self.rank, _lib.STRIP_SCHEMES[self.part.scheme], _lib.STRIP_RESET_HISTORY if reset_history else 0)

    def has_c_frame(self):
        """the library call can serve this strip: one strip (the whole frame), or the RCCL transport (a torch.distributed exchange is Python's)"""
        return isinstance(self.svgf, SVGFDenoiser) and (self.part.N == 1 or self._comm is not None)

    def submit_strip_frame_local(self, phase, up, down, with_gi=True, stream=None):
        """The same frame for a host that drives ALL strips from one thread (several contexts, any devices): `phase` "begin" for every strip,
        then "finish" for every strip; up / down = the neighbouring StripRenderers (None at the image's edges).  Rows travel by
        hipMemcpyPeerAsync between the contexts (scheme "once").  -> whether SVGF runs this frame."""
        import ctypes as C

        from . import _lib
        st = C.c_void_p(self.info.stream if stream is None else stream)
        skip = self.dynamic_scene_this_frame and not self.denoise_while_moving
        if skip:
            if phase == "begin" and with_gi:
                self.submit_commands_gi_pathtrace(stream=stream)
            return False
        peers = _lib.StripPeers(up._ctx if up is not None else None, down._ctx if down is not None else None)
        if phase == "begin":
            self._plan = self.strip_plan(self.reset_history)
            self.reset_history = False
            c = self.global_constants() if with_gi else None
            self._check(self._lib.neb_strip_frame_begin(self._ctx, C.byref(c) if c is not None else None, C.byref(self._plan), C.byref(peers), st),
                        "neb_strip_frame_begin")
        else:
            self._check(self._lib.neb_strip_frame_finish(self._ctx, None, C.byref(self._plan), C.byref(peers), st), "neb_strip_frame_finish")
        return True

    def submit_strip_frame(self, with_gi=True, stream=None):
        """GI rows -> temporal rows -> exchange(s) -> levels as ONE library call (neb_strip_frame); the frame policy stays here.
        -> whether SVGF ran (src/DeferredRenderer.cpp:595)."""
        import ctypes as C
        st = C.c_void_p(self.info.stream if stream is None else stream)
        skip = self.dynamic_scene_this_frame and not self.denoise_while_moving
        if skip:
            if with_gi:
                self.submit_commands_gi_pathtrace(stream=stream)
            return False
        plan = self.strip_plan(self.reset_history)
        self.reset_history = False
        c = self.global_constants() if with_gi else None
        self._check(self._lib.neb_strip_frame(self._ctx, C.byref(c) if c is not None else None, self._comm, C.byref(plan), st), "neb_strip_frame")
        return True

    def _swap_rows_rccl_begin(self, planes, plan):
        """The same exchange through neb_strips_exchange, on a side stream so that work enqueued on the launch stream between
        begin and finish runs beside it."""
        import ctypes as C

        import torch

        from . import _lib
        if not plan:
            return lambda: None
        if self._xstream is None:
            self._xstream = torch.cuda.Stream()
        P = (_lib.HaloPlane * len(planes))(*[_lib.HaloPlane(p, sl) for p, sl in planes])
        S_ = (_lib.HaloSwap * len(plan))(*[_lib.HaloSwap(peer, s0, s1, r0, r1) for peer, (s0, s1), (r0, r1) in plan])
        launch = torch.cuda.ExternalStream(self.info.stream) if self.info.stream else torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(launch)
        self._xstream.wait_event(ready)
        self._check(self._lib.neb_strips_exchange(self._ctx, self._comm, P, len(planes), S_, len(plan), C.c_void_p(self._xstream.cuda_stream)),
                    "neb_strips_exchange")
        done = torch.cuda.Event()
        done.record(self._xstream)
        return lambda: launch.wait_event(done)

    def _plane_rows(self, plane, slot, row0, row1):
        key = (plane, slot)
        if key not in self._views:
            self._views[key] = self.svgf.plane_tensor(plane, slot)
        base = self.svgf.row_begin
        return self._views[key][row0 - base:row1 - base]

    def submit_commands_gi_pathtrace(self, rows=None, stream=None):
        super().submit_commands_gi_pathtrace(rows=self.part.gi_rows(self.rank) if rows is None else rows, stream=stream)

    def _staging_mode(self, sample):
        """How halo rows travel, decided ONCE and COLLECTIVELY (every rank must post the same operations):
        None = zero copy, rows go straight out of / into the context's planes (RCCL: the planes are ordinary hipMalloc'ed
        device memory); "host" = through host memory (a gloo group over GPU planes: the 1-GPU rehearsal);
        "device" = through torch-allocated device buffers (NEB_STRIPS_STAGING=device).  The choice is a pure function
        of the backend and of NEB_STRIPS_STAGING, and the ranks check with an all_reduce that they agree -- a rank
        never changes mode on its own (an exception while posting is raised, not retried: its peer would hang)."""
        if self._staging_decided:
            return self._staging
        import torch
        import torch.distributed as dist
        mode = self._staging
        if mode not in (None, "host", "device"):
            raise ValueError(f"NEB_STRIPS_STAGING={mode!r}: expected 'host' or 'device'")
        if mode is None and sample.is_cuda and dist.get_backend(self.group) == "gloo":
            mode = "host"
        code = {None: 0, "host": 1, "device": 2}[mode]
        on_cuda = sample.is_cuda and dist.get_backend(self.group) != "gloo"
        v = torch.tensor([code, -code], dtype=torch.int32, device=sample.device if on_cuda else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
        if int(v[0]) != code or int(v[1]) != -code:
            raise RuntimeError("strip ranks disagree on the halo staging mode (NEB_STRIPS_STAGING must be the same on every rank)")
        self._staging, self._staging_decided = mode, True
        return mode

    def _exchange_stream(self):
        """The exchange (P2P posts, staging copies, waits) must be ordered on the stream the kernels are enqueued on,
        not on whatever torch's current stream happens to be."""
        import contextlib
        import torch
        st = self.info.stream if self.info is not None else 0
        if not torch.cuda.is_available() or st == torch.cuda.current_stream().cuda_stream:
            return contextlib.nullcontext()
        return torch.cuda.stream(torch.cuda.ExternalStream(st))

    def _swap_rows_begin(self, planes, plan):
        """Starts one batched P2P exchange -- for every (plane, slot) and every (peer, send rows, recv rows) of the plan --
        and returns the function that completes it (makes the launch stream wait for the transfer).  Work enqueued on
        the launch stream between the two runs beside the transfer; it must not touch the rows being received."""
        import torch
        import torch.distributed as dist
        if not plan:
            return lambda: None
        if self._comm is not None:
            return self._swap_rows_rccl_begin(planes, plan)
        send = [self._plane_rows(p, sl, s0, s1) for p, sl in planes for _, (s0, s1), _ in plan]
        recv = [self._plane_rows(p, sl, r0, r1) for p, sl in planes for _, _, (r0, r1) in plan]
        peers = [peer for _ in planes for peer, _, _ in plan]
        mode = self._staging_mode(send[0])
        with self._exchange_stream():
            dst = recv
            if mode == "host":
                torch.cuda.current_stream().synchronize()
                send = [t.cpu() for t in send]
                dst = [torch.empty(t.shape, dtype=t.dtype) for t in recv]
            elif mode == "device":
                send = [t.clone() for t in send]
                dst = [torch.empty_like(t) for t in recv]
            ops = []
            for k, peer in enumerate(peers):
                ops.append(dist.P2POp(dist.isend, send[k], peer, group=self.group))
                ops.append(dist.P2POp(dist.irecv, dst[k], peer, group=self.group))
            works = dist.batch_isend_irecv(ops)

        def finish():
            with self._exchange_stream():
                for w in works:
                    w.wait()
                if mode is not None:
                    for d, s_ in zip(recv, dst):
                        d.copy_(s_)
        return finish

    def _swap_rows(self, planes, plan):
        self._swap_rows_begin(planes, plan)()

    def exchange_frame_halo_begin(self):
        """scheme "once": start swapping the boundary rows of the temporally accumulated radiance and of the variance plane"""
        cur = self.svgf.get_current_resource_index()
        return self._swap_rows_begin([(PLANE_RADIANCE, cur), (PLANE_VARIANCE, 0)], self.part.frame_exchange(self.rank))

    def exchange_halo(self, level):
        """scheme "per_level": swap the boundary rows of `level`'s source plane with the neighbouring strips"""
        (sp, ss), _ = self.svgf.atrous_level_planes(level)
        self._swap_rows([(sp, ss)], self.part.level_exchange(self.rank, level))

    def gather_frame(self, dst=0, plane=PLANE_RADIANCE, slot=None):
        """SURVEY.md 8e "final image gather": the owned rows of every strip to rank `dst` (display / tonemap / readback),
        one [H/N, W, C] block per rank over `torch.distributed.gather` (RCCL: point-to-point into `dst`, up to seven links
        in parallel).  Returns the full image tensor on `dst`, None elsewhere.  Not part of the timed frame: the path's
        output is the per-GPU strip, `bench.py --gather` reports the rate with it."""
        import torch
        import torch.distributed as dist
        if slot is None:
            slot = self.svgf.get_current_resource_index()
        own = self.part.owned(self.rank)
        mine = self._plane_rows(plane, slot, *own)
        if self.part.N == 1:
            return mine
        staged = mine.is_cuda and dist.get_backend(self.group) == "gloo"
        src = mine.cpu() if staged else mine.contiguous()
        out = [torch.empty_like(src) for _ in range(self.part.N)] if self.rank == dst else None
        dist.gather(src, out, dst=dst, group=self.group)
        return torch.cat(out, dim=0) if self.rank == dst else None

    def destroy(self):
        if self._comm is not None:
            self._lib.neb_strips_comm_destroy(self._comm)
            self._comm = None
        super().destroy()

    def submit_commands_svgf_denoising(self, events=None):
        if self.dynamic_scene_this_frame and not self.denoise_while_moving:  # src/DeferredRenderer.cpp:595
            return False
        st = self.info.stream
        own = self.part.owned(self.rank)
        if events is None and self.part.N > 1 and self._comm is not None:
            # the steady-state frame over RCCL: one library call for temporal rows -> exchange(s) -> levels (the sequence below, in C)
            return self.submit_strip_frame(with_gi=False)
        if self.reset_history:
            self.reset_history = False
            self.svgf.reset_history(st)
        if self.part.N == 1:
            # one strip = the whole frame: the two whole-frame calls of SubmitCommandsSVGFDenoising (DeferredRenderer.cpp:593-614),
            # which the library runs as one fused chain (the temporal call is held back: "t0".."t1" brackets nothing then,
            # and per-level times come from option svgf_profile / SVGFDenoiser.level_times)
            if events is not None:
                events["t0"].record()
            self.svgf.submit_temporal_accumulation(st)
            if events is not None:
                events["t1"].record()
            self.svgf.submit_atrous_compute_wavelet(st)
            if events is not None and "a1" in events:
                events["a1"].record()
            return True
        if events is not None:
            events["t0"].record()
        self.svgf.submit_temporal_accumulation(st, rows=self.part.gi_rows(self.rank))
        if events is not None:
            events["t1"].record()
        per_level = events.get("levels") if events is not None else None
        L = self.part.L
        once = self.part.N > 1 and self.part.scheme == "once"
        for level in range(L):
            rows = self.part.atrous_rows(self.rank, level)
            if self.part.level_exchange(self.rank, level):  # "per_level": every level; "overlap": the widest one
                self.exchange_halo(level)
            if per_level is not None:
                per_level[level][0].record()
            if once and level == 0:
                # the exchange runs beside level 0 on the rows that need none of the incoming halo (taps reach 2 rows)
                finish = self.exchange_frame_halo_begin()
                top = own[0] + 2 if self.rank > 0 else rows[0]
                bot = own[1] - 2 if self.rank < self.part.N - 1 else rows[1]
                if top < bot:
                    self.svgf.submit_atrous_level(0, (top, bot), st)
                finish()
                if rows[0] < top:
                    self.svgf.submit_atrous_level(0, (rows[0], top), st)
                if bot < rows[1]:
                    self.svgf.submit_atrous_level(0, (bot, rows[1]), st)
            else:
                self.svgf.submit_atrous_level(level, rows, st)
            if per_level is not None:
                per_level[level][1].record()
        if self.part.history_exchange(self.rank):  # "overlap": both copies of next frame's history must hold the FINAL rows
            (dp, ds) = self.svgf.atrous_level_planes(L - 1)[1] if L > 1 else (PLANE_RADIANCE, self.svgf.get_current_resource_index())
            self._swap_rows([(dp, ds)], self.part.history_exchange(self.rank))
        if events is not None and "a1" in events:
            events["a1"].record()  # (with "t1": brackets all levels with two events only)
        if L == 1:  # single level filters into the scratch plane: copy the owned rows back (api.hip: neb_svgf_atrous)
            cur = self.svgf.get_current_resource_index()
            with self._exchange_stream():
                self._plane_rows(PLANE_RADIANCE, cur, *own).copy_(self._plane_rows(PLANE_SCRATCH, 0, *own))
        return True

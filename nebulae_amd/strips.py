"""Multi-GPU screen-space sharding of the hot path: one process per GPU, row strips, RCCL halo rows.

The reference is single-GPU; this is new functionality whose oracle is
"N-GPU output == 1-GPU output, bit for bit".  Every stage of the path is per-pixel except the
a-trous wavelet, whose level l reads rows up to 2 * 2^l away (svgf_atrous.hlsl:51-65), so:

* rank r owns image rows [r*H/N, (r+1)*H/N) and keeps rows [own0 - h, own1 + h) resident,
  h = 2 * 2^(L-1) (the widest level's reach), clipped to the image; the G-buffer producer fills
  depth/normal for all resident rows (no exchange: variance is only read at the centre texel,
  quirk 4, and depth/normal halos come from the producer);
* GI, temporal accumulation and every a-trous level run on the OWNED rows only;
* before level l each rank swaps the 2 * 2^l boundary rows of that level's source radiance plane
  with its up/down neighbour -- contiguous row blocks sent straight out of the plane
  (``torch.distributed`` P2P, backend "nccl" = RCCL over xGMI; one direct link per neighbour);
* history needs no exchange: next frame's temporal pass reads history at owned rows only.

Per frame and direction that is sum_l 2*2^l rows (62 for L=5) of W*16 B.  The alternative of
SURVEY.md 8e (recompute the narrow levels on overlapped rows, exchange only the widest level plus a
history-consistency block) moves the same number of rows in 2 messages instead of L; it is not
implemented yet (DESIGN.md "Multi-GPU").

The compute backend is injected (``denoiser_factory``): the product uses the HIP ``SVGFDenoiser``;
the world_size-2 gloo tests inject an oracle-backed stand-in to exercise exactly this file on CPU.
"""
import math

from .renderer import DeferredRenderer
from .svgf import PLANE_RADIANCE, PLANE_SCRATCH, SVGFDenoiser


def frame_factors(n):
    """(a, b) with a*b == n, as square as possible, b >= a: the weak-scaling frame is (W*a) x (H*b)."""
    a = int(math.isqrt(n))
    while n % a:
        a -= 1
    return a, n // a


class StripPartition:
    def __init__(self, width, height, world, levels):
        if height % world:
            raise ValueError(f"image height {height} is not divisible by {world} strips")
        self.W, self.H, self.N, self.L = width, height, world, levels
        self.halo = 2 * (1 << (levels - 1)) if (world > 1 and levels > 0) else 0
        if world > 1 and height // world < self.halo:
            raise ValueError("strips are shorter than the a-trous reach; use fewer GPUs or fewer levels")

    def owned(self, r):
        h = self.H // self.N
        return r * h, (r + 1) * h

    def resident(self, r):
        a, b = self.owned(r)
        return max(0, a - self.halo), min(self.H, b + self.halo)

    def rows_temporal(self, r):
        a, b = self.owned(r)
        return b - a

    def atrous_rows(self, r, level):
        a, b = self.owned(r)
        return b - a

    def level_exchange(self, r, level):
        """[(peer, (send_row0, send_row1), (recv_row0, recv_row1))] for the source plane of `level`."""
        n = 2 * (1 << level)
        a, b = self.owned(r)
        out = []
        if r > 0:
            out.append((r - 1, (a, a + n), (a - n, a)))
        if r < self.N - 1:
            out.append((r + 1, (b - n, b), (b, b + n)))
        return out

    def exchanged_bytes_per_frame(self):
        """radiance bytes a middle rank sends per frame (both neighbours)."""
        return 2 * sum(2 * (1 << l) for l in range(self.L)) * self.W * 16 if self.N > 1 else 0


class StripRenderer(DeferredRenderer):
    """DeferredRenderer over one row strip.  With world == 1 it is exactly the single-GPU renderer."""

    def __init__(self, part, rank, device=0, group=None, denoiser_factory=SVGFDenoiser):
        super().__init__()
        self.svgf = denoiser_factory()
        self.part, self.rank, self.group = part, rank, group
        res0, res1 = part.resident(rank)
        self.init(part.W, part.H, atrous_levels=part.L, device=device, row_begin=res0, row_end=res1 if part.N > 1 else 0)
        self._views = {}
        import os
        self._staging = os.environ.get("NEB_STRIPS_STAGING") or None  # None | "host" | "device"

    def _plane_rows(self, plane, slot, row0, row1):
        key = (plane, slot)
        if key not in self._views:
            self._views[key] = self.svgf.plane_tensor(plane, slot)
        base = self.svgf.row_begin
        return self._views[key][row0 - base:row1 - base]

    def submit_commands_gi_pathtrace(self, rows=None, stream=None):
        super().submit_commands_gi_pathtrace(rows=self.part.owned(self.rank) if rows is None else rows, stream=stream)

    def exchange_halo(self, level):
        """Swap the boundary rows of `level`'s source plane with the neighbouring strips."""
        import torch.distributed as dist
        (sp, ss), _ = self.svgf.atrous_level_planes(level)
        plan = self.part.level_exchange(self.rank, level)
        if not plan:
            return
        import torch
        send = [self._plane_rows(sp, ss, s0, s1) for _, (s0, s1), _ in plan]
        recv = [self._plane_rows(sp, ss, r0, r1) for _, _, (r0, r1) in plan]
        # staging modes: None = zero copy (rows go straight out of / into the plane: the default on RCCL);
        # "host" = through host memory (gloo rehearsal on a box without RCCL peers);
        # "device" = through torch-allocated device buffers (NEB_STRIPS_STAGING=device: for RCCL builds that insist on
        # memory from the framework's allocator; the planes are hipMalloc'ed by the context and only viewed by torch)
        mode = self._staging
        if mode is None and send[0].is_cuda and dist.get_backend(self.group) == "gloo":
            mode = "host"
        dst = recv
        if mode == "host":
            torch.cuda.current_stream().synchronize()
            send = [t.cpu() for t in send]
            dst = [torch.empty(t.shape, dtype=t.dtype) for t in recv]
        elif mode == "device":
            send = [t.clone() for t in send]
            dst = [torch.empty_like(t) for t in recv]
        ops = []
        for k, (peer, _, _) in enumerate(plan):
            ops.append(dist.P2POp(dist.isend, send[k], peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, dst[k], peer, group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if mode is not None:
            for d, s_ in zip(recv, dst):
                d.copy_(s_)

    def submit_commands_svgf_denoising(self, events=None):
        if self.dynamic_scene_this_frame:  # src/DeferredRenderer.cpp:595
            return False
        st = self.info.stream
        own = self.part.owned(self.rank)
        if self.reset_history:
            self.reset_history = False
            self.svgf.reset_history(st)
        if events is not None:
            events["t0"].record()
        self.svgf.submit_temporal_accumulation(st, rows=own)
        if events is not None:
            events["t1"].record()
        L = self.part.L
        for level in range(L):
            if self.part.N > 1:
                self.exchange_halo(level)
            if events is not None:
                events["levels"][level][0].record()
            self.svgf.submit_atrous_level(level, own, st)
            if events is not None:
                events["levels"][level][1].record()
        if L == 1:  # single level filters into the scratch plane: copy the owned rows back (api.hip: neb_svgf_atrous)
            cur = self.svgf.get_current_resource_index()
            self._plane_rows(PLANE_RADIANCE, cur, *own).copy_(self._plane_rows(PLANE_SCRATCH, 0, *own))
        return True

"""N strip contexts of nebulae_amd.strips on ONE GPU, run in lock step with the RCCL exchange replaced by direct
device-to-device row copies between the contexts (the rendezvous itself is covered by the gloo tests in
test_strips_cpu.py and by the driver's multi-GPU run).  Test infrastructure."""
import numpy as np
import torch

from nebulae_amd import strips
from nebulae_amd.svgf import PLANE_RADIANCE, PLANE_SCRATCH, PLANE_VARIANCE


class LockstepStrips:
    def __init__(self, W, H, N, L, scheme="once"):
        self.part = strips.StripPartition(W, H, N, L, scheme=scheme)
        self.rs = [strips.StripRenderer(self.part, k) for k in range(N)]
        self.N, self.L, self.scheme = N, L, scheme

    def _pull(self, me, planes, plan):
        for p, sl in planes:
            for peer, _, (r0, r1) in plan:  # pull what the peer would send: its owned rows [r0, r1) of the same plane
                self.rs[me]._plane_rows(p, sl, r0, r1).copy_(self.rs[peer]._plane_rows(p, sl, r0, r1))

    def each(self, fn):
        for r in self.rs:
            fn(r)

    def denoise(self):
        """submit_commands_svgf_denoising of every strip, exchange emulated; -> the per-strip 'ran' flags"""
        part, rs, L = self.part, self.rs, self.L
        ran = []
        for r in rs:
            skip = r.dynamic_scene_this_frame and not r.denoise_while_moving  # src/DeferredRenderer.cpp:595
            ran.append(not skip)
        if not all(ran):
            assert not any(ran)
            return ran
        for r in rs:
            if r.reset_history:
                r.reset_history = False
                r.svgf.reset_history()
            r.svgf.submit_temporal_accumulation(rows=part.gi_rows(r.rank))
        torch.cuda.synchronize()
        if self.scheme == "once":
            for k, r in enumerate(rs):
                cur = r.svgf.get_current_resource_index()
                self._pull(k, [(PLANE_RADIANCE, cur), (PLANE_VARIANCE, 0)], part.frame_exchange(k))
        for level in range(L):
            torch.cuda.synchronize()
            for k, r in enumerate(rs):  # "per_level": every level; "overlap": the widest one
                (sp, ss), _ = r.svgf.atrous_level_planes(level)
                self._pull(k, [(sp, ss)], part.level_exchange(k, level))
            torch.cuda.synchronize()
            for r in rs:
                r.svgf.submit_atrous_level(level, part.atrous_rows(r.rank, level))
        torch.cuda.synchronize()
        for k, r in enumerate(rs):  # "overlap": the final rows inside the neighbours' bands (next frame's history)
            (dp, ds) = r.svgf.atrous_level_planes(L - 1)[1] if L > 1 else (PLANE_RADIANCE, r.svgf.get_current_resource_index())
            self._pull(k, [(dp, ds)], part.history_exchange(k))
        if L == 1:
            for r in rs:
                cur = r.svgf.get_current_resource_index()
                own = part.owned(r.rank)
                r._plane_rows(PLANE_RADIANCE, cur, *own).copy_(r._plane_rows(PLANE_SCRATCH, 0, *own))
        torch.cuda.synchronize()
        return ran

    def image(self, plane=PLANE_RADIANCE):
        part = self.part
        return np.concatenate([r.svgf.download(plane, row0=part.owned(r.rank)[0], nrows=part.owned(r.rank)[1] - part.owned(r.rank)[0])
                               for r in self.rs], axis=0)

    def destroy(self):
        for r in self.rs:
            r.destroy()

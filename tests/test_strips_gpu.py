"""GPU test of the row-strip path with the HIP backend: two strip contexts on ONE GPU, the RCCL exchange
replaced by a direct device-to-device row copy between them (the rendezvous itself is covered by the gloo
tests in test_strips_cpu.py).  GI + SVGF on strips must equal the full-image run bit for bit."""
import numpy as np
import pytest
import torch

from nebulae_amd import scene as S
from nebulae_amd import strips
from nebulae_amd.renderer import RenderInfo
from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE, PLANE_VARIANCE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scheme", ["once", "per_level"])
def test_two_strips_equal_full_image_gi_plus_svgf(scheme):
    W, H, L, N = 256, 192, 5, 2
    sc = S.atrium_standin(target_triangles=20000, n_submeshes=40, tex_size=32)
    cam = S.sponza_camera()
    full = strips.StripRenderer(strips.StripPartition(W, H, 1, L), 0)
    part = strips.StripPartition(W, H, N, L, scheme=scheme)
    rs = [strips.StripRenderer(part, k) for k in range(N)]

    def pull(me, planes, plan):
        for p, sl in planes:
            for peer, _, (r0, r1) in plan:
                # pull what the peer would send: its owned rows [r0, r1) of the same plane
                rs[me]._plane_rows(p, sl, r0, r1).copy_(rs[peer]._plane_rows(p, sl, r0, r1))

    for f in range(1, 6):
        info = RenderInfo(scene=sc, camera=cam, frame_index=f)
        for r in [full] + rs:
            r.begin_frame(info)
            r.submit_commands_gbuffer()
            cur = r.svgf.get_current_resource_index()
            r.svgf.plane_tensor(PLANE_RADIANCE, cur).zero_()
            r.submit_commands_gi_pathtrace()
        torch.cuda.synchronize()
        ran_full = full.submit_commands_svgf_denoising()
        # strips: run the denoiser in lock-step so the emulated exchange sees finished rows
        if not rs[0].dynamic_scene_this_frame:
            for r in rs:
                if r.reset_history:
                    r.reset_history = False
                    r.svgf.reset_history()
                r.svgf.submit_temporal_accumulation(rows=part.owned(r.rank))
            torch.cuda.synchronize()
            if scheme == "once":
                for k, r in enumerate(rs):
                    cur = r.svgf.get_current_resource_index()
                    pull(k, [(PLANE_RADIANCE, cur), (PLANE_VARIANCE, 0)], part.frame_exchange(k))
            for level in range(L):
                torch.cuda.synchronize()
                if scheme == "per_level":
                    for k, r in enumerate(rs):
                        (sp, ss), _ = r.svgf.atrous_level_planes(level)
                        pull(k, [(sp, ss)], part.level_exchange(k, level))
                torch.cuda.synchronize()
                for r in rs:
                    r.svgf.submit_atrous_level(level, part.atrous_rows(r.rank, level))
            ran = [True] * N
        else:
            ran = [False] * N
        assert all(x == ran_full for x in ran)
        torch.cuda.synchronize()
    want = full.svgf.download(PLANE_RADIANCE)
    got = np.concatenate([r.svgf.download(PLANE_RADIANCE, row0=part.owned(r.rank)[0],
                                          nrows=part.owned(r.rank)[1] - part.owned(r.rank)[0]) for r in rs], axis=0)
    assert np.isfinite(want).all() and float(np.abs(want[..., :3]).max()) > 0.0
    assert np.array_equal(got, want)
    for r in [full] + rs:
        r.destroy()

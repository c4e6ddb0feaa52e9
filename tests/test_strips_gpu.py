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


@pytest.mark.parametrize("scheme", ["once", "per_level"])
def test_two_strips_equal_full_image_gi_plus_svgf(scheme):
    W, H, L, N = 256, 192, 5, 2
    sc = S.atrium_standin(target_triangles=20000, n_submeshes=40, tex_size=32)
    cam = S.sponza_camera()
    full = strips.StripRenderer(strips.StripPartition(W, H, 1, L), 0)
    ls = LockstepStrips(W, H, N, L, scheme)
    for f in range(1, 6):
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

"""Diagnostics: what each halo-exchange scheme of nebulae_amd/strips.py costs in COMPUTE, measured on ONE GPU: the N strip
contexts of a frame run one after the other on the device with the exchange replaced by device-to-device row copies
(tests/strip_harness.py), so the figure is the sum over strips of GI + SVGF kernel time -- the redundant work of "once"
(extra a-trous row-levels) and of "overlap" (GI + temporal + narrow levels on the overlap rows) shows, exchange latency does
not (that needs N GPUs).  usage (GPU box): python tools/strip_schemes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from nebulae_amd import scene as S  # noqa: E402
from nebulae_amd.renderer import RenderInfo  # noqa: E402
from nebulae_amd.svgf import PLANE_RADIANCE  # noqa: E402
from strip_harness import LockstepStrips  # noqa: E402

sc, cam = S.atrium_standin(), S.sponza_camera()
for (W, H, N) in ((1920, 1080, 8), (1920, 1080, 4), (3840, 2160, 4), (3840, 2160, 8)):
    line = []
    for scheme in ("once", "per_level", "overlap"):
        ls = LockstepStrips(W, H, N, 5, scheme)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_gi, t_svgf = [], []
        for f in range(1, 10):
            info = RenderInfo(scene=sc, camera=cam, frame_index=f)
            for r in ls.rs:
                r.begin_frame(info)
                if f == 1:
                    r.submit_commands_gbuffer()
                    cur = r.svgf.get_current_resource_index()
                    r.svgf.plane_tensor(1, cur ^ 1).copy_(r.svgf.plane_tensor(1, cur))  # normal / depth of the other slot: static camera
                    r.svgf.plane_tensor(2, cur ^ 1).copy_(r.svgf.plane_tensor(2, cur))
                r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).zero_()
            torch.cuda.synchronize()
            e[0].record()
            for r in ls.rs:
                r.submit_commands_gi_pathtrace()
            e[1].record()
            ls.denoise()
            e[2].record()
            torch.cuda.synchronize()
            if f >= 4:
                t_gi.append(e[0].elapsed_time(e[1]) * 1e3)
                t_svgf.append(e[1].elapsed_time(e[2]) * 1e3)
        ls.destroy()
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        line.append(f"{scheme}: GI {med(t_gi):.0f} + SVGF {med(t_svgf):.0f} = {med(t_gi) + med(t_svgf):.0f} us")
    print(f"{W}x{H} in {N} strips (sum over the strips, one device): " + " | ".join(line), flush=True)

"""Cameras the bench never uses: the sky only (every G-buffer pixel a miss: worldPos 0, normal (0, 0, 1), albedo 0 -- the rays still fly), far
away, nose against a wall, inside a wall.  GI dispatch time with the table on and off, finite output, same bits.   python tools/odd_cameras.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nebulae_amd import scene as S  # noqa: E402
from nebulae_amd.renderer import DeferredRenderer, RenderInfo  # noqa: E402
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT  # noqa: E402

W, H = 1920, 1080
sc, base = S.atrium_standin(), S.sponza_camera()


def cam(eye, target):
    c = S.moved_camera(base)
    c.eye[:] = eye
    c.target[:] = target
    return c


e = [float(x) for x in base.eye]
cams = {"bench": base,
        "sky only": cam((e[0], e[1] + 200.0, e[2]), (e[0], e[1] + 400.0, e[2] + 1.0)),
        "far away (the scene a few pixels)": cam((e[0] + 3000.0, e[1] + 1000.0, e[2]), (e[0], e[1], e[2])),
        "nose against the floor": cam((e[0], -0.999 * abs(e[1]) if e[1] < 0 else 0.0005, e[2]), (e[0] + 0.3, -10.0, e[2] + 0.3)),
        "straight up from the court": cam((e[0], e[1], e[2]), (e[0], e[1] + 100.0, e[2] + 0.01))}
rs = {}
for table in (1, 0):
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=5)
    r.begin_frame(RenderInfo(scene=sc, camera=base, frame_index=1))
    r.svgf.set_option("gi_sun_table", table)
    rs[table] = r
for name, c in cams.items():
    res = {}
    for table, r in rs.items():
        t = []
        for f in range(2, 8):
            r.begin_frame(RenderInfo(scene=sc, camera=c, frame_index=f))
            r.submit_commands_gbuffer()
            r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
            r.ray_count(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r.submit_commands_gi_pathtrace()
            torch.cuda.synchronize()
            t.append((time.perf_counter() - t0) * 1e6)
        res[table] = (r.svgf.download(PLANE_RADIANCE), r.ray_count(), float(np.median(t[2:])))
    (a, rays, us_on), (b, rays_b, us_off) = res[1], res[0]
    print(f"{name}: {rays} queries; GI dispatch {us_on:.0f} us with the table, {us_off:.0f} without; finite {bool(np.isfinite(a).all())}; "
          f"same bits {bool(np.array_equal(a, b)) and rays == rays_b}", flush=True)

"""Which shadow rays does the sun table leave, and how long are they?  Node visits per shadow ray (any-hit walk) of the bench
frame with the table on: all traced rays, split by outcome; the longest ray bounds the list-walking pass (one wave cannot be
shorter than its longest ray).   python tools/shadow_tail_stats.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nebulae_amd import scene as S  # noqa: E402
from nebulae_amd.renderer import DeferredRenderer, RenderInfo  # noqa: E402
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT  # noqa: E402

W, H = 1920, 1080
sc, cam = S.atrium_standin(), S.sponza_camera()
for mode in (1, 0):
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=5)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
    r.svgf.set_option("gi_sun_table", mode)
    r.submit_commands_gbuffer()
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.set_debug_hits(True)
    r.ray_count(reset=True)
    r.submit_commands_gi_pathtrace()
    hits = r.download_hits()
    rays = r.ray_count()
    st = r.sun_table_stats()
    visits = (hits["flags"] >> 20).astype(np.int64)
    traced = visits > 0
    vis = (hits["flags"] & 1).astype(bool)
    print(f"gi_sun_table={mode}: {rays} rays, {st['rays_answered']} answered by the table; shadow rays walked: {int(traced.sum())}")
    for label, m in (("all walked", traced), ("  unoccluded", traced & vis), ("  occluded", traced & ~vis)):
        v = visits[m]
        if v.size:
            print(f"  {label:12s} n={v.size:8d}  node visits mean {v.mean():6.1f}  p50 {np.percentile(v, 50):5.0f}  p90 {np.percentile(v, 90):5.0f}  "
                  f"p99 {np.percentile(v, 99):5.0f}  p99.9 {np.percentile(v, 99.9):5.0f}  max {v.max():5d}")
    r.destroy()

"""Does the ORDER in which the closest-hit pass takes its tiles matter?  The launch ends when its last wave does; a wave that starts late and is long (an interior
tile: 40+ iterations) runs on a draining chip.  Per-tile cost of the bench frame (wave iterations = the longest ray of the tile, from the debug hits), then the
GI dispatch timed with (a) the default XCD-aware order, (b) tiles sorted by that cost, longest first, dealt round-robin to the XCDs, (c) shortest first (the worst
case), (d) a random permutation.  python tools/gi_tile_order.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np
import torch

from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE, SLOT_CURRENT

W, H = 1920, 1080
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer()
r.init(W, H, atrous_levels=5)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
r.submit_commands_gbuffer()
torch.cuda.synchronize()
for pl in (PLANE_NORMAL, PLANE_DEPTH):
    r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
r.set_debug_hits(True)
r.ray_count(reset=True)
r.submit_commands_gi_pathtrace()
r.ray_count()
hits = r.download_hits()
r.set_debug_hits(False)
it = ((hits["flags"] >> 8) & 0xFFF).astype(np.int64).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
cost = it.max(axis=1)  # wave iterations of tile t (tile = ty * tiles_x + tx)
n = len(cost)
print(f"{n} tiles; wave iterations mean {cost.mean():.1f}, p50 {np.percentile(cost, 50):.0f}, p90 {np.percentile(cost, 90):.0f}, p99 {np.percentile(cost, 99):.0f}, max {cost.max()}")
orders = {"default": None,
          "longest first": np.argsort(-cost, kind="stable").astype(np.uint32),
          "shortest first": np.argsort(cost, kind="stable").astype(np.uint32),
          "random": np.random.default_rng(1).permutation(n).astype(np.uint32)}


def timed(frames=40):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frames)]
    for f in range(frames):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=10 + f))
        ev[f][0].record()
        r.submit_commands_gi_pathtrace()
        ev[f][1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return float(np.mean(t[4:-4]))


for rep in range(2):
    for name, order in orders.items():
        rc = r._lib.neb_gi_debug_set_tile_order(r._ctx, order.ctypes.data_as(C.c_void_p) if order is not None else None, n if order is not None else 0)
        assert rc == 0
        timed(8)
        print(f"  {name:15s}: GI dispatch {timed():.1f} us", flush=True)
r.destroy()

"""How many of a closest-hit wave's loop iterations run with few lanes live?  Per-ray iteration counts of the bench frame (debug hits), grouped by wave
(8x8 pixel tile): for every iteration k of a wave, live(k) = rays of the tile that need more than k iterations.  python tools/gi_tail_lanes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT

W, H = 1920, 1080
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer()
r.init(W, H, atrous_levels=5)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
r.submit_commands_gbuffer()
r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
r.set_debug_hits(True)
r.ray_count(reset=True)
r.submit_commands_gi_pathtrace()
r.ray_count()
hits = r.download_hits()
it = ((hits["flags"] >> 8) & 0xFFF).astype(np.int64).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)  # [tile, lane]
it.sort(axis=1)
wave_iters = it[:, -1]
total = int(wave_iters.sum())
print(f"{len(it)} waves, {total / len(it):.1f} iterations per wave (a ray needs {it.mean():.1f})")
# live lanes at iteration k of a wave = 64 - (number of rays with iters <= k)
kmax = int(wave_iters.max())
hist = np.zeros(65, np.int64)
for k in range(kmax):
    running = wave_iters > k
    live = (it[running] > k).sum(axis=1)
    hist += np.bincount(live, minlength=65)
cum = np.cumsum(hist)
for t in (4, 8, 16, 24, 32):
    print(f"  iterations with <= {t:2d} lanes live: {cum[t] / total:.3f} of all wave-iterations")
print(f"  mean live lanes {float((hist * np.arange(65)).sum()) / total:.1f}")
r.destroy()

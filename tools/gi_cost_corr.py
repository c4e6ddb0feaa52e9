"""Diagnostics: how well does a pixel's traversal cost in one frame predict its cost in the next (same origin, new random direction)?
And what would reordering rays inside 32x32 blocks by the PREVIOUS frame's cost buy?  (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
W, H = 1920, 1080
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer(); r.init(W, H, atrous_levels=5)
its = []
for f in (1, 2):
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
    r.submit_commands_gbuffer()
    r.set_debug_hits(True); r.submit_commands_gi_pathtrace()
    its.append((r.download_hits()["flags"] >> 8).astype(np.int64))
    r.end_frame()
a, b = its
print("per-pixel cost correlation between two frames: %.3f" % np.corrcoef(a.reshape(-1), b.reshape(-1))[0, 1])
def blocks(x): return x[: (H // 32) * 32, : (W // 32) * 32].reshape(H // 32, 32, W // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 1024)
def tiles(x): return x[: (H // 8) * 8, : (W // 8) * 8].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
base = tiles(b).max(1).sum()
A, B = blocks(a), blocks(b)
order = np.argsort(A, axis=1, kind="stable")
pred = np.take_along_axis(B, order, axis=1).reshape(-1, 64).max(1).sum() * (tiles(b).size / B.size)
best = np.sort(B, axis=1).reshape(-1, 64).max(1).sum() * (tiles(b).size / B.size)
print("wave-iterations of frame 2: 8x8 tiles %.3e | 32x32 blocks ordered by frame 1's cost %.3e (x%.2f) | by its own cost %.3e (x%.2f)" % (base, pred, base / pred, best, base / best))

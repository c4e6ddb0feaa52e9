import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from nebulae_amd.svgf import *
from svgf_cases import frame_inputs
W, H, L = 128, 96, 5
def make():
    d = SVGFDenoiser(); d.init(W, H, atrous_levels=L); return d
a, b = make(), make()
g, rad = frame_inputs(W, H, 2, None)
for d in (a, b):
    d.begin_frame(2)
    d.upload(PLANE_DEPTH, SLOT_CURRENT, g["depth"]); d.upload(PLANE_NORMAL, SLOT_CURRENT, g["normal"])
    d.upload(PLANE_VARIANCE, 0, np.full((H, W), 0.1, np.float16))
for lvl in range(L):
    (sp, ss), (dp, ds) = a.atrous_level_planes(lvl)
    for d in (a, b):
        d.upload(sp, ss, rad)
        d.upload(dp, ds, np.zeros_like(rad))
    a.submit_atrous_level(lvl, (0, H))
    for r in ((0, 17), (17, 64), (64, H)):
        b.submit_atrous_level(lvl, r)
    x, y = a.download(dp, ds), b.download(dp, ds)
    diff = np.abs(x - y).max(axis=2)
    ys, xs = np.nonzero(diff)
    print("level", lvl, "maxdiff", diff.max(), "n", len(ys), "rows", sorted(set(ys.tolist()))[:20], "cols", sorted(set(xs.tolist()))[:10])
    # repeat the full run twice for determinism
    a.upload(dp, ds, np.zeros_like(rad)); a.submit_atrous_level(lvl, (0, H)); x2 = a.download(dp, ds)
    print("   full-vs-full equal:", np.array_equal(x, x2))

"""How long are the sun-table build's walks?  NEB_SUN_WALK_STATS=1 python tools/sun_table_walks.py [long_thin]"""
import ctypes as C
import os
import sys

os.environ["NEB_SUN_WALK_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT

W, H = 640, 360
sc, cam = S.atrium_standin(long_thin=len(sys.argv) > 1 and sys.argv[1] == "long_thin"), S.sponza_camera()
r = DeferredRenderer()
r.init(W, H, atrous_levels=5)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
r.submit_commands_gbuffer()
r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
r.submit_commands_gi_pathtrace()  # (builds the table)
out = (C.c_uint64 * 12)()
assert r._lib.neb_gi_debug_sun_walk_stats(r._ctx, out) == 0
n = sc.num_triangles
for p, name in ((0, "lit pass"), (1, "hint pass")):
    v, mx, tested, ticks, longest, waves = (int(out[6 * p + k]) for k in range(6))
    print(f"{name}: {v} node visits = {v / n:.1f} per triangle of the scene, longest walk {mx}; {tested} candidate triangles tested = {tested / n:.1f} per triangle; "
          f"{waves} waves, {ticks / 100 / max(waves, 1):.1f} us per wave, longest {longest / 100:.1f} us, sum {ticks / 1e5:.1f} wave-ms")
print(r.sun_table_stats())
r.destroy()

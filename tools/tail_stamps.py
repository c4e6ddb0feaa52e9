"""Per-wave stamps of gi_shadow_list_kernel (a -DNEB_TAIL_STAMPS=1 build selected with NEB_LIB_PATH): when each wave starts and ends
(s_memrealtime, 100 MHz), how many rays it walked and its longest ray.  python tools/tail_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nebulae_amd import _lib, scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo

W, H = 1920, 1080
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer()
r.init(W, H, atrous_levels=5)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
r.submit_commands_gbuffer()
for f in range(4, 12):
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
    r.submit_commands_gi_pathtrace()
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_uint64 * (8192 * 4))()
assert lib.neb_debug_tail_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
t0 = st[:, 0].min()
start, end, rays, visits = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, st[:, 2], st[:, 3]
work = rays > 0
print(f"waves {len(st)}, with rays {int(work.sum())}; rays {int(rays.sum())}")
print(f"kernel span by stamps: {end.max():.1f} us; empty waves: start p50 {np.percentile(start[~work], 50):.1f} max {start[~work].max():.1f} us")
d = end - start
for label, m in (("working waves", work),):
    print(f"{label}: start p50 {np.percentile(start[m], 50):.1f} p99 {np.percentile(start[m], 99):.1f} max {start[m].max():.1f} us; duration p50 {np.percentile(d[m], 50):.1f} "
          f"p90 {np.percentile(d[m], 90):.1f} p99 {np.percentile(d[m], 99):.1f} max {d[m].max():.1f} us; end p50 {np.percentile(end[m], 50):.1f} max {end[m].max():.1f} us")
    print(f"  longest ray of a wave: p50 {np.percentile(visits[m], 50):.0f} max {visits[m].max()} node visits; us per node visit of the longest ray: p50 {np.percentile(d[m] / np.maximum(visits[m], 1), 50):.2f}")
order = np.argsort(-d)[:8]
for k in order:
    print(f"  wave {k}: start {start[k]:.1f} dur {d[k]:.1f} rays {rays[k]} max visits {visits[k]}")

"""How a freshly created context approaches its steady frame time: per-frame GPU time (events) of the bench frame, frames 2..120.
python tools/frame_ramp.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

args = bench.parse(["--cpu-frames", "0"])
rt = bench.GpuRuntime()
rt.set_device(0)
from nebulae_amd import scene as S
sc, cam = S.atrium_standin(), S.sponza_camera()
w = bench.Workload(rt, args, 1920, 1080, 5, 1, sc, cam, 0, 1, 0, None)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(121)]
ev[0].record()
for k in range(120):
    w.step()
    ev[k + 1].record()
torch.cuda.synchronize()
t = np.array([ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(120)])
for a in range(0, 120, 10):
    print(f"frames {a + 2:3d}..{a + 11:3d}: " + " ".join(f"{x:6.0f}" for x in t[a:a + 10]), "us")

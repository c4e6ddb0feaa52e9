"""Scenes eight and ten times the bench's (2.0 and 2.6 M triangles; hints index triangles with 23 bits, 8.4 M -- round 4's 21 bits lost nearly every occluder at 2.6 M): BVH build, sun-table build,
a few 1080p frames with the table on and off -- same bits -- and their times.   python tools/big_scene.py [triangles [long_thin]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nebulae_amd import scene as S  # noqa: E402
from nebulae_amd.renderer import DeferredRenderer, RenderInfo  # noqa: E402
from nebulae_amd.svgf import PLANE_RADIANCE  # noqa: E402

W, H = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
long_thin = len(sys.argv) > 2 and sys.argv[2] == "long_thin"  # (the real Sponza's pathology: full-length strips and beams, aspect ratios up to 190 : 1)
sc, cam = S.atrium_standin(target_triangles=n, tex_size=256, long_thin=long_thin), S.sponza_camera()
out = {}
for table in (1, 0):
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=5)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    r.svgf.set_option("gi_sun_table", table)
    t = []
    for f in range(1, 12):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
        r.submit_commands_gbuffer()
        r.submit_commands_pbr_lighting()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.submit_commands_gi_pathtrace()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) * 1e6)
        r.submit_commands_svgf_denoising()
        r.end_frame()
    out[table] = r.svgf.download(PLANE_RADIANCE)
    r.ray_count(reset=True)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=12))
    r.submit_commands_gi_pathtrace()
    rays = r.ray_count()
    st = r.sun_table_stats()
    st["queries_last_frame"] = rays
    print(f"{sc.num_triangles} triangles, table {table}: BVH {r.scene_info()[1]} nodes, depth {r.bvh_depth()}, build {r.build_ms():.1f} ms; sun table {st}, "
          f"build {r.sun_table_build_ms()} ms; GI dispatch (host-timed, synchronised) frames 5-11: {np.median(t[4:]):.0f} us", flush=True)
    r.destroy()
print("table on == off:", bool(np.array_equal(out[1], out[0])))

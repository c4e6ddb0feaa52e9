"""One-time costs on the bench scene: scene upload + texture tables + BVH build (wall), the BVH build alone (the library's own
timer) and the tree it gives.  python tools/time_build.py   (NEB_LIB_PATH selects an A/B build, e.g. -DNEB_SAH_BIG=0)"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from nebulae_amd import scene as S  # noqa: E402
from nebulae_amd.renderer import DeferredRenderer, RenderInfo  # noqa: E402

for long_thin in (False, True):
    sc, cam = S.atrium_standin(long_thin=long_thin), S.sponza_camera()
    for rep in range(2):
        r = DeferredRenderer()
        r.init(256, 144, atrous_levels=1)
        t0 = time.time()
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
        torch.cuda.synchronize()
        print(f"[{sc.name}] set_scene + build_bvh: {(time.time() - t0) * 1e3:.0f} ms; BVH build {r.build_ms():.2f} ms in {r.build_passes()} SAH passes; "
              f"triangles / BVH4 nodes {r.scene_info()}, depth {r.bvh_depth()}", flush=True)
        r.destroy()

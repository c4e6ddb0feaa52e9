import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer(); r.init(256, 144, atrous_levels=1)
t0 = time.time()
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
torch.cuda.synchronize()
print("set_scene + build_bvh: %.0f ms" % ((time.time() - t0) * 1e3), r.scene_info())

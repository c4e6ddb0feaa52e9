"""One-off check at the bench size: GI of the HIP path against the oracle on sponza-standin 1920x1080 (run on the GPU box)."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT
from oracle_lib import OracleTracer
from test_gi_gpu import upload_gbuffer
W, H = 1920, 1080
sc, cam = S.atrium_standin(), S.sponza_camera()
o = OracleTracer(sc); gb = o.gbuffer(W, H, cam)
r = DeferredRenderer(); r.init(W, H, atrous_levels=5)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3)); upload_gbuffer(r, gb)
base = np.full((H, W, 4), 0.25, np.float32); r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, base)
r.set_debug_hits(True); r.submit_commands_gi_pathtrace()
got = r.svgf.download(PLANE_RADIANCE); hits = r.download_hits()
want, ohits, orays = o.gi(gb, r.global_constants(), radiance=base.copy())
idm = (hits["geometry"] != ohits["geometry"]) | (hits["primitive"] != ohits["primitive"])
fl = ((hits["flags"] & 1) != (ohits["flags"] & 1)) & ~idm
same = ~idm & ~fl
num = np.sqrt(((got[same][:, :3] - want[same][:, :3]) ** 2).sum()); den = np.sqrt((want[same][:, :3] ** 2).sum())
dt = np.abs(hits["t"][idm] - ohits["t"][idm])
print(f"1080p: bounce-hit id mismatches {idm.sum()} of {W*H} (max |dt| {dt.max() if dt.size else 0:.3g}), shadow-flag-only mismatches {fl.sum()}, "
      f"radiance rel-L2 where hits agree {num/den:.3e}, whole image {np.sqrt(((got[...,:3]-want[...,:3])**2).sum())/np.sqrt((want[...,:3]**2).sum()):.3e}")
ys, xs = np.nonzero(idm)
for y, x in zip(ys, xs):
    print(y, x, "gpu t %.6f geom %d prim %d | oracle t %.6f geom %d prim %d" % (hits["t"][y, x], hits["geometry"][y, x], hits["primitive"][y, x], ohits["t"][y, x], ohits["geometry"][y, x], ohits["primitive"][y, x]))

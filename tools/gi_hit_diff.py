"""Diagnostics: per-pixel differences between the HIP tracer's bounce hits and the oracle's on the small atrium (run on the GPU box).
A closer oracle hit means the GPU traversal lost a triangle (that is how a bug in the BVH relinking was found)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT
from oracle_lib import OracleTracer
from test_gi_gpu import upload_gbuffer
sc = S.atrium_standin(target_triangles=30000, n_submeshes=60, tex_size=64); cam = S.sponza_camera(); W, H = 320, 184
o = OracleTracer(sc); gb = o.gbuffer(W, H, cam)
r = DeferredRenderer(); r.init(W, H, atrous_levels=4)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=5)); upload_gbuffer(r, gb)
base = np.full((H, W, 4), 0.25, np.float32); r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, base)
r.set_debug_hits(True); r.svgf.set_option("gi_sort_rays", 0); r.submit_commands_gi_pathtrace()
hits = r.download_hits(); want, ohits, _ = o.gi(gb, r.global_constants(), radiance=base.copy())
idm = (hits["geometry"] != ohits["geometry"]) | (hits["primitive"] != ohits["primitive"])
fl = (hits["flags"] & 1) != (ohits["flags"] & 1)
print("id mismatches", idm.sum(), "shadow-flag-only mismatches", (fl & ~idm).sum())
dt = np.abs(hits["t"][idm] - ohits["t"][idm])
print("t differences at id mismatches:", np.sort(dt)[-10:], "exactly equal:", (dt == 0).sum())
ys, xs = np.nonzero(idm)
for y, x in list(zip(ys, xs))[:15]:
    print(y, x, "gpu t %.5f geom %d prim %d | oracle t %.5f geom %d prim %d" % (hits["t"][y, x], hits["geometry"][y, x], hits["primitive"][y, x], ohits["t"][y, x], ohits["geometry"][y, x], ohits["primitive"][y, x]))

"""What the few pixels are made of in which the HIP path and the CPU oracle disagree at a given frame size: closest-hit ties (another
triangle, |dt| ~ 1e-6), sun-visibility flips (the same hit, the shadow ray grazing a silhouette), and what each weighs in the frame's
L2 norm -- with the default (1-ulp hardware) shading arithmetic and with the oracle's ("gi_exact_shade").
python tools/mismatch_kinds.py [W H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT
from oracle_lib import OracleTracer, oracle_pbr_direct
from test_gi_gpu import upload_gbuffer

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
sc, cam = S.atrium_standin(), S.sponza_camera()
o = OracleTracer(sc)
gb = o.gbuffer(W, H, cam)
for exact in (0, 1):
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=5)
    for f in (1, 2, 3, 4, 5):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
        if f == 1:
            r.set_debug_hits(True)
            r.svgf.set_option("gi_exact_shade", exact)
        upload_gbuffer(r, gb)
        # the direct term first (row f1: its own sun shadow ray per pixel), as the frame has it
        r.submit_commands_pbr_lighting()
        direct = r.svgf.download(PLANE_RADIANCE)
        odirect, _ = oracle_pbr_direct(o, gb, r.global_constants())
        dd = direct[..., :3] - odirect[..., :3]
        lit_flip = (direct[..., 0] > 0) != (odirect[..., 0] > 0)
        print(f"exact_shade={exact} frame {f}: direct term: {int(lit_flip.sum())} px lit on one side only, whole image {np.linalg.norm(dd) / max(np.linalg.norm(odirect[..., :3]), 1e-20):.2e}")
        r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
        r.submit_commands_gi_pathtrace()
        got, hits = r.svgf.download(PLANE_RADIANCE), r.download_hits()
        want, ohits, _ = o.gi(gb, r.global_constants())
        tie = (hits["geometry"] != ohits["geometry"]) | (hits["primitive"] != ohits["primitive"])
        flip = ~tie & ((hits["flags"] & 1) != (ohits["flags"] & 1))
        d = got[..., :3] - want[..., :3]
        n = np.linalg.norm(want[..., :3])
        print(f"exact_shade={exact} frame {f}: other triangle {int(tie.sum())} px (weight {np.linalg.norm(d[tie]) / n:.2e}, max |dt|/t "
              f"{(np.abs(hits['t'] - ohits['t'])[tie] / np.maximum(ohits['t'][tie], 1e-6)).max() if tie.any() else 0:.1e}), visibility flips {int(flip.sum())} px "
              f"(weight {np.linalg.norm(d[flip]) / n:.2e}), the rest {np.linalg.norm(d[~tie & ~flip]) / n:.2e}; whole image {np.linalg.norm(d) / n:.2e}", flush=True)
    r.destroy()

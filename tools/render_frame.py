"""Render sponza-standin through the whole path on the GPU and write PNGs (noisy 1-spp frame and denoised frame)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_LDR, PLANE_RADIANCE

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 540)
L = 5
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer()
r.init(W, H, atrous_levels=L)
for f in range(1, 12):
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
    r.submit_commands_gbuffer()
    r.submit_commands_pbr_lighting()
    r.submit_commands_gi_pathtrace()
    if f == 11:
        r.submit_commands_hdr_tonemapping()
        noisy = r.svgf.download(PLANE_LDR).view(np.uint8).reshape(H, W, 4)[..., :3].copy()
    r.submit_commands_svgf_denoising()
    r.end_frame()
r.submit_commands_hdr_tonemapping()
den = r.svgf.download(PLANE_LDR).view(np.uint8).reshape(H, W, 4)[..., :3]
os.makedirs(out, exist_ok=True)
Image.fromarray(noisy).save(os.path.join(out, "frame_noisy.png"))
Image.fromarray(den).save(os.path.join(out, "frame_denoised.png"))
print("wrote", out, "mean noisy", noisy.mean(), "mean denoised", den.mean())

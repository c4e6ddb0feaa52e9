"""Diagnostics: traversal steps per ray on sponza-standin at 1080p (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
W, H = 1920, 1080
sc = S.atrium_standin()
cam = S.sponza_camera()
r = DeferredRenderer(); r.init(W, H, atrous_levels=5)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
r.submit_commands_gbuffer()
r.set_debug_hits(True)
r.ray_count(reset=True)
r.submit_commands_gi_pathtrace()
rays = r.ray_count()
st = r.traversal_stats()
px = W * H
hits = r.download_hits()
nb = px; ns = int((hits["t"] > 0).sum())
print("tris/nodes", r.scene_info(), "rays", rays, st)
print("bounce: nodes/ray %.1f tris/ray %.1f | shadow (%d rays): nodes/ray %.1f tris/ray %.1f" % (st["bounce_nodes"]/nb, st["bounce_tris"]/nb, ns, st["shadow_nodes"]/max(ns,1), st["shadow_tris"]/max(ns,1)))

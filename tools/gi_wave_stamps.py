"""Where the closest-hit pass's waves spend their loop iterations (the verdict's "in-kernel stamps" for the bounce rays): wave-uniform
counters kept by the STATS build of the traversal loop (gi_device.h) -- iterations, iterations that ran a node phase / a leaf phase,
and the lanes live in them -- for the bench frame.  python tools/gi_wave_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT

W, H = 1920, 1080
cam = S.sponza_camera()
for long_thin in (False, True):
    sc = S.atrium_standin(long_thin=long_thin)
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=5)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
    r.submit_commands_gbuffer()
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.set_debug_hits(True)
    r.ray_count(reset=True)
    r.submit_commands_gi_pathtrace()
    rays = r.ray_count()
    st, ws = r.traversal_stats(), r.wave_stats()
    hits = r.download_hits()
    it = ((hits["flags"] >> 8) & 0xFFF).astype(np.int64)  # per-ray loop iterations (node visits + leaf steps)
    print(f"[{sc.name}] closest-hit pass: {ws['waves']} waves, {W * H} rays: {st['bounce_nodes'] / (W * H):.1f} node visits + {st['bounce_tris'] / (W * H):.1f} triangle tests per ray")
    I = ws["iterations"]
    print(f"  loop iterations per wave {I / ws['waves']:.1f} (a ray needs {it.mean():.1f} on average, p99 {np.percentile(it, 99):.0f}, max {it.max()})")
    print(f"  node phase: ran in {ws['node_iterations'] / I:.3f} of the iterations with {ws['node_lanes'] / max(ws['node_iterations'], 1):.1f} of 64 lanes live")
    print(f"  leaf phase: ran in {ws['leaf_iterations'] / I:.3f} of the iterations with {ws['leaf_lanes'] / max(ws['leaf_iterations'], 1):.1f} of 64 lanes live")
    ni = r.node_index_stats()
    nv = max(st["bounce_nodes"], 1)
    print("  node visits by index in the breadth-first array: " + ", ".join(f"< {k}: {ni[f'below_{k}'] / nv:.3f}" for k in (64, 256, 1024, 4096))
          + f"; node phases that ended with > 12 stack entries: {ni['deep_stack_phases'] / nv:.4f}")
    node_cost, leaf_cost = 130.0, 110.0  # wave-instructions of one node phase / one leaf phase (two triangles), from the ISA
    total = ws["node_iterations"] * node_cost + ws["leaf_iterations"] * leaf_cost
    useful = ws["node_lanes"] / 64.0 * node_cost + ws["leaf_lanes"] / 64.0 * leaf_cost
    print(f"  issue slots by phase (at ~{node_cost:.0f} / ~{leaf_cost:.0f} wave-instructions per node / leaf phase): node {ws['node_iterations'] * node_cost / total:.2f}, "
          f"leaf {ws['leaf_iterations'] * leaf_cost / total:.2f}; lane utilisation of those slots {useful / total:.2f}")
    r.destroy()

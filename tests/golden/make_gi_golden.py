#!/usr/bin/env python3
"""Generates tests/golden/gi_*.npz from oracle/gi_np.py -- the independent numpy restatement of the GI shaders with
brute-force ray/triangle intersection (no acceleration structure).  Both the C++ oracle (oracle/trace_ref.cpp) and the HIP
path are compared with these vectors (tests/test_oracle_gi.py, tests/test_gi_gpu.py).

Each file holds the inputs (G-buffer planes in the reference's formats, the GlobalConstants fields, the input radiance)
and the expected outputs (radiance, and for the last sample's first bounce ray: t, GeometryIndex, PrimitiveIndex, whether
the sun ray was unoccluded).  The scenes are rebuilt by name (`case_scene`): procedural ones from nebulae_amd.scene, the
real ones from the committed glTF binaries next to this file.  The G-buffers are inputs, produced here with the C++
oracle's primary-visibility pass (any G-buffer would do)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = {
    # name: (scene key, camera kwargs, W, H, frameIndex, spp, maxPathVertices, albedo override)
    "gi_cornell_tex_40x32": ("cornell_standin_textured", dict(), 40, 32, 3, 1, 2, None),
    "gi_cornell_tex_multibounce_32x24": ("cornell_standin_textured", dict(yaw_deg=10.0, pitch_deg=80.0, distance=2.6), 32, 24, 7, 2, 4, None),
    "gi_cornell_box_real_32x32": ("cornell_box.glb", dict(origin=(0.0, 1.0, 0.0), distance=3.5), 32, 32, 5, 1, 3, (0.725, 0.71, 0.68)),
    "gi_damaged_helmet_48x32": ("DamagedHelmet_256.glb", dict(yaw_deg=15.0, pitch_deg=75.0, distance=2.2), 48, 32, 2, 1, 2, None),
    # the helmet with its ORIGINAL 2048^2 JPEG maps (the un-filtered texture content: many more mirror-like roughness texels)
    "gi_damaged_helmet_full_80x64": ("DamagedHelmet_jpeg.glb", dict(yaw_deg=25.0, pitch_deg=80.0, distance=1.7), 80, 64, 4, 1, 2, None),
}
GB_KEYS = ("albedo", "rough_metal", "world_pos", "normal", "depth")


def case_scene(key):
    from nebulae_amd import scene as S
    if key == "cornell_standin_textured":
        return S.cornell_standin(textured=True)
    return S.load_gltf(os.path.join(HERE, key))


def case_constants(d):
    from nebulae_amd import scene as S
    c = S.default_constants(frame_index=int(d["frame_index"]), spp=int(d["spp"]), eye=tuple(float(v) for v in d["eye"]),
                            max_path_vertices=int(d["max_path_vertices"]))
    return c


def main():
    import ctypes as C

    from nebulae_amd import scene as S
    from oracle import gi_np
    from oracle_lib import OracleTracer
    only = set(sys.argv[1:])
    for name, (key, camkw, W, H, frame, spp, mv, alb) in CASES.items():
        if only and name not in only:
            continue
        sc = case_scene(key)
        cam = S.orbit_camera(**camkw)
        o = OracleTracer(sc)
        gb = o.gbuffer(W, H, cam)
        if alb is not None:
            packed = o.L.trace_ref_pack_r11g11b10((C.c_float * 3)(*alb))
            gb["albedo"] = np.where((gb["depth"] >> 24) == 0xFF, np.uint32(packed), np.uint32(0)).astype(np.uint32)
        c = S.default_constants(frame_index=frame, spp=spp, eye=tuple(cam.eye), max_path_vertices=mv)
        rng = np.random.default_rng(frame)
        rad_in = rng.uniform(0.0, 0.5, (H, W, 4)).astype(np.float32)
        rad_in[..., 3] = 1.0
        out = gi_np.trace(sc, gb, c, radiance_in=rad_in)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), scene=key, eye=np.array(list(cam.eye), np.float32), frame_index=frame, spp=spp,
                            max_path_vertices=mv, radiance_in=rad_in, radiance=out["radiance"], t=out["t"], geometry=out["geometry"],
                            primitive=out["primitive"], unoccluded=out["unoccluded"], rays=out["rays"], **{k: gb[k] for k in GB_KEYS})
        cov = float(((gb["depth"] >> 24) == 0xFF).mean())
        print(f"{name}: {sc.num_triangles} triangles, {W}x{H}, coverage {cov:.2f}, bounce hits {(out['t'] > 0).mean():.2f}, "
              f"sun reached {out['unoccluded'].mean():.2f}, rays {out['rays']}, mean radiance {out['radiance'].mean():.4f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Writes the scene fixtures of BASELINE.json configs 1 and 2 from the reference checkout's assets.

Runs in the BUILD CONTAINER ONLY (it reads /root/reference/assets, which does not exist on the GPU box); the outputs
are committed and travel:

  tests/golden/cornell_box.glb        assets/cornell_box/cornell_box.gltf + .bin re-packed as one binary glTF:
                                      34 triangles, 3 factor-only submeshes, node rotation 90 deg about X.  Exact.
  tests/golden/DamagedHelmet_256.glb  assets/DamagedHelmet/DamagedHelmet.gltf + .bin: 15 452 triangles, 14 556 vertices,
                                      uint16 indices, one submesh, no TANGENT stream (the loader generates tangents as
                                      GLTFSceneImporter.cpp:626-727 does).  Geometry exact.  The three maps the reference
                                      uses (albedo, metalRoughness, normal; emissive and AO are ignored by it) are
                                      decoded from the JPEGs here and box-filtered 2048^2 -> 256^2 (8x8 means), stored as
                                      PNG: a 16 MB-per-map working set does not belong in a repository.  Tests load them
                                      with tex_upscale=8 to restore the 2048^2 footprint (each texel repeated 8x8).

  tests/golden/DamagedHelmet_jpeg.glb the same geometry with the three maps as the ORIGINAL JPEG files, embedded byte for byte
                                      (2.75 MB): the real 2048^2 texture content -- with far more mirror-like roughness
                                      texels than any box-filtered copy keeps -- for BASELINE.json config 2 and the
                                      gi_damaged_helmet_full golden.  Decoded by PIL wherever the tests run; the
                                      reference decodes with stb_image: at most 3 grey levels apart, mean 0.02
                                      (tests/test_scene_ref_cpu.py).

All are data in the scenes' own on-disk format (glTF 2.0 binary), read back through nebulae_amd.scene.load_gltf -- the
same code path (GLB container parsing included) the reference's default scene takes (src/Nebulae.cpp:36).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
ASSETS = "/root/reference/assets"


def box_downsample(px, k):
    h, w, c = px.shape
    v = px.reshape(h // k, k, w // k, k, c).astype(np.float64).mean(axis=(1, 3))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def main():
    from nebulae_amd import scene as S
    sc = S.load_gltf(os.path.join(ASSETS, "cornell_box", "cornell_box.gltf"))
    assert sc.num_triangles == 34 and len(sc.geometries) == 3
    S.save_glb(sc, os.path.join(HERE, "cornell_box.glb"))
    sc = S.load_gltf(os.path.join(ASSETS, "DamagedHelmet", "DamagedHelmet.gltf"))
    assert sc.num_triangles == 15452 and len(sc.geometries) == 1 and [t.shape for t in sc.textures] == [(2048, 2048, 4)] * 3
    import json
    doc = json.load(open(os.path.join(ASSETS, "DamagedHelmet", "DamagedHelmet.gltf")))
    files = [(open(os.path.join(ASSETS, "DamagedHelmet", doc["images"][i]["uri"]), "rb").read(), "image/jpeg") for i in sc.source["images"]]
    S.save_glb(sc, os.path.join(HERE, "DamagedHelmet_jpeg.glb"), encoded_images=files)
    full = S.load_gltf(os.path.join(HERE, "DamagedHelmet_jpeg.glb"))
    assert all(np.array_equal(a, b) for a, b in zip(full.textures, sc.textures)) and np.array_equal(full.geometries[0]["positions"], sc.geometries[0]["positions"])
    sc.textures = [box_downsample(t, 8) for t in sc.textures]
    S.save_glb(sc, os.path.join(HERE, "DamagedHelmet_256.glb"))
    for f in ("cornell_box.glb", "DamagedHelmet_256.glb", "DamagedHelmet_jpeg.glb"):
        back = S.load_gltf(os.path.join(HERE, f))
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes;", back.num_triangles, "triangles,", len(back.textures), "textures")


if __name__ == "__main__":
    main()

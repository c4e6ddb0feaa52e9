"""nebulae_amd.scene.load_gltf held against the reference's OWN glTF parser: TinyGLTF + stb_image, compiled from the
reference checkout where it lies (oracle/ref_tinygltf: a Makefile and a small dumper, outputs in oracle/_ref/).  This is the
one piece of the reference that can be built here, so it is the one place where parity is PINNED to the
reference rather than to a restatement: accessor payloads (strides, offsets, GLB chunks), node order and transforms, material
factors / texture -> image wiring, and the decoded RGBA8 texels (stb_image there, PIL here).

BUILD CONTAINER ONLY: needs /root/reference; on the GPU box (no reference, no oracle/_ref) the module skips."""
import json
import os
import subprocess

import numpy as np
import pytest

from nebulae_amd import scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DUMP = os.path.join(ROOT, "oracle", "_ref", "dump_gltf")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "vendor", "TinyGLTF")), reason="reference checkout not present (GPU box)")


@pytest.fixture(scope="module")
def dump_tool():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_tinygltf")])
    assert os.path.exists(DUMP)
    return DUMP


def tinygltf(tool, path, tmp_path, zeros=False):
    blob = str(tmp_path / (os.path.basename(path) + ".bin"))
    out = subprocess.run([tool, path, blob] + (["--missing-buffers-as-zeros"] if zeros else []), capture_output=True, text=True, check=True)
    return json.loads(out.stdout), np.memmap(blob, np.uint8, "r") if os.path.getsize(blob) else np.zeros(0, np.uint8)


def first_mesh_node(doc):
    """node order of ImportScene / ImportGLTFNode (GLTFSceneImporter.cpp:86,442-474): roots in order, depth first"""
    order = []

    def visit(i):
        order.append(i)
        for c in doc["nodes"][i]["children"]:
            visit(c)
    for r in doc["scenes"][max(doc["default_scene"], 0)]:
        visit(r)
    return next(i for i in order if doc["nodes"][i]["mesh"] >= 0)


def node_dict(n):
    d = {}
    for k in ("matrix", "translation", "rotation", "scale"):
        if n[k]:
            d[k] = n[k]
    return d


def check_structure(sc, doc, payload, with_payload):
    ni = first_mesh_node(doc)
    assert sc.source["node"] == ni
    prims = doc["meshes"][doc["nodes"][ni]["mesh"]]
    assert len(sc.geometries) == len(prims)
    M = S._node_matrix(node_dict(doc["nodes"][ni]))
    for g, p, src in zip(sc.geometries, prims, sc.source["primitives"]):
        assert np.array_equal(g["M"], M)
        assert src["material"] == p["material"]
        at = p["attributes"]
        assert g["positions"].shape == (at["POSITION"]["count"], 3) and g["normals"].shape == (at["NORMAL"]["count"], 3)
        assert g["uvs"].shape == (at["TEXCOORD_0"]["count"], 2) and len(g["indices"]) == p["indices"]["count"]
        assert src["has_tangents"] == ("TANGENT" in at)
        if not with_payload:
            continue
        for name, key, n in (("POSITION", "positions", 3), ("NORMAL", "normals", 3), ("TEXCOORD_0", "uvs", 2), ("TANGENT", "tangents", 4)):
            if name not in at:
                continue
            a = at[name]
            want = np.frombuffer(payload[a["offset"]:a["offset"] + 4 * a["floats"]], np.float32).reshape(-1, n)
            assert np.array_equal(g[key], want), name
        i = p["indices"]
        want = np.frombuffer(payload[i["offset"]:i["offset"] + 4 * i["count"]], np.uint32)
        assert np.array_equal(g["indices"].astype(np.uint32), want)
    # materials: factors and texture -> image wiring as the importer reads them
    for m, mi in zip(sc.materials, sc.source["materials"]):
        ref = doc["materials"][mi]
        want_img = (ref["base_color_image"], ref["normal_image"], ref["metallic_roughness_image"])
        got_img = tuple(sc.source["images"][t] if t >= 0 else -1 for t in m["textures"])
        assert got_img == want_img
        if want_img[0] < 0:
            assert np.allclose(m["albedo"], ref["base_color_factor"], rtol=0, atol=0)
        if want_img[2] < 0:
            assert m["rm"] == (ref["roughness_factor"], ref["metallic_factor"])


def texel_report(sc, doc, payload):
    """-> {"png": max abs difference over the PNG images, "jpg": (max, mean) over the JPEG images}"""
    worst = {"png": 0, "jpg": 0}
    mean_jpg = []
    for t, src in zip(sc.textures, sc.source["images"]):
        im = doc["images"][src]
        assert t.shape == (im["height"], im["width"], 4) and im["component"] == 4 and im["bits"] == 8
        want = np.asarray(payload[im["offset"]:im["offset"] + im["bytes"]]).reshape(t.shape)
        d = np.abs(t.astype(np.int16) - want.astype(np.int16))
        kind = "png" if im["uri"].lower().endswith(".png") else "jpg"
        worst[kind] = max(worst[kind], int(d.max()))
        if kind == "jpg":
            mean_jpg.append(float(d.mean()))
    return worst, (float(np.mean(mean_jpg)) if mean_jpg else 0.0)


def test_cornell_box_matches_tinygltf(dump_tool, tmp_path):
    path = os.path.join(REF, "assets", "cornell_box", "cornell_box.gltf")
    doc, payload = tinygltf(dump_tool, path, tmp_path)
    sc = S.load_gltf(path)
    check_structure(sc, doc, payload, with_payload=True)
    assert sc.num_triangles == 34 and len(sc.textures) == 0


def test_damaged_helmet_matches_tinygltf_and_its_jpegs_decode_like_stb_image(dump_tool, tmp_path):
    path = os.path.join(REF, "assets", "DamagedHelmet", "DamagedHelmet.gltf")
    doc, payload = tinygltf(dump_tool, path, tmp_path)
    sc = S.load_gltf(path)
    check_structure(sc, doc, payload, with_payload=True)
    assert sc.num_triangles == 15452 and [t.shape for t in sc.textures] == [(2048, 2048, 4)] * 3
    worst, mean = texel_report(sc, doc, payload)
    # the three maps are JPEGs: stb_image (the reference) and libjpeg (PIL) share the DCT coefficients but not the
    # inverse transform's rounding nor the chroma upsampling filter.  Measured here: max 3 grey levels, mean 0.02 (Sponza's
    # JPEGs: 3 and 0.01); the bound below is what this test pins (and DESIGN.md quotes)
    print(f"DamagedHelmet JPEG texels, PIL vs stb_image: max |d| = {worst['jpg']}, mean |d| = {mean:.3f}")
    assert worst["jpg"] <= 4 and mean < 0.05


def test_the_committed_glb_fixtures_parse_like_tinygltf(dump_tool, tmp_path):
    """the binary container (the reference's default scene is a .glb, src/Nebulae.cpp:36): chunks, embedded PNG images"""
    for name in ("cornell_box.glb", "DamagedHelmet_256.glb"):
        path = os.path.join(ROOT, "tests", "golden", name)
        doc, payload = tinygltf(dump_tool, path, tmp_path)
        sc = S.load_gltf(path)
        check_structure(sc, doc, payload, with_payload=True)
        worst, _ = texel_report(sc, doc, payload)
        assert worst["png"] == 0 and worst["jpg"] == 0  # embedded PNGs: lossless, both decoders agree to the bit


def test_sponza_gltf_structure_matches_tinygltf_without_its_stripped_bin(dump_tool, tmp_path):
    """assets/sponza/Sponza.gltf: 103 primitives, 25 materials, 69 images; Sponza.bin is stripped from the checkout
    (.MISSING_LARGE_BLOBS), so the payload is read as zeros on both sides and only the structure and the textures are
    compared -- with the .bin dropped beside the .gltf the same load_gltf call (and bench.py --scene) takes the real geometry."""
    path = os.path.join(REF, "assets", "sponza", "Sponza.gltf")
    if os.path.exists(os.path.join(REF, "assets", "sponza", "Sponza.bin")):
        pytest.skip("the .bin is present: covered by the payload tests")
    doc, payload = tinygltf(dump_tool, path, tmp_path, zeros=True)
    with pytest.raises((FileNotFoundError, OSError)):
        S.load_gltf(path)  # the default stays loud
    sc = S.load_gltf(path, missing_buffers="zeros")
    check_structure(sc, doc, payload, with_payload=False)
    assert len(sc.geometries) == 103 and len(sc.materials) == 25 and sc.num_triangles == 262267
    assert len(sc.textures) == len({i for m in doc["materials"] for i in (m["base_color_image"], m["normal_image"], m["metallic_roughness_image"]) if i >= 0})
    worst, mean = texel_report(sc, doc, payload)
    print(f"Sponza texels, PIL vs stb_image: PNG max |d| = {worst['png']}, JPEG max |d| = {worst['jpg']}, mean |d| = {mean:.3f}")
    assert worst["png"] == 0 and worst["jpg"] <= 4 and mean < 0.05

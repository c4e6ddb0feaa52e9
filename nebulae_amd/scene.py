"""Scene tables for the GI path: the host-side equivalent of GLTFSceneImporter + GIProcessedScene.

* ``load_gltf`` extracts what ``GLTFSceneImporter::ImportStaticMesh``
  (/root/reference/src/core/GLTFSceneImporter.cpp:476-775) extracts per primitive: SoA attribute
  streams (position float3, normal float3, texcoord float2, tangent float4 -- generated per
  :626-727 when absent), uint16/uint32 indices, one material per submesh (factors or RGBA8
  textures, no sRGB decode, :156), and the node transform ``s * r * t`` (:777-802).
* ``Scene.descs()`` flattens it to the C structs of include/nebulae_hip.h, which mirror
  ``StaticMeshGeometryData`` / ``StaticMeshMaterialData`` (src/nri/GIProcessedScene.h:17-39).
* ``cornell_standin`` / ``atrium_standin`` are procedural scenes.  The Sponza geometry blobs are
  stripped from the reference checkout (.MISSING_LARGE_BLOBS); ``atrium_standin`` matches the
  statistics of assets/sponza/Sponza.gltf (103 submeshes, ~262 k triangles, 25 materials with its
  69-image texture topology at 1024^2, node scale 0.008, same world AABB) and every number measured on it is labelled "sponza-standin".
"""
import ctypes as C
import json
import math
import os

import numpy as np

F = np.float32


class GeometryDesc(C.Structure):
    _fields_ = [("surfaceToWorld", C.c_float * 16), ("materialIndex", C.c_int32), ("indexStride", C.c_uint32),
                ("numIndices", C.c_uint32), ("numVertices", C.c_uint32), ("indices", C.c_void_p),
                ("attributes", C.c_void_p * 4), ("attributeStrides", C.c_uint32 * 4), ("_pad", C.c_uint32)]


class MaterialDesc(C.Structure):
    _fields_ = [("textureIndices", C.c_int32 * 3), ("albedo", C.c_float * 4), ("roughnessMetalness", C.c_float * 2),
                ("_pad", C.c_uint32)]


class TextureDesc(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class GIConstants(C.Structure):
    """GlobalConstants (src/DeferredRenderer.h:219-238) without the NRC-only members."""
    _fields_ = [("frameIndex", C.c_uint32), ("samplesPerPixel", C.c_uint32), ("maxPathVertices", C.c_uint32),
                ("cameraWorldPos", C.c_float * 3), ("skyColor", C.c_float * 3), ("sunLightDirection", C.c_float * 3),
                ("sunLightRadiance", C.c_float * 3), ("sunTanHalfAngle", C.c_float), ("throughputThreshold", C.c_float)]


class CameraDesc(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("target", C.c_float * 3), ("up", C.c_float * 3), ("vfov_deg", C.c_float),
                ("znear", C.c_float), ("zfar", C.c_float)]


def default_constants(frame_index=1, spp=1, eye=(0.0, 0.0, 3.0), max_path_vertices=2):
    """UI defaults of src/DeferredRenderer.h:111-125; sunTanHalfAngle per DeferredRenderer.cpp:418."""
    c = GIConstants()
    c.frameIndex, c.samplesPerPixel, c.maxPathVertices = frame_index, spp, max_path_vertices
    c.cameraWorldPos[:] = eye
    c.skyColor[:] = (8.0, 8.0, 8.0)
    c.sunLightDirection[:] = (0.5, -1.0, -0.2)
    c.sunLightRadiance[:] = (20.0, 20.0, 20.0)
    c.sunTanHalfAngle = math.tan(math.radians(0.58 * 0.5))
    c.throughputThreshold = 0.01
    return c


def orbit_camera(origin=(0.0, 0.0, 0.0), yaw_deg=0.0, pitch_deg=90.0, distance=3.0):
    """InspectCamera::GetEyePos (src/core/InspectCamera.h:31-42,52-55); projection per
    src/DeferredRenderer.cpp:147-148 (RH, vfov 60 deg, near 0.1, far 100)."""
    rx, ry = math.radians(yaw_deg), math.radians(pitch_deg)
    v = np.array([math.cos(ry) * math.cos(rx), math.sin(rx), math.sin(ry) * math.cos(rx)])
    v = v / np.linalg.norm(v) * distance
    cam = CameraDesc()
    cam.eye[:] = [float(origin[k] + v[k]) for k in range(3)]
    cam.target[:] = [float(x) for x in origin]
    cam.up[:] = (0.0, 1.0, 0.0)
    cam.vfov_deg, cam.znear, cam.zfar = 60.0, 0.1, 100.0
    return cam


class Scene:
    def __init__(self, name="scene"):
        self.name = name
        self.geometries = []  # dicts: M[4,4], material, indices, positions, normals, uvs, tangents
        self.materials = []   # dicts: textures[3] (-1 = none), albedo[4], rm[2]
        self.textures = []    # uint8 [h, w, 4]

    def add_geometry(self, positions, normals, uvs, indices, material, M=None, tangents=None, omit=()):
        """omit: attribute streams to leave out of the submesh ("normals", "uvs", "tangents"): their bindless index is
        then invalid and hits on the submesh end the path (pathtracer.hlsl:313-318)."""
        positions = np.ascontiguousarray(positions, F)
        normals = np.ascontiguousarray(normals, F)
        uvs = np.ascontiguousarray(uvs, F)
        idx = np.ascontiguousarray(indices).reshape(-1)
        idx = idx.astype(np.uint16 if positions.shape[0] <= 65535 else np.uint32)
        if tangents is None:
            tangents = generate_tangents(positions, normals, uvs, idx)
        gm = dict(M=np.eye(4, dtype=F) if M is None else np.ascontiguousarray(M, F), material=material,
                  indices=idx, positions=positions, normals=normals, uvs=uvs, tangents=np.ascontiguousarray(tangents, F))
        for key in omit:
            if key not in ("normals", "uvs", "tangents"):
                raise ValueError(f"cannot omit {key!r}")
            gm[key] = None
        self.geometries.append(gm)

    def add_material(self, albedo=(0, 0, 0, 1), rm=(1.0, 0.0), textures=(-1, -1, -1)):
        self.materials.append(dict(textures=tuple(int(t) for t in textures), albedo=tuple(float(a) for a in albedo),
                                   rm=tuple(float(r) for r in rm)))
        return len(self.materials) - 1

    def add_texture(self, rgba8):
        self.textures.append(np.ascontiguousarray(rgba8, np.uint8))
        return len(self.textures) - 1

    @property
    def num_triangles(self):
        return sum(len(g["indices"]) // 3 for g in self.geometries)

    def world_aabb(self):
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for g in self.geometries:
            w = g["positions"] @ g["M"][:3, :3] + g["M"][3, :3]
            lo, hi = np.minimum(lo, w.min(0)), np.maximum(hi, w.max(0))
        return lo, hi

    def descs(self):
        """-> (GeometryDesc[], n, MaterialDesc[], n, TextureDesc[], n); arrays borrow this Scene's memory."""
        G = (GeometryDesc * len(self.geometries))()
        for d, g in zip(G, self.geometries):
            d.surfaceToWorld[:] = [float(v) for v in g["M"].reshape(-1)]
            d.materialIndex = g["material"]
            d.indexStride = g["indices"].dtype.itemsize
            d.numIndices = g["indices"].size
            d.numVertices = g["positions"].shape[0]
            d.indices = g["indices"].ctypes.data
            for k, (key, n) in enumerate((("positions", 3), ("normals", 3), ("uvs", 2), ("tangents", 4))):
                d.attributes[k] = g[key].ctypes.data if g[key] is not None else None
                d.attributeStrides[k] = 4 * n
        Mt = (MaterialDesc * max(1, len(self.materials)))()
        for d, m in zip(Mt, self.materials):
            d.textureIndices[:] = m["textures"]
            d.albedo[:] = m["albedo"]
            d.roughnessMetalness[:] = m["rm"]
        T = (TextureDesc * max(1, len(self.textures)))()
        for d, t in zip(T, self.textures):
            d.rgba8 = t.ctypes.data
            d.height, d.width = t.shape[0], t.shape[1]
        return G, len(self.geometries), Mt, len(self.materials), T, len(self.textures)


def generate_tangents(positions, normals, uvs, indices):
    """Per-vertex tangents as GLTFSceneImporter.cpp:626-727 (accumulated sdir/tdir, Gram-Schmidt, handedness)."""
    tri = indices.reshape(-1, 3).astype(np.int64)
    p0, p1, p2 = positions[tri[:, 0]], positions[tri[:, 1]], positions[tri[:, 2]]
    w0, w1, w2 = uvs[tri[:, 0]], uvs[tri[:, 1]], uvs[tri[:, 2]]
    d1, d2 = p1 - p0, p2 - p0
    u1, u2 = w1 - w0, w2 - w0
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (F(1.0) / (u1[:, 0] * u2[:, 1] - u1[:, 1] * u2[:, 0])).astype(F)
        sd = ((d1 * u2[:, 1:2] - d2 * u1[:, 1:2]) * r[:, None]).astype(F)
        td = ((d2 * u1[:, 0:1] - d1 * u2[:, 0:1]) * r[:, None]).astype(F)
    sd = np.nan_to_num(sd, nan=0.0, posinf=0.0, neginf=0.0)  # degenerate UV triangles (the reference lets NaNs through)
    td = np.nan_to_num(td, nan=0.0, posinf=0.0, neginf=0.0)
    tan1 = np.zeros_like(positions)
    tan2 = np.zeros_like(positions)
    for k in range(3):
        np.add.at(tan1, tri[:, k], sd)
        np.add.at(tan2, tri[:, k], td)
    n = normals
    t = tan1 - n * np.sum(n * tan1, axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = t / np.linalg.norm(t, axis=1, keepdims=True)
    t = np.nan_to_num(t, nan=0.0, posinf=0.0, neginf=0.0)
    w = np.where(np.sum(np.cross(n, tan1) * tan2, axis=1) < 0.0, F(-1.0), F(1.0))
    return np.concatenate([t, w[:, None]], axis=1).astype(F)


# ---------------------------------------------------------------------------------------------
# glTF (ASCII .gltf + .bin, or binary .glb) -- what the reference gets from TinyGLTF
# ---------------------------------------------------------------------------------------------
_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _node_matrix(node):
    """GetTransformationMatrix (GLTFSceneImporter.cpp:777-802): row-vector s * r * t."""
    if "matrix" in node:
        return np.array(node["matrix"], F).reshape(4, 4)  # column-major glTF == row-vector row-major
    T = np.eye(4, dtype=np.float64)
    T[3, :3] = node.get("translation", [0, 0, 0])
    x, y, z, w = node.get("rotation", [0, 0, 0, 1])
    Rc = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    R = np.eye(4)
    R[:3, :3] = Rc.T  # row-vector convention
    S = np.diag(list(node.get("scale", [1, 1, 1])) + [1.0])
    return (S @ R @ T).astype(F)


_GLB_MAGIC, _GLB_JSON, _GLB_BIN = 0x46546C67, 0x4E4F534A, 0x004E4942  # 'glTF', 'JSON', 'BIN\0'


def _read_container(path):
    """-> (json dict, embedded BIN chunk or None).  EGLTFType::AsciiFile / ::Binary of
    GLTFSceneImporter::ImportScenesFromFile (src/core/GLTFSceneImporter.cpp:20-34); the reference's default scene is a
    .glb (src/Nebulae.cpp:36).  GLB 2.0: 12-byte header {magic, version, length}, then chunks {length, type, data}, the
    first one JSON, an optional second one BIN (the buffer without a uri)."""
    raw = open(path, "rb").read()
    if len(raw) >= 12 and int.from_bytes(raw[:4], "little") == _GLB_MAGIC:
        version, length = int.from_bytes(raw[4:8], "little"), int.from_bytes(raw[8:12], "little")
        if version != 2 or length > len(raw):
            raise ValueError(f"{path}: unsupported or truncated GLB (version {version}, length {length} of {len(raw)})")
        off, doc, blob = 12, None, None
        while off + 8 <= length:
            n, kind = int.from_bytes(raw[off:off + 4], "little"), int.from_bytes(raw[off + 4:off + 8], "little")
            data = raw[off + 8:off + 8 + n]
            if len(data) != n:
                raise ValueError(f"{path}: truncated GLB chunk")
            if kind == _GLB_JSON and doc is None:
                doc = json.loads(data.decode("utf-8"))
            elif kind == _GLB_BIN and blob is None:
                blob = np.frombuffer(data, np.uint8)
            off += 8 + ((n + 3) & ~3)
        if doc is None:
            raise ValueError(f"{path}: GLB without a JSON chunk")
        return doc, blob
    return json.loads(raw.decode("utf-8")), None


def _read_uri(base, uri):
    if uri.startswith("data:"):
        import base64
        return np.frombuffer(base64.b64decode(uri.split(",", 1)[1]), np.uint8)
    return np.fromfile(os.path.join(base, uri), np.uint8)


def load_gltf(path, first_mesh_only=True, tex_upscale=1, missing_buffers="error"):
    """ASCII .gltf (+ .bin / image files) or binary .glb.  Nodes are visited as ImportScene / ImportGLTFNode do
    (GLTFSceneImporter.cpp:86,442-474): the scene's root nodes in order, then children, each mesh node with its OWN
    local transform (the reference does not concatenate parents).  first_mesh_only mirrors InitRTAccelerationStructures,
    which builds the BLAS from the first StaticMesh only (src/DeferredRenderer.cpp:992-995).  tex_upscale > 1 repeats
    every texel k x k times (fixtures that carry down-sampled maps get their original working-set size back).
    missing_buffers="zeros": an external .bin that is not there (assets/sponza/Sponza.bin is stripped from the reference
    checkout) reads as zeros of its declared length -- the file's STRUCTURE (primitives, accessors, materials, images) can
    then still be loaded and checked; with the .bin beside the .gltf the same call loads the real geometry.
    The Scene remembers where things came from: Scene.source = {"node": index of the mesh node, "primitives": [...],
    "images": glTF image index per Scene texture, "materials": glTF material index per Scene material}."""
    import io

    from PIL import Image
    base = os.path.dirname(path)
    g, blob = _read_container(path)

    def read_buffer(b):
        if "uri" not in b:
            return blob
        if missing_buffers == "zeros" and not b["uri"].startswith("data:") and not os.path.exists(os.path.join(base, b["uri"])):
            return np.zeros(int(b["byteLength"]), np.uint8)
        return _read_uri(base, b["uri"])
    buffers = [read_buffer(b) for b in g.get("buffers", [])]

    def view_bytes(i):
        bv = g["bufferViews"][i]
        off = bv.get("byteOffset", 0)
        return buffers[bv["buffer"]][off:off + bv["byteLength"]]

    def accessor(i):
        a = g["accessors"][i]
        bv = g["bufferViews"][a["bufferView"]]
        dt, n = _COMP[a["componentType"]], _NCOMP[a["type"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or np.dtype(dt).itemsize * n
        raw = buffers[bv["buffer"]]
        rows = np.lib.stride_tricks.as_strided(raw[off:], shape=(a["count"], np.dtype(dt).itemsize * n), strides=(stride, 1))
        return np.ascontiguousarray(rows).view(dt).reshape(a["count"], n)

    sc = Scene(os.path.basename(path))
    sc.source = {"node": None, "primitives": [], "images": [], "materials": []}
    tex_cache = {}

    def texture(ref):
        if ref is None or ref.get("index", -1) < 0:
            return -1
        src = g["textures"][ref["index"]]["source"]
        if src not in tex_cache:
            im = g["images"][src]
            data = view_bytes(im["bufferView"]) if "bufferView" in im else _read_uri(base, im["uri"])
            px = np.asarray(Image.open(io.BytesIO(data.tobytes())).convert("RGBA"))  # RGBA8, no sRGB decode (:156)
            if tex_upscale > 1:
                px = np.repeat(np.repeat(px, tex_upscale, axis=0), tex_upscale, axis=1)
            tex_cache[src] = sc.add_texture(px)
            sc.source["images"].append(src)
        return tex_cache[src]

    mat_map = {}
    order = []

    def visit(ni):
        order.append(ni)
        for c in g["nodes"][ni].get("children", []):
            visit(c)

    scenes = g.get("scenes") or [{"nodes": list(range(len(g["nodes"])))}]
    for root in scenes[g.get("scene", 0)].get("nodes", []):
        visit(root)
    for ni in order:
        node = g["nodes"][ni]
        if "mesh" not in node:
            continue
        M = _node_matrix(node)
        for prim in g["meshes"][node["mesh"]]["primitives"]:
            at = prim["attributes"]
            pos, nrm, uv = accessor(at["POSITION"]), accessor(at["NORMAL"]), accessor(at["TEXCOORD_0"])
            idx = accessor(prim["indices"]).reshape(-1)
            tang = accessor(at["TANGENT"]) if "TANGENT" in at else None
            mi = prim.get("material", -1)
            if mi >= 0 and mi not in mat_map:
                m = g["materials"][mi]
                pbr = m.get("pbrMetallicRoughness", {})
                tex = (texture(pbr.get("baseColorTexture")), texture(m.get("normalTexture")),
                       texture(pbr.get("metallicRoughnessTexture")))
                mat_map[mi] = sc.add_material(albedo=pbr.get("baseColorFactor", [1, 1, 1, 1]) if tex[0] < 0 else (0, 0, 0, 1),
                                              rm=(pbr.get("roughnessFactor", 1.0), pbr.get("metallicFactor", 1.0))
                                              if tex[2] < 0 else (1.0, 0.0), textures=tex)
                sc.source["materials"].append(mi)
            sc.add_geometry(pos, nrm, uv, idx, mat_map.get(mi, -1), M=M, tangents=tang)
            sc.source["primitives"].append({"mesh": node["mesh"], "material": mi, "has_tangents": tang is not None})
        if sc.source["node"] is None:
            sc.source["node"] = ni
        if first_mesh_only:
            break
    return sc


def save_glb(scene, path, with_tangents=False, encoded_images=None):
    """Writes a Scene as one binary glTF 2.0 file: one mesh node (matrix = the geometries' shared instance transform), one
    primitive per geometry, textures as embedded PNGs.  The fixtures under tests/golden/ are written with it; tangents
    are left out by default so the loader regenerates them exactly as it does for a file that has none.
    encoded_images: per Scene texture the bytes of an already encoded image file (and its mime type) to embed as they are
    instead of a PNG of the decoded texels -- [(bytes, "image/jpeg"), ...]: a JPEG-textured asset keeps its original files."""
    import io

    from PIL import Image
    M0 = scene.geometries[0]["M"]
    if any(not np.array_equal(g["M"], M0) for g in scene.geometries):
        raise ValueError("save_glb: all geometries must share one instance transform (one mesh node)")
    blob = bytearray()
    views, accs = [], []

    def add_view(data):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob.extend(data)
        return len(views) - 1

    def add_acc(arr, ctype, kind, minmax=False):
        a = {"bufferView": add_view(np.ascontiguousarray(arr).tobytes()), "componentType": ctype, "count": int(arr.shape[0]), "type": kind}
        if minmax:
            a["min"], a["max"] = [float(v) for v in arr.min(0)], [float(v) for v in arr.max(0)]
        accs.append(a)
        return len(accs) - 1

    prims = []
    for gm in scene.geometries:
        at = {"POSITION": add_acc(gm["positions"], 5126, "VEC3", True), "NORMAL": add_acc(gm["normals"], 5126, "VEC3"),
              "TEXCOORD_0": add_acc(gm["uvs"], 5126, "VEC2")}
        if with_tangents and gm["tangents"] is not None:
            at["TANGENT"] = add_acc(gm["tangents"], 5126, "VEC4")
        idx = gm["indices"]
        pr = {"attributes": at, "indices": add_acc(idx.reshape(-1, 1), 5123 if idx.dtype == np.uint16 else 5125, "SCALAR")}
        if gm["material"] >= 0:
            pr["material"] = int(gm["material"])
        prims.append(pr)
    images, textures = [], []
    for k, t in enumerate(scene.textures):
        if encoded_images is not None:
            data, mime = encoded_images[k]
            images.append({"bufferView": add_view(bytes(data)), "mimeType": mime})
        else:
            buf = io.BytesIO()
            Image.fromarray(t, "RGBA").save(buf, format="PNG", optimize=True)
            images.append({"bufferView": add_view(buf.getvalue()), "mimeType": "image/png"})
        textures.append({"source": len(images) - 1})
    mats = []
    for m in scene.materials:
        pbr = {}
        ta, tn, tr = m["textures"]
        if ta >= 0:
            pbr["baseColorTexture"] = {"index": ta}
        else:
            pbr["baseColorFactor"] = [float(v) for v in m["albedo"]]
        if tr >= 0:
            pbr["metallicRoughnessTexture"] = {"index": tr}
        else:
            pbr["roughnessFactor"], pbr["metallicFactor"] = float(m["rm"][0]), float(m["rm"][1])
        d = {"pbrMetallicRoughness": pbr}
        if tn >= 0:
            d["normalTexture"] = {"index": tn}
        mats.append(d)
    doc = {"asset": {"version": "2.0", "generator": "nebulae_amd.scene.save_glb"}, "scene": 0, "scenes": [{"nodes": [0]}],
           "nodes": [{"mesh": 0, "name": scene.name, "matrix": [float(v) for v in M0.reshape(-1)]}],
           "meshes": [{"primitives": prims}], "accessors": accs, "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    if mats:
        doc["materials"] = mats
    if images:
        doc["images"], doc["textures"] = images, textures
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * (-len(js) % 4)
    while len(blob) % 4:
        blob.append(0)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(_GLB_MAGIC.to_bytes(4, "little") + (2).to_bytes(4, "little") + total.to_bytes(4, "little"))
        f.write(len(js).to_bytes(4, "little") + _GLB_JSON.to_bytes(4, "little") + js)
        f.write(len(blob).to_bytes(4, "little") + _GLB_BIN.to_bytes(4, "little") + bytes(blob))


# ---------------------------------------------------------------------------------------------
# Procedural scenes
# ---------------------------------------------------------------------------------------------
def _quad(p0, p1, p2, p3, uv_scale=1.0):
    """Two CCW triangles p0,p1,p2,p3 (normal = (p1-p0) x (p3-p0))."""
    P = np.array([p0, p1, p2, p3], F)
    n = np.cross(P[1] - P[0], P[3] - P[0])
    n = n / np.linalg.norm(n)
    N = np.tile(n.astype(F), (4, 1))
    UV = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], F) * F(uv_scale)
    return P, N, UV, np.array([0, 1, 2, 0, 2, 3], np.uint32)


def _merge(parts):
    P, N, UV, I, base = [], [], [], [], 0
    for p, n, uv, i in parts:
        P.append(p)
        N.append(n)
        UV.append(uv)
        I.append(i + base)
        base += p.shape[0]
    return np.concatenate(P), np.concatenate(N), np.concatenate(UV), np.concatenate(I)


def _box(lo, hi, uv_scale=1.0, inward=False):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),  # +z
             ((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),  # -z
             ((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),  # +x
             ((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),  # -x
             ((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)),  # +y
             ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1))]  # -y
    parts = []
    for f in faces:
        q = _quad(*(f if not inward else f[::-1]), uv_scale=uv_scale)
        parts.append(q)
    return _merge(parts)


def _grid_surface(fn, nu, nv, uv_scale=(1.0, 1.0)):
    """Tessellated parametric surface fn(u, v) -> (pos[...,3], normal[...,3]) on an (nu+1) x (nv+1) grid."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    pos, nrm = fn(u, v)
    P = pos.reshape(-1, 3).astype(F)
    N = nrm.reshape(-1, 3)
    N = (N / np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)).astype(F)
    UV = np.stack([u * uv_scale[0], v * uv_scale[1]], -1).reshape(-1, 2).astype(F)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).reshape(-1)
    b, c, d = a + (nv + 1), a + (nv + 1) + 1, a + 1
    I = np.stack([a, b, c, a, c, d], 1).reshape(-1).astype(np.uint32)
    return P, N, UV, I


def _proc_texture(kind, seed, size=256):
    """Deterministic RGBA8 textures: 'albedo' (tinted value noise + grout lines), 'normal'
    (tangent-space bumps), 'rm' (G = roughness, B = metalness).  The height field is a sum of separable waves, so a
    1024^2 map costs a few outer products (69 of them make the Sponza-sized bench scene)."""
    rng = np.random.default_rng(seed)
    ax = (np.arange(size, dtype=np.float64) / size * 2 * np.pi)
    ph = rng.uniform(0, 2 * np.pi, 8)
    k = rng.integers(1, 7, 8)
    sx, cy = np.sin(k[0] * ax + ph[0]).astype(F), np.cos(k[1] * ax + ph[1]).astype(F)
    s2, c2 = np.sin(k[2] * ax + ph[2]).astype(F), np.cos(k[2] * ax + ph[2]).astype(F)
    s3, c3 = np.sin(k[3] * ax).astype(F), np.cos(k[3] * ax).astype(F)
    c4, s4 = np.cos(k[4] * ax + ph[3]).astype(F), np.sin(k[4] * ax + ph[3]).astype(F)
    c5, s5 = np.cos(k[5] * ax).astype(F), np.sin(k[5] * ax).astype(F)
    # h[y, x] = sin(k0 x + p0) cos(k1 y + p1) + 0.5 sin(k2 x + k3 y + p2) + 0.25 cos(k4 x - k5 y + p3)
    h = np.outer(cy, sx) + F(0.5) * (np.outer(c3, s2) + np.outer(s3, c2)) + F(0.25) * (np.outer(c5, c4) + np.outer(s5, s4))
    lo, hi = float(h.min()), float(h.max())
    h = (h - F(lo)) * F(1.0 / (hi - lo))
    out = np.zeros((size, size, 4), np.uint8)
    out[..., 3] = 255
    if kind == "albedo":
        tint = rng.uniform(0.25, 0.95, 3).astype(F)
        i = np.arange(size)
        gx, gy = (i % max(1, size // 4) < 2), (i % max(1, size // 8) < 2)
        grout = (gx[None, :] | gy[:, None])
        val = (F(0.55) + F(0.45) * h) * np.where(grout, F(0.5), F(1.0))
        for c in range(3):
            out[..., c] = np.clip(val * (tint[c] * F(255.0)), 0, 255).astype(np.uint8)
    elif kind == "normal":
        gx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * F(-6.0)
        gy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * F(-6.0)
        inv = F(1.0) / np.sqrt(gx * gx + gy * gy + F(1.0))
        out[..., 0] = np.clip((gx * inv * F(0.5) + F(0.5)) * F(255.0), 0, 255).astype(np.uint8)
        out[..., 1] = np.clip((gy * inv * F(0.5) + F(0.5)) * F(255.0), 0, 255).astype(np.uint8)
        out[..., 2] = np.clip((inv * F(0.5) + F(0.5)) * F(255.0), 0, 255).astype(np.uint8)
    else:
        out[..., 1] = np.clip((F(0.35) + F(0.6) * h) * F(255.0), 0, 255).astype(np.uint8)
        out[..., 2] = 255 if seed % 7 == 0 else 0
    return out


def cornell_standin(textured=False):
    """A Cornell-box-like room of 34 triangles in 3 factor-only submeshes (white shell + boxes, red
    wall, green wall) under the reference's 90-degree-about-X node rotation -- the statistics of
    assets/cornell_box/cornell_box.gltf, generated, not copied."""
    sc = Scene("cornell-standin")
    s = math.sqrt(0.5)
    M = _node_matrix({"rotation": [s, 0, 0, s]})
    if textured:
        ta, tn, tr = sc.add_texture(_proc_texture("albedo", 11)), sc.add_texture(_proc_texture("normal", 12)), \
            sc.add_texture(_proc_texture("rm", 13))
        white = sc.add_material(textures=(ta, tn, tr))
    else:
        white = sc.add_material(albedo=(0.725, 0.71, 0.68, 1), rm=(1.0, 0.0))
    red = sc.add_material(albedo=(0.63, 0.065, 0.05, 1), rm=(1.0, 0.0))
    green = sc.add_material(albedo=(0.14, 0.45, 0.091, 1), rm=(1.0, 0.0))
    # room in WORLD space: x,y in [-1,1], z in [-2,0], open towards the default camera at (0,0,3);
    # vertices are stored in the node's local frame (world * M^-1), as a Blender Z-up export would.
    Minv = np.linalg.inv(M.astype(np.float64)).astype(F)

    def to_local(part):
        P, N, UV, I = part
        return (P @ Minv[:3, :3] + Minv[3, :3]).astype(F), (N @ Minv[:3, :3]).astype(F), UV, I

    shell = [_quad((-1, -1, -2), (1, -1, -2), (1, 1, -2), (-1, 1, -2)),      # back wall, faces +z
             _quad((-1, -1, 0), (1, -1, 0), (1, -1, -2), (-1, -1, -2)),      # floor, faces +y
             _quad((-1, 1, -2), (1, 1, -2), (1, 1, 0), (-1, 1, 0))]          # ceiling, faces -y
    short = _box((-0.7, -1.0, -1.3), (-0.1, -0.4, -0.7))
    tall = _box((0.1, -1.0, -1.9), (0.7, 0.2, -1.3))
    sc.add_geometry(*to_local(_merge(shell + [short, tall])), material=white, M=M)   # 6 + 12 + 12 = 30 triangles
    sc.add_geometry(*to_local(_quad((-1, -1, 0), (-1, -1, -2), (-1, 1, -2), (-1, 1, 0))), material=red, M=M)
    sc.add_geometry(*to_local(_quad((1, -1, -2), (1, -1, 0), (1, 1, 0), (1, 1, -2))), material=green, M=M)
    return sc


def atrium_standin(target_triangles=262267, n_submeshes=103, n_materials=25, tex_size=1024, seed=2025, long_thin=False):
    """'sponza-standin': a colonnaded two-storey atrium with arches, drapes and clutter, in Sponza's
    local units under the node scale 0.00800000038 and inside its local AABB
    (assets/sponza/Sponza.gltf accessor min/max: [-1921,-126,-1183] .. [1800,1429,1105]).
    long_thin: the pathology of the real asset that a uniformly gridded stand-in lacks ("long thin triangles") --
    the outer walls, gallery floors, their undersides and the roofs become full-length strips (aspect ratios of 50:1 to 120:1,
    ~1 500 triangles that each cross a large part of the atrium) and 40 thin beams span the court, while the floor keeps the
    bulk of the triangle budget: boxes of such triangles overlap everything along their length, which is what a BVH built from
    centroids handles worst."""
    rng = np.random.default_rng(seed)
    sc = Scene("sponza-standin-longthin" if long_thin else "sponza-standin")
    S = 0.00800000037997961
    M = np.diag([S, S, S, 1.0]).astype(F)
    lo = np.array([-1920.9, -126.4, -1182.8])
    hi = np.array([1799.9, 1429.4, 1105.4])
    # Material / texture topology of assets/sponza/Sponza.gltf: 25 materials, 24 of them with all three maps, material 2
    # with an albedo map only (the one 4x4 image); materials 14-16 and 17-19 each share one roughness-metalness map:
    # 24 + 24 + 20 + 1 = 69 images (68 of them 1024^2 in the original; `tex_size` here).
    mats, shared_rm = [], {}
    for m in range(n_materials):
        fac = dict(albedo=tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,), rm=(float(rng.uniform(0.3, 1.0)), 0.0))
        if m == 2:
            tex = [sc.add_texture(_proc_texture("albedo", 100 + m, 4)), -1, -1]
        else:
            group = 14 if 14 <= m <= 16 else 17 if 17 <= m <= 19 else m
            if group not in shared_rm:
                shared_rm[group] = None
            tex = [sc.add_texture(_proc_texture("albedo", 100 + m, tex_size)),
                   sc.add_texture(_proc_texture("normal", 200 + m, tex_size)), -1]
            if shared_rm[group] is None:
                shared_rm[group] = sc.add_texture(_proc_texture("rm", 300 + m, tex_size))
            tex[2] = shared_rm[group]
        mats.append(sc.add_material(textures=tex, **fac))
    parts = []  # (P, N, UV, I) per submesh

    def plane(x0, x1, z0, z1, y, nu, nv, up=True, bump=0.0):
        def fn(u, v):
            x = x0 + (x1 - x0) * u
            z = z0 + (z1 - z0) * (v if up else 1 - v)
            yy = y + bump * np.sin(u * 37.0) * np.cos(v * 29.0)
            pos = np.stack([x, yy, z], -1)
            nrm = np.zeros_like(pos)
            nrm[..., 1] = 1.0 if up else -1.0
            return pos, nrm
        return _grid_surface(fn, nu, nv, uv_scale=(12.0, 8.0))

    def wall(p0, p1, y0, y1, nu, nv, flip=False):
        p0, p1 = np.array(p0, float), np.array(p1, float)
        t = p1 - p0
        n = np.array([t[1], -t[0]]) / np.linalg.norm(t)
        if flip:
            n = -n

        def fn(u, v):
            uu = (1 - u) if flip else u
            x = p0[0] + t[0] * uu
            z = p0[1] + t[1] * uu
            pos = np.stack([x, y0 + (y1 - y0) * v, z], -1)
            nrm = np.zeros_like(pos)
            nrm[..., 0], nrm[..., 2] = n[0], n[1]
            return pos, nrm
        return _grid_surface(fn, nu, nv, uv_scale=(10.0, 4.0))

    def column(cx, cz, y0, y1, r, nu, nv, flute=0.06):
        def fn(u, v):
            a = u * 2 * np.pi
            rr = r * (1.0 + flute * np.cos(a * 12.0)) * (1.0 - 0.08 * v)
            pos = np.stack([cx + rr * np.cos(a), y0 + (y1 - y0) * v, cz - rr * np.sin(a)], -1)
            nrm = np.stack([np.cos(a), np.zeros_like(a), -np.sin(a)], -1)
            return pos, nrm
        return _grid_surface(fn, nu, nv, uv_scale=(4.0, 6.0))

    def arch(cx, cz, y, span, depth, axis, nu, nv):
        def fn(u, v):
            a = np.pi * u
            r = span * 0.5
            along = -r * np.cos(a)
            yy = y + r * np.sin(a)
            d = (v - 0.5) * depth
            if axis == 0:
                pos = np.stack([cx + along, yy, cz + d], -1)
                nrm = np.stack([np.cos(a), -np.sin(a), np.zeros_like(a)], -1)
            else:
                pos = np.stack([cx + d, yy, cz + along], -1)
                nrm = np.stack([np.zeros_like(a), -np.sin(a), np.cos(a)], -1)
            return pos, nrm
        return _grid_surface(fn, nu, nv, uv_scale=(3.0, 1.0))

    def drape(x, z0, z1, y0, y1, nu, nv, phase):
        def fn(u, v):
            zz = z0 + (z1 - z0) * u
            xx = x + 25.0 * np.sin(u * 21.0 + phase) * (0.3 + v)
            pos = np.stack([xx, y1 - (y1 - y0) * v, zz], -1)
            dx = 25.0 * 21.0 * np.cos(u * 21.0 + phase) * (0.3 + v) / (z1 - z0)
            nrm = np.stack([np.ones_like(dx), np.zeros_like(dx), -dx], -1)
            return pos, nrm
        return _grid_surface(fn, nu, nv, uv_scale=(2.0, 3.0))

    def blob(c, r, nu, nv, sq):
        def fn(u, v):
            a, b = u * 2 * np.pi, (v - 0.5) * np.pi
            rr = r * (1.0 + sq * np.cos(3 * a) * np.cos(b))
            d = np.stack([np.cos(b) * np.cos(a), np.sin(b), -np.cos(b) * np.sin(a)], -1)
            return np.array(c) + rr[..., None] * d, d
        return _grid_surface(fn, nu, nv, uv_scale=(2.0, 1.0))

    x0, x1, z0, z1 = lo[0], hi[0], lo[2], hi[2]
    yf, y2, yc = lo[1] + 126.0, 560.0, hi[1]
    ix0, ix1, iz0, iz1 = x0 + 700, x1 - 700, z0 + 520, z1 - 520  # inner court

    def grid(nu, nv, strips_along_u):  # long_thin: one cell along the long side, the strips across it
        return ((1, nv) if strips_along_u else (nu, 1)) if long_thin else (nu, nv)
    parts.append(plane(x0, x1, z0, z1, yf, 160, 96, bump=0.6))                      # floor
    parts.append(plane(x0, ix0, z0, z1, y2, *grid(40, 64, False)))                 # gallery floors (4)
    parts.append(plane(ix1, x1, z0, z1, y2, *grid(40, 64, False)))
    parts.append(plane(ix0, ix1, z0, iz0, y2, *grid(56, 24, True)))
    parts.append(plane(ix0, ix1, iz1, z1, y2, *grid(56, 24, True)))
    parts.append(plane(x0, ix0, z0, z1, y2 - 20, *grid(24, 40, False), up=False))  # gallery undersides (4)
    parts.append(plane(ix1, x1, z0, z1, y2 - 20, *grid(24, 40, False), up=False))
    parts.append(plane(ix0, ix1, z0, iz0, y2 - 20, *grid(40, 16, True), up=False))
    parts.append(plane(ix0, ix1, iz1, z1, y2 - 20, *grid(40, 16, True), up=False))
    parts.append(plane(x0, ix0, z0, z1, yc, *grid(16, 24, False), up=False))       # roof over galleries (4); court open
    parts.append(plane(ix1, x1, z0, z1, yc, *grid(16, 24, False), up=False))
    parts.append(plane(ix0, ix1, z0, iz0, yc, *grid(24, 12, True), up=False))
    parts.append(plane(ix0, ix1, iz1, z1, yc, *grid(24, 12, True), up=False))
    parts.append(wall((x0, z0), (x1, z0), yf, yc, *grid(120, 48, True), flip=True))  # outer walls (4), facing inward
    parts.append(wall((x1, z1), (x0, z1), yf, yc, *grid(120, 48, True), flip=True))
    parts.append(wall((x0, z1), (x0, z0), yf, yc, *grid(80, 48, True), flip=True))
    parts.append(wall((x1, z0), (x1, z1), yf, yc, *grid(80, 48, True), flip=True))
    if long_thin:  # 40 beams across the court, 2 290 x 12 x 12 units each (12 triangles of 190:1), at the two storeys
        # (one texture repeat per ~20 beam widths: a map that repeated eight times across a 10 cm face would put 85 k texels on a world
        # unit and turn the last ulp of a bounce direction -- ocml's against glibc's sinf -- into a different texel)
        beams = [_box((ix0, y, z - 6.0), (ix1, y + 12.0, z + 6.0), uv_scale=0.05)
                 for y in (y2 - 60.0, yc - 90.0) for z in np.linspace(iz0 + 40.0, iz1 - 40.0, 20)]
        parts.append(_merge(beams))
    ncol = 0
    for storey, (ya, yb) in enumerate(((yf, y2 - 20), (y2, yc - 60))):
        for cx in np.linspace(ix0, ix1, 9):
            for cz in (iz0, iz1):
                parts.append(column(cx, cz, ya, yb, 42.0 - 8 * storey, 40, 24))
                ncol += 1
        for cz in np.linspace(iz0, iz1, 5)[1:-1]:
            for cx in (ix0, ix1):
                parts.append(column(cx, cz, ya, yb, 42.0 - 8 * storey, 40, 24))
                ncol += 1
    for cx in (np.linspace(ix0, ix1, 9)[:-1] + (ix1 - ix0) / 16):
        for cz in (iz0, iz1):
            parts.append(arch(cx, cz, y2 - 150, (ix1 - ix0) / 8 - 84, 90.0, 0, 24, 6))
    for k in range(6):
        zc = iz0 + (k + 0.5) * (iz1 - iz0) / 6
        parts.append(drape(ix0 + 60 + 40 * (k % 2), zc - 150, zc + 150, y2 + 40, yc - 120, 48, 40, 0.7 * k))
    # clutter (vases / lion heads / pots) up to the requested submesh count
    while len(parts) < n_submeshes:
        c = (rng.uniform(ix0 + 100, ix1 - 100), yf + rng.uniform(30, 90), rng.uniform(iz0 + 80, iz1 - 80))
        parts.append(blob(c, rng.uniform(25, 70), 28, 18, rng.uniform(0.0, 0.25)))
    parts = parts[:n_submeshes]
    # refine the largest-area-per-triangle parts until the triangle budget is met: rebuild the floor denser
    total = sum(p[3].size // 3 for p in parts)
    if total < target_triangles:
        need = target_triangles - (total - parts[0][3].size // 3)
        nu = int(math.sqrt(need / 2 * 160 / 96))
        nv = max(1, int(round(need / 2 / nu)))
        parts[0] = plane(x0, x1, z0, z1, yf, nu, nv, bump=0.6)
    for k, (P, N, UV, I) in enumerate(parts):
        sc.add_geometry(P, N, UV, I, material=mats[k % n_materials], M=M)
    return sc


def moved_scene(sc, scale=1.0, shift=(0.0, 0.0, 0.0)):
    """The same scene scaled about the origin and then translated (world' = world * scale + shift): a copy that shares the vertex
    streams and differs in the per-geometry surfaceToWorld matrices (row-vector convention, as GIProcessedScene.cpp:47-93 uploads
    them).  For the tests that put a scene far from the origin or blow it up."""
    out = Scene(f"{sc.name}-x{scale:g}+{tuple(float(v) for v in shift)}")
    out.materials, out.textures = sc.materials, sc.textures
    for g in sc.geometries:
        h = dict(g)
        M = g["M"].astype(np.float64).copy()
        M[:, :3] *= scale
        M[3, :3] += np.asarray(shift, np.float64)
        h["M"] = np.ascontiguousarray(M, F)
        out.geometries.append(h)
    return out


def moved_camera(cam, scale=1.0, shift=(0.0, 0.0, 0.0)):
    """the camera that sees moved_scene(sc, scale, shift) as `cam` saw sc (near / far planes scale along)"""
    out = CameraDesc()
    out.eye[:] = [float(cam.eye[k] * scale + shift[k]) for k in range(3)]
    out.target[:] = [float(cam.target[k] * scale + shift[k]) for k in range(3)]
    out.up[:] = list(cam.up)
    out.vfov_deg, out.znear, out.zfar = cam.vfov_deg, cam.znear * scale, cam.zfar * scale
    return out


def framing_camera(sc):
    """A view of a whole scene from outside its box (bench.py --scene with a file that is not Sponza): the reference's orbit
    camera (InspectCamera.h:31-42) about the box centre, far enough for the vertical field of view to hold the box."""
    lo, hi = sc.world_aabb()
    centre, radius = 0.5 * (lo + hi), 0.5 * float(np.linalg.norm(hi - lo))
    return orbit_camera(origin=tuple(float(v) for v in centre), yaw_deg=12.0, pitch_deg=70.0, distance=max(2.2 * radius, 0.2))


def sponza_camera():
    """Reference default orbit camera (InspectCamera.h:52-55): origin (0,0,0), yaw 0, pitch 90 deg, distance 3 ->
    eye (0,0,3) looking down -z; moved up to eye height inside the atrium so the view is not degenerate."""
    cam = orbit_camera(origin=(0.0, 2.0, 0.0), yaw_deg=12.0, pitch_deg=60.0, distance=9.0)
    return cam

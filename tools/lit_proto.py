"""CPU prototype of the sun-visibility table (nebulae_amd/csrc/lit_predicate.h through tools/lit_proto.cpp): which share of the
shadow rays of a frame start on a (triangle, side) that the certificate proves fully lit, and -- the safety check -- that no
ray from such a (triangle, side) is occluded in the oracle's own trace.   python tools/lit_proto.py [W H]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nebulae_amd import scene as S  # noqa: E402
from oracle_lib import OracleTracer  # noqa: E402


def world_triangles(sc):
    V, N, first = [], [], []
    for g in sc.geometries:
        M = g["M"].astype(np.float64)
        P = g["positions"].astype(np.float32) @ g["M"][:3, :3].astype(np.float32) + g["M"][3, :3].astype(np.float32)
        n = g["normals"].astype(np.float64) @ M[:3, :3]
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        idx = g["indices"].astype(np.int64).reshape(-1, 3)
        first.append(sum(len(v) for v in V))
        V.append(P[idx].astype(np.float32))
        N.append(n[idx].astype(np.float32))
    return np.concatenate(V), np.concatenate(N), np.array(first)


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 360)
    sc = S.atrium_standin()
    cam = S.sponza_camera()
    V, N, first = world_triangles(sc)
    n = len(V)
    lib = C.CDLL(os.path.join(ROOT, "build_variants", "liblit_proto.so"))
    consts = S.default_constants(frame_index=3)
    consts.cameraWorldPos[:] = tuple(cam.eye)
    sun = np.array(list(consts.sunLightDirection), np.float32)
    flags = np.zeros(n, np.uint8)
    stats = np.zeros(2, np.float64)
    t0 = time.time()
    lib.lit_proto_flags(n, V.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), sun.ctypes.data_as(C.c_void_p),
                        C.c_float(consts.sunTanHalfAngle), flags.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p))
    print(f"{n} triangles: lit(+) {np.mean(flags & 1):.3f}, lit(-) {np.mean((flags >> 1) & 1):.3f}; {stats[0] / n:.0f} pair tests per triangle, "
          f"{stats[1]:.0f} receiver sides without a certificate; {time.time() - t0:.1f} s")
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    rad, hits, rays = o.gi(gb, consts)
    hit = hits["t"] > 0
    tri = first[np.minimum(hits["geometry"], len(first) - 1)] + hits["primitive"]
    tri = np.where(hit, tri, 0)
    L = -sun.astype(np.float64)
    L /= np.linalg.norm(L)
    # the side a ray starts on is decided per ray (transition = dot(GN, inc) <= 0, pathtracer.hlsl:558-560); here it is known
    # only where all three vertex normals agree about the sun by more than the disk's half angle
    nl = N[tri].astype(np.float64) @ L                       # [H, W, 3]
    plus, minus = (nl > 0.02).all(axis=2), (nl < -0.02).all(axis=2)
    known = plus | minus
    side_plus = plus
    lit = np.where(side_plus, flags[tri] & 1, (flags[tri] >> 1) & 1).astype(bool) & hit & known
    lit |= (flags[tri] == 3) & hit
    print(f"side known for {(known & hit).sum() / hit.sum():.3f} of the shadow rays")
    visible = (hits["flags"] & 1).astype(bool) & hit
    print(f"{W}x{H}: bounce hits {hit.mean():.3f} of pixels; shadow rays unoccluded {visible.sum() / hit.sum():.3f}; "
          f"proven lit by the table {lit.sum() / hit.sum():.3f} of the shadow rays = {(lit & visible).sum() / max(visible.sum(), 1):.3f} of the unoccluded ones")
    # how far would ONE (or two) precomputed occluder hints per (triangle, side) go for the occluded rays?
    cover = np.zeros((n, 2, 2), np.float32)
    lib.lit_proto_hints(n, V.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), sun.ctypes.data_as(C.c_void_p),
                        C.c_float(consts.sunTanHalfAngle), cover.ctypes.data_as(C.c_void_p))
    occl = hit & ~visible & known
    side = np.where(side_plus, 0, 1)
    c1, c2 = cover[tri, side, 0], cover[tri, side, 1]
    print(f"occluded rays (side known): {occl.sum()}; expected share answered by the best single occluder above: {c1[occl].mean():.3f}; by the two best: {c2[occl].mean():.3f}; "
          f"rays whose triangle has NO occluder wholly above its footprint: {(c1[occl] == 0).mean():.3f}")
    bad = lit & ~visible
    print(f"SAFETY: rays from a proven-lit (triangle, side) that the oracle found occluded: {int(bad.sum())}")
    if bad.any():
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:10]:
            print("  pixel", x, y, "tri", tri[y, x], "geom", hits["geometry"][y, x], "flags", flags[tri[y, x]], "side+", side_plus[y, x])


if __name__ == "__main__":
    main()

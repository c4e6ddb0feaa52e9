"""Deterministic synthetic G-buffers and noisy radiance for SVGF tests and benches.

Inputs follow SURVEY.md section 8d ("Synthetic inputs for SVGF-only benchmarking"):
analytic depth (two tilted planes + a sphere, z in [.2,.98], quantised to D24),
analytic normals packed with Oct16_FastPack
(/root/reference/assets/shaders/octahedron_encoding.hlsli:16-23) and rounded to
fp16 with .xy == .zw, and radiance = smooth base x log-normal noise from a
counter-based generator keyed on (seed, frame, pixel).  Everything is plain numpy
integer/float32 arithmetic so the same bytes come out on every machine.
"""
import numpy as np

F = np.float32


def jenkins_hash(x):
    """Jenkins one-at-a-time hash on uint32 arrays (same mixing as
    /root/reference/assets/shaders/rand.hlsli:6-14)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x += x << np.uint32(10)
        x ^= x >> np.uint32(6)
        x += x << np.uint32(3)
        x ^= x >> np.uint32(11)
        x += x << np.uint32(15)
    return x


def uniform01(seed, frame, index, stream):
    """Counter-based uniform [0,1) float32 keyed on (seed, frame, pixel index, stream)."""
    fk = np.uint32((int(frame) * 0x9E3779B9 + int(stream)) & 0xFFFFFFFF)
    k = jenkins_hash(np.uint32(seed) ^ jenkins_hash(fk))
    h = jenkins_hash(np.asarray(index, np.uint32) ^ k)
    with np.errstate(over="ignore"):
        h = jenkins_hash(h + np.uint32((int(stream) * 0x85EBCA6B) & 0xFFFFFFFF))
    return ((h >> np.uint32(9)) | np.uint32(0x3F800000)).view(F) - F(1.0)


def oct16_fast_pack(n):
    """[..., 3] unit vectors -> [..., 2] in [-1,1] (octahedron_encoding.hlsli:16-23)."""
    n = n.astype(F)
    inv = F(1.0) / (np.abs(n[..., 0]) + np.abs(n[..., 1]) + np.abs(n[..., 2]))
    px, py = n[..., 0] * inv, n[..., 1] * inv
    sx = np.where(px > 0, F(1), F(-1))
    sy = np.where(py > 0, F(1), F(-1))
    fx = (F(1) - np.abs(py)) * sx
    fy = (F(1) - np.abs(px)) * sy
    lower = n[..., 2] <= 0
    return np.stack([np.where(lower, fx, px), np.where(lower, fy, py)], axis=-1).astype(F)


def pack_depth_stencil(z, stencil=0xFF):
    """float depth in [0,1] -> R24G8 uint32 (D24_UNORM in bits 0..23, stencil in 24..31)."""
    d = np.rint(np.clip(z.astype(np.float64), 0.0, 1.0) * 16777215.0).astype(np.uint32)
    return d | (np.uint32(stencil) << np.uint32(24))


def synth_gbuffer(W, H, seed=1234, camera_shift=0.0):
    """Returns dict(depth uint32[H,W], normal float16[H,W,4], base float32[H,W,3]).

    camera_shift slides the sphere horizontally (fraction of the width) so that
    consecutive frames can disagree in depth/normal (temporal-stability weight < 1)."""
    ys, xs = np.meshgrid(np.arange(H, dtype=F), np.arange(W, dtype=F), indexing="ij")
    u = (xs + F(0.5)) / F(W) * F(2) - F(1)
    v = (ys + F(0.5)) / F(H) * F(2) - F(1)
    # two tilted planes meeting in a crease at u == 0.15 v
    left = u < F(0.15) * v
    z_plane = np.where(left, F(0.55) + F(0.30) * u + F(0.08) * v, F(0.62) - F(0.22) * u + F(0.10) * v)
    nl = np.array([-0.45, -0.12, 0.88], F)
    nr = np.array([0.35, -0.15, 0.92], F)
    nl /= np.linalg.norm(nl)
    nr /= np.linalg.norm(nr)
    n = np.where(left[..., None], nl, nr).astype(F)
    # sphere in front
    cx, cy, r = F(0.25 + 2.0 * camera_shift), F(-0.1), F(0.42)
    aspect = F(W) / F(H)
    dx, dy = (u - cx) * aspect, (v - cy)
    rr = dx * dx + dy * dy
    inside = rr < r * r
    hz = np.sqrt(np.maximum(r * r - rr, F(0)))
    z = np.where(inside, F(0.40) - F(0.35) * hz, z_plane)
    ns = np.stack([dx, dy, hz], axis=-1) / r
    n = np.where(inside[..., None], ns, n).astype(F)
    z = np.clip(z, F(0.2), F(0.98)).astype(F)
    # a strip of "sky" (no geometry): depth 1.0, stencil 0, normal oct(0,0)
    sky = v < F(-0.92)
    depth = pack_depth_stencil(z)
    depth = np.where(sky, np.uint32(0x00FFFFFF), depth).astype(np.uint32)
    e = oct16_fast_pack(n)
    e = np.where(sky[..., None], F(0), e)
    e16 = e.astype(np.float16)
    normal = np.concatenate([e16, e16], axis=-1)  # .xy == .zw
    # smooth base radiance with a texture-like pattern and a bright region
    base = np.stack([
        F(0.6) + F(0.4) * np.sin(F(6.0) * u + F(1.0)) * np.cos(F(4.0) * v),
        F(0.5) + F(0.3) * np.cos(F(5.0) * u - F(2.0) * v),
        F(0.4) + F(0.35) * np.sin(F(3.0) * v + F(0.5)),
    ], axis=-1).astype(F)
    base = base * np.where(inside, F(2.5), F(1.0))[..., None]
    base = np.where(sky[..., None], F(8.0), base).astype(F)
    _ = seed
    return dict(depth=depth, normal=normal, base=base)


def synth_radiance(base, frame, seed=1234, sigma=1.0):
    """base[H,W,3] -> noisy RGBA32F radiance[H,W,4] (alpha = 1), log-normal(sigma) noise with mean 1."""
    H, W, _ = base.shape
    idx = np.arange(H * W, dtype=np.uint32).reshape(H, W)
    u1 = np.maximum(uniform01(seed, frame, idx, 1), F(1e-7))
    u2 = uniform01(seed, frame, idx, 2)
    g = np.sqrt(F(-2.0) * np.log(u1)) * np.cos(F(2.0 * np.pi) * u2)
    eta = np.exp(F(sigma) * g - F(0.5 * sigma * sigma)).astype(F)
    out = np.empty((H, W, 4), F)
    out[..., :3] = base * eta[..., None]
    out[..., 3] = F(1.0)
    return out

"""oracle/svgf_np.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED.

Independent numpy restatement of the reference SVGF shaders, written from the HLSL
text (not from oracle/svgf_ref.c) so the two restatements cross-check each other:

  assets/shaders/svgf_temporal.hlsl:24-68   -> temporal()
  assets/shaders/svgf_atrous.hlsl:29-85     -> atrous()
  assets/shaders/svgf_common.hlsli:4-35     -> weights, luminance
  assets/shaders/octahedron_encoding.hlsli:27-34 -> oct16_fast_unpack()
  src/SVGFDenoiser.cpp:39-64,133-203        -> SVGFStateNP frame logic

The reference has no tests or golden vectors for this path; only
tests/ may import this module.  All arithmetic is float32; fp16 planes are
numpy float16 (IEEE RNE with denormals, the D3D typed-store rule).
"""
import numpy as np

F = np.float32

DEFAULT_PARAMS = dict(
    depthSigma=F(0.002), alpha=F(0.9), varianceEps=F(1e-4),      # SVGFDenoiser.h:79-81
    phiColor=F(4.0) / F(255.0), phiNormal=F(128.0), phiDepth=F(0.002),  # SVGFDenoiser.h:89-91
)


def lerp(a, b, t):
    return a + t * (b - a)


def luminance(rgb):
    return rgb[..., 0] * F(0.2126) + rgb[..., 1] * F(0.7152) + rgb[..., 2] * F(0.0722)


def oct16_fast_unpack(e):
    """e[..., 2] float32 -> unit normal [..., 3] (octahedron_encoding.hlsli:27-34)."""
    ex, ey = e[..., 0], e[..., 1]
    vz = F(1.0) - np.abs(ex) - np.abs(ey)
    neg = vz < 0
    sx = np.where(ex > 0, F(1.0), F(-1.0))
    sy = np.where(ey > 0, F(1.0), F(-1.0))
    vx = np.where(neg, (F(1.0) - np.abs(ey)) * sx, ex)
    vy = np.where(neg, (F(1.0) - np.abs(ex)) * sy, ey)
    ln = np.sqrt(vx * vx + vy * vy + vz * vz)
    return np.stack([vx / ln, vy / ln, vz / ln], axis=-1).astype(F)


def depth_unorm24(d):
    return (d & np.uint32(0xFFFFFF)).astype(F) / F(16777215.0)


def shading_normal(normal_f16):
    """normal plane [H,W,4] float16 -> decoded .zw shading normal."""
    return oct16_fast_unpack(normal_f16[..., 2:4].astype(F))


def _dispatch_region(H, W):
    return (H // 8) * 8, (W // 8) * 8  # Dispatch(W/8, H/8): SVGFDenoiser.cpp:116,185


def temporal(rad_cur, rad_hist, depth_cur, depth_hist, normal_cur, normal_hist, mom_hist, p=DEFAULT_PARAMS,
             mom_cur=None, variance=None):
    """Returns (radiance_out[H,W,4] f32, moments_out[H,W,2] f16, variance[H,W] f16)."""
    H, W = depth_cur.shape
    Hd, Wd = _dispatch_region(H, W)
    out = rad_cur.copy()
    mom_out = np.zeros((H, W, 2), np.float16) if mom_cur is None else mom_cur.copy()
    var_out = np.zeros((H, W), np.float16) if variance is None else variance.copy()
    s = (slice(0, Hd), slice(0, Wd))
    Cc = rad_cur[s][..., :3].astype(F)
    Ch = rad_hist[s][..., :3].astype(F)
    Dc, Dh = depth_unorm24(depth_cur[s]), depth_unorm24(depth_hist[s])
    Nc, Nh = shading_normal(normal_cur[s]), shading_normal(normal_hist[s])
    Mh = mom_hist[s].astype(F)
    dz = np.abs(Dc - Dh)
    sig = F(p["depthSigma"])
    wD = np.exp(-dz * dz / (F(2.0) * sig * sig))
    wN = np.clip(np.sum(Nc * Nh, axis=-1, dtype=F), F(0), F(1))
    w = wD * wN
    a = lerp(F(1.0), F(p["alpha"]), w)
    Y = luminance(Cc)
    M1 = lerp(Y, Mh[..., 0], a)
    M2 = lerp(Y * Y, Mh[..., 1], a)
    var = np.maximum(M2 - M1 * M1, F(p["varianceEps"]))
    out[s][..., :3] = lerp(Cc, Ch, a[..., None])
    with np.errstate(over="ignore"):
        mom_out[s] = np.stack([M1, M2], axis=-1).astype(np.float16)
        var_out[s] = var.astype(np.float16)
    return out, mom_out, var_out


def atrous(rad_src, variance, depth_cur, normal_cur, step, p=DEFAULT_PARAMS, dst=None):
    """One edge-stopping 5x5 wavelet level.  Returns radiance_dst [H,W,4] f32."""
    H, W = depth_cur.shape
    Hd, Wd = _dispatch_region(H, W)
    out = np.zeros_like(rad_src) if dst is None else dst.copy()
    K = [F(1.0 / 16.0), F(1.0 / 4.0), F(3.0 / 8.0)]  # K[abs(d)] (quirk 1)
    c_all = rad_src[..., :3].astype(F)
    lum_all = luminance(c_all)
    z_all = depth_unorm24(depth_cur)
    n_all = shading_normal(normal_cur)
    ys, xs = np.arange(Hd), np.arange(Wd)
    c0, lum0 = c_all[:Hd, :Wd], lum_all[:Hd, :Wd]
    var = variance[:Hd, :Wd].astype(F)
    varScale = F(p["phiColor"]) * np.sqrt(np.maximum(var, F(1e-8)))
    denomL = np.maximum(varScale, F(1e-6))
    z0, n0 = z_all[:Hd, :Wd], n_all[:Hd, :Wd]
    denomZ = F(p["phiDepth"]) * F(step)
    sumC = np.zeros((Hd, Wd, 3), F)
    sumW = np.zeros((Hd, Wd), F)
    for dy in range(-2, 3):
        qy = np.clip(ys + dy * int(step), 0, H - 1)
        for dx in range(-2, 3):
            qx = np.clip(xs + dx * int(step), 0, W - 1)
            idx = np.ix_(qy, qx)
            c, lum, z, n = c_all[idx], lum_all[idx], z_all[idx], n_all[idx]
            wz = np.exp(-np.abs(z0 - z) / denomZ)
            d = n0[..., 0] * n[..., 0] + n0[..., 1] * n[..., 1] + n0[..., 2] * n[..., 2]
            wn = np.power(np.maximum(F(0), d), F(p["phiNormal"]))
            wl = np.exp(-np.abs(lum0 - lum) / denomL)
            w = K[abs(dx)] * K[abs(dy)] * wz * wn * wl
            sumC += w[..., None] * c
            sumW += w
    out[:Hd, :Wd, :3] = sumC / np.maximum(sumW, F(1e-4))[..., None]
    out[:Hd, :Wd, 3] = rad_src[:Hd, :Wd, 3]
    return out


class SVGFStateNP:
    """Frame-level mirror of SVGFDenoiser (src/SVGFDenoiser.cpp:39-64,66-203)."""

    def __init__(self, W, H, levels=4, params=None):
        self.W, self.H, self.levels = W, H, levels
        self.radiance = [np.zeros((H, W, 4), F) for _ in range(2)]
        self.normal = [np.zeros((H, W, 4), np.float16) for _ in range(2)]
        self.depth = [np.zeros((H, W), np.uint32) for _ in range(2)]
        self.moments = [np.zeros((H, W, 2), np.float16) for _ in range(2)]
        self.variance = np.zeros((H, W), np.float16)
        self.params = dict(DEFAULT_PARAMS if params is None else params)
        self.cur, self.hist = 0, 1

    def begin_frame(self, frame_index):
        self.cur = frame_index & 1
        self.hist = self.cur ^ 1

    def reset_history(self):
        self.radiance[self.hist] = self.radiance[self.cur].copy()

    def temporal_pass(self):
        c, h = self.cur, self.hist
        self.radiance[c], self.moments[c], self.variance = temporal(
            self.radiance[c], self.radiance[h], self.depth[c], self.depth[h], self.normal[c], self.normal[h],
            self.moments[h], self.params, mom_cur=self.moments[c], variance=self.variance)

    def atrous_pass(self):
        """src = cur, dst = hist, swap per level (SVGFDenoiser.cpp:146-196).  Odd level
        counts route through a third plane so the result still lands in radiance[cur]
        (SURVEY.md quirk 5); pixels outside the floor-dispatched region stay untouched
        in every destination plane."""
        c, h, L = self.cur, self.hist, self.levels
        if not hasattr(self, "scratch"):
            self.scratch = np.zeros((self.H, self.W, 4), F)
        bufs = {"cur": self.radiance[c], "hist": self.radiance[h], "scratch": self.scratch}
        if L == 1:
            chain = ["cur", "scratch"]
        elif L % 2 == 0:
            chain = ["cur", "hist"] * (L // 2) + ["cur"]
        else:
            chain = ["cur", "hist"] + ["scratch", "hist"] * ((L - 3) // 2) + ["scratch", "cur"]
        for i in range(L):
            s, d = chain[i], chain[i + 1]
            bufs[d] = atrous(bufs[s], self.variance, self.depth[c], self.normal[c], 1 << i, self.params, dst=bufs[d])
        if L == 1:
            bufs["cur"] = bufs["scratch"].copy()
        self.radiance[c], self.radiance[h], self.scratch = bufs["cur"], bufs["hist"], bufs["scratch"]

"""gi_np.py -- an INDEPENDENT numpy restatement of the GI path's per-ray arithmetic (TEST INFRASTRUCTURE ONLY).

Written from the text of the reference's shaders, not from oracle/trace_ref.cpp or the HIP kernels, so that a shared
misreading of the HLSL in those two (they were developed together) cannot pass unnoticed: tests compare BOTH against
this file (tests/test_oracle_gi.py, tests/test_gi_gpu.py) and against the golden vectors generated from it
(tests/golden/make_gi_golden.py -> tests/golden/gi_*.npz).  Nothing under nebulae_amd/ may import it.

Parity status: UNPINNED, like the rest of oracle/ -- the reference has no tests or fixtures for this path and cannot
be built or run here (Win32 + D3D12/DXR + DXC + the closed NVIDIA NRC DLL); this is a second reading of the same text.

What is restated (paths relative to the reference checkout):
  assets/shaders/pathtracer.hlsl:397-625   PathtracerRG, query variant, with the NRC calls replaced by the reference's own
                                           ENABLE_NRC = 0 stubs (rtxgi/Nrc.hlsli:579-621: NrcUpdateOnHit returns Continue)
  assets/shaders/pathtracer.hlsl:132-143   QueryReconstructedHemisphereRay
  assets/shaders/pathtracer.hlsl:180-259   PrepareBRDFData, EvaluateDirectBRDF, EvaluateIndirectBRDF (rng BY VALUE)
  assets/shaders/pathtracer.hlsl:299-395   ReconstructSurfaceData
  assets/shaders/brdf.hlsli                Brdf_*, Ndf_GGXTrowbridgeReitz, Gsf_*, Brdf_GetSpecularProbability,
                                           CosineSampleHemisphereSurfaceAligned
  assets/shaders/rand.hlsli:6-55           JenkinsHash, InitRNG, XorShift, UintToFloat, Rand, Rand2
  assets/shaders/sun_disk_sampling.hlsli:45-52  GetPerpendicularVector
  assets/shaders/octahedron_encoding.hlsli:27-41  Oct16_FastUnpack, UnpackOct16Normals
Ray/triangle intersection is brute force over all triangles (no acceleration structure at all), Moeller-Trumbore in
float64 on the float32 world-space triangles; TraceRay reports the CLOSEST hit (the build's documented divergence from
RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH with a closest-hit shader), the sun RayQuery any hit;
TMin < t < TMax (DXR: both ends exclusive).  Triangles are placed with the correct instance transform (quirk 12).
Everything else is float32 numpy; libm-level differences against the C oracle / the GPU are ~1e-7 relative.
Two deliberate divergences carried over from DESIGN.md 4, so that the three implementations are comparable: a
Cook-Torrance term whose denominator 4 VdotN LdotN is zero evaluates to 0 (the shader computes 0 * inf = NaN, which
NRC discards on the reference's screen), and the frame result is the mean over the samples, ADDED to the input radiance.
"""
import numpy as np

F = np.float32
U = np.uint32


# ---- rand.hlsli ----
def jenkins_hash(x):
    x = x.astype(U).copy()
    x += x << U(10)
    x ^= x >> U(6)
    x += x << U(3)
    x ^= x >> U(11)
    x += x << U(15)
    return x


def rand(state):
    """Rand(inout rng): advances `state` (uint32 array, in place) and returns floats in [0, 1)."""
    state ^= state << U(13)
    state ^= state >> U(17)
    state ^= state << U(5)
    return (U(0x3F800000) | (state >> U(9))).view(F) - F(1.0)


def rand2(state):
    a = rand(state)  # float2(Rand(rng), Rand(rng)): left to right
    b = rand(state)
    return a, b


# ---- small vector helpers (rows are vectors) ----
def dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]).astype(F)


def normalize(v):
    n = np.sqrt(np.sum(v * v, axis=-1, keepdims=True, dtype=F))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (v / n).astype(F)


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(F)


def saturate(x):
    return np.clip(x, F(0.0), F(1.0)).astype(F)


def lerp(a, b, t):
    return (a + t * (b - a)).astype(F)


# ---- octahedron_encoding.hlsli:27-34 ----
def oct16_fast_unpack(e):
    e = e.astype(F)
    z = F(1.0) - np.abs(e[..., 0]) - np.abs(e[..., 1])
    x, y = e[..., 0].copy(), e[..., 1].copy()
    neg = z < 0
    sx = np.where(x > 0, F(1.0), F(-1.0))
    sy = np.where(y > 0, F(1.0), F(-1.0))
    nx = (F(1.0) - np.abs(y)) * sx  # V.xy = (1 - abs(V.yx)) * sign(V.xy)
    ny = (F(1.0) - np.abs(x)) * sy
    x = np.where(neg, nx, x)
    y = np.where(neg, ny, y)
    return normalize(np.stack([x, y, z], -1).astype(F))


# ---- G-buffer formats (src/DeferredRenderer.cpp:758-770) ----
def unpack_r11g11b10(v):
    """DXGI_FORMAT_R11G11B10_FLOAT: unsigned small floats, 5-bit exponent (bias 15), 6 / 6 / 5 mantissa bits."""
    def small(bits, mbits):
        e = (bits >> U(mbits)).astype(np.int64)
        m = (bits & U((1 << mbits) - 1)).astype(np.float64) / float(1 << mbits)
        val = np.where(e == 0, m * 2.0 ** -14, (1.0 + m) * np.exp2(e - 15.0))
        val = np.where(e == 31, np.where(m == 0, np.inf, np.nan), val)
        return val.astype(F)
    v = v.astype(U)
    return np.stack([small(v & U(0x7FF), 6), small((v >> U(11)) & U(0x7FF), 6), small((v >> U(22)) & U(0x3FF), 5)], -1)


# ---- brdf.hlsli ----
PI = F(3.14159265)
PI_INV = F(1.0) / PI
PI_TWO = F(2.0) * PI


def luminance(c):
    return (c[..., 0] * F(0.2126) + c[..., 1] * F(0.7152) + c[..., 2] * F(0.0722)).astype(F)


def specular_f0(albedo, metalness):
    return lerp(np.full_like(albedo, F(0.04)), albedo, metalness[..., None])


def diffuse_reflectance(albedo, metalness):
    return (albedo * (F(1.0) - metalness)[..., None]).astype(F)


def fresnel_schlick(f0, vdoth):
    # F0 + (1 - F0) * (1 - pow(VdotH, 5)): as written (brdf.hlsli:22-25)
    k = (F(1.0) - np.power(vdoth, F(5.0), dtype=F))[..., None]
    return (f0 + (F(1.0) - f0) * k).astype(F)


def specular_probability(vdotn, f0, albedo):
    dr = luminance(albedo)
    fres = saturate(luminance(fresnel_schlick(f0, saturate(vdotn))))
    diff = dr * (F(1.0) - fres)
    p = diff / np.maximum(F(0.0001), fres + diff)
    return np.clip(p, F(0.1), F(0.9)).astype(F)


def cosine_sample_hemisphere_surface_aligned(u0, u1, sn):
    a = np.sqrt(u0, dtype=F)
    b = PI_TWO * u1
    zx, zy, zz = a * np.cos(b, dtype=F), a * np.sin(b, dtype=F), np.sqrt(F(1.0) - u0, dtype=F)
    up = np.where((np.abs(sn[..., 2]) < F(0.999))[..., None], np.array([0, 0, 1], F), np.array([1, 0, 0], F)).astype(F)
    tx = normalize(cross(up, sn))
    ty = cross(sn, tx)
    return normalize((zx[..., None] * tx + zy[..., None] * ty + zz[..., None] * sn).astype(F))


def evaluate_direct_brdf(albedo, roughness, metalness, sn, v, l):
    """EvaluateDirectBRDF (pathtracer.hlsl:209-228) through PrepareBRDFData (:180-206) and Brdf_Specular_CookTorrance."""
    h = normalize((v + l).astype(F))
    ldotn, vdoth, vdotn, ndoth = dot(l, sn), saturate(dot(v, h)), dot(v, sn), dot(sn, h)
    f0 = specular_f0(albedo, metalness)
    fr = fresnel_schlick(f0, saturate(vdoth))
    kd = F(1.0) - fr
    vn, ln, nh = saturate(vdotn), saturate(ldotn), saturate(ndoth)
    alpha = roughness * roughness
    a2 = alpha * alpha
    dist = (nh * nh) * (a2 - F(1.0)) + F(1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        ndf = a2 / (PI * dist * dist)
        k = alpha * F(0.5)
        gsf = (vn * (F(1.0) / (vn * (F(1.0) - k) + k))) * (ln * (F(1.0) / (ln * (F(1.0) - k) + k)))
        den = F(4.0) * vn * ln
        spec = (ndf * gsf)[..., None] * fr * (F(1.0) / den)[..., None]
    spec = np.where((den > 0)[..., None], spec, F(0.0)).astype(F)  # deliberate divergence: 0, not 0 * inf = NaN
    return (kd * (albedo * PI_INV) + spec).astype(F)


def perpendicular(u):
    a = np.abs(u)
    xm = ((a[..., 0] - a[..., 1]) < 0) & ((a[..., 0] - a[..., 2]) < 0)
    ym = np.where((a[..., 1] - a[..., 2]) < 0, ~xm, False)
    zm = ~(xm | ym)
    return cross(u, np.stack([xm, ym, zm], -1).astype(F))


# ---- scene ----
class FlatScene:
    """World-space triangle soup + the tables ReconstructSurfaceData reads, from a nebulae_amd.scene.Scene."""

    def __init__(self, scene):
        v0, v1, v2, geom, prim = [], [], [], [], []
        for gi, g in enumerate(scene.geometries):
            M = g["M"].astype(F)
            p = g["positions"].astype(F)
            w = (p[:, 0:1] * M[0, :3] + p[:, 1:2] * M[1, :3] + p[:, 2:3] * M[2, :3] + M[3, :3]).astype(F)  # (p, 1) * M, row vectors
            tri = g["indices"].reshape(-1, 3).astype(np.int64)
            v0.append(w[tri[:, 0]])
            v1.append(w[tri[:, 1]])
            v2.append(w[tri[:, 2]])
            geom.append(np.full(len(tri), gi, np.int64))
            prim.append(np.arange(len(tri), dtype=np.int64))
        cat = lambda xs, shape: np.concatenate(xs) if xs else np.zeros(shape)  # noqa: E731
        self.v0, self.v1, self.v2 = (cat(a, (0, 3)).astype(np.float64) for a in (v0, v1, v2))
        self.geom, self.prim = cat(geom, (0,)).astype(np.int64), cat(prim, (0,)).astype(np.int64)
        self.scene = scene

    def intersect(self, o, d, tmin, tmax, any_hit=False):
        """-> (t [inf = miss], triangle index, u, v) per ray; brute force."""
        n = o.shape[0]
        best_t = np.full(n, np.inf)
        best_i = np.full(n, -1, np.int64)
        best_u, best_v = np.zeros(n), np.zeros(n)
        if len(self.v0) == 0 or n == 0:
            return best_t, best_i, best_u, best_v
        o64, d64 = o.astype(np.float64), d.astype(np.float64)
        e1, e2 = self.v1 - self.v0, self.v2 - self.v0
        for a in range(0, n, 256):  # 256 rays x all triangles at a time
            oo, dd = o64[a:a + 256, None, :], d64[a:a + 256, None, :]
            p = np.cross(dd, e2[None])
            det = np.sum(e1[None] * p, -1)
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / det
                tv = oo - self.v0[None]
                u = np.sum(tv * p, -1) * inv
                q = np.cross(tv, e1[None])
                v = np.sum(dd * q, -1) * inv
                t = np.sum(e2[None] * q, -1) * inv
            lo = np.asarray(tmin, np.float64).reshape(-1)[a:a + 256, None] if np.ndim(tmin) else tmin
            ok = (det != 0) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > lo) & (t < tmax)
            t = np.where(ok, t, np.inf)
            i = np.argmin(t, axis=1)
            r = np.arange(t.shape[0])
            best_t[a:a + 256], best_i[a:a + 256] = t[r, i], np.where(np.isfinite(t[r, i]), i, -1)
            best_u[a:a + 256], best_v[a:a + 256] = u[r, i], v[r, i]
        return best_t, best_i, best_u, best_v

    # SampleLevel(s_MaterialSampler, uv, 0): linear filter, wrap addressing, mip 0, R8G8B8A8_UNORM (no sRGB decode)
    def sample(self, ti, uv):
        out = np.zeros((uv.shape[0], 4), F)
        for t in np.unique(ti):
            img = self.scene.textures[int(t)]
            h, w = img.shape[:2]
            m = ti == t
            x = uv[m, 0] * F(w) - F(0.5)
            y = uv[m, 1] * F(h) - F(0.5)
            x0, y0 = np.floor(x), np.floor(y)
            fx, fy = (x - x0).astype(F)[:, None], (y - y0).astype(F)[:, None]
            xi, yi = x0.astype(np.int64) % w, y0.astype(np.int64) % h
            x1, y1 = (xi + 1) % w, (yi + 1) % h
            tex = img.astype(F) / F(255.0)
            top = tex[yi, xi] + fx * (tex[yi, x1] - tex[yi, xi])
            bot = tex[y1, xi] + fx * (tex[y1, x1] - tex[y1, xi])
            out[m] = top + fy * (bot - top)
        return out

    def reconstruct(self, tri, bu, bv):
        """ReconstructSurfaceData (pathtracer.hlsl:299-395) for hits `tri` (indices into the soup) -> dict + valid mask"""
        n = tri.shape[0]
        GN, SN = np.zeros((n, 3), F), np.zeros((n, 3), F)
        albedo, rough, metal = np.zeros((n, 3), F), np.zeros(n, F), np.zeros(n, F)
        valid = np.zeros(n, bool)
        b1, b2 = bu.astype(F), bv.astype(F)
        b0 = F(1.0) - (b1 + b2)
        for gi in np.unique(self.geom[tri]):
            g = self.scene.geometries[int(gi)]
            m = self.geom[tri] == gi
            if any(g[k] is None for k in ("positions", "normals", "uvs", "tangents")):
                continue  # an InvalidIndex attribute buffer: return false (:313-318)
            idx = g["indices"].reshape(-1, 3).astype(np.int64)[self.prim[tri[m]]]
            w0, w1, w2 = b0[m, None], b1[m, None], b2[m, None]
            nrm = g["normals"].astype(F)
            gn = normalize((nrm[idx[:, 0]] * w0 + nrm[idx[:, 1]] * w1 + nrm[idx[:, 2]] * w2).astype(F))
            M = g["M"].astype(F)
            gn = normalize((gn[:, 0:1] * M[0, :3] + gn[:, 1:2] * M[1, :3] + gn[:, 2:3] * M[2, :3]).astype(F))  # mul(float4(GN, 0), surfaceToWorld)
            GN[m] = gn
            uvs = g["uvs"].astype(F)
            uv = (uvs[idx[:, 0]] * w0 + uvs[idx[:, 1]] * w1 + uvs[idx[:, 2]] * w2).astype(F)
            if g["material"] < 0:
                continue  # no material: return false (:349)
            mat = self.scene.materials[g["material"]]
            k = int(m.sum())
            ta, tn, tr = mat["textures"]
            albedo[m] = np.array(mat["albedo"][:3], F) if ta < 0 else self.sample(np.full(k, ta), uv)[:, :3]
            if tn < 0:
                SN[m] = gn
            else:
                tg = g["tangents"].astype(F)
                t4 = (tg[idx[:, 0]] * w0 + tg[idx[:, 1]] * w1 + tg[idx[:, 2]] * w2).astype(F)
                t4 = (t4 / np.sqrt(np.sum(t4 * t4, -1, keepdims=True, dtype=F))).astype(F)  # normalize(float4)
                bt = normalize((cross(gn, t4[:, :3]) * t4[:, 3:4]).astype(F))
                N = (self.sample(np.full(k, tn), uv)[:, :3] * F(2.0) - F(1.0)).astype(F)
                SN[m] = normalize((N[:, 0:1] * t4[:, :3] + N[:, 1:2] * bt + N[:, 2:3] * gn).astype(F))  # mul(N, float3x3(T, B, GN))
            if tr < 0:
                rough[m], metal[m] = F(mat["rm"][0]), F(mat["rm"][1])
            else:
                rm = self.sample(np.full(k, tr), uv)
                rough[m], metal[m] = rm[:, 1], rm[:, 2]  # .gb
            valid[m] = True
        return dict(GN=GN, SN=SN, albedo=albedo, roughness=rough, metalness=metal), valid


TRACING_MAX_DISTANCE = 10000.0


def trace(scene, gb, c, radiance_in=None):
    """PathtracerRG over the whole G-buffer `gb` (planes as nebulae_amd stores them) with constants `c`
    (nebulae_amd.scene.GIConstants).  -> dict(radiance [H,W,3] = radiance_in + mean over spp, and for the LAST sample's
    first bounce ray: t [< 0 miss], geometry, primitive, unoccluded [sun ray reached the sun])."""
    fs = scene if isinstance(scene, FlatScene) else FlatScene(scene)
    H, W = gb["albedo"].shape
    n = H * W
    yy, xx = np.meshgrid(np.arange(H, dtype=U), np.arange(W, dtype=U), indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    # InitRNG(loc, dim, frameIndex): dot(pixel, uint2(1, resolution.x)) ^ JenkinsHash(frame), hashed again (rand.hlsli:26-30)
    rng = jenkins_hash((xx + yy * U(W)) ^ jenkins_hash(np.array([c.frameIndex], U)))
    albedo = unpack_r11g11b10(gb["albedo"].reshape(-1))
    world_pos = gb["world_pos"].reshape(n, 4)[:, :3].astype(F)
    SN = oct16_fast_unpack(gb["normal"].reshape(n, 4)[:, 2:4])
    rm = gb["rough_metal"].reshape(n, 2).astype(F)
    metalness = rm[:, 1]
    cam = np.array(list(c.cameraWorldPos), F)
    sky, sun_dir, sun_rad = (np.array(list(v), F) for v in (c.skyColor, c.sunLightDirection, c.sunLightRadiance))
    V = (cam - world_pos).astype(F)  # NOT normalised, and it survives across the samples of a pixel (:431,522)
    total = np.zeros((n, 3), F)
    rec_t, rec_g, rec_p = np.full(n, -1.0, F), np.full(n, 0xFFFFFFFF, U), np.full(n, 0xFFFFFFFF, U)
    rec_un = np.zeros(n, bool)
    n_rays = 0
    for _ in range(int(c.samplesPerPixel)):
        rand(rng)  # NrcCreatePathState(g_NrcConstants, Rand(rng))
        throughput = np.ones((n, 3), F)
        radiance = np.zeros((n, 3), F)
        f0 = specular_f0(albedo, metalness)
        throughput *= diffuse_reflectance(albedo, metalness)
        with np.errstate(invalid="ignore"):
            dp = F(1.0) - specular_probability(saturate(dot(normalize(V), SN)), f0, albedo)
        take = rand(rng) < dp
        throughput[take] = (throughput[take] / dp[take, None]).astype(F)
        u0, u1 = rand2(rng)
        org = (world_pos + SN * F(1e-2)).astype(F)
        dirn = cosine_sample_hemisphere_surface_aligned(u0, u1, SN)
        tmin = np.full(n, 0.01)
        alive = np.ones(n, bool)
        rec_t[:], rec_g[:], rec_p[:], rec_un[:] = -1.0, 0xFFFFFFFF, 0xFFFFFFFF, False
        for bounce in range(1, int(c.maxPathVertices)):
            ids = np.nonzero(alive)[0]
            n_rays += len(ids)
            t, tri, bu, bv = fs.intersect(org[ids], dirn[ids], tmin[ids], TRACING_MAX_DISTANCE)
            hit = tri >= 0
            miss_ids = ids[~hit]
            radiance[miss_ids] += sky * throughput[miss_ids]
            alive[miss_ids] = False
            ids, t, tri, bu, bv = ids[hit], t[hit], tri[hit], bu[hit], bv[hit]
            if bounce == 1:
                rec_t[ids], rec_g[ids], rec_p[ids] = t, fs.geom[tri], fs.prim[tri]
            surf, ok = fs.reconstruct(tri, bu, bv)
            alive[ids[~ok]] = False  # ReconstructSurfaceData returned false: break
            ids, t = ids[ok], t[ok]
            surf = {k: v[ok] for k, v in surf.items()}
            if len(ids) == 0:
                break
            hitP = (org[ids] + dirn[ids] * t.astype(F)[:, None]).astype(F)
            V[ids] = normalize(-dirn[ids])
            # sun next-event estimation (:546-576)
            sub = rng[ids]
            a0, a1 = rand2(sub)
            rng[ids] = sub
            angle, dist = a0 * F(2.0) * F(3.1415926535), np.sqrt(a1, dtype=F)
            L = normalize(-sun_dir[None])[0]
            B = normalize(perpendicular(L[None]))[0]
            T = cross(B[None], L[None])[0]
            inc = normalize((L + (B * np.sin(angle, dtype=F)[:, None] + T * np.cos(angle, dtype=F)[:, None]) * F(c.sunTanHalfAngle) * dist[:, None]).astype(F))
            transition = dot(surf["GN"], inc) <= 0
            so = (hitP + np.where(transition[:, None], -surf["GN"], surf["GN"]) * F(1e-2)).astype(F)
            n_rays += len(ids)
            st, _, _, _ = fs.intersect(so, inc, 0.001, TRACING_MAX_DISTANCE)
            un = ~np.isfinite(st)
            O = evaluate_direct_brdf(surf["albedo"], surf["roughness"], surf["metalness"], surf["SN"], V[ids], np.broadcast_to(L, (len(ids), 3)))
            add = (O * sun_rad * throughput[ids]).astype(F)  # BRDF at the disk CENTRE, no N.L factor (:573-574)
            radiance[ids[un]] += add[un]
            if bounce == 1:
                rec_un[ids] = un
            if bounce == int(c.maxPathVertices) - 1:
                break
            # EvaluateIndirectBRDF (:230-259): rng by value -- its draws do not advance the path's stream
            copy = rng[ids].copy()
            SNn = normalize(surf["SN"])
            e0, e1 = rand2(copy)
            Ld = cosine_sample_hemisphere_surface_aligned(e0, e1, SNn)
            dprob = F(1.0) - specular_probability(saturate(dot(V[ids], SNn)), specular_f0(surf["albedo"], surf["metalness"]), surf["albedo"])
            org[ids] = (hitP + surf["GN"] * F(1e-2)).astype(F)
            dirn[ids] = Ld
            tmin[ids] = 0.001
            tp = (throughput[ids] * diffuse_reflectance(surf["albedo"], surf["metalness"])).astype(F)
            sub = rng[ids]
            take = rand(sub) < dprob
            rng[ids] = sub
            tp[take] = (tp[take] / dprob[take, None]).astype(F)
            throughput[ids] = tp
        total += radiance
    mean = (total * (F(1.0) / F(int(c.samplesPerPixel)))).astype(F)
    base = np.zeros((H, W, 3), F) if radiance_in is None else radiance_in[..., :3].astype(F)
    return dict(radiance=(base + mean.reshape(H, W, 3)).astype(F), t=rec_t.reshape(H, W), geometry=rec_g.reshape(H, W),
                primitive=rec_p.reshape(H, W), unoccluded=rec_un.reshape(H, W), rays=n_rays)

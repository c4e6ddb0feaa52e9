"""GPU parity tests for the GI path: HIP LBVH traverser + shading (through the C ABI) against the CPU oracle
(oracle/trace_ref.cpp, its own SAH BVH).  True closest hits and any-hit occlusion do not depend on the tree,
so hit ids must agree except where a ray grazes a triangle edge; RNG is integer arithmetic and bit-exact.
Tolerances: hit-id mismatches <= 2e-4 of pixels; radiance relative L2 <= 2e-3 overall (a mismatched pixel
contributes its whole value) and <= 2e-5 over the pixels whose hits agree (1 spp)."""
import ctypes as C

import numpy as np
import pytest
import torch

from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import (PLANE_ALBEDO, PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE, PLANE_ROUGH_METAL, PLANE_WORLDPOS,
                              SLOT_CURRENT, NebError)
from oracle_lib import OracleTracer
from svgf_cases import rel_l2

pytestmark = pytest.mark.gpu


def upload_gbuffer(r, gb):
    r.svgf.upload(PLANE_ALBEDO, 0, gb["albedo"])
    r.svgf.upload(PLANE_ROUGH_METAL, 0, gb["rough_metal"])
    r.svgf.upload(PLANE_WORLDPOS, 0, gb["world_pos"])
    r.svgf.upload(PLANE_NORMAL, SLOT_CURRENT, gb["normal"])
    r.svgf.upload(PLANE_DEPTH, SLOT_CURRENT, gb["depth"])


def _atrium_mixed_textures():
    """atrium_small with every other textured material given a normal map of another, non-square size: those materials
    cannot use the interleaved footprint table (DevMat::bundle) and sample their three maps separately."""
    sc = S.atrium_standin(target_triangles=30000, n_submeshes=60, tex_size=64)
    for k, m in enumerate(sc.materials):
        if m["textures"][1] >= 0 and k % 2 == 0:
            sc.textures[m["textures"][1]] = np.ascontiguousarray(S._proc_texture("normal", 900 + k, 48)[:32])  # 32 rows x 48 columns
    return sc


def scenes():
    return {
        "atrium_mixed_tex": (_atrium_mixed_textures, S.sponza_camera(), 320, 184),
        "cornell": (lambda: S.cornell_standin(textured=True), S.orbit_camera(), 256, 256),
        "cornell_factors": (lambda: S.cornell_standin(textured=False), S.orbit_camera(yaw_deg=10.0, pitch_deg=80.0, distance=2.6), 200, 152),
        "atrium_small": (lambda: S.atrium_standin(target_triangles=30000, n_submeshes=60, tex_size=64), S.sponza_camera(), 320, 184),
        # the real Sponza's pathology: full-length wall / floor / roof strips and beams across the court (aspect ratios up to 190:1)
        "atrium_longthin": (lambda: S.atrium_standin(target_triangles=60000, n_submeshes=70, tex_size=64, long_thin=True), S.sponza_camera(), 320, 184),
    }


@pytest.mark.parametrize("name,spp,sort_rays",
                         [(n, s, m) for n in ("cornell", "cornell_factors", "atrium_small") for s, m in ((1, 0), (3, 1), (1, 3), (2, 2))]
                         + [("atrium_mixed_tex", 1, 1), ("atrium_longthin", 1, 1), ("atrium_longthin", 2, 0)])
def test_gi_matches_oracle(name, spp, sort_rays):
    """sort_rays: the "gi_sort_rays" mask (bit 0 shadow rays, bit 1 bounce rays)."""
    make, cam, W, H = scenes()[name]
    sc = make()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=4)
    r.gi_ui.gi_samples_per_pixel = spp
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=5))
    assert r.scene_info()[0] == sc.num_triangles == o.triangles
    # reference splitting (gi_build.hip): only the long-thin scene has triangles far larger than the rest -- its strips and beams are
    # referenced by several leaves (176 bytes of triangle + shading record per reference); every other scene keeps one per triangle
    refs = r.scene_bytes()["triangles"] // 176
    assert (refs > sc.num_triangles) if name == "atrium_longthin" else (refs == sc.num_triangles), (refs, sc.num_triangles)
    upload_gbuffer(r, gb)
    base = np.full((H, W, 4), 0.25, np.float32)
    base[..., 3] = 1.0
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, base)
    r.set_debug_hits(True)
    r.svgf.set_option("gi_sort_rays", sort_rays & 3)
    r.ray_count(reset=True)
    r.submit_commands_gi_pathtrace()
    got = r.svgf.download(PLANE_RADIANCE)
    hits = r.download_hits()
    rays = r.ray_count()
    want, ohits, orays = o.gi(gb, r.global_constants(), radiance=base.copy())
    same = (hits["geometry"] == ohits["geometry"]) & (hits["primitive"] == ohits["primitive"]) & \
           ((hits["flags"] & 1) == (ohits["flags"] & 1))
    assert 1.0 - same.mean() <= 2e-4, f"hit mismatch fraction {1.0 - same.mean():.2e}"
    assert abs(rays - orays) <= max(4, 4e-4 * orays)
    assert np.array_equal(got[..., 3], base[..., 3])  # alpha untouched
    assert rel_l2(got[..., :3], want[..., :3]) <= 2e-3
    # (the hit record covers the last sample only: with spp > 1 an earlier sample may still differ)
    # long-thin: a 30-unit triangle puts 1e-7 * 30 of rounding into a hit point (tv = o - v0 cancels large coordinates), and the
    # device's and the host's bounce directions differ by an ulp of sinf / cosf: on this scene's 64^2 high-contrast maps the texel
    # fractions move by ~1e-3 on a few grazing hits (measured 3.7e-5 overall; the full-size scene with its 1024^2 maps: 1.2e-5 at
    # 1080p, 7 of 2 073 600 hits / flags different, tools/gi_stats.py)
    bar = 1e-4 if name == "atrium_longthin" else 2e-5
    assert rel_l2(got[same][:, :3], want[same][:, :3]) <= (bar if spp == 1 else 2e-4)
    t_err = np.abs(hits["t"][same] - ohits["t"][same]) / np.maximum(np.abs(ohits["t"][same]), 1e-6)
    assert t_err.max() <= 1e-4
    r.destroy()


def test_gbuffer_raycast_matches_oracle():
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    r.submit_commands_gbuffer()
    d = r.svgf.download(PLANE_DEPTH)
    # identical visibility except at triangle edges; depth within 2 D24 steps where visibility agrees
    covered = (d >> 24) == (gb["depth"] >> 24)
    assert covered.mean() >= 1.0 - 2e-4
    dz = np.abs((d & 0xFFFFFF).astype(np.int64) - (gb["depth"] & 0xFFFFFF).astype(np.int64))
    assert np.percentile(dz, 99.9) <= 4
    n = r.svgf.download(PLANE_NORMAL).astype(np.float32)
    assert np.percentile(np.abs(n - gb["normal"].astype(np.float32)), 99.5) <= 2e-3
    wp = r.svgf.download(PLANE_WORLDPOS).astype(np.float32)
    assert np.percentile(np.abs(wp - gb["world_pos"].astype(np.float32)), 99.5) <= 1e-2
    a = r.svgf.download(PLANE_ALBEDO)
    assert (a == gb["albedo"]).mean() >= 0.995
    rm = r.svgf.download(PLANE_ROUGH_METAL).astype(np.float32)
    assert np.percentile(np.abs(rm - gb["rough_metal"].astype(np.float32)), 99.5) <= 2e-3
    r.destroy()


def test_edge_cases_empty_scene_single_triangle_and_missing_attributes():
    W, H = 64, 48
    cam = S.orbit_camera()
    # (a) empty scene: every bounce ray misses -> sky * throughput
    empty = S.Scene("empty")
    r = DeferredRenderer()
    r.init(W, H)
    r.begin_frame(RenderInfo(scene=empty, camera=cam, frame_index=1))
    o = OracleTracer(S.cornell_standin(textured=True))
    gb = o.gbuffer(W, H, cam)
    upload_gbuffer(r, gb)
    r.submit_commands_gi_pathtrace()
    got = r.svgf.download(PLANE_RADIANCE)
    oe = OracleTracer(empty)
    want, _, _ = oe.gi(gb, r.global_constants())
    assert rel_l2(got[..., :3], want[..., :3]) <= 1e-6
    # (b) one triangle (LBVH with no inner node)
    one = S.Scene("one")
    m = one.add_material(albedo=(0.5, 0.5, 0.5, 1))
    one.add_geometry([[-5, 1.5, -5], [5, 1.5, -5], [0, 1.5, 5]], [[0, -1, 0]] * 3, [[0, 0], [1, 0], [0, 1]], [0, 1, 2], m)
    r.begin_frame(RenderInfo(scene=one, camera=cam, frame_index=2))
    upload_gbuffer(r, gb)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.set_debug_hits(True)
    r.submit_commands_gi_pathtrace()
    got = r.svgf.download(PLANE_RADIANCE)
    hits = r.download_hits()
    o1 = OracleTracer(one)
    want, ohits, _ = o1.gi(gb, r.global_constants())
    assert (hits["t"] > 0).any() and np.array_equal(hits["t"] > 0, ohits["t"] > 0)
    assert rel_l2(got[..., :3], want[..., :3]) <= 1e-5
    # (c) constants outside the supported range are rejected, not silently clamped
    r.gi_ui.max_path_vertices = 9
    with pytest.raises(NebError):
        r.submit_commands_gi_pathtrace()
    r.gi_ui.max_path_vertices = 2
    # (d) a vertex position that is not a finite number is refused at upload (the context then has no scene); a good scene after it works
    for bad in (float("nan"), float("inf")):
        broken = S.Scene("broken")
        mb = broken.add_material(albedo=(0.5, 0.5, 0.5, 1))
        broken.add_geometry([[-5, 1.5, -5], [5, bad, -5], [0, 1.5, 5]], [[0, -1, 0]] * 3, [[0, 0], [1, 0], [0, 1]], [0, 1, 2], mb)
        with pytest.raises(NebError):
            r.begin_frame(RenderInfo(scene=broken, camera=cam, frame_index=3))
    r.begin_frame(RenderInfo(scene=one, camera=cam, frame_index=3))
    upload_gbuffer(r, gb)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.submit_commands_gi_pathtrace()
    assert np.isfinite(r.svgf.download(PLANE_RADIANCE)).all()
    r.destroy()


def test_trace_before_scene_is_an_error():
    r = DeferredRenderer()
    r.init(64, 48)
    r.info = RenderInfo(scene=None, camera=S.orbit_camera(), frame_index=1)
    r.svgf.begin_frame(1)
    with pytest.raises(NebError):
        r.submit_commands_gi_pathtrace()
    r.destroy()


def test_direct_light_and_tonemap_match_oracle():
    """Rows f1 + f3: PBR direct light (overwrites radiance, one any-hit ray per pixel) and the ACES tonemap."""
    from nebulae_amd.svgf import PLANE_LDR
    from oracle_lib import oracle_pbr_direct, oracle_tonemap
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=9))
    upload_gbuffer(r, gb)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.full((H, W, 4), 7.0, np.float32))
    r.ray_count(reset=True)
    r.submit_commands_pbr_lighting()
    got = r.svgf.download(PLANE_RADIANCE)
    want, orays = oracle_pbr_direct(o, gb, r.global_constants())
    assert r.ray_count() == orays == W * H
    lit_same = (got[..., 0] > 0) == (want[..., 0] > 0)
    assert lit_same.mean() >= 1.0 - 3e-4                       # shadow-ray visibility agrees except at triangle edges
    assert np.array_equal(got[..., 3], np.ones((H, W), np.float32))
    assert rel_l2(got[lit_same][:, :3], want[lit_same][:, :3]) <= 2e-5
    # GI on top of the direct term, then tonemap
    r.submit_commands_gi_pathtrace()
    hdr = r.svgf.download(PLANE_RADIANCE)
    r.submit_commands_hdr_tonemapping()
    ldr = r.svgf.download(PLANE_LDR).view(np.uint8).reshape(H, W, 4)
    ref = oracle_tonemap(hdr)
    assert np.abs(ldr.astype(np.int16) - ref.astype(np.int16)).max() <= 1  # UNORM8 rounding of ~1e-7-different floats
    assert (ldr == ref).mean() >= 0.999
    r.destroy()


@pytest.mark.parametrize("sort_rays", [1, 3])
@pytest.mark.parametrize("max_vertices,spp", [(3, 1), (5, 2), (8, 1), (1, 1)])
def test_multi_bounce_matches_oracle(max_vertices, spp, sort_rays):
    """Row f4: the shader's bounce loop (pathtracer.hlsl:495-621) with the NRC stubs, up to 8 path vertices, including the
    by-value rng of EvaluateIndirectBRDF.  maxPathVertices = 1 traces nothing and adds nothing."""
    make, cam, W, H = scenes()["cornell"]
    sc = make()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H)
    r.gi_ui.gi_samples_per_pixel = spp
    r.gi_ui.max_path_vertices = max_vertices
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
    upload_gbuffer(r, gb)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.set_debug_hits(True)
    r.svgf.set_option("gi_sort_rays", sort_rays & 3)
    r.ray_count(reset=True)
    r.submit_commands_gi_pathtrace()
    got = r.svgf.download(PLANE_RADIANCE)
    hits = r.download_hits()
    rays = r.ray_count()
    want, ohits, orays = o.gi(gb, r.global_constants())
    if max_vertices == 1:
        assert rays == orays == 0 and float(np.abs(got).max()) == 0.0
    else:
        same = (hits["geometry"] == ohits["geometry"]) & (hits["primitive"] == ohits["primitive"])
        assert same.mean() >= 1.0 - 2e-4
        assert abs(rays - orays) <= max(8, 1e-3 * orays)
        assert rel_l2(got[..., :3], want[..., :3]) <= 5e-3      # a path that forks at a triangle edge changes its pixel entirely
        assert rel_l2(got[same][:, :3], want[same][:, :3]) <= 3e-3
        assert np.median(np.abs(got[..., :3] - want[..., :3])) <= 1e-6
    r.destroy()


@pytest.mark.parametrize("depth", [1, 2])
def test_deferred_resolve_and_two_stream_pipelining_are_bit_exact(depth):
    """neb_gi_trace with "gi_defer_resolve" + neb_gi_resolve == the fused dispatch, and running the GI stages of frame
    f+1 on a side stream while frame f is denoised gives the same frames bit for bit -- with one record set and one side stream, and with
    two of each ("gi_defer_resolve" = 2: the GI stages of frames f+1 and f+2 may both be in flight)."""
    import torch
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    outs = []
    for mode in ("fused", "pipelined"):
        r = DeferredRenderer()
        r.init(W, H, atrous_levels=4)
        main = torch.cuda.current_stream()
        sides = [torch.cuda.Stream() for _ in range(depth)]
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=main.cuda_stream))
        r.submit_commands_gbuffer()
        torch.cuda.synchronize()
        from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL
        for pl in (PLANE_NORMAL, PLANE_DEPTH):
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        rad = [r.svgf.plane_tensor(PLANE_RADIANCE, 0), r.svgf.plane_tensor(PLANE_RADIANCE, 1)]
        direct = torch.full_like(rad[0], 0.125)
        if mode == "pipelined":
            r.set_defer_resolve(depth)
        resolved = [None] * depth
        for f in range(2, 11):
            side, slot = sides[f % depth], f % depth
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
            cur = r.svgf.get_current_resource_index()
            if mode == "pipelined":
                if resolved[slot] is not None:
                    side.wait_event(resolved[slot])
                r.submit_commands_gi_pathtrace(stream=side.cuda_stream)
                rad[cur].copy_(direct, non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
                main.wait_event(done)
                r.submit_commands_gi_resolve()
                resolved[slot] = torch.cuda.Event()
                resolved[slot].record(main)
            else:
                rad[cur].copy_(direct, non_blocking=True)
                r.submit_commands_gi_pathtrace()
            r.submit_commands_svgf_denoising()
            r.end_frame()
        torch.cuda.synchronize()
        outs.append(r.svgf.download(PLANE_RADIANCE))
        r.destroy()
    assert float(np.abs(outs[0][..., :3]).max()) > 0.2
    assert np.array_equal(outs[0], outs[1])


def test_dispatch_in_two_calls_with_the_next_walk_beside_this_frames_shadow_pass_is_bit_exact():
    """neb_gi_trace_begin (ray generation + closest-hit walk) + neb_gi_trace_finish (shade + shadow passes) == neb_gi_trace, also when the walk of
    frame f + 1 runs on a side stream from the end of frame f's shade pass, beside its shadow pass and SVGF chain (bench.py's default form)."""
    from nebulae_amd.svgf import NebError
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    outs = []
    for mode in ("one call", "two calls"):
        r = DeferredRenderer()
        r.init(W, H, atrous_levels=4)
        main, side = torch.cuda.current_stream(), torch.cuda.Stream()
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=main.cuda_stream))
        r.submit_commands_gbuffer()
        torch.cuda.synchronize()
        for pl in (PLANE_NORMAL, PLANE_DEPTH):
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        rad = [r.svgf.plane_tensor(PLANE_RADIANCE, 0), r.svgf.plane_tensor(PLANE_RADIANCE, 1)]
        direct = torch.full_like(rad[0], 0.125)
        shaded = None
        for f in range(2, 11):
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
            cur = r.svgf.get_current_resource_index()
            if mode == "two calls":
                if shaded is not None:
                    side.wait_event(shaded)
                r.submit_commands_gi_pathtrace_begin(stream=side.cuda_stream)
                walked = torch.cuda.Event()
                walked.record(side)
                rad[cur].copy_(direct, non_blocking=True)
                main.wait_event(walked)
                shaded = torch.cuda.Event()
                shaded.record(main)
                r.submit_commands_gi_pathtrace_finish(after_shade_event=shaded.cuda_event)
            else:
                rad[cur].copy_(direct, non_blocking=True)
                r.submit_commands_gi_pathtrace()
            r.submit_commands_svgf_denoising()
            r.end_frame()
        torch.cuda.synchronize()
        outs.append(r.svgf.download(PLANE_RADIANCE))
        if mode == "two calls":  # misuse is refused, not guessed at
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=11, stream=main.cuda_stream))
            with pytest.raises(NebError):
                r.submit_commands_gi_pathtrace_finish()  # nothing begun
            r.submit_commands_gi_pathtrace_begin()
            r.submit_commands_gi_pathtrace_begin()
            with pytest.raises(NebError):
                r.submit_commands_gi_pathtrace_begin()  # both record sets taken
            with pytest.raises(NebError):
                r.submit_commands_gi_pathtrace()  # the one-call form while a dispatch is begun
            r.submit_commands_gi_pathtrace_finish()
            r.submit_commands_gi_pathtrace_finish()
            r.gi_ui.gi_samples_per_pixel = 2
            with pytest.raises(NebError):
                r.submit_commands_gi_pathtrace_begin()  # one sample per pixel only
            torch.cuda.synchronize()
        r.destroy()
    assert float(np.abs(outs[0][..., :3]).max()) > 0.2
    assert np.array_equal(outs[0], outs[1])


def test_two_record_sets_hold_two_dispatches_and_refuse_a_third():
    """"gi_defer_resolve" = 2: two traced dispatches may wait for their neb_gi_resolve at once (each on its own set of records, retired in the
    order they were traced); a third is refused, and so is a resolve with nothing pending or a change of the option in between."""
    from nebulae_amd.svgf import NebError
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    outs = []
    for mode in ("fused", "two sets"):
        r = DeferredRenderer()
        r.init(W, H, atrous_levels=2)
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
        r.submit_commands_gbuffer()
        rad = r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index())
        rad.zero_()
        if mode == "fused":
            r.submit_commands_gi_pathtrace()
        else:
            r.set_defer_resolve(2)
            r.submit_commands_gi_pathtrace()
            r.submit_commands_gi_pathtrace()  # (the same frame again: the same indirect term, in the other set)
            with pytest.raises(NebError):
                r.submit_commands_gi_pathtrace()
            with pytest.raises(NebError):
                r.set_defer_resolve(1)
            torch.cuda.synchronize()
            assert float(rad.abs().max()) == 0.0  # nothing has touched radiance[cur] yet
            r.submit_commands_gi_resolve()
            r.submit_commands_gi_resolve()
            with pytest.raises(NebError):
                r.submit_commands_gi_resolve()
            r.set_defer_resolve(0)
        torch.cuda.synchronize()
        outs.append(rad.cpu().numpy().copy())
        r.destroy()
    assert float(np.abs(outs[0][..., :3]).max()) > 0.05
    assert np.array_equal(outs[1][..., :3], 2.0 * outs[0][..., :3])  # s + s, exactly


def test_resize_recreates_planes_and_gi_buffers():
    """SVGFDenoiser::Resize re-creates (zeroes) the resources (src/SVGFDenoiser.cpp:28-37); after it a frame at the new
    size must equal the same frame on a fresh context."""
    sc = S.cornell_standin(textured=True)
    cam = S.orbit_camera()
    a = DeferredRenderer()
    a.init(96, 64, atrous_levels=3)
    a.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    a.submit_commands_gbuffer()
    a.submit_commands_gi_pathtrace()
    a.svgf.resize(160, 120)
    a.width, a.height = 160, 120
    b = DeferredRenderer()
    b.init(160, 120, atrous_levels=3)
    b.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))  # same dynamic-scene / reset-history state as `a`
    for r in (a, b):
        for f in (4, 5, 6):
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
            r.submit_commands_gbuffer()
            r.submit_commands_pbr_lighting()
            r.submit_commands_gi_pathtrace()
            r.submit_commands_svgf_denoising()
    ga, gb_ = a.svgf.download(PLANE_RADIANCE), b.svgf.download(PLANE_RADIANCE)
    assert ga.shape == (120, 160, 4) and float(np.abs(ga[..., :3]).max()) > 0
    assert np.array_equal(ga, gb_)
    a.destroy()
    b.destroy()


def test_gi_matches_oracle_at_the_bench_size():
    """BASELINE.json configs[2]: sponza-standin 1920x1080, 1 spp -- the full-size frame bench.py times."""
    W, H = 1920, 1080
    sc, cam = S.atrium_standin(), S.sponza_camera()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=5)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
    upload_gbuffer(r, gb)
    base = np.full((H, W, 4), 0.25, np.float32)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, base)
    r.set_debug_hits(True)
    r.ray_count(reset=True)
    r.submit_commands_gi_pathtrace()
    got, hits, rays = r.svgf.download(PLANE_RADIANCE), r.download_hits(), r.ray_count()
    want, ohits, orays = o.gi(gb, r.global_constants(), radiance=base.copy())
    same = (hits["geometry"] == ohits["geometry"]) & (hits["primitive"] == ohits["primitive"]) & \
           ((hits["flags"] & 1) == (ohits["flags"] & 1))
    # (measured: 4 of 2 073 600 -- ties on shared edges and silhouette-grazing rays whose direction differs by an ulp
    # between the device's and the host's sinf / cosf)
    assert 1.0 - same.mean() <= 2e-5, f"hit mismatch fraction {1.0 - same.mean():.2e}"
    assert abs(rays - orays) <= 8
    assert rel_l2(got[same][:, :3], want[same][:, :3]) <= 2e-5
    assert rel_l2(got[..., :3], want[..., :3]) <= 2e-3
    r.destroy()


@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 2160)])
def test_exact_policy_equals_the_oracle_bit_for_bit_except_exact_ties(W, H):
    """Round 4: the device and the C++ oracle compute sin / cos by the same fixed sequence of IEEE operations (det_sincosf) and x^5 as
    the same product, so under the oracle's arithmetic policy ("gi_exact_shade" = 1; ray generation always uses it) every ray is the same
    bits on both sides, every triangle test has the same operands, and the frames are EQUAL -- the only pixels that may differ are exact
    ties: a ray through a shared edge or vertex meets two triangles at the same t to the last bit, and which one is reported depends on
    the order the tree is walked in.  Under the default policy (1-ulp hardware rcp / rsq in the BRDF; the shadow ray's geometry exact
    either way) the discrete outcomes are the same and the radiance agrees to ~4e-6 of the frame's L2 norm."""
    sc, cam = S.atrium_standin(), S.sponza_camera()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    for exact in (1, 0):
        r = DeferredRenderer()
        r.init(W, H, atrous_levels=5)
        for f in (2, 7):
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
            r.svgf.set_option("gi_exact_shade", exact)
            upload_gbuffer(r, gb)
            r.submit_commands_pbr_lighting()
            direct = r.svgf.download(PLANE_RADIANCE)
            from oracle_lib import oracle_pbr_direct
            odirect, _ = oracle_pbr_direct(o, gb, r.global_constants())
            assert np.array_equal(direct, odirect)  # the direct term (its own shadow ray per pixel): the same bits
            r.set_debug_hits(True)
            r.submit_commands_gi_pathtrace()
            got, hits = r.svgf.download(PLANE_RADIANCE), r.download_hits()
            want, ohits, _ = o.gi(gb, r.global_constants(), radiance=odirect.copy())
            other = (hits["geometry"] != ohits["geometry"]) | (hits["primitive"] != ohits["primitive"])
            assert other.mean() <= 5e-6 and np.array_equal(hits["t"][other], ohits["t"][other]), (int(other.sum()), "only exact ties may differ")
            assert np.array_equal(hits["flags"] & 1, ohits["flags"] & 1)  # every sun-visibility flag
            if exact:
                assert np.array_equal(got[~other], want[~other])
            whole = rel_l2(got[..., :3], want[..., :3])
            print(f"[{W}x{H} exact_shade={exact} frame {f}] exact ties {int(other.sum())} px, whole-image rel-L2 {whole:.2e}")
            assert whole <= (1e-8 if exact else 2e-5)
        r.destroy()
    o.close()


def test_rebuild_and_refused_build_keep_a_valid_tree():
    """neb_gi_build_bvh commits only on success: a second build gives the same frame, and a build the depth limit refuses
    (NEB_ERR_OUT_OF_RANGE instead of a traversal stack that silently drops nodes) leaves the previous tree in place."""
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=7))
    depth = r.bvh_depth()
    assert 1 <= depth <= 21  # 3 * depth <= the 64-entry traversal stack
    upload_gbuffer(r, gb)

    def frame():
        r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
        r.submit_commands_gi_pathtrace()
        return r.svgf.download(PLANE_RADIANCE)

    first = frame()
    assert float(np.abs(first).max()) > 0
    r._check(r._lib.neb_gi_build_bvh(r._ctx, C.c_void_p(0)), "rebuild")     # a second build of the same scene
    assert r.bvh_depth() == depth and np.array_equal(frame(), first)
    r.svgf.set_option("gi_max_bvh_depth", 2)
    assert r._lib.neb_gi_build_bvh(r._ctx, C.c_void_p(0)) == -5            # NEB_ERR_OUT_OF_RANGE
    assert b"depth" in r._lib.neb_last_error(r._ctx)
    assert r.bvh_depth() == depth and np.array_equal(frame(), first)       # the old tree still answers
    r.svgf.set_option("gi_max_bvh_depth", 21)
    r._check(r._lib.neb_gi_build_bvh(r._ctx, C.c_void_p(0)), "rebuild")
    assert np.array_equal(frame(), first)
    with pytest.raises(NebError):
        r.svgf.set_option("gi_max_bvh_depth", 22)
    r.destroy()


@pytest.mark.parametrize("name", ["gi_cornell_tex_40x32", "gi_cornell_tex_multibounce_32x24", "gi_cornell_box_real_32x32",
                                  "gi_damaged_helmet_48x32", "gi_damaged_helmet_full_80x64"])
def test_hip_matches_the_numpy_restatement_golden(name):
    """The HIP path against the committed vectors of oracle/gi_np.py (tests/golden/make_gi_golden.py): an independent
    reading of the shaders with brute-force intersection -- not the C++ oracle the kernels were developed against."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_gi_golden as mk
    d = np.load(os.path.join(mk.HERE, name + ".npz"))
    sc = mk.case_scene(str(d["scene"]))
    H, W = d["albedo"].shape
    c = mk.case_constants(d)
    cam = S.orbit_camera()
    cam.eye[:] = [float(v) for v in d["eye"]]
    r = DeferredRenderer()
    r.init(W, H)
    r.gi_ui.gi_samples_per_pixel = int(d["spp"])
    r.gi_ui.max_path_vertices = int(d["max_path_vertices"])
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=int(d["frame_index"])))
    upload_gbuffer(r, {k: d[k] for k in mk.GB_KEYS})
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, d["radiance_in"])
    r.set_debug_hits(True)
    r.ray_count(reset=True)
    gc = r.global_constants()
    assert (gc.frameIndex, gc.samplesPerPixel, gc.maxPathVertices) == (c.frameIndex, c.samplesPerPixel, c.maxPathVertices)
    assert list(gc.cameraWorldPos) == list(c.cameraWorldPos) and gc.sunTanHalfAngle == c.sunTanHalfAngle
    r.submit_commands_gi_pathtrace()
    got, hits, rays = r.svgf.download(PLANE_RADIANCE), r.download_hits(), r.ray_count()
    same = (hits["geometry"] == d["geometry"]) & (hits["primitive"] == d["primitive"]) & (((hits["flags"] & 1) == 1) == d["unoccluded"])
    assert same.mean() >= 0.998, f"hit / sun-visibility mismatch on {(~same).sum()} of {same.size} pixels"
    assert abs(rays - int(d["rays"])) <= 2
    assert np.array_equal(got[..., 3], d["radiance_in"][..., 3])
    assert rel_l2(got[same][:, :3], d["radiance"][same]) <= 2e-5
    r.destroy()


def test_device_build_handles_degenerate_inputs():
    """The device SAH build on scenes that defeat spatial splitting: thousands of IDENTICAL triangles (all centroids
    coincide: the builder falls back to halving the run) and a long row of equal quads.  Which of the coincident
    triangles a ray reports is arbitrary, so distances and radiance are compared, not primitive ids."""
    W, H = 96, 64
    cam = S.orbit_camera()
    gb = OracleTracer(S.cornell_standin(textured=True)).gbuffer(W, H, cam)
    for kind in ("identical", "row", "two", "three", "five"):
        sc = S.Scene(kind)
        m = sc.add_material(albedo=(0.6, 0.5, 0.4, 1))
        if kind in ("two", "three", "five"):  # the smallest trees: a single leaf, a leaf + a pair, ...
            n = {"two": 2, "three": 3, "five": 5}[kind]
            x = np.arange(n, dtype=np.float32)[:, None] * 2.5 - 5.0
            q = np.array([[0, 1.8, -6], [2.4, 1.8, -6], [1.2, 1.8, 6]], np.float32)
            P = (q[None] + np.concatenate([x, np.zeros((n, 2), np.float32)], 1)[:, None, :]).reshape(-1, 3)
            I = np.arange(3 * n)
        elif kind == "identical":
            n = 3000
            P = np.tile(np.array([[-6, 1.8, -6], [6, 1.8, -6], [0, 1.8, 6]], np.float32), (n, 1))
            I = np.arange(3 * n)
        else:
            n = 4096
            x = np.arange(n, dtype=np.float32)[:, None] * 0.01 - 20.0
            q = np.array([[0, 1.8, -3], [0.01, 1.8, -3], [0.01, 1.8, 3], [0, 1.8, 3]], np.float32)
            P = (q[None] + np.concatenate([x, np.zeros((n, 2), np.float32)], 1)[:, None, :]).reshape(-1, 3)
            I = (np.arange(n)[:, None] * 4 + np.array([0, 1, 2, 0, 2, 3])[None]).reshape(-1)
        sc.add_geometry(P, np.tile(np.array([[0, -1, 0]], np.float32), (len(P), 1)), np.zeros((len(P), 2), np.float32), I, m)
        r = DeferredRenderer()
        r.init(W, H)
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=2))
        assert r.scene_info()[0] == len(I) // 3
        assert 3 * r.bvh_depth() <= 64 and r.build_passes() <= 64, (r.bvh_depth(), r.build_passes())
        upload_gbuffer(r, gb)
        r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
        r.set_debug_hits(True)
        r.submit_commands_gi_pathtrace()
        got, hits = r.svgf.download(PLANE_RADIANCE), r.download_hits()
        want, ohits, _ = OracleTracer(sc).gi(gb, r.global_constants())
        assert np.array_equal(hits["t"] > 0, ohits["t"] > 0) and (hits["t"] > 0).mean() > 0.02
        hit = hits["t"] > 0
        assert np.abs(hits["t"][hit] - ohits["t"][hit]).max() <= 1e-4 * np.abs(ohits["t"][hit]).max()
        assert rel_l2(got[..., :3], want[..., :3]) <= 2e-5
        r.destroy()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind", ["far", "tiny_far", "huge_ground"])
def test_quantised_nodes_stay_conservative_far_from_the_origin(kind):
    """The traversal walks a 64-byte copy of the tree whose child boxes are 8-bit offsets from the node's corner
    (Bvh4NodeQ, gi_internal.h), rounded outwards in the decoder's own arithmetic.  Where that rounding is hardest: a scene
    away from the origin (64 units: as far as the reference's fp16 world-position plane still resolves a 2-unit box; an
    ulp there is 8e-6) with a patch of 1e-3-sized quads, whose nodes have quantisation steps of 4e-6 -- below the ulp.
    The build must terminate and no hit may be lost: hit ids, visibility and distances equal the oracle's, which walks
    its own full-precision tree."""
    W, H = 128, 96
    off = np.array([64.0, 16.0, -32.0], np.float32)
    sc = S.cornell_standin(textured=True)
    for g in sc.geometries:  # (row-vector convention: world = p @ M[:3,:3] + M[3,:3])
        g["M"] = g["M"].copy()
        g["M"][3, :3] += off
    if kind == "tiny_far":  # 32 x 32 quads of edge 1e-3 hovering in the box
        n = 32
        ij = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 2).astype(np.float32)
        q = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1]], np.float32) * 1e-3
        base = np.concatenate([ij[:, :1] * 1.5e-3, np.zeros((n * n, 1), np.float32), ij[:, 1:] * 1.5e-3], 1)
        P = (q[None] + base[:, None, :]).reshape(-1, 3) + off[None, :] + np.array([[0.0, 0.4, 0.0]], np.float32)
        I = (np.arange(n * n)[:, None] * 4 + np.array([0, 1, 2, 0, 2, 3])[None]).reshape(-1)
        sc.add_geometry(P.astype(np.float32), np.tile(np.array([[0, 1, 0]], np.float32), (len(P), 1)), np.zeros((len(P), 2), np.float32), I,
                        sc.add_material(albedo=(0.7, 0.7, 0.2, 1)))
    if kind == "huge_ground":  # a 20 km ground quad under the box: the root's quantisation step is 80 units, 40 boxes wide
        g = 1.0e4
        P = np.array([[-g, -1.5, -g], [g, -1.5, -g], [g, -1.5, g], [-g, -1.5, g]], np.float32) + off[None, :]
        sc.add_geometry(P, np.tile(np.array([[0, 1, 0]], np.float32), (4, 1)), np.zeros((4, 2), np.float32), np.array([0, 2, 1, 0, 3, 2]),
                        sc.add_material(albedo=(0.3, 0.5, 0.3, 1)))
    cam = S.orbit_camera(origin=tuple(float(x) for x in off))
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
    assert r.scene_info()[0] == sc.num_triangles and 3 * r.bvh_depth() <= 64
    upload_gbuffer(r, gb)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.set_debug_hits(True)
    r.submit_commands_gi_pathtrace()
    got, hits = r.svgf.download(PLANE_RADIANCE), r.download_hits()
    want, ohits, _ = o.gi(gb, r.global_constants())
    assert (ohits["t"] > 0).mean() > 0.1
    # (the box has coincident coplanar triangles; which of two hits at the SAME distance is reported depends on the shape of
    # the tree, which the extra geometry changes: equal distance in the same geometry counts as the same hit)
    tie = (hits["geometry"] == ohits["geometry"]) & (hits["t"] == ohits["t"])
    same = (hits["geometry"] == ohits["geometry"]) & ((hits["primitive"] == ohits["primitive"]) | tie) & ((hits["flags"] & 1) == (ohits["flags"] & 1))
    assert 1.0 - same.mean() <= 1e-3, f"hit mismatch fraction {1.0 - same.mean():.2e}"
    # no hit may be LOST (a missed box would turn a hit into a miss or a farther hit): mismatches may only be ties
    lost = (ohits["t"] > 0) & ((hits["t"] <= 0) | (hits["t"] > ohits["t"] * (1 + 1e-3)))
    assert not lost.any(), int(lost.sum())
    assert rel_l2(got[same][:, :3], want[same][:, :3]) <= 2e-4
    r.destroy()


@pytest.mark.parametrize("direction", [(0.0, -1.0, 0.0), (1.0, 0.0, 0.0), (0.0, -1.0, -1.0)])
def test_axis_parallel_sun_rays(direction):
    """Shadow rays with exact zeros in their direction (a sun straight overhead or level with an axis, no disk: every ray is
    the same vector): 1/d is infinite on those axes and a plane distance becomes inf - inf.  The box test must then ignore
    the axis (v_min3 / v_max3 drop NaNs) -- conservatively, in the quantised decode (q * inf + (origin * inf - o * inf)) as
    in the float one: visibility and radiance equal the oracle's."""
    make, cam, W, H = scenes()["cornell"]
    sc = make()
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    r = DeferredRenderer()
    r.init(W, H)
    r.sun.direction = direction
    r.sun.rough_diameter = 0.0
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=7))
    upload_gbuffer(r, gb)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.set_debug_hits(True)
    r.submit_commands_gi_pathtrace()
    got, hits = r.svgf.download(PLANE_RADIANCE), r.download_hits()
    want, ohits, _ = o.gi(gb, r.global_constants())
    same = (hits["geometry"] == ohits["geometry"]) & (hits["primitive"] == ohits["primitive"]) & ((hits["flags"] & 1) == (ohits["flags"] & 1))
    assert 1.0 - same.mean() <= 2e-4, f"hit / visibility mismatch fraction {1.0 - same.mean():.2e}"
    assert rel_l2(got[same][:, :3], want[same][:, :3]) <= 2e-5
    # the direct-light pass traces the same kind of ray from the primary surfaces
    r.submit_commands_pbr_lighting()
    direct = r.svgf.download(PLANE_RADIANCE)
    from oracle_lib import oracle_pbr_direct
    dwant, _ = oracle_pbr_direct(o, gb, r.global_constants())
    lit_same = (direct[..., 0] > 0) == (dwant[..., 0] > 0)
    assert lit_same.mean() >= 1.0 - 1e-3, f"direct-light visibility mismatch fraction {1.0 - lit_same.mean():.2e}"
    if (dwant[..., 0] > 0).any():  # (a level sun leaves the closed box dark)
        assert rel_l2(direct[lit_same][:, :3], dwant[lit_same][:, :3]) <= 2e-5
    r.destroy()


def test_degenerate_normals_do_not_walk_the_whole_tree():
    """A G-buffer whose covered pixels carry normals that are not numbers (a slot nobody wrote, a broken producer) makes bounce
    rays with a NaN direction: every plane distance is NaN, the box test -- which drops NaNs -- passes every node, and the ray visits
    the whole tree without ever hitting a triangle: 0.1 s per wave on a 262 k-triangle scene (seen in a tool that forgot to fill the
    second G-buffer slot).  Such a ray does not start (it hits nothing): the dispatch takes its usual time, no node is visited."""
    import time
    import torch
    sc = S.atrium_standin(target_triangles=120000, n_submeshes=60, tex_size=64)
    cam = S.sponza_camera()
    W, H = 512, 288
    r = DeferredRenderer()
    r.init(W, H)
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
    r.submit_commands_gbuffer()
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    r.set_debug_hits(True)
    r.submit_commands_gi_pathtrace()  # (first launch of the process: not timed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.submit_commands_gi_pathtrace()
    torch.cuda.synchronize()
    usual = time.perf_counter() - t0
    r.ray_count()
    before = r.traversal_stats()
    assert before["bounce_nodes"] > 0
    n = r.svgf.download(PLANE_NORMAL)
    r.svgf.upload(PLANE_NORMAL, SLOT_CURRENT, np.full_like(n, np.nan))
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
    t0 = time.perf_counter()
    r.submit_commands_gi_pathtrace()
    torch.cuda.synchronize()
    degenerate = time.perf_counter() - t0
    r.ray_count()
    after = r.traversal_stats()
    assert after["bounce_nodes"] == before["bounce_nodes"], (before, after)  # no bounce ray walked the tree
    assert degenerate < 5.0 * usual + 2e-3, (degenerate, usual)
    r.destroy()


def test_two_call_dispatch_with_a_deferred_resolve_is_bit_exact():
    """neb_gi_trace_begin / _finish together with "gi_defer_resolve" = 1 (round 5): the GI chain of frame f on a side stream, back to back, its sums left in the
    record set the two-call form alternates; neb_gi_resolve takes the oldest set; the denoiser of frame f - 1 enqueued AFTER the walk of frame f (the context
    switched back to frame f - 1 for it).  The schedule of tools/frame_stagger.py -- measured slower than bench.py's form, kept as a supported order: same bits."""
    import torch
    from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    outs = []
    for mode in ("serial", "stagger"):
        r = DeferredRenderer()
        r.init(W, H, atrous_levels=4)
        main, side = torch.cuda.current_stream(), torch.cuda.Stream()
        info = lambda f: RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream)
        r.begin_frame(info(1))
        r.submit_commands_gbuffer()
        torch.cuda.synchronize()
        for pl in (PLANE_NORMAL, PLANE_DEPTH):
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        rad = [r.svgf.plane_tensor(PLANE_RADIANCE, 0), r.svgf.plane_tensor(PLANE_RADIANCE, 1)]
        direct = torch.full_like(rad[0], 0.125)
        if mode == "serial":
            for f in range(2, 10):
                r.begin_frame(info(f))
                rad[r.svgf.get_current_resource_index()].copy_(direct, non_blocking=True)
                r.submit_commands_gi_pathtrace()
                r.submit_commands_svgf_denoising()
                r.end_frame()
        else:
            r.set_defer_resolve(1)
            gi_done = None
            for f in range(2, 11):  # (the last pass only flushes frame 9's denoiser)
                walked = None
                if f < 10:
                    r.begin_frame(info(f))
                    r.submit_commands_gi_pathtrace_begin(stream=side.cuda_stream)
                    walked = torch.cuda.Event()
                    walked.record(side)
                if gi_done is not None:
                    r.begin_frame(info(f - 1))
                    rad[r.svgf.get_current_resource_index()].copy_(direct, non_blocking=True)
                    main.wait_event(gi_done)
                    r.submit_commands_gi_resolve()
                    if walked is not None:
                        main.wait_event(walked)
                    r.submit_commands_svgf_denoising()
                    r.end_frame()
                if f < 10:
                    r.begin_frame(info(f))
                    r.submit_commands_gi_pathtrace_finish(stream=side.cuda_stream)
                    gi_done = torch.cuda.Event()
                    gi_done.record(side)
        torch.cuda.synchronize()
        outs.append(r.svgf.download(PLANE_RADIANCE, slot=1))  # (frame 9's slot, whatever the context's current frame is)
        r.destroy()
    assert float(np.abs(outs[0][..., :3]).max()) > 0.2
    assert np.array_equal(outs[0], outs[1])

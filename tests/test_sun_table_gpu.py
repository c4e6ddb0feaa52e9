"""The sun-visibility table (nebulae_amd/csrc/gi_sun_table.hip, lit_predicate.h): shadow rays that start on a (triangle, side)
PROVEN lit by the whole sun disk are answered without a traversal.  Its bar is not a tolerance: with the table on and off the
frame must be the same bits -- radiance, per-pixel sun-visibility flags, ray counts -- on every GI scene, with several samples
and bounces, at the bench size, and after the sun has moved (the table is rebuilt for the new sun)."""
import os

import numpy as np
import pytest
import torch

from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
from nebulae_amd.svgf import PLANE_RADIANCE, SLOT_CURRENT
from test_gi_gpu import scenes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frame(r, sc, cam, W, H, f, spp=1, vertices=2):
    r.gi_ui.gi_samples_per_pixel = spp
    r.gi_ui.max_path_vertices = vertices
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
    r.submit_commands_gbuffer()
    base = np.full((H, W, 4), 0.125, np.float32)
    r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, base)
    r.set_debug_hits(True)
    r.ray_count(reset=True)
    r.submit_commands_gi_pathtrace()
    rad, hits, rays = r.svgf.download(PLANE_RADIANCE), r.download_hits(), r.ray_count()
    return rad, hits, rays


def _pair(W, H):
    on, off = DeferredRenderer(), DeferredRenderer()
    on.init(W, H)
    off.init(W, H)
    return on, off


def test_three_way_table_lists_sorted_tail_and_no_table():
    """"gi_sun_table" = 1 (table + compacted ray lists, the default), 2 (table, the remaining rays through the sorted pass) and
    0 (every ray traced): one frame sequence with 2 spp and 3 path vertices, the same bits from all three."""
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    rs = [DeferredRenderer() for _ in range(3)]
    for r, mode in zip(rs, (1, 2, 0)):
        r.init(W, H)
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
        r.svgf.set_option("gi_sun_table", mode)
        if mode == 2:
            r.svgf.set_option("gi_sort_rays", 1)
    for f in (2, 3):
        out = [_frame(r, sc, cam, W, H, f, 2, 3) for r in rs]
        _same(out[0], out[1])
        _same(out[0], out[2])
    for r in rs:
        r.destroy()


def _same(a, b):
    (ra, ha, na), (rb, hb, nb) = a, b
    assert np.array_equal(ra, rb), float(np.abs(ra - rb).max())
    for k in ("t", "geometry", "primitive"):
        assert np.array_equal(ha[k], hb[k]), k
    assert np.array_equal(ha["flags"] & 1, hb["flags"] & 1)  # sun visible: the table's answer == the traversal's
    assert na == nb                                         # a ray = a visibility query, however it is answered


@pytest.mark.parametrize("name,spp,vertices", [("atrium_small", 1, 2), ("atrium_small", 3, 2), ("atrium_small", 2, 4), ("atrium_mixed_tex", 1, 2), ("atrium_longthin", 1, 2),
                                               ("cornell", 2, 3), ("cornell_factors", 1, 2)])
def test_table_on_and_off_give_the_same_bits(name, spp, vertices):
    make, cam, W, H = scenes()[name]
    sc = make()
    on, off = _pair(W, H)
    for f in (3, 4):
        a = _frame(on, sc, cam, W, H, f, spp, vertices)
        if f == 3:
            off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
            off.svgf.set_option("gi_sun_table", 0)
        b = _frame(off, sc, cam, W, H, f, spp, vertices)
        _same(a, b)
    st, st_off = on.sun_table_stats(), off.sun_table_stats()
    assert st["builds"] == 1 and st_off["builds"] == 0 and st_off["rays_answered"] == 0
    assert 0.0 < on.sun_table_build_ms() < 1e3 and off.sun_table_build_ms() is None  # (device time of the build; nothing to report without one)
    if name.startswith("atrium"):  # an open court under the sun: a good share of the queries is answered by the table
        vis = (a[1]["flags"] & 1).astype(bool) & (a[1]["t"] > 0)
        assert st["lit_plus"] > 0.01 * sc.num_triangles and st["rays_answered"] >= 0.5 * vis.sum() * (1 if spp * (vertices - 1) == 1 else 0), (st, int(vis.sum()))
        print(f"[{name} spp={spp} vertices={vertices}] sides proven lit {st['lit_plus']} + {st['lit_minus']} of {sc.num_triangles} triangles; "
              f"{st['rays_answered']} of {a[2]} rays answered by the table; unoccluded (last sample / vertex 1) {int(vis.sum())}")
    on.destroy()
    off.destroy()


@pytest.mark.parametrize("file,tex", [("cornell_box.glb", 1), ("DamagedHelmet_256.glb", 2)])
def test_table_on_real_scenes(file, tex):
    """the reference's own assets: the Cornell box's node rotation, the helmet's curved surface with interpolated normals"""
    if not os.path.exists(os.path.join(GOLDEN, file)):
        pytest.skip(f"tests/golden/{file} is not present (an optional third-party asset: tests/golden/README.md)")
    sc = S.load_gltf(os.path.join(GOLDEN, file), tex_upscale=tex)
    cam = S.orbit_camera(origin=(0.0, 1.0, 0.0), distance=3.5) if "cornell" in file else S.orbit_camera(yaw_deg=20.0, pitch_deg=70.0, distance=2.6)
    W, H = 320, 200
    on, off = _pair(W, H)
    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    off.svgf.set_option("gi_sun_table", 0)
    for f in (2, 5):
        _same(_frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f))
    print(f"[{file}] {on.sun_table_stats()}")
    on.destroy()
    off.destroy()


def test_the_table_follows_the_sun():
    """sunLightDirection / the disk's diameter are per-frame constants (src/DeferredRenderer.cpp:403-421): a frame with another sun
    gets another table -- the same bits as a fresh context that never had one."""
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    on, off = _pair(W, H)
    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    off.svgf.set_option("gi_sun_table", 0)
    on.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    on.svgf.set_option("gi_sun_hold", 2)  # (a table for every sun that is seen twice, however short the last one lived: the default would wait for 32 sightings here)
    suns = [((0.5, -1.0, -0.2), 0.58), ((-0.3, -1.0, 0.4), 0.58), ((-0.3, -1.0, 0.4), 3.0), ((0.0, -1.0, 0.0), 0.0), ((0.5, -1.0, -0.2), 0.58)]
    f, builds = 2, 0
    for idx, (direction, diameter) in enumerate(suns):
        for r in (on, off):
            r.sun.direction, r.sun.rough_diameter = direction, diameter
        # a sun that has just moved is traced the plain way (no rebuild for a sun that is being dragged: the table follows once the
        # same sun is seen a second time); the first sun of a scene is built at once
        for rep in range(3):
            _same(_frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f))
            st = on.sun_table_stats()
            if (idx == 0 and rep == 0) or (idx > 0 and rep == 1):
                builds += 1
            assert st["builds"] == builds, (direction, rep, st)
            assert (st["rays_answered"] > 0) == (idx == 0 or rep >= 1), (direction, rep, st)
            f += 1
    # a sun dragged through five positions, one per frame: not one rebuild
    for k in range(5):
        for r in (on, off):
            r.sun.direction = (0.5 - 0.1 * k, -1.0, -0.2)
        _same(_frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f))
        f += 1
    assert on.sun_table_stats()["builds"] == builds + 0 and on.sun_table_stats()["rays_answered"] == 0
    for r in (on, off):
        r.sun.direction, r.sun.rough_diameter = suns[0]
    _frame(on, sc, cam, W, H, f)
    _frame(off, sc, cam, W, H, f)
    _same(_frame(on, sc, cam, W, H, f + 1), _frame(off, sc, cam, W, H, f + 1))
    builds = on.sun_table_stats()["builds"]
    # switching the option off clears the flags; on again rebuilds them
    on.svgf.set_option("gi_sun_table", 0)
    _same(_frame(on, sc, cam, W, H, 9), _frame(off, sc, cam, W, H, 9))
    assert on.sun_table_stats()["rays_answered"] == 0
    on.svgf.set_option("gi_sun_table", 1)
    _same(_frame(on, sc, cam, W, H, 10), _frame(off, sc, cam, W, H, 10))
    assert on.sun_table_stats()["rays_answered"] > 0 and on.sun_table_stats()["builds"] == builds + 1
    on.destroy()
    off.destroy()


def test_table_at_the_bench_size():
    """BASELINE.json configs[2]: sponza-standin 1920x1080 -- 2 M visibility queries per frame, three frames (three RNG streams)."""
    W, H = 1920, 1080
    sc, cam = S.atrium_standin(), S.sponza_camera()
    on, off = _pair(W, H)
    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    off.svgf.set_option("gi_sun_table", 0)
    # the tree of the one-workgroup-per-segment SAH build of rounds 2-3 (profiles/r03m_bench.json: 54 506 wide nodes, 14 levels):
    # the first levels' splits, now spread over the chip in slices (sah_big_* kernels), must give the very same tree
    assert off.scene_info() == (262244, 54506) and off.bvh_depth() == 14 and 0.0 < off.build_ms() < 200.0
    for f in (2, 3, 4):
        a, b = _frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f)
        _same(a, b)
    st = on.sun_table_stats()
    vis = (a[1]["flags"] & 1).astype(bool) & (a[1]["t"] > 0)
    print(f"[bench size] sides proven lit {st['lit_plus']} + {st['lit_minus']} of {sc.num_triangles} triangles; {st['rays_answered']} of {a[2]} rays "
          f"answered by the table = {st['rays_answered'] / max(int(vis.sum()), 1):.3f} of the {int(vis.sum())} unoccluded shadow rays")
    assert st["rays_answered"] >= 0.8 * vis.sum()
    torch.cuda.synchronize()
    on.destroy()
    off.destroy()


@pytest.mark.parametrize("scale,shift,table", [(1.0, (150.0, 40.0, -90.0), True), (100.0, (2000.0, 500.0, -1000.0), False)])
def test_table_far_from_the_origin_and_large(scale, shift, table):
    """The certificate's slack is kMarginUlps ulps of the scene's largest coordinate (lit_predicate.h), not round 4's absolute 2e-4: at
    +-170 units (an ulp is 1.5e-5; slack 1.9e-3 of the 1e-2 ray offset) the table is built and still the traversal's answer bit for bit;
    a scene x100 at +-3500 units (an ulp is 2.4e-4: the offset is 40 ulps) gets NO table -- nothing can be proven there with a margin
    worth the name -- and is traced the plain way.  Flags against the oracle's own trace of the same frame as well."""
    from oracle_lib import OracleTracer
    from test_gi_gpu import upload_gbuffer
    make, cam0, W, H = scenes()["atrium_small"]
    sc, cam = S.moved_scene(make(), scale, shift), S.moved_camera(cam0, scale, shift)
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    on, off = _pair(W, H)

    def frame(r, f):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
        upload_gbuffer(r, gb)
        r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.full((H, W, 4), 0.125, np.float32))
        r.set_debug_hits(True)
        r.ray_count(reset=True)
        r.submit_commands_gi_pathtrace()
        return r.svgf.download(PLANE_RADIANCE), r.download_hits(), r.ray_count()

    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    off.svgf.set_option("gi_sun_table", 0)
    for f in (3, 4):
        a, b = frame(on, f), frame(off, f)
        _same(a, b)
    st = on.sun_table_stats()
    assert (st["builds"] == 1) == table and (st["rays_answered"] > 0) == table, st
    if table:
        assert st["lit_plus"] > 0.01 * sc.num_triangles, st
    _, ohits, _ = o.gi(gb, on.global_constants())
    o.close()
    hit = (a[1]["t"] > 0) & (ohits["t"] > 0) & (a[1]["geometry"] == ohits["geometry"]) & (a[1]["primitive"] == ohits["primitive"])
    assert hit.mean() > 0.2
    wrong = hit & ((a[1]["flags"] & 1) != (ohits["flags"] & 1))
    assert wrong.sum() <= 1e-4 * hit.sum(), (int(wrong.sum()), int(hit.sum()))
    print(f"[x{scale:g} + {shift}] table {st}; {int(hit.sum())} common hits, {int(wrong.sum())} flags differ from the oracle's")
    on.destroy()
    off.destroy()


def test_no_table_for_a_20_km_ground_with_the_camera_5_km_out():
    """test_gi_gpu's huge_ground scene (a 20-km quad under the Cornell box) with a second box 5 km along x and the camera over it: hit points
    are org + dir * t at |coordinate| ~ 5000, an ulp there is 5e-4 -- the scene's size refuses the certificate (no build), table on == off
    and the flags are the oracle's."""
    from oracle_lib import OracleTracer
    from test_gi_gpu import upload_gbuffer
    W, H = 160, 96
    far = np.array([5000.0, 0.0, 0.0], np.float32)
    sc = S.cornell_standin(textured=True)
    for g in list(sc.geometries):
        h = dict(g)
        h["M"] = g["M"].copy()
        h["M"][3, :3] += far
        sc.geometries.append(h)
    gq = 1.0e4
    P = np.array([[-gq, -1.5, -gq], [gq, -1.5, -gq], [gq, -1.5, gq], [-gq, -1.5, gq]], np.float32)
    sc.add_geometry(P, np.tile(np.array([[0, 1, 0]], np.float32), (4, 1)), np.zeros((4, 2), np.float32), np.array([0, 2, 1, 0, 3, 2]),
                    sc.add_material(albedo=(0.3, 0.5, 0.3, 1)))
    cam = S.orbit_camera(origin=(5000.0, 0.0, -1.0), yaw_deg=25.0, pitch_deg=60.0, distance=6.0)
    o = OracleTracer(sc)
    gb = o.gbuffer(W, H, cam)
    on, off = _pair(W, H)
    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    off.svgf.set_option("gi_sun_table", 0)
    out = []
    for r in (on, off):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=5))
        upload_gbuffer(r, gb)
        r.svgf.upload(PLANE_RADIANCE, SLOT_CURRENT, np.zeros((H, W, 4), np.float32))
        r.set_debug_hits(True)
        r.ray_count(reset=True)
        r.submit_commands_gi_pathtrace()
        out.append((r.svgf.download(PLANE_RADIANCE), r.download_hits(), r.ray_count()))
    _same(out[0], out[1])
    assert on.sun_table_stats()["builds"] == 0 and on.sun_table_stats()["rays_answered"] == 0
    _, ohits, _ = o.gi(gb, on.global_constants())
    o.close()
    hits = out[0][1]
    hit = (hits["t"] > 0) & (ohits["t"] > 0) & (hits["geometry"] == ohits["geometry"]) & (hits["primitive"] == ohits["primitive"])
    assert hit.sum() > 200
    assert ((hits["flags"] & 1) == (ohits["flags"] & 1))[hit].mean() >= 1.0 - 1e-3
    on.destroy()
    off.destroy()


def test_a_new_sun_with_two_dispatches_in_flight_on_two_streams():
    """The table is rewritten in place, in the shading records, by a 15-ms launch on the stream of the dispatch that noticed the new sun, and
    the host calls it valid from the moment of the enqueue: the NEXT dispatch goes to the other side stream ("gi_defer_resolve" = 2) and
    must be ordered behind that launch (gi_sun_table_order), or its shade pass reads the old sun's lit bits of records the build has not
    reached yet.  No synchronisation between the frames around the change; the denoised sequence equals a single-stream run without a table."""
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    suns = {2: ((0.5, -1.0, -0.2), 0.58), 6: ((-0.3, -1.0, 0.4), 0.58), 11: ((0.2, -1.0, 0.1), 1.5)}
    outs = []
    for mode in ("plain", "two_streams"):
        r = DeferredRenderer()
        r.init(W, H, atrous_levels=4)
        main = torch.cuda.current_stream()
        sides = [torch.cuda.Stream() for _ in range(2)]
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=main.cuda_stream))
        r.submit_commands_gbuffer()
        torch.cuda.synchronize()
        from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL
        for pl in (PLANE_NORMAL, PLANE_DEPTH):
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        rad = [r.svgf.plane_tensor(PLANE_RADIANCE, 0), r.svgf.plane_tensor(PLANE_RADIANCE, 1)]
        direct = torch.full_like(rad[0], 0.125)
        if mode == "two_streams":
            r.set_defer_resolve(2)
            r.svgf.set_option("gi_sun_hold", 2)  # (three suns in fourteen frames: a table for each that is seen twice)
        else:
            r.svgf.set_option("gi_sun_table", 0)
        resolved = [None, None]
        for f in range(2, 16):
            if f in suns:
                r.sun.direction, r.sun.rough_diameter = suns[f]
            side, slot = sides[f % 2], f % 2
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
            cur = r.svgf.get_current_resource_index()
            if mode == "two_streams":
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
        if mode == "two_streams":
            st = r.sun_table_stats()
            assert st["builds"] == 3, st  # the first sun at once, the two later ones when seen a second time
        outs.append(r.svgf.download(PLANE_RADIANCE))
        r.destroy()
    assert float(np.abs(outs[0][..., :3]).max()) > 0.2
    assert np.array_equal(outs[0], outs[1])


def _tiny_scene(kind):
    """scenes whose tree is a single leaf or a root over two leaves: the build's walk starts (and ends) at a leaf code"""
    sc = S.Scene("tiny-" + kind)
    m = sc.add_material(albedo=(0.7, 0.6, 0.5, 1), rm=(0.8, 0.0))
    floor = S._quad((-2, -1, 1), (2, -1, 1), (2, -1, -3), (-2, -1, -3))  # faces +y, under the default sun (.5, -1, -.2)
    if kind == "one_triangle":
        P, N, UV, I = floor
        sc.add_geometry(P[:3].copy(), N[:3].copy(), UV[:3].copy(), np.array([0, 1, 2], np.uint32), material=m)
    elif kind == "quad":
        sc.add_geometry(*floor, material=m)
    elif kind == "roofed":  # a floor with a small roof over a part of it: lit and shadowed receivers, occluder hints
        sc.add_geometry(*S._merge([floor, S._quad((-0.5, 0.2, -0.5), (-0.5, 0.2, -1.5), (0.5, 0.2, -1.5), (0.5, 0.2, -0.5))]), material=m)
    else:  # the roofed floor + triangles without area: a point, a doubled vertex, three collinear vertices, and one a millionth of a unit wide
        P = np.array([[0.3, 0.0, -1.0]] * 3 + [[0.0, 0.1, -1.0], [0.0, 0.1, -1.0], [0.4, 0.1, -1.2]] + [[-1.0, 0.3, -2.0], [0.0, 0.3, -2.0], [1.0, 0.3, -2.0]]
                     + [[-0.2, 0.15, -0.8], [0.2, 0.15, -0.8], [0.0, 0.15, -0.800001]], np.float32)
        N = np.tile(np.array([[0.0, 1.0, 0.0]], np.float32), (12, 1))
        UV = np.zeros((12, 2), np.float32)
        slivers = (P, N, UV, np.arange(12, dtype=np.uint32))
        sc.add_geometry(*S._merge([floor, S._quad((-0.5, 0.2, -0.5), (-0.5, 0.2, -1.5), (0.5, 0.2, -1.5), (0.5, 0.2, -0.5)), slivers]), material=m)
    return sc


@pytest.mark.parametrize("kind", ["one_triangle", "quad", "roofed", "roofed_with_degenerate_triangles"])
def test_trees_of_one_or_two_leaves(kind):
    sc, cam, W, H = _tiny_scene(kind), S.orbit_camera(yaw_deg=15.0, pitch_deg=70.0, distance=3.0), 160, 120
    on, off = _pair(W, H)
    for f in (3, 4):
        a = _frame(on, sc, cam, W, H, f, 2, 3)
        if f == 3:
            off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
            off.svgf.set_option("gi_sun_table", 0)
        b = _frame(off, sc, cam, W, H, f, 2, 3)
        _same(a, b)
    st = on.sun_table_stats()
    assert st["builds"] == 1 and st["lit_plus"] + st["lit_minus"] >= 1, st  # (the open floor is proven lit on its upper side)
    assert np.isfinite(a[0]).all()
    print(f"[tiny {kind}] {sc.num_triangles} triangles, {on.scene_info()[1]} nodes: {st}")
    on.destroy()
    off.destroy()


def test_hints_reach_occluders_in_a_scene_of_2_6_million_triangles():
    """A hint is a 23-bit triangle index.  In leaf order the TOP of a scene -- what shadows a sun ray -- sorts last: with round 4's 21 bits nearly every
    occluder of a 2.6 M-triangle scene was out of range and the hints answered nothing (0.59 M of 4.0 M queries, the lit bits alone, against 1.75 M)."""
    W, H = 640, 360
    sc, cam = S.atrium_standin(target_triangles=2_600_000, tex_size=64), S.sponza_camera()
    assert sc.num_triangles > (1 << 21)
    on, off = _pair(W, H)
    a = _frame(on, sc, cam, W, H, 3)
    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=3))
    off.svgf.set_option("gi_sun_table", 0)
    b = _frame(off, sc, cam, W, H, 3)
    _same(a, b)
    st = on.sun_table_stats()
    shadow_queries = a[2] - W * H  # (a ray = a query: one bounce ray per pixel + one sun-visibility query per hit)
    print(f"[2.6 M triangles] {st}; {st['rays_answered']} of {shadow_queries} sun-visibility queries answered by the table")
    assert st["rays_answered"] > 0.7 * shadow_queries, (st, shadow_queries)
    on.destroy()
    off.destroy()


def test_no_table_for_a_wide_sun_disk_and_the_tail_is_chosen_by_measurement():
    """A disk of 10 degrees gets no table (its columns widen with the disk: 8 ms of build and no gain at 5 degrees on the bench scene, 70 ms at 30) -- every ray
    is traced, same bits.  Back under the reference's 0.58-degree sun the table is built, and which pass takes the rays it leaves -- the compacted lists or the
    sorted pass -- is decided from two timed dispatches: undecided for the first two frames of a new table, decided afterwards, the same bits throughout."""
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    on, off = _pair(W, H)
    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    off.svgf.set_option("gi_sun_table", 0)
    for r in (on, off):
        r.sun.rough_diameter = 10.0
    for f in (2, 3, 4):
        _same(_frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f))
    st = on.sun_table_stats()
    assert st["builds"] == 0 and st["rays_answered"] == 0 and on.sun_table_build_ms() is None, st
    for r in (on, off):
        r.sun.rough_diameter = 0.58
    modes = []
    for f in range(5, 12):
        _same(_frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f))
        torch.cuda.synchronize()
        modes.append(on.shadow_tail_mode())
    st = on.sun_table_stats()
    assert st["builds"] == 1 and st["rays_answered"] > 0, st
    # frame 5: built at once (there was no table to keep meanwhile), the lists run for the first time (untimed: they are allocated there); 6: the lists
    # timed; 7: the sorted pass timed; 8: both times read
    assert [m for m, _ in modes[:3]] == [-1, -1, -1] and all(m in (0, 1) for m, _ in modes[3:]), modes
    lists_us, sorted_us = modes[-1][1]
    assert lists_us > 0.0 and sorted_us > 0.0 and (modes[-1][0] == 1) == (sorted_us < 0.95 * lists_us), modes[-1]
    print(f"[tail] lists {lists_us:.0f} us, sorted pass {sorted_us:.0f} us -> {'sorted pass' if modes[-1][0] else 'lists'}")
    # the forced forms give the same bits
    for mode in (3, 2, 1):
        on.svgf.set_option("gi_sun_table", mode)
        for f in (12, 13):
            _same(_frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f))
    on.destroy()
    off.destroy()


def test_a_sun_that_moves_in_steps_is_not_built_for_at_every_step():
    """A build costs five frames' time and earns a tenth of a frame per dispatch.  A new sun gets its table when it has been seen twice -- unless the table it
    replaces served fewer than 32 dispatches: then when it has held for 32 (a sun stepping every four frames with a build per step made the frames twice as
    slow as no table at all).  Same bits whatever the table does."""
    make, cam, W, H = scenes()["atrium_small"]
    sc = make()
    on, off = _pair(W, H)
    off.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
    off.svgf.set_option("gi_sun_table", 0)
    f = 2

    def run(direction, frames, check_every=1):
        nonlocal f
        for r in (on, off):
            r.sun.direction = direction
        for k in range(frames):
            if k % check_every == 0 or k == frames - 1:
                _same(_frame(on, sc, cam, W, H, f), _frame(off, sc, cam, W, H, f))
            else:
                _frame(on, sc, cam, W, H, f)
            f += 1
        return on.sun_table_stats()["builds"]

    assert run((0.5, -1.0, -0.2), 4) == 1            # the first sun: at once
    assert run((0.4, -1.0, -0.2), 4) == 1            # its table served 3 dispatches when the sun moved on: from now a sun must hold for 32
    assert run((0.3, -1.0, -0.2), 4) == 1
    assert run((0.2, -1.0, -0.2), 4) == 1
    assert run((0.1, -1.0, -0.2), 31, check_every=10) == 1
    assert run((0.1, -1.0, -0.2), 2) == 2            # seen 32 times: built
    assert run((0.1, -1.0, -0.2), 40, check_every=13) == 2   # a table that serves long ...
    assert run((0.0, -1.0, -0.2), 2) == 3            # ... is followed by one built at the second sighting again
    on.destroy()
    off.destroy()

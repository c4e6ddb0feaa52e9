"""CPU tests: the C oracle against the numpy restatement and the committed golden frames."""
import numpy as np
import pytest

from oracle import svgf_np
from oracle_lib import OracleSVGF, lib
from svgf_cases import GOLDEN, frame_inputs, half_ulp_mismatch, load_golden, rel_l2


def test_fp16_conversions_match_ieee():
    L = lib()
    rng = np.random.default_rng(7)
    xs = (rng.standard_normal(20000) * 10.0 ** rng.integers(-9, 6, 20000)).astype(np.float32)
    edge = np.array([0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, 65504, 65519.9, 65520, 1e9, -1e-7, np.inf,
                     -np.inf], np.float32)
    xs = np.concatenate([xs, edge])
    with np.errstate(over="ignore"):
        ref = xs.astype(np.float16).view(np.uint16)
    got = np.array([L.svgf_ref_f32_to_f16(float(x)) for x in xs], np.uint16)
    assert (ref == got).all()
    hs = np.arange(0, 65536, 7, dtype=np.uint16)
    back = np.array([L.svgf_ref_f16_to_f32(int(h)) for h in hs], np.float32)
    refb = hs.view(np.float16).astype(np.float32)
    assert ((back == refb) | (np.isnan(back) & np.isnan(refb))).all()


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.split("/")[-1])
def test_c_oracle_matches_golden(path):
    z, W, H, L, frames, shift = load_golden(path)
    o = OracleSVGF(W, H, L)
    for f in range(1, frames + 1):
        g, rad = frame_inputs(W, H, f, shift)
        o.begin_frame(f)
        c = o.cur
        o.depth[c][...] = g["depth"]
        o.normal[c][...] = g["normal"]
        o.radiance[c][...] = rad
        o.temporal_pass()
        assert rel_l2(o.radiance[c], z[f"temporal_{f}"]) < 1e-6
        assert half_ulp_mismatch(o.moments[c], z[f"moments_{f}"]) == 0.0
        assert half_ulp_mismatch(o.variance, z[f"variance_{f}"]) == 0.0
        o.atrous_pass()
        assert rel_l2(o.radiance[c], z[f"denoised_{f}"]) < 2e-6, f
    # frame 1 starts from zero history: quirk 2 keeps 100 % history -> black frame
    assert float(np.abs(z["denoised_1"][..., :3]).max()) == 0.0


@pytest.mark.parametrize("W,H,L", [(40, 24, 1), (40, 24, 2), (70, 53, 3), (64, 40, 6)])
def test_c_oracle_matches_numpy_odd_sizes_and_levels(W, H, L):
    """Non-multiple-of-8 sizes (floor dispatch), 1 level, and 6 levels (step 32 > image/2)."""
    o, n = OracleSVGF(W, H, L), svgf_np.SVGFStateNP(W, H, L)
    for f in range(1, 4):
        g, rad = frame_inputs(W, H, f, 3)
        for s in (o, n):
            s.begin_frame(f)
            c = s.cur
            s.depth[c][...] = g["depth"]
            s.normal[c][...] = g["normal"]
            s.radiance[c][...] = rad
            s.temporal_pass()
            s.atrous_pass()
        assert rel_l2(o.radiance[o.cur], n.radiance[n.cur]) < 2e-6
    if W % 8:
        # columns beyond (W/8)*8 are never written by the floor-dispatched passes
        assert np.array_equal(o.radiance[o.cur][:, (W // 8) * 8:], rad[:, (W // 8) * 8:])


def test_reset_history_copies_radiance_only():
    o = OracleSVGF(32, 16, 4)
    g, rad = frame_inputs(32, 16, 1, None)
    o.begin_frame(1)
    o.radiance[o.cur][...] = rad
    o.moments[o.hist][...] = 3.0
    o.reset_history()
    assert np.array_equal(o.radiance[o.hist], rad)
    assert float(o.moments[o.hist].min()) == 3.0  # moments are NOT reset (SVGFDenoiser.cpp:57)


def test_threaded_oracle_equals_scalar():
    a, b = OracleSVGF(64, 48, 4, threads=1), OracleSVGF(64, 48, 4, threads=4)
    for f in (1, 2, 3):
        g, rad = frame_inputs(64, 48, f, None)
        for s in (a, b):
            s.begin_frame(f)
            c = s.cur
            s.depth[c][...] = g["depth"]
            s.normal[c][...] = g["normal"]
            s.radiance[c][...] = rad
            s.temporal_pass()
            s.atrous_pass()
    assert np.array_equal(a.radiance[a.cur], b.radiance[b.cur])

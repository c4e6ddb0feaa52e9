"""raysort.hip (the GI ray-reordering sort) against numpy's stable sort, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(lib, ctx, keys, vals, bits):
    out = np.empty_like(vals)
    rc = lib.neb_debug_sort_pairs(ctx, keys.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), C.c_uint32(keys.size), bits,
                                  out.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.neb_last_error(ctx)
    return out


@pytest.fixture(scope="module")
def ctx():
    from nebulae_amd.svgf import SVGFDenoiser
    d = SVGFDenoiser()
    d.init(64, 64)
    yield d._lib, d._ctx
    d.destroy()


@pytest.mark.parametrize("n", [1, 63, 64, 4095, 4096, 4097, 100_003, 1920 * 1080])
@pytest.mark.parametrize("bits", [16, 12, 8, 5])
def test_sort_is_the_stable_sort(ctx, n, bits):
    lib, c = ctx
    rng = np.random.default_rng(n * 31 + bits)
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)  # bits at and above `bits` must be ignored
    vals = np.arange(n, dtype=np.uint32)
    got = _sort(lib, c, keys, vals, bits)
    want = vals[np.argsort(keys & np.uint32((1 << bits) - 1), kind="stable")]
    assert np.array_equal(got, want)


def test_sort_skewed_keys(ctx):
    """Ray keys are not uniform: most pixels of a sky tile carry the 'no ray' key, others cluster in a few cells."""
    lib, c = ctx
    n = 300_000
    rng = np.random.default_rng(5)
    keys = np.where(rng.random(n) < 0.6, 0xFFFF, rng.integers(0, 40, n)).astype(np.uint32)
    keys[: 3 * 4096] = 0xFFFF  # whole tiles of one digit
    vals = rng.permutation(n).astype(np.uint32)
    got = _sort(lib, c, keys, vals, 16)
    assert np.array_equal(got, vals[np.argsort(keys, kind="stable")])

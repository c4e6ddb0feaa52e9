"""Shared SVGF test drivers: run a denoiser state (oracle, numpy or HIP) over seeded frames."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import frame_inputs  # noqa: E402

GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "svgf_*.npz")))


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def half_ulp_mismatch(a, b, max_ulp=1, abs_tol=1e-6):
    """fraction of fp16 values that differ by more than max_ulp fp16 ulps of the expected value
    (values whose difference is below abs_tol -- fp16-denormal rounding noise -- always agree)"""
    af = a.astype(np.float64)
    bf = b.astype(np.float64)
    ulp = np.maximum(np.abs(bf) * 2.0 ** -10, 2.0 ** -24)
    return float((np.abs(af - bf) > np.maximum(max_ulp * ulp, abs_tol)).mean())


def load_golden(path):
    z = np.load(path)
    W, H, L, frames, shift = [int(v) for v in z["meta"]]
    return z, W, H, L, frames, (None if shift < 0 else shift)

"""Generates tests/golden/svgf_*.npz from the numpy restatement (oracle/svgf_np.py).

The reference ships no golden vectors for this path ("parity unpinned"), so
these fixtures pin OUR two restatements (numpy here, C in oracle/svgf_ref.c) and the HIP
kernels against each other on fixed seeds.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nebulae_amd import synth  # noqa: E402
from oracle import svgf_np  # noqa: E402

CASES = [
    # name, W, H, levels, frames, frame with a camera shift (disocclusion) or None
    ("svgf_64x48_L4", 64, 48, 4, 4, 3),
    ("svgf_96x64_L5", 96, 64, 5, 3, None),
    ("svgf_72x40_L3", 72, 40, 3, 3, 2),
]


def frame_inputs(W, H, f, shift_frame):
    g = synth.synth_gbuffer(W, H, camera_shift=0.03 if f == shift_frame else 0.0)
    return g, synth.synth_radiance(g["base"], f)


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, W, H, L, frames, shift in CASES:
        st = svgf_np.SVGFStateNP(W, H, L)
        data = {"meta": np.array([W, H, L, frames, -1 if shift is None else shift], np.int32)}
        h = hashlib.sha256()
        for f in range(1, frames + 1):
            g, rad = frame_inputs(W, H, f, shift)
            for a in (g["depth"], g["normal"], rad):
                h.update(np.ascontiguousarray(a).tobytes())
            st.begin_frame(f)
            c = st.cur
            st.depth[c][...] = g["depth"]
            st.normal[c][...] = g["normal"]
            st.radiance[c][...] = rad
            st.temporal_pass()
            data[f"temporal_{f}"] = st.radiance[c].copy()
            data[f"moments_{f}"] = st.moments[c].copy()
            data[f"variance_{f}"] = st.variance.copy()
            st.atrous_pass()
            data[f"denoised_{f}"] = st.radiance[c].copy()
        data["input_sha256"] = np.frombuffer(h.digest(), np.uint8)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **data)
        print(name, "written")


if __name__ == "__main__":
    main()

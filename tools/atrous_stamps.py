#!/usr/bin/env python3
"""Diagnostic only: where a wave of svgf_atrous_lds_kernel spends its time.

Needs a library built with -DNEB_ATROUS_STAMPS=1 (tools/build_variant.sh stamps -DNEB_ATROUS_STAMPS=1; select it with
NEB_LIB_PATH): every wave then sums s_memtime deltas per phase of its tile loop and writes them out at the end
(nebulae_amd/csrc/svgf.hip, NEB_STAMP).  Runs a few SVGF frames on synthetic 1080p inputs and prints, per step, the mean
share of each phase and the spread of wave start / end times (s_memrealtime, 100 MHz).  The stamps cost ~10 % themselves.
usage (GPU box): NEB_LIB_PATH=$PWD/build_variants/lib_stamps.so python tools/atrous_stamps.py [W H]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nebulae_amd import synth  # noqa: E402
from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE, SLOT_CURRENT, SVGFDenoiser  # noqa: E402

PHASES = ["prologue", "stage (DMA issue, lum, ds_write)", "barrier 1 (+DMA landing)", "next-tile set-up", "filter", "stores", "barrier 2"]


def xcc_of(s):
    return s[:, 11].astype(np.int64) & 15


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    L = 5
    d = SVGFDenoiser()
    d.init(W, H, atrous_levels=L)
    g = synth.synth_gbuffer(W, H)
    for f in range(1, 5):
        d.begin_frame(f)
        d.upload(PLANE_DEPTH, SLOT_CURRENT, g["depth"])
        d.upload(PLANE_NORMAL, SLOT_CURRENT, g["normal"])
        d.upload(PLANE_RADIANCE, SLOT_CURRENT, synth.synth_radiance(g["base"], f))
        d.submit_temporal_accumulation()
        d.submit_atrous_compute_wavelet()
        d.synchronize()
        d.end_frame()
    lib = d._lib
    words = 2048 * 4 * 16
    buf = np.zeros(6 * words, np.uint64)
    grids = (C.c_uint32 * 6)()
    lib.neb_debug_atrous_stamps.argtypes = [C.c_void_p, C.c_void_p]
    rc = lib.neb_debug_atrous_stamps(buf.ctypes.data_as(C.c_void_p), grids)
    if rc != 0:
        raise SystemExit(f"neb_debug_atrous_stamps -> {rc} (library not built with -DNEB_ATROUS_STAMPS=1?)")
    for lvl in range(L):
        n = grids[lvl] * 4
        s = buf[lvl * words: lvl * words + n * 16].reshape(n, 16).astype(np.float64)
        wg = np.arange(n) // 4
        keep = s[:, 10] > 0  # waves that had a tile
        s, wg = s[keep], wg[keep]
        tot = s[:, 7]
        t0 = s[:, 8].min()
        start, end = (s[:, 8] - t0) * 0.01, (s[:, 9] - t0) * 0.01  # us
        clock = tot.mean() / ((s[:, 9] - s[:, 8]).mean() * 10.0)  # shader cycles per ns
        print(f"step {1 << lvl}: {len(s)} waves, {s[:, 10].mean():.2f} tiles per wave, wave lifetime {tot.mean():.0f} cycles "
              f"= {(end - start).mean():.1f} us (clock {clock:.2f} GHz); starts {start.min():.1f}..{np.percentile(start, 99):.1f} us, "
              f"ends p1 {np.percentile(end, 1):.1f} p50 {np.percentile(end, 50):.1f} p99 {np.percentile(end, 99):.1f} max {end.max():.1f} us")
        for i, name in enumerate(PHASES):
            print(f"    {name:34s} {s[:, i].mean():9.0f} cycles  {100 * s[:, i].sum() / tot.sum():5.1f} %   per tile {s[:, i].sum() / s[:, 10].sum():8.0f}")
        # per XCD balance
        xcc = s[:, 11].astype(int) & 15
        hw, lds = s[:, 12].astype(np.int64), s[:, 13].astype(np.int64)
        cu = (xcc_of(s) << 8) | ((hw >> 8) & 0xff)  # XCC | se_id, sh_id, cu_id
        bases = sorted(set(lds & 0xff))
        per_cu = {}
        for c, b in zip(cu, lds & 0xff):
            per_cu.setdefault(int(c), set()).add(int(b))
        print(f"    HW_ID census: {len(per_cu)} CUs; LDS_ALLOC raw example {int(lds[0]):#x}; distinct LDS bases {bases[:8]}; workgroups per CU "
              f"min {min(len(v) for v in per_cu.values())} max {max(len(v) for v in per_cu.values())}")
        if lvl == 0:
            for c in sorted(per_cu)[:2]:
                m = cu == c
                rows = sorted({(int(w), int(b) & 0xff, round(float(st), 1), round(float(en), 1), int(nt)) for w, b, st, en, nt in zip(wg[m], lds[m], start[m], end[m], s[m, 10])})
                print(f"    CU {c:#x}: (workgroup, LDS base, start us, end us, tiles) {rows}")
        print("    waves / mean end (us) per XCC:", " ".join(f"{x}:{(xcc == x).sum()}/{end[xcc == x].mean():.1f}" for x in sorted(set(xcc))))
    d.destroy()


if __name__ == "__main__":
    main()

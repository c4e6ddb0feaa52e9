"""How much of a strip frame is the chip waiting between launches?  One interior strip of the 1080p frame alone on the device (exchange stubbed), one frame
in flight; run under `rocprofv3 --kernel-trace` the trace says what the launches take and what lies between them:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/strip_gaps -- python tools/strip_gaps.py 8      then      python tools/strip_gaps.py --read gpurun_out/strip_gaps"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import csv
    f = glob.glob(os.path.join(sys.argv[2], "*", "*kernel_trace.csv"))[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    # the steady window: the last 100 frames = the last 100 launches of the closest-hit kernel
    starts = [i for i, r in enumerate(rows) if "gi_raygen_trace" in r[2]]
    first = starts[-101]
    win = rows[first:starts[-1]]
    frames = 100
    busy = sum(e - s for s, e, _ in win)
    span = win[-1][1] - win[0][0]
    # (kernels of one stream run one after the other: what is not inside a kernel is between two)
    per = {}
    for s, e, n in win:
        k = n.split("(")[0].replace("void neb::", "").replace("neb::", "")
        per.setdefault(k, [0, 0.0])
        per[k][0] += 1
        per[k][1] += (e - s) / 1e3
    print(f"{len(win) / frames:.1f} launches per frame; {span / frames / 1e3:.1f} us per frame, of which {busy / frames / 1e3:.1f} us inside kernels and "
          f"{(span - busy) / frames / 1e3:.1f} us between them ({(span - busy) / max(len(win) - 1, 1) / 1e3:.2f} us per gap)")
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t / frames:7.1f} us per frame  x{c / frames:4.1f}  {k[:100]}")
    sys.exit(0)

import torch  # noqa: E402

from nebulae_amd import scene as S, strips  # noqa: E402
from nebulae_amd.renderer import RenderInfo  # noqa: E402
from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc, cam = S.atrium_standin(), S.sponza_camera()
stream = torch.cuda.Stream()
part = strips.StripPartition(1920, 1080, N, 5, scheme="once")
r = strips.StripRenderer(part, N // 2)
r._swap_rows_begin = lambda planes, plan: (lambda: None)  # no peers here
with torch.cuda.stream(stream):
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=stream.cuda_stream))
    r.submit_commands_gbuffer()
    stream.synchronize()
    for pl in (PLANE_NORMAL, PLANE_DEPTH):
        r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
    r.submit_commands_pbr_lighting()
    stream.synchronize()
    direct = r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).clone()
    t0 = None
    for f in range(2, 262):
        if f == 162:
            stream.synchronize()
            t0 = time.perf_counter()
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=stream.cuda_stream))
        r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
        r.submit_commands_gi_pathtrace()
        r.submit_commands_svgf_denoising()
        r.end_frame()
    stream.synchronize()
    print(f"N = {N}: one strip, one frame in flight: wall {(time.perf_counter() - t0) / 100 * 1e6:.0f} us per frame")

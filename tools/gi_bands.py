"""Experiment: the GI dispatch of one frame cut into K row bands issued alternately on two streams, so that the bandwidth-bound
shade pass of one band runs beside the instruction-bound traversal of the next.  Prints us per frame (wall clock over N frames,
streams joined at every frame boundary)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
W, H = 1920, 1080
N = 40
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer(); r.init(W, H, atrous_levels=5)
main = torch.cuda.current_stream()
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=main.cuda_stream))
r.submit_commands_gbuffer()
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def bands_of(K, first_frac=None):
    edges = [round(H * k / K / 8) * 8 for k in range(K + 1)]
    edges[-1] = H
    if first_frac is not None:  # a shorter first band staggers the two streams
        edges[1] = max(8, round(H * first_frac / 8) * 8)
    return [(edges[k], edges[k + 1]) for k in range(K) if edges[k + 1] > edges[k]]

def run(label, fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    print("%-60s %7.1f us per frame" % (label, (time.perf_counter() - t) / N * 1e6), flush=True)

def whole():
    r.submit_commands_gi_pathtrace()
run("whole frame, one stream (library default sort)", whole)
r.svgf.set_option("gi_sort_rays", 0)
run("whole frame, one stream, no sort", whole)
for K in (2, 4, 8, 16):
    bs = bands_of(K)
    def serial():
        for b in bs:
            r.submit_commands_gi_pathtrace(rows=b)
    run("%d bands, one stream, no sort" % K, serial)
    for ff in (None, 0.5 / K):
        bs2 = bands_of(K, ff)
        def two():
            sA.wait_stream(main); sB.wait_stream(main)
            for k, b in enumerate(bs2):
                r.submit_commands_gi_pathtrace(rows=b, stream=(sA if k % 2 == 0 else sB).cuda_stream)
            main.wait_stream(sA); main.wait_stream(sB)
        run("%d bands, two streams, no sort, first band %s" % (K, "equal" if ff is None else "half"), two)
# three streams
sC = torch.cuda.Stream()
for K in (6, 12):
    bs = bands_of(K)
    def three():
        ss = (sA, sB, sC)
        for s in ss:
            s.wait_stream(main)
        for k, b in enumerate(bs):
            r.submit_commands_gi_pathtrace(rows=b, stream=ss[k % 3].cuda_stream)
        for s in ss:
            main.wait_stream(s)
    run("%d bands, three streams, no sort" % K, three)

"""Diagnostics: how much of a wave's traversal is idle lanes, and what would direction binning recover?
Per-ray loop iterations come from the debug hit records; waves are then re-formed on the host."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nebulae_amd import scene as S
from nebulae_amd.renderer import DeferredRenderer, RenderInfo
W, H = 1920, 1080
sc, cam = S.atrium_standin(), S.sponza_camera()
r = DeferredRenderer(); r.init(W, H, atrous_levels=5)
r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
r.submit_commands_gbuffer(); r.set_debug_hits(True); r.submit_commands_gi_pathtrace()
hits = r.download_hits()
it = (hits["flags"] >> 8).astype(np.int64)
def wave_cost(order_it):
    w = order_it[: (order_it.size // 64) * 64].reshape(-1, 64)
    return w.max(axis=1).sum(), w.sum() / 64.0
# (a) as launched: 8x8 tiles
t = it[: (H // 8) * 8, : (W // 8) * 8].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
mx, mean = wave_cost(t)
print("8x8 tiles: sum of wave maxima %.3e vs sum of lane means %.3e -> lane utilisation %.2f" % (mx, mean, mean / mx))
# (b) oracle-best: sort rays globally by their own iteration count (upper bound of any reordering)
mx2, _ = wave_cost(np.sort(it.reshape(-1)))
print("perfect sort by cost: utilisation %.2f (gain x%.2f)" % (mean / mx2, mx / mx2))
# (c) binning by iteration count inside 32x32 blocks (what an in-block reorder could at best do)
b = it[: (H // 32) * 32, : (W // 32) * 32].reshape(H // 32, 32, W // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 1024)
mx3 = np.sort(b, axis=1).reshape(-1, 64).max(axis=1).sum()
mx3_base = b.reshape(-1, 16, 64)
print("sorted inside 32x32 blocks: gain x%.2f" % (wave_cost(t)[0] / (mx3 * (t.size / b.size))))
print("iterations per ray: mean %.1f p50 %d p90 %d p99 %d max %d" % (it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), it.max()))

# (d) workgroup-level periodic compaction: a workgroup of G rays runs rounds of at most N iterations; after each round the
# unfinished rays are packed into as few waves as possible (state handed over through LDS) and the other waves retire.
def compaction_cost(G, N, overhead):
    tiles = it[: (H // 8) * 8, : (W // 8) * 8].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    nw = G // 64
    tiles = tiles[: (tiles.shape[0] // nw) * nw].reshape(-1, G).astype(np.int64)  # consecutive tiles form a workgroup
    rem = tiles.copy()
    total = 0.0
    first = True
    while True:
        alive = rem > 0
        if not alive.any():
            break
        if first:
            packed = rem  # initial assignment: pixel order
            first = False
        else:
            # pack unfinished rays to the front (order preserved)
            idx = np.argsort(~alive, axis=1, kind="stable")
            packed = np.take_along_axis(rem, idx, axis=1)
        waves = packed.reshape(packed.shape[0], nw, 64)
        wmax = waves.max(axis=2)
        run = np.minimum(wmax, N)
        total += run.sum() + overhead * (wmax > 0).sum()
        rem = np.maximum(packed - N, 0)
    return total / tiles.size * t.size
base = wave_cost(t)[0]
for G in (256, 512, 1024):
    for N in (8, 16, 24, 32):
        c = compaction_cost(G, N, 2.0)
        print("workgroup %4d rays, rounds of %2d iterations (+2 per live wave and round): cost x%.2f of baseline" % (G, N, c / base))

# (e) tail suspension: a wave stops as soon as at most T of its lanes are still traversing (and it has run at least M
# iterations); the stragglers' state goes to a queue and a follow-up launch finishes them in dense waves (in queue order).
tiles = it[: (H // 8) * 8, : (W // 8) * 8].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64).astype(np.int64)
srt = np.sort(tiles, axis=1)[:, ::-1]  # descending per wave
for T in (2, 4, 8, 12, 16):
    for M in (0, 16):
        stop = np.maximum(srt[:, T], M)                      # iteration at which only T lanes remain
        stop = np.minimum(stop, srt[:, 0])
        first = stop.sum()
        rem = np.maximum(tiles - stop[:, None], 0)
        q = rem[rem > 0]                                      # queue order = wave order
        pad = (-q.size) % 64
        q = np.concatenate([q, np.zeros(pad, np.int64)]).reshape(-1, 64)
        second = q.max(axis=1).sum()
        print("suspend at <= %2d lanes (min %2d iterations): first pass x%.3f + follow-up x%.3f = x%.3f of baseline; %.1f %% of the rays are queued"
              % (T, M, first / base, second / base, (first + second) / base, 100.0 * (rem > 0).mean()))

# (f) dynamic refill: a wave owns K consecutive tiles; whenever at least T of its lanes are idle (and rays are left in its
# pool) the idle lanes start the next rays of the pool in place (no state moves).  Cost = loop iterations with a live
# lane + `setup` iterations per refill event (ray generation / result write-out run by the refilling lanes only).
def refill_cost(K, T, setup):
    nw = tiles.shape[0] // K
    pool = tiles[: nw * K].reshape(nw, K * 64)
    rem = pool[:, :64].copy()
    pos = np.full(nw, 64, np.int64)
    end = K * 64
    total = 0.0
    while True:
        live = rem > 0
        any_live = live.any(axis=1)
        idle = ~live
        n_idle = idle.sum(axis=1)
        want = (n_idle >= T) & (pos < end) | (~any_live & (pos < end))
        if want.any():
            w = np.nonzero(want)[0]
            rank = np.cumsum(idle[w], axis=1) - 1
            src = pos[w, None] + rank
            ok = idle[w] & (src < end)
            vals = np.take_along_axis(pool[w], np.minimum(src, end - 1), axis=1)
            r = rem[w]
            r[ok] = vals[ok]
            rem[w] = r
            pos[w] = np.minimum(pos[w] + n_idle[w], end)
            total += setup * w.size
            live = rem > 0
            any_live = live.any(axis=1)
        if not any_live.any():
            break
        total += any_live.sum()
        rem = np.maximum(rem - 1, 0)
    return total * (tiles.shape[0] / (nw * K))
for K in (2, 4, 8, 16):
    for T in (8, 16, 32, 48):
        print("refill: %2d tiles per wave, refill at >= %2d idle lanes (+3 per refill): cost x%.3f of baseline" % (K, T, refill_cost(K, T, 3.0) / base))

"""World-size-2 (and 3) gloo tests of the multi-GPU strip path on CPU: nebulae_amd/strips.py drives the
partition and the halo exchange; an oracle-backed denoiser does the arithmetic.  The N-strip result must
equal the single-process result BIT FOR BIT, and no strip may ever read a non-resident row."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nebulae_amd import scene as S
from nebulae_amd import strips
from nebulae_amd.renderer import RenderInfo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_geometry():
    # default scheme: one exchange per frame, levels run on owned +- (rows the later levels still reach into)
    p = strips.StripPartition(3840, 2160, 4, 5)
    assert p.scheme == "once" and p.halo == 62 and p.owned(1) == (540, 1080)
    assert p.resident(0) == (0, 602) and p.resident(3) == (1558, 2160)
    assert [p.level_extension(l) for l in range(5)] == [60, 56, 48, 32, 0]
    assert p.atrous_rows(0, 0) == (0, 600) and p.atrous_rows(2, 1) == (1080 - 56, 1620 + 56) and p.atrous_rows(3, 4) == (1620, 2160)
    assert p.frame_exchange(0) == [(1, (478, 540), (540, 602))]
    assert p.frame_exchange(2) == [(1, (1080, 1142), (1018, 1080)), (3, (1558, 1620), (1620, 1682))]
    assert p.level_exchange(2, 0) == []
    assert p.exchanged_bytes_per_frame() == 2 * 62 * 3840 * 18  # 62 rows per direction of radiance + variance
    # SURVEY.md 8e's per-level scheme
    q = strips.StripPartition(3840, 2160, 4, 5, scheme="per_level")
    assert q.halo == 32 and q.resident(0) == (0, 572) and q.resident(3) == (1588, 2160)
    assert q.level_exchange(0, 4) == [(1, (508, 540), (540, 572))]
    assert q.level_exchange(2, 0) == [(1, (1080, 1082), (1078, 1080)), (3, (1618, 1620), (1620, 1622))]
    assert q.frame_exchange(1) == [] and q.atrous_rows(1, 0) == q.owned(1)
    assert q.exchanged_bytes_per_frame() == 2 * 62 * 3840 * 16  # 62 rows per direction
    assert strips.frame_factors(1) == (1, 1) and strips.frame_factors(4) == (2, 2) and strips.frame_factors(8) == (2, 4)
    # SURVEY.md 8e's overlapped scheme: GI / temporal / levels 0..3 on +-30 rows, 32 rows before the widest level, 30 rows of history after it
    v = strips.StripPartition(3840, 2160, 4, 5, scheme="overlap")
    assert v.band == 30 and v.halo == 32 and v.gi_rows(0) == (0, 570) and v.gi_rows(2) == (1050, 1650) and v.resident(1) == (508, 1112)
    assert [v.level_extension(l) for l in range(5)] == [28, 24, 16, 0, 0] and v.atrous_rows(1, 0) == (512, 1108) and v.atrous_rows(1, 4) == (540, 1080)
    assert v.level_exchange(1, 3) == [] and v.level_exchange(1, 4) == [(0, (540, 572), (508, 540)), (2, (1048, 1080), (1080, 1112))]
    assert v.history_exchange(1) == [(0, (540, 570), (510, 540)), (2, (1050, 1080), (1080, 1110))] and v.frame_exchange(1) == []
    assert v.exchanged_bytes_per_frame() == 2 * 62 * 3840 * 16 and q.history_exchange(1) == [] and p.gi_rows(1) == p.owned(1)
    one = strips.StripPartition(1920, 1080, 1, 5)
    assert one.halo == 0 and one.resident(0) == (0, 1080) and one.level_exchange(0, 3) == [] and one.frame_exchange(0) == []
    assert one.atrous_rows(0, 2) == (0, 1080)
    with pytest.raises(ValueError):
        strips.StripPartition(64, 48, 4, 5)  # 12-row strips cannot feed the a-trous reach


def _frames(W, H, n):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import frame_inputs
    return [frame_inputs(W, H, f, 3) for f in range(1, n + 1)]


def _run_frames(r, part, rank, frames, moving_frame=None):
    from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE
    res0, res1 = part.resident(rank)
    cam0, cam1 = S.orbit_camera(), S.orbit_camera(yaw_deg=5.0)
    ran = []
    for f, (g, rad) in enumerate(frames, start=1):
        cam = cam1 if (moving_frame is not None and f >= moving_frame) else cam0
        r.begin_frame(RenderInfo(scene=None, camera=cam, frame_index=f))
        cur = r.svgf.get_current_resource_index()
        r.svgf.plane_tensor(PLANE_DEPTH, cur).copy_(torch.from_numpy(g["depth"][res0:res1].view(np.int32)))
        r.svgf.plane_tensor(PLANE_NORMAL, cur).copy_(torch.from_numpy(g["normal"][res0:res1]))
        g0, g1 = part.gi_rows(rank)  # the rows this rank "renders": its strip (+- the band of the overlap scheme)
        r.svgf.plane_tensor(PLANE_RADIANCE, cur)[g0 - res0:g1 - res0].copy_(torch.from_numpy(rad[g0:g1]))
        ran.append(r.submit_commands_svgf_denoising())
        r.end_frame()
    cur = r.svgf.get_current_resource_index()
    own0, own1 = part.owned(rank)
    return r.svgf.plane_tensor(PLANE_RADIANCE, cur)[own0 - res0:own1 - res0].clone(), ran


def _worker(rank, world, port, W, H, L, nframes, moving_frame, out_dir, scheme):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleDenoiser
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    part = strips.StripPartition(W, H, world, L, scheme=scheme)
    r = strips.StripRenderer(part, rank, group=dist.group.WORLD, denoiser_factory=OracleDenoiser)
    out, ran = _run_frames(r, part, rank, _frames(W, H, nframes), moving_frame)
    full = r.gather_frame(dst=0)  # SURVEY.md 8e final gather: rank 0 ends up with the whole image
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), full.numpy())
    else:
        assert full is None
    np.save(os.path.join(out_dir, f"strip_{rank}.npy"), out.numpy())
    np.save(os.path.join(out_dir, f"ran_{rank}.npy"), np.array(ran))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,L,moving,scheme", [(2, 64, 144, 5, None, "once"), (3, 72, 120, 4, None, "once"), (2, 64, 80, 3, 2, "once"),
                                                        (2, 40, 64, 1, None, "once"), (2, 64, 144, 5, None, "per_level"),
                                                        (3, 72, 120, 4, 3, "per_level"), (2, 64, 144, 5, None, "overlap"),
                                                        (3, 72, 120, 4, 3, "overlap"), (2, 40, 64, 1, None, "overlap")])
def test_strips_equal_single_process_bit_for_bit(tmp_path, world, W, H, L, moving, scheme):
    from oracle_backend import OracleDenoiser
    nframes = 4
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, W, H, L, nframes, moving, str(tmp_path), scheme), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"strip_{k}.npy") for k in range(world)], axis=0)
    part1 = strips.StripPartition(W, H, 1, L)
    r1 = strips.StripRenderer(part1, 0, denoiser_factory=OracleDenoiser)
    want, ran1 = _run_frames(r1, part1, 0, _frames(W, H, nframes), moving)
    assert not np.isnan(got).any(), "a strip read a row that was not resident"
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), got)
    assert np.array_equal(got, want.numpy())
    for k in range(world):  # every rank takes the same skip/reset decisions as the single process
        assert np.array_equal(np.load(tmp_path / f"ran_{k}.npy"), np.array(ran1))


def test_frame_policy_reference_and_always_on():
    """src/DeferredRenderer.cpp:133-146,593-614: SVGF is skipped on frames whose camera moved and the history is reset on
    the first still frame; `denoise_while_moving` (beyond the reference) keeps it on throughout."""
    from oracle_backend import OracleDenoiser
    W, H, L = 48, 40, 2
    frames = _frames(W, H, 5)
    part = strips.StripPartition(W, H, 1, L)
    ref = strips.StripRenderer(part, 0, denoiser_factory=OracleDenoiser)
    out_ref, ran_ref = _run_frames(ref, part, 0, frames, moving_frame=3)
    assert ran_ref == [False, True, False, True, True]  # frame 1: first camera; frame 3: the camera moves; frame 4: reset + denoise
    on = strips.StripRenderer(part, 0, denoiser_factory=OracleDenoiser)
    on.denoise_while_moving = True
    out_on, ran_on = _run_frames(on, part, 0, frames, moving_frame=3)
    assert ran_on == [True] * 5
    assert not np.array_equal(out_on.numpy(), out_ref.numpy())  # the history was never reset

"""Multi-process CPU test (gloo, world_size 2) of the N>1 path used by bench.py: disjoint frame
shards, barrier, MAX-over-ranks timing, whole-job throughput."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vieo_slam_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, w, _ = sharding.env_rank()
    b, e = sharding.frame_range(2273, r, w)  # MH05 has 2273 frames
    dist.barrier()
    elapsed = 1.0 + 0.5 * rank  # rank 1 is slower
    mx = sharding.max_over_ranks(dist, elapsed)
    # every rank gathers all shards to check disjointness/coverage
    shards = [None] * w
    dist.all_gather_object(shards, (b, e))
    q.put((rank, b, e, mx, shards, sharding.rank_seed(rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    out = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (r0, b0, e0, m0, s0, seed0), (r1, b1, e1, m1, s1, seed1) = out
    assert (b0, e0, b1, e1) == (0, 1137, 1137, 2273)
    assert m0 == m1 == 1.5  # MAX over ranks
    assert s0 == s1 == [(0, 1137), (1137, 2273)]
    assert seed0 != seed1
    assert sharding.aggregate_throughput(64, 10, 2, 1.5) == pytest.approx(64 * 10 * 2 / 1.5)


def test_frame_range_properties():
    for n in (0, 1, 7, 2273, 5990):
        for w in (1, 2, 3, 8):
            rs = [sharding.frame_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in rs]
            assert max(sizes) - min(sizes) <= 1


# ---- landmark-sharded LocalBundleAdjustmentNavStatePRV ---------------------------------
def test_shard_window_partitions_points_and_observations():
    import numpy as np
    from vieo_slam_amd import synth_ba
    win = synth_ba.make_lba_vio_problem(40, n_local=4, n_fixed=2, n_points=200)[:6]
    params, kfs, pts, close, obs, imu = win
    for world in (1, 2, 3):
        seen_pts, seen_obs = [], 0
        for r in range(world):
            (p2, k2, pts2, close2, obs2, imu2), mine = sharding.shard_window(win, r, world)
            assert k2 is kfs and imu2 is imu  # key frames and inertial edges replicated
            assert np.array_equal(pts2, pts[mine]) and np.array_equal(close2, np.asarray(close)[mine])
            assert len(obs2) and obs2["mp"].min() == 0 and obs2["mp"].max() == len(mine) - 1
            assert (np.diff(obs2["mp"]) >= 0).all()  # still grouped by point
            # the shard's observations are exactly the window's observations of its points
            back = obs[np.isin(obs["mp"], mine)]
            assert np.array_equal(back["u"], obs2["u"]) and np.array_equal(mine[obs2["mp"]], back["mp"])
            seen_pts.append(mine)
            seen_obs += len(obs2)
        assert np.array_equal(np.sort(np.concatenate(seen_pts)), np.arange(len(pts))) and seen_obs == len(obs)


def _reduce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.arange(20, dtype=torch.float64) * (rank + 1)
    fn = sharding.torch_allreduce(buf)
    assert fn(0, 12) == 0   # the system part ...
    assert fn(16, 4) == 0   # ... and the scalars, as the library calls it
    q.put((rank, buf.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduction_callback_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_reduce_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    out = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for r in range(world):
        v = out[r]
        for i in range(20):
            summed = i < 12 or i >= 16
            assert v[i] == (i * 3 if summed else i * (r + 1))


def _reduce8_worker(rank, world, port, q):
    """One rank of an 8-rank landmark-sharded window whose point count is below the world size: some ranks own no
    point.  Everything the sharded entry needs from the host side: the shard (possibly empty), the buffer size every
    rank computes alike, the reduction callback summing over all ranks -- empty ranks included -- in both exchanges."""
    import numpy as np
    from vieo_slam_amd import synth_ba
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    win = synth_ba.make_lba_vio_problem(44, n_local=3, n_fixed=1, n_points=40)[:6]
    params, kfs, pts, close, obs, imu = win
    keep = np.arange(5)  # a window of five points over eight ranks
    sel = np.isin(obs["mp"], keep)
    small = (params, kfs, pts[:5], np.asarray(close)[:5], obs[sel], imu)
    (p2, k2, pts2, close2, obs2, imu2), mine = sharding.shard_window(small, rank, world)
    assert len(pts2) == (1 if rank < 5 else 0) and len(obs2) == int(np.isin(small[4]["mp"], mine).sum())
    assert k2 is kfs and imu2 is imu
    nf = int((kfs["fixed"] == 0).sum())
    n_sys = 6 * nf * (6 * nf + 1) + 42 * nf  # the packed reduced visual system
    buf = torch.zeros(n_sys + 4, dtype=torch.float64)
    buf[:n_sys] = float(len(pts2))       # a rank without points contributes zeros
    buf[n_sys:n_sys + 3] = float(rank)
    fn = sharding.torch_allreduce(buf)
    assert fn(0, n_sys) == 0 and fn(n_sys, 4) == 0
    q.put((rank, float(buf[0]), float(buf[n_sys - 1]), float(buf[n_sys]), len(pts2)))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_reduction_with_empty_shards_gloo():
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_reduce8_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    out = sorted(q.get(timeout=240) for _ in range(world))
    [p.join(120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert [o[0] for o in out] == list(range(8)) and sum(o[4] for o in out) == 5
    for r, first, last, sc, npts in out:
        assert first == last == 5.0      # five ranks own one point each, three none: the sum is the same everywhere
        assert sc == float(sum(range(8)))

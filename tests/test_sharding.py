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

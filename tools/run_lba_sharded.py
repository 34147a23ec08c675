#!/usr/bin/env python3
"""One LocalBundleAdjustmentNavStatePRV window sharded by landmark over the ranks of a
torch.distributed job, RCCL all-reduce of the reduced pose system per LM trial:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29511 tools/run_lba_sharded.py [--points 4000] [--local 10] [--reps 5]
        [--gba ITERATIONS]   full BA instead (BASELINE configs[4]): --local key frames all free but one,
                             e.g. --gba 5 --local 200 --fixed 1 --points 20000

Every rank builds the same seeded window, keeps the points with index % N == rank, and calls the
sharded C-ABI entry with a reduction buffer that lives in a torch tensor.  Rank 0 prints the time per
call and the pose agreement with the unsharded call on its own GPU."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from vieo_slam_amd import sharding, synth_ba
from vieo_slam_amd.optimizer import Optimizer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4000)
    ap.add_argument("--local", type=int, default=10)
    ap.add_argument("--fixed", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--gba", type=int, default=0, help="full BA with this many LM iterations instead of the local BA")
    ap.add_argument("--callback", action="store_true",
                    help="exchange through the torch.distributed callback (two host synchronisations per LM trial) "
                         "instead of the in-library RCCL all-reduce on the bundle-adjustment stream")
    a = ap.parse_args()
    rank, world, local = sharding.env_rank()
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if a.gba:
        win = synth_ba.make_lba_vio_problem(777, n_local=a.local, n_fixed=a.fixed, n_points=a.points,
                                            anchors=max(1, a.local // 2), span=5)[:6]
    else:
        win = synth_ba.make_lba_vio_problem(777, n_local=a.local, n_fixed=a.fixed, n_points=a.points)[:6]
    shard, mine = sharding.shard_window(win, rank, world)
    n = Optimizer.sharded_buffer_doubles([shard])

    def call():
        if a.gba:
            navs, pts, res = Optimizer.GlobalBundleAdjustmentNavStatePRVSharded(shard, buf.data_ptr(), n, cb, a.gba, True,
                                                                                comm=comm)
            return navs, pts, None, res
        return Optimizer.LocalBundleAdjustmentNavStatePRVSharded([shard], buf.data_ptr(), n, cb, comm=comm)[0]

    buf = torch.zeros(n, dtype=torch.float64, device="cuda")
    cb = sharding.torch_allreduce(buf) if a.callback else None
    comm = None if a.callback else sharding.RcclComm(rank, world).handle
    out = None
    for i in range(a.reps + 1):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = call()
        dt = time.perf_counter() - t
        if i == 0:
            times = []
        else:
            times.append(dt)
    if rank == 0:
        if a.gba:
            ref = Optimizer.GlobalBundleAdjustmentNavStatePRV(win[0], win[1], win[2], win[4], win[5], a.gba, True)
        else:
            ref = Optimizer.LocalBundleAdjustmentNavStatePRV(*win)
        dmax = max(max(synth_ba.pose_error(ref[0][k], out[0][k])) for k in range(len(win[1])))
        print({"mode": "full BA" if a.gba else "local BA", "ranks": world, "points_total": len(win[2]), "points_this_rank": len(mine),
               "observations_total": len(win[4]), "ms_per_call": 1e3 * float(np.mean(times)),
               "lm_trials": int(out[3]["lm_trials"]), "max_pose_diff_vs_unsharded": dmax,
               "exchange": "torch.distributed callback" if a.callback else "in-library RCCL on the BA stream"})
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

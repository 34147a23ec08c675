"""Frame sharding across GPUs: frames / sequences are independent units, so ranks
own disjoint frame ranges and the data path needs NO collective; torch.distributed (RCCL on GPUs,
gloo in the CPU tests) is only used for the barrier around the timed region and the MAX-over-ranks
of the elapsed time, as bench.py's contract requires."""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def frame_range(n_frames_total, rank, world):
    """Contiguous, balanced shard [begin, end) of a global frame sequence for `rank`."""
    base, rem = divmod(n_frames_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_seed(rank, base=1):
    """Distinct synthetic scenes per rank (weak scaling: every rank renders its own batch)."""
    return base + 1000 * rank


def max_over_ranks(dist, value, device=None):
    """MAX all-reduce of a python float (elapsed seconds)."""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(frames_per_rank_per_step, steps, world, elapsed_max_s):
    """Whole-job frames/s: all ranks' frames over the slowest rank's time."""
    return frames_per_rank_per_step * steps * world / elapsed_max_s


# ---- one LocalBundleAdjustmentNavStatePRV window over several GPUs ----------------
def shard_window(window, rank, world):
    """Landmark shard of a visual-inertial window (params, kfs, points, close, obs, imu): the points
    with index % world == rank and their observations (renumbered, still grouped by point); key
    frames and inertial edges are replicated.  Returns (window_shard, point_index) where point_index
    maps the shard's points back to the window's."""
    import numpy as np
    params, kfs, points, close, obs, imu = window
    mine = np.nonzero(np.arange(len(points)) % world == rank)[0]
    remap = -np.ones(len(points), np.int64)
    remap[mine] = np.arange(len(mine))
    sel = remap[obs["mp"]] >= 0
    o = obs[sel].copy()
    o["mp"] = remap[o["mp"]]
    return (params, kfs, np.ascontiguousarray(points[mine]), np.ascontiguousarray(np.asarray(close)[mine]), o,
            imu), mine


def torch_allreduce(buf):
    """Reduction callback for Optimizer.LocalBundleAdjustmentNavStatePRVSharded on top of
    torch.distributed (backend nccl = RCCL over xGMI; gloo with a CPU tensor in the tests): `buf` is the
    float64 tensor whose storage was handed to the C-ABI as the reduction buffer."""
    import torch
    import torch.distributed as dist

    def fn(offset, n):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(buf[offset:offset + n], op=dist.ReduceOp.SUM)
        if buf.is_cuda:
            torch.cuda.synchronize(buf.device)
        return 0
    return fn


class RcclComm:
    """In-library RCCL communicator (vieo_rccl_*): rank 0 draws the unique id, `bcast(bytes_or_None) -> bytes`
    carries it to the other ranks (default: torch.distributed broadcast of a uint8 tensor), every rank creates its
    communicator on its current GPU.  `.handle` goes to the sharded Optimizer entries as `comm=`."""

    def __init__(self, rank, world, bcast=None):
        import ctypes
        import numpy as np
        from ._lib import check, lib
        if not lib().vieo_rccl_available():
            raise RuntimeError("librccl.so could not be loaded")
        ident = np.zeros(128, np.uint8)
        if rank == 0:
            check(lib().vieo_rccl_unique_id(ident.ctypes.data), "vieo_rccl_unique_id")
        if world > 1:
            if bcast is None:
                import torch
                import torch.distributed as dist
                t = torch.from_numpy(ident).cuda() if dist.get_backend() == "nccl" else torch.from_numpy(ident)
                dist.broadcast(t, src=0)
                ident = t.cpu().numpy()
            else:
                ident = np.frombuffer(bcast(ident.tobytes() if rank == 0 else None), np.uint8).copy()
        h = ctypes.c_void_p()
        check(lib().vieo_rccl_comm_create(ctypes.byref(h), ident.ctypes.data, world, rank), "vieo_rccl_comm_create")
        self.handle = h.value

    def close(self):
        from ._lib import lib
        if getattr(self, "handle", None):
            lib().vieo_rccl_comm_destroy(self.handle)
            self.handle = None

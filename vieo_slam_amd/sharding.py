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

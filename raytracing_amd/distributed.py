"""spp sharding across the GPUs of a node: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Rank r of G traces the pattern columns r, r+G, r+2G, ... into an un-normalised FP32 accumulator (H*W*3 floats),
the accumulators are summed by ONE all-reduce per frame, and every rank finalises (x Float32(1/N), sqrt).  The
reference has no multi-device path; this is the only exchange step the path needs (src/Ray_Tracing.jl:82-91 is a sum
over pattern columns, which is associative up to FP32 rounding).
"""
import numpy as np


def shard_columns(n, rank, world_size):
    """Pattern columns of one rank: (first, stride, count)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    count = len(range(rank, n, world_size))
    return rank, world_size, count


def render_sharded(accumulate, all_reduce, finalize, n, rank, world_size):
    """Generic driver so that the same sharding logic runs on GPUs (DeviceScene.accumulate + dist.all_reduce +
    DeviceScene.finalize) and in the CPU tests (any accumulate callable + gloo).

    accumulate(first, stride) -> accumulator (un-normalised sums of that shard, or None when the shard is empty)
    all_reduce(acc)           -> sums the accumulators of all ranks in place
    finalize(acc, n)          -> image
    """
    first, stride, count = shard_columns(n, rank, world_size)
    acc = accumulate(first, stride) if count > 0 else None
    acc = all_reduce(acc)
    return finalize(acc, n)


def finalize_host(acc, n):
    """col /= N (a multiply by Float32(1/N), src/vec3.jl:31) and the per-channel sqrt (src/Ray_Tracing.jl:90-91)."""
    return np.sqrt(np.asarray(acc, np.float32) * np.float32(1.0 / n))

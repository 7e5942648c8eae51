"""spp sharding across the GPUs of a node: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Rank r of G traces the pattern columns r, r+G, r+2G, ... into an un-normalised FP32 accumulator (H*W*3 floats),
the accumulators are summed by ONE all-reduce per frame, and every rank finalises (x Float32(1/N), sqrt).  The
reference has no multi-device path; this is the only exchange step the path needs (src/Ray_Tracing.jl:82-91 is a sum
over pattern columns, which is associative up to FP32 rounding).

sharded_frame() is that frame, written once: bench.py runs it with a DeviceScene and dist.all_reduce over RCCL, the CPU
test (tests/test_dist_gloo.py) with a stand-in scene and gloo.  (A single process that drives several GPUs itself uses
rt_group_create / rt_group_render of the C ABI instead, see DeviceGroup in render.py.)
"""
import numpy as np


def shard_columns(n, rank, world_size):
    """Pattern columns of one rank: (first, stride, count)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    count = len(range(rank, n, world_size))
    return rank, world_size, count


def sharded_frame(scene, camera, pattern, background, width, height, accum, out, rank, world_size, all_reduce=None,
                  stream=None, flags=0, after_accumulate=None, **render_args):
    """One frame on rank `rank` of `world_size`.

    scene            anything with DeviceScene's accumulate(...) and finalize(...) (they take device pointers)
    accum, out       tensors of 3 * height * width float32 on the scene's device (out may be accum)
    all_reduce(t)    sums t over all ranks in place (dist.all_reduce); not called when world_size == 1
    after_accumulate called between the shard's kernels and the all-reduce (bench.py records an event there)

    A rank whose shard is empty (more ranks than pattern columns) still takes part: its accumulator is zero.
    """
    first, stride, _ = shard_columns(pattern.shape[1], rank, world_size)
    scene.accumulate(camera, pattern, background, width, height, accum.data_ptr(), sample_first=first, sample_stride=stride,
                     stream=stream, flags=flags, **render_args)
    if after_accumulate is not None:
        after_accumulate()
    if world_size > 1:
        all_reduce(accum)
    scene.finalize(accum.data_ptr(), pattern.shape[1], width, height, out.data_ptr(), stream=stream)
    return out


def finalize_host(acc, n):
    """col /= N (a multiply by Float32(1/N), src/vec3.jl:31) and the per-channel sqrt (src/Ray_Tracing.jl:90-91)."""
    return np.sqrt(np.asarray(acc, np.float32) * np.float32(1.0 / n))

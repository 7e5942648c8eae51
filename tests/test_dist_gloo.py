"""The multi-rank path on CPU: several processes (gloo), each accumulates its share of the pattern columns, one
all-reduce, finalise -- through raytracing_amd.distributed.sharded_frame, the function bench.py runs on the GPUs.  The
scene here is a stand-in with DeviceScene's accumulate / finalize signatures whose accumulate step is the CPU oracle
(tensors in host memory); what is under test is the sharding, the reduction and the normalisation."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_ffi as O
import raytracing_amd as rt
from raytracing_amd.distributed import finalize_host, shard_columns, sharded_frame

F = np.float32
W, H, SPP = 20, 12, 7


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _host_array(ptr, count):
    return np.ctypeslib.as_array((ctypes.c_float * count).from_address(ptr))


class HostScene:
    """DeviceScene's accumulate / finalize over host memory: the shard's columns from per-path radiances of the oracle,
    added in column order; buffers in the library's image layout (column, row, rgb)."""

    def __init__(self, world):
        self.ref = O.OracleScene(world)

    def accumulate(self, camera, pattern, background, width, height, d_accum, sample_first=0, sample_stride=1, stream=None,
                   flags=0):
        cols = list(range(sample_first, pattern.shape[1], sample_stride))
        acc = np.zeros((width, height, 3), F)         # an empty shard is a zero sum, like rt_render_accumulate's
        for j in range(height):
            ids = np.array([(i, j, n) for i in range(width) for n in cols], np.int32)
            rad = self.ref.trace_paths(camera, pattern, background, width, height, ids).reshape(width, len(cols), 3)
            for c in range(len(cols)):
                acc[:, height - 1 - j] = acc[:, height - 1 - j] + rad[:, c]
        _host_array(d_accum, 3 * width * height)[:] = acc.reshape(-1)

    def finalize(self, d_accum, n_total, width, height, d_out, stream=None):
        _host_array(d_out, 3 * width * height)[:] = finalize_host(_host_array(d_accum, 3 * width * height), n_total)


def _worker(rank, world_size, port, out_dir, spp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    world, kw = rt.get_scene("cornell_smoke_box", spp=spp)
    accum, out = torch.full((3 * W * H,), 7.0), torch.zeros(3 * W * H)
    sharded_frame(HostScene(world), kw["camera"], kw["pattern"], kw["background"], W, H, accum, out, rank, world_size,
                  all_reduce=dist.all_reduce)
    np.save(os.path.join(out_dir, f"img{rank}.npy"), out.numpy().reshape(W, H, 3).transpose(1, 0, 2))
    dist.destroy_process_group()


def test_shard_columns_partition():
    for n in (1, 2, 7, 1000):
        for g in (1, 2, 4, 8):
            cols = []
            for r in range(g):
                first, stride, count = shard_columns(n, r, g)
                cols += list(range(first, n, stride))
                assert count == len(range(first, n, stride))
            assert sorted(cols) == list(range(n))
    with pytest.raises(ValueError):
        shard_columns(4, 2, 2)


@pytest.mark.parametrize("world_size,spp", [(2, SPP), (3, 2)])      # (3 ranks, 2 columns: the third rank's shard is empty)
def test_ranks_over_gloo_match_single_process(world_size, spp, tmp_path):
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path), spp), nprocs=world_size, join=True)
    imgs = [np.load(tmp_path / f"img{r}.npy") for r in range(world_size)]
    for other in imgs[1:]:
        np.testing.assert_array_equal(imgs[0], other)         # every rank ends with the same image
    world, kw = rt.get_scene("cornell_smoke_box", spp=spp)
    want = O.OracleScene(world).render(kw["camera"], kw["pattern"], kw["background"], W, H)
    # same set of paths, only the FP32 summation order differs (partial sums per rank instead of one running sum)
    np.testing.assert_allclose(imgs[0], want, rtol=2e-6, atol=1e-7)

"""The multi-rank path on CPU: two processes (gloo), each accumulates its share of the pattern columns, one all-reduce,
finalise.  The accumulate step here is the CPU oracle (the GPU test does the same with the HIP path); what is under test
is the sharding / reduction / normalisation logic of raytracing_amd.distributed."""
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
from raytracing_amd.distributed import finalize_host, render_sharded, shard_columns

F = np.float32
W, H, SPP = 20, 12, 7


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_accumulate(ref, cam, pattern, bg, first, stride):
    """Un-normalised sums of the shard's columns, from per-path radiances added in column order."""
    cols = list(range(first, pattern.shape[1], stride))
    acc = np.zeros((H, W, 3), F)
    for j in range(H):
        ids = np.array([(i, j, n) for i in range(W) for n in cols], np.int32)
        rad = ref.trace_paths(cam, pattern, bg, W, H, ids).reshape(W, len(cols), 3)
        for c in range(len(cols)):
            acc[H - 1 - j] = acc[H - 1 - j] + rad[:, c]
    return acc


def _worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    world, kw = rt.get_scene("cornell_smoke_box", spp=SPP)
    ref = O.OracleScene(world)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]

    def all_reduce(acc):
        t = torch.zeros(H, W, 3) if acc is None else torch.from_numpy(acc)
        dist.all_reduce(t)
        return t.numpy()

    img = render_sharded(lambda first, stride: _oracle_accumulate(ref, cam, pat, bg, first, stride), all_reduce,
                         finalize_host, SPP, rank, world_size)
    np.save(os.path.join(out_dir, f"img{rank}.npy"), img)
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


def test_two_ranks_gloo_match_single_process(tmp_path):
    world_size = 2
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    a, b = np.load(tmp_path / "img0.npy"), np.load(tmp_path / "img1.npy")
    np.testing.assert_array_equal(a, b)                       # every rank ends with the same image
    world, kw = rt.get_scene("cornell_smoke_box", spp=SPP)
    want = O.OracleScene(world).render(kw["camera"], kw["pattern"], kw["background"], W, H)
    # same set of paths, only the FP32 summation order differs (two partial sums instead of one running sum)
    np.testing.assert_allclose(a, want, rtol=2e-6, atol=1e-7)

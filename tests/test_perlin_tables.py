"""rt_perlin_generate against a plain restatement of the Perlin3 constructors (src/perlin.jl:17-33) over the scene
random stream.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from raytracing_amd.scenes import SceneRng

F = np.float32


def tables_by_hand(rng, vector):
    random = np.zeros((256, 3), dtype=F)
    for i in range(256):
        if vector:   # normalize(rand(Vec3{F}, -1, 1)): a * F(1/length(a)), src/vec3.jl:31,52,57,85
            v = np.array([rng.rand32() * F(2) + F(-1) for _ in range(3)], dtype=F)
            random[i] = v * (F(1) / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
        else:
            random[i, 0] = rng.rand32()
    perm_y = rng.shuffle(range(256))      # src/perlin.jl:20-22: perm_y, perm_z, then perm_x in place
    perm_z = rng.shuffle(range(256))
    perm_x = rng.shuffle(range(256))
    return random, perm_x, perm_y, perm_z


@pytest.mark.parametrize("vector", [True, False])
@pytest.mark.parametrize("seed,skip", [(1, 0), (7, 12345), (0xFFFFFFFF, 2 ** 32 - 100)])   # the last crosses the low counter word
def test_native_tables_are_the_restated_ones(seed, skip, vector):
    a, b = SceneRng(seed), SceneRng(seed)
    a.ctr = b.ctr = skip
    want = tables_by_hand(a, vector)
    got = rt.Perlin3(b, vector=vector)
    assert b.ctr == a.ctr == skip + (768 if vector else 256) + 3 * 255
    np.testing.assert_array_equal(got.random.view(np.uint32), want[0].view(np.uint32))
    assert (got.perm_x, got.perm_y, got.perm_z) == want[1:]
    for p in want[1:]:
        assert sorted(p) == list(range(256))
    if vector:
        np.testing.assert_allclose(np.linalg.norm(got.random.astype(np.float64), axis=1), 1.0, atol=2e-7)


def test_null_arguments_are_errors():
    L = ffi.lib()
    ctr, table = C.c_uint64(0), ffi.rtPerlin()
    assert L.rt_perlin_generate(1, None, 1, C.byref(table)) == -1
    assert L.rt_perlin_generate(1, C.byref(ctr), 1, None) == -1
    assert ctr.value == 0

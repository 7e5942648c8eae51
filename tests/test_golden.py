"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py).
On the GPU box the same vectors are what the HIP path is compared with (tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

import oracle_ffi as O
from conftest import SCENES

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_golden(name):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import golden_inputs, W, H
    g = np.load(os.path.join(HERE, "golden", f"golden_{name}.npz"))
    world, cam, pat, bg, rays, ids = golden_inputs(name)
    np.testing.assert_array_equal(rays, g["rays"])
    ref = O.OracleScene(world)
    hits = ref.trace_rays(rays)
    for k, v in hits.items():
        np.testing.assert_array_equal(v, g["hit_" + k], err_msg=k)
    np.testing.assert_array_equal(ref.trace_paths(cam, pat, bg, W, H, ids), g["radiance"])
    np.testing.assert_array_equal(ref.render(cam, pat, bg, W, H), g["image"])
    # threading must not change anything (paths are keyed, not streamed)
    np.testing.assert_array_equal(ref.render(cam, pat, bg, W, H, threads=1), g["image"])

"""Native OBJ loader (rt_obj_load, the host-side step of render(filename, material), main.jl:53-81) against the plain
Python statement of the same rules, on the shipped teapot and on small files that exercise the corner cases."""
import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from raytracing_amd.obj_loader import load_obj, load_obj_py
from raytracing_amd.scenes import ASSETS
import os


def same(path):
    v1, i1 = load_obj(path)
    v2, i2 = load_obj_py(path)
    assert v1.dtype == np.float32 and i1.dtype == np.int32
    np.testing.assert_array_equal(v1.view(np.uint32), v2.view(np.uint32))
    np.testing.assert_array_equal(i1, i2)
    return v1, i1


def test_teapot():
    v, i = same(os.path.join(ASSETS, "teapot.obj"))
    assert i.shape == (6320, 4) and i[:, :3].max() == len(v) - 1 and np.all(i[:, 3] == 0)


def test_corner_cases(tmp_path):
    p = tmp_path / "a.obj"
    p.write_text("""# comment
o thing
v 0 0 0
v 1 0 0.1
v 1 1 1e-3
v 0 1 -2.5E+1

vt 0 0
vt 1 0.25
vn 0 0 1
usemtl whatever
f 1 2 3 4
f 1/1 2/2 3/1
f 1//1 2//1 -1//1
f -4/1/1 -3/2/1 -2/1/1 -1/2/1 1/2/1
f 1   2\t3
s off
""")
    v, i = same(p)
    # quad -> 2 triangles, pentagon -> 3, fan order (1,2,3),(1,3,4),...
    assert len(i) == 2 + 1 + 1 + 3 + 1
    np.testing.assert_array_equal(i[0, :3], [0, 1, 2])
    np.testing.assert_array_equal(i[1, :3], [0, 2, 3])
    # position-only vertices: zero normal, (-1, -1) texture coordinates
    np.testing.assert_array_equal(v[0], [0, 0, 0, 0, 0, 0, -1, -1])
    # the same position with different vt / vn is a different vertex; a repeated triple is reused
    assert len(v) == 4 + 3 + 3 + 5
    np.testing.assert_array_equal(i[-1, :3], [0, 1, 2])
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\n")
    v, i = same(empty)
    assert v.shape == (0, 8) and i.shape == (0, 4)


def test_errors(tmp_path):
    with pytest.raises(ffi.RtError, match="cannot open"):
        load_obj(tmp_path / "missing.obj")
    for text, msg in (("v 1 2\n", "too few"), ("v 1 2 x\n", "malformed number"), ("v 0 0 0\nf 1 2 3\n", "out of range"),
                      ("v 0 0 0\nf 1/a 1 1\n", "malformed face"), ("v 0 0 0\nvt 0 0\nf /1 1 1\n", "without a position")):
        p = tmp_path / "bad.obj"
        p.write_text(text)
        with pytest.raises(ffi.RtError, match=msg):
            load_obj(p)

"""Scene descriptions in a file (rt_scene_desc_save / rt_scene_desc_load): what comes back builds the same device
layout as the description it was written from, and damaged files are turned down.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _ffi as ffi
from raytracing_amd.render import scene_layout


def layout_of(desc):
    info = ffi.rtLayout()
    L = ffi.lib()
    ffi.check(L.rt_scene_layout(desc, C.byref(info), None, 0, None, 0))
    nodes = np.zeros((8 * info.n_sets, info.nodes_per_layout, 4), dtype=np.uint32)
    prim_top = np.zeros(info.n_prims, dtype=np.int32)
    ffi.check(L.rt_scene_layout(desc, C.byref(info), nodes.ctypes.data_as(C.POINTER(C.c_uint32)), nodes.size // 4,
                                ffi.iptr(prim_top), prim_top.size))
    return info, nodes, prim_top


@pytest.mark.parametrize("name", ["cornell_smoke_box", "cover2", "teapot"])   # media + transforms, images + noise + nested tree, mesh
def test_round_trip_builds_the_same_device_layout(name, tmp_path):
    world, kw = rt.get_scene(name, spp=1)
    path = tmp_path / (name + ".rtscene")
    rt.save_scene(path, world)
    assert path.stat().st_size > 64
    info, nodes, prim_top = scene_layout(world)
    desc = C.POINTER(ffi.rtSceneDesc)()
    L = ffi.lib()
    ffi.check(L.rt_scene_desc_load(os.fsencode(path), C.byref(desc)))
    try:
        info2, nodes2, prim_top2 = layout_of(desc)
        assert desc.contents.n_top == len(world)
    finally:
        L.rt_scene_desc_free(desc)
    assert (info2.nodes_per_layout, info2.n_prims, info2.n_sets, info2.surface_size) == \
           (info.nodes_per_layout, info.n_prims, info.n_sets, info.surface_size)
    np.testing.assert_array_equal(nodes2, nodes)
    np.testing.assert_array_equal(prim_top2, prim_top)
    # saving what was loaded gives the same bytes
    desc = C.POINTER(ffi.rtSceneDesc)()
    ffi.check(L.rt_scene_desc_load(os.fsencode(path), C.byref(desc)))
    again = tmp_path / "again.rtscene"
    try:
        ffi.check(L.rt_scene_desc_save(desc, os.fsencode(again)))
    finally:
        L.rt_scene_desc_free(desc)
    assert again.read_bytes() == path.read_bytes()


def test_damaged_files_are_turned_down(tmp_path):
    world, kw = rt.get_scene("cornell_box", spp=1)
    path = tmp_path / "c.rtscene"
    rt.save_scene(path, world)
    data = path.read_bytes()
    L = ffi.lib()
    desc = C.POINTER(ffi.rtSceneDesc)()
    cases = {"cut": data[:len(data) // 2], "tail": data + b"\0\0\0\0", "magic": b"NOTASCENE" + data[9:],
             "count": data[:8] + (2 ** 40).to_bytes(8, "little") + data[16:], "empty": b""}
    for tag, blob in cases.items():
        bad = tmp_path / (tag + ".rtscene")
        bad.write_bytes(blob)
        assert L.rt_scene_desc_load(os.fsencode(bad), C.byref(desc)) == -1, tag   # RT_EINVAL
        assert not desc
    assert L.rt_scene_desc_load(os.fsencode(tmp_path / "missing.rtscene"), C.byref(desc)) == -1
    assert b"cannot open" in L.rt_last_error()
    L.rt_scene_desc_free(None)   # a null description is ignored


@pytest.mark.parametrize("name", ["cornell_smoke_box", "cover2", "teapot"])
def test_float64_descriptions_round_trip(name, tmp_path):
    """rt_scene_desc_save_f64 / _load_f64: the tables come back value for value, saving them again gives the same bytes,
    and each loader turns the other element type's file down"""
    from raytracing_amd import hittables as H
    from raytracing_amd.scene_manager import SceneManager, build_bvh_scene, flatten_scene
    world, kw = rt.get_scene(name, spp=1, F=np.float64)
    path = tmp_path / (name + ".rtscene64")
    rt.save_scene(path, world, F=np.float64)
    with H.element_type(np.float64):
        want, keep = flatten_scene(world, build_bvh_scene(SceneManager(world)))
    L = ffi.lib()
    desc = C.POINTER(ffi.rtSceneDesc64)()
    ffi.check(L.rt_scene_desc_load_f64(os.fsencode(path), C.byref(desc)))
    try:
        d = desc.contents
        for f in ("n_objects", "n_top", "n_transforms", "n_materials", "n_textures", "n_images", "n_perlins", "n_meshes",
                  "n_groups", "n_buckets"):
            assert getattr(d, f) == getattr(want, f), f
        assert C.string_at(d.objects, d.n_objects * C.sizeof(ffi.rtObject64)) == \
               C.string_at(want.objects, want.n_objects * C.sizeof(ffi.rtObject64))
        for b in range(d.n_buckets):
            got_t, want_t = d.buckets[b], want.buckets[b]
            assert (got_t.kind, got_t.n_objects, got_t.n_nodes) == (want_t.kind, want_t.n_objects, want_t.n_nodes)
            assert C.string_at(got_t.node_bounds, 48 * got_t.n_nodes) == C.string_at(want_t.node_bounds, 48 * want_t.n_nodes)
            assert C.string_at(got_t.node_leaf, got_t.n_nodes) == C.string_at(want_t.node_leaf, want_t.n_nodes)
        again = tmp_path / "again.rtscene64"
        ffi.check(L.rt_scene_desc_save_f64(desc, os.fsencode(again)))
    finally:
        L.rt_scene_desc_free_f64(desc)
    assert again.read_bytes() == path.read_bytes()
    # the file carries its element type
    d32, d64 = C.POINTER(ffi.rtSceneDesc)(), C.POINTER(ffi.rtSceneDesc64)()
    assert L.rt_scene_desc_load(os.fsencode(path), C.byref(d32)) == -1 and not d32
    assert b"element type" in L.rt_last_error()
    rt.save_scene(tmp_path / "f32.rtscene", rt.get_scene(name, spp=1)[0])
    assert L.rt_scene_desc_load_f64(os.fsencode(tmp_path / "f32.rtscene"), C.byref(d64)) == -1 and not d64
    assert L.rt_scene_desc_load_f64(os.fsencode(tmp_path / "missing"), C.byref(d64)) == -1
    assert L.rt_scene_desc_save_f64(None, os.fsencode(again)) == -1
    L.rt_scene_desc_free_f64(None)
    del keep

"""The Julia side of the boundary (julia/*.jl) cannot run in the build image (no julia), so it is held to the C ABI by
text: the isbits structs of julia/librt_amd.jl against the C header (compiled here with gcc) and the ctypes mirror, field
by field; every ccall against the declared entry points; block structure of the files; and the overlay script against a
copy of the reference when the reference is at hand."""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys

import pytest

from raytracing_amd import _ffi as ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JULIA = os.path.join(ROOT, "julia")
REFERENCE = "/root/reference"

SCALARS = {"Int32": (4, 4), "UInt32": (4, 4), "Float32": (4, 4), "Float64": (8, 8), "UInt64": (8, 8), "Int64": (8, 8),
           "UInt8": (1, 1), "Cint": (4, 4)}
PAIRS = {"RtObject": "rtObject", "RtTransform": "rtTransform", "RtMaterial": "rtMaterial", "RtTexture": "rtTexture",
         "RtImage": "rtImage", "RtPerlin": "rtPerlin", "RtMesh": "rtMesh", "RtBvh": "rtBvh", "RtSceneDesc": "rtSceneDesc",
         "RtCamera": "rtCamera", "RtStats": "rtStats",
         "RtObject64": "rtObject64", "RtTransform64": "rtTransform64", "RtMaterial64": "rtMaterial64", "RtTexture64": "rtTexture64",
         "RtPerlin64": "rtPerlin64", "RtMesh64": "rtMesh64", "RtBvh64": "rtBvh64", "RtSceneDesc64": "rtSceneDesc64",
         "RtCamera64": "rtCamera64"}


def julia_structs():
    text = open(os.path.join(JULIA, "librt_amd.jl")).read()
    out = {}
    for m in re.finditer(r"^struct (\w+)\n(.*?)^end", text, flags=re.M | re.S):
        fields = []
        for line in m.group(2).splitlines():
            line = line.split("#")[0].strip()
            if line:
                name, typ = line.split("::")
                fields.append((name.strip(), typ.strip()))
        out[m.group(1)] = fields
    return out


def size_align(typ):
    if typ in SCALARS:
        return SCALARS[typ]
    if typ.startswith("Ptr{"):
        return 8, 8
    m = re.fullmatch(r"NTuple\{(\d+),\s*(\w+)\}", typ)
    if m:
        s, a = SCALARS[m.group(2)]
        return int(m.group(1)) * s, a
    raise AssertionError("unknown Julia field type " + typ)


def layout(fields):
    """offsets and total size of an isbits struct (C layout rules, which Julia follows for such structs)"""
    off, worst, res = 0, 1, []
    for name, typ in fields:
        s, a = size_align(typ)
        off = (off + a - 1) // a * a
        res.append((name, off, s))
        off += s
        worst = max(worst, a)
    return res, (off + worst - 1) // worst * worst


def c_layouts(tmp_path):
    """sizeof / offsetof of every struct of include/rt_amd.h, from gcc"""
    fields = {c: [f for f, _ in getattr(ffi, c)._fields_] for c in PAIRS.values()}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_amd.h"', 'int main(void) {']
    for c, names in fields.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (c, c))
        for f in names:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (c, f, c, f))
    src.append("return 0; }")
    cfile, exe = tmp_path / "layout.c", tmp_path / "layout"
    cfile.write_text("\n".join(src))
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)])
    out = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        k, v = line.split()
        out[k] = int(v)
    return out


def test_julia_structs_match_the_header_and_ctypes(tmp_path):
    structs = julia_structs()
    in_c = c_layouts(tmp_path)
    assert set(PAIRS) <= set(structs), set(PAIRS) - set(structs)
    for jl, c in PAIRS.items():
        ct = getattr(ffi, c)
        offsets, size = layout(structs[jl])
        assert [n for n, _, _ in offsets] == [n for n, _ in ct._fields_], (jl, "field names / order")
        assert size == C.sizeof(ct) == in_c[c], (jl, size, C.sizeof(ct), in_c[c])
        for name, off, fsize in offsets:
            cf = getattr(ct, name)
            assert off == cf.offset == in_c["%s.%s" % (c, name)], (jl, name, off, cf.offset)
            assert fsize == cf.size, (jl, name, fsize, cf.size)
    # Camera{Float32} (src/camera.jl:3-14: seven Vec3 and three scalars) is passed as it is where the library wants an rtCamera
    assert C.sizeof(ffi.rtCamera) == 7 * 12 + 3 * 4
    assert C.sizeof(ffi.rtCamera64) == 7 * 24 + 3 * 8       # ... and Camera{Float64} an rtCamera64


def julia_sources():
    return {n: open(os.path.join(JULIA, n)).read() for n in sorted(os.listdir(JULIA)) if n.endswith(".jl")}


def test_status_codes_and_kinds_match_the_header():
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = julia_sources()["librt_amd.jl"]
    status = re.search(r"typedef enum rtStatus \{(.*?)\}", header, flags=re.S).group(1)
    codes = re.findall(r"(RT_\w+) = (-?\d+)", status)
    assert len(codes) == 7
    for name, value in codes:
        m = re.search(r"const %s = Cint\((-?\d+)\)" % name, text)
        assert m and int(m.group(1)) == int(value), name
    kinds = dict(re.findall(r"(RT_[A-Z_]+) = (\d+)", header))
    for names, values in re.findall(r"^const ((?:RT_\w+, )+RT_\w+) = (.+)$", text, flags=re.M):
        for n, v in zip(names.split(", "), re.findall(r"Int32\((\d+)\)", values)):
            assert kinds[n] == v, (n, kinds[n], v)


def test_every_ccall_names_a_declared_entry_point_with_matching_arguments():
    kind = {"Cint": C.c_int, "UInt64": C.c_uint64, "UInt32": C.c_uint32, "Cstring": C.c_char_p}
    seen = set()
    for name, text in julia_sources().items():
        for m in re.finditer(r"ccall\(\(:(\w+), librt\), (\w+),\s*\(([^()]*)\)", text):
            fn, ret, args = m.group(1), m.group(2), m.group(3)
            assert fn in ffi.SYMBOLS, (name, fn)
            seen.add(fn)
            res, argtypes = ffi.SYMBOLS[fn]
            assert (ret == "Cstring") == (res is C.c_char_p) and (ret == "Cint") == (res is C.c_int), (fn, ret)
            # split the argument tuple at top-level commas
            parts, depth, cur = [], 0, ""
            for ch in args:
                if ch == "," and depth == 0:
                    parts.append(cur.strip()); cur = ""
                else:
                    depth += ch == "{"
                    depth -= ch == "}"
                    cur += ch
            if cur.strip():
                parts.append(cur.strip())
            assert len(parts) == len(argtypes), (fn, parts, len(argtypes))
            for p, a in zip(parts, argtypes):
                if p.startswith(("Ptr{", "Ref{")):
                    assert a is C.c_void_p or hasattr(a, "contents") or a is C.c_char_p, (fn, p, a)
                else:
                    assert kind[p] is a, (fn, p, a)
    assert {"rt_scene_create", "rt_scene_destroy", "rt_render", "rt_group_create", "rt_group_render", "rt_group_destroy",
            "rt_last_error", "rt_device_count", "rt_scene_create_f64", "rt_scene_destroy_f64", "rt_render_f64", "rt_group_create_f64",
            "rt_group_render_f64", "rt_group_destroy_f64"} <= seen


def strip_code(text):
    """Julia source without comments, strings and bracketed expressions (where `end` and `for` are not block words)"""
    text = re.sub(r'"""(?:.|\n)*?"""', '""', text)
    text = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', text)
    text = re.sub(r"#[^\n]*", "", text)
    prev = None
    while prev != text:
        prev = text
        text = re.sub(r"\[[^\[\]]*\]", "", text)
    return text


@pytest.mark.parametrize("name", sorted(n for n in os.listdir(JULIA) if n.endswith(".jl")))
def test_blocks_are_closed(name):
    code = strip_code(open(os.path.join(JULIA, name)).read())
    opens = len(re.findall(r"(?<![\w.:])(?:function|struct|if|for|while|try|begin|do|let|module|quote|macro)(?![\w!])", code))
    opens -= len(re.findall(r"(?<![\w.:])mutable\s+struct", code)) * 0      # (`mutable struct` opens one block: counted once)
    closes = len(re.findall(r"(?<![\w.:])end(?![\w!])", code))
    assert opens == closes, (name, opens, closes)
    assert code.count("(") == code.count(")") and code.count("{") == code.count("}"), name


def test_every_hittable_of_the_reference_has_an_add_method():
    text = julia_sources()["cuscene_manager.jl"]
    handled = set(re.findall(r"add_new!\(t::SceneTables, o::(\w+)\)", text))
    assert handled == {"Sphere", "MovingSphere", "xy_rect", "xz_rect", "yz_rect", "Box", "Transform", "ConstantMedium", "BVHTree", "Mesh"}
    textures = set(re.findall(r"add_texture!\(t::SceneTables, tex::(\w+)", text))
    assert textures == {"SolidColor", "CheckerTexture", "ImageTexture", "NoiseTexture"}


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src")), reason="the reference checkout is not on this machine")
def test_overlay_applies_to_the_reference(tmp_path):
    """julia/apply_overlay.py on a copy of the reference: every anchor is found, nothing of CUDA.jl / Adapt.jl is left in
    code, the device file is included, and the field names the glue reads exist in the reference's structs."""
    work = tmp_path / "ref"
    shutil.copytree(REFERENCE, work)
    for d, _, files in os.walk(work):
        os.chmod(d, 0o755)
        for f in files:
            os.chmod(os.path.join(d, f), 0o644)
    subprocess.check_call([sys.executable, os.path.join(JULIA, "apply_overlay.py"), str(work)])
    with pytest.raises(subprocess.CalledProcessError):          # a second time the anchors are gone: it must say so, not mangle
        subprocess.check_call([sys.executable, os.path.join(JULIA, "apply_overlay.py"), str(work)], stderr=subprocess.DEVNULL)
    for d in (work / "src", work):
        for f in sorted(os.listdir(d)):
            if (not f.endswith(".jl") and f != "Project.toml") or f in ("librt_amd.jl", "cuBVH.jl", "cuscene_manager.jl", "Ray_Tracing_device.jl"):
                continue            # (the four files of julia/ name what they replace in their comments)
            code = re.sub(r"#[^\n]*", "", open(d / f).read())
            code = re.sub(r'"""(?:.|\n)*?"""', "", code)
            for word in ("CUDA", "Adapt", "CuArray", "CuDeviceArray", "CuBVHTree", "@cuda", "cu_trace", "draw_kernel"):
                assert word not in code, (f, word)
    rt = open(work / "src" / "Ray_Tracing.jl").read()
    assert 'include("Ray_Tracing_device.jl")' in rt and "function trace(" in rt and "function draw!(img::Array{RGB{F},2}" in rt
    assert 'include("hittable.jl")' in open(work / "src" / "BVH.jl").read()
    assert "DeviceImage(F, height, width)" in open(work / "main.jl").read()
    # the fields the glue reads, in the reference's own struct definitions
    src = "".join(open(work / "src" / f).read() for f in os.listdir(work / "src") if f.endswith(".jl") and not f.startswith(("cu", "librt", "Ray_Tracing_device")))
    want = {"Sphere": ["center", "radius", "material"], "MovingSphere": ["center0", "t0", "center1", "t1", "radius", "material"],
            "xy_rect": ["x0", "x1", "y0", "y1", "k", "material"], "xz_rect": ["x0", "x1", "z0", "z1", "k", "material"],
            "yz_rect": ["y0", "y1", "z0", "z1", "k", "material"], "Box": ["box_min", "box_max", "sides"],
            "Transform": ["object", "xform", "inv_xform", "bbox"], "ConstantMedium": ["boundary", "phase_function", "neg_inv_density"],
            "BVHTree": ["nodes", "objects"], "BVHNode": ["box", "left", "right", "is_leaf"], "Mesh": ["vertices", "materials", "indices"],
            "Vertex": ["position", "normal", "u", "v"], "Material": ["type", "albedo", "fuzz", "eta"], "Perlin3": ["random", "perm_x", "perm_y", "perm_z"],
            "NoiseTexture": ["noise", "scale"], "CheckerTexture": ["odd", "even"], "ImageTexture": ["image"], "SolidColor": ["color"],
            "Camera": ["origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w", "lens_radius", "t0", "t1"]}
    for struct, names in want.items():
        m = re.search(r"struct %s(?:\{[^\n]*)?(?: <: [^\n]*)?\n(.*?)\nend" % struct, src, flags=re.S)
        assert m, struct
        have = re.findall(r"^\s+(\w+)::", m.group(1), flags=re.M)
        assert have[:len(names)] == names, (struct, have)


# ---- the call path: which keywords reach draw!(::DeviceImage, ...) and whether its methods take them ----------------

def julia_signature(text, head):
    """The keyword part of the method whose definition starts with `head`: (named keywords, name of the slurp or None)"""
    a = text.index(head)
    depth, i = 0, a + head.index("(")        # the opening parenthesis of the argument list
    for j in range(i, len(text)):
        depth += text[j] == "("
        depth -= text[j] == ")"
        if depth == 0:
            break
    args = text[i + 1:j]
    if ";" not in args:
        return set(), None
    kw = args.split(";", 1)[1]
    parts, depth, cur = [], 0, ""
    for ch in kw:
        if ch == "," and depth == 0:
            parts.append(cur.strip()); cur = ""
        else:
            depth += ch in "({["
            depth -= ch in ")}]"
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    named, slurp = set(), None
    for p in parts:
        if p.endswith("..."):
            slurp = p[:-3]
        else:
            named.add(re.match(r"\w+", p).group(0))
    return named, slurp


def call_keywords(call):
    """names of the keywords written in a Julia call `render(...; a = 1, b = 2)`"""
    return set(re.findall(r"(\w+)\s*=", call.split(";", 1)[1])) if ";" in call else set()


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src")), reason="the reference checkout is not on this machine")
def test_documented_render_calls_reach_draw_with_keywords_it_takes(tmp_path):
    """render(:cornell_box; spp = 10000, gpu = true) is the reference's own GPU example (README.md:79).  Its keywords
    travel: render(F, key; gpu, kwargs...) (main.jl:90-94) hands all of them to the scene function (which takes
    `kwargs...`), merges them over the scene's own and calls render(F, world; ...) (main.jl:21-29), which names six and
    collects the rest in `kwargs...`; the patched gpu branch passes that collection on to draw!(::DeviceImage, ...).
    Every keyword that can arrive there -- `spp` among them -- must be taken by the method, or the call is a MethodError.
    Walked here on the overlay applied to a copy of the reference, for the README's calls and INTEGRATION.md's."""
    work = tmp_path / "ref"
    shutil.copytree(REFERENCE, work)
    for d, _, files in os.walk(work):
        os.chmod(d, 0o755)
        for f in files:
            os.chmod(os.path.join(d, f), 0o644)
    subprocess.check_call([sys.executable, os.path.join(JULIA, "apply_overlay.py"), str(work)])
    main = open(work / "main.jl").read()
    device = open(work / "src" / "Ray_Tracing_device.jl").read()

    # the documented calls
    readme = open(os.path.join(REFERENCE, "README.md")).read()
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    calls = re.findall(r"render\(:\w+[^)\n]*\)", readme) + re.findall(r"render\(:\w+[^)\n]*\)", integration)
    assert any("spp=10000, gpu=true" in c for c in calls) and any("devices" in c for c in calls) and any("seed" in c for c in calls), calls

    # render(F, key::Symbol; gpu, kwargs...): what it names, what it passes on
    by_key, slurp = julia_signature(main, "function render(::Type{F}, key::Symbol")
    assert by_key == {"gpu"} and slurp == "kwargs"
    assert "render(F, world; kwargs..., gpu=gpu)" in main and "kwargs = merge(scene_kwargs, kwargs)" in main
    scenes = open(work / "scenes.jl").read()
    scene_kwargs = set(re.findall(r"(\w+)=\w+", re.search(r"return world, \((.*?)\)", scenes).group(1)))
    assert scene_kwargs == {"width", "height", "camera", "background", "pattern"}
    for f in os.listdir(work / "samples"):                      # the scene functions take whatever they are handed
        head = re.search(r"^scenes\[:\w+\] = \(;(.*?)\) -> begin", open(work / "samples" / f).read(), flags=re.M).group(1)
        assert "kwargs..." in head, f

    # render(F, world::Vector; ...): what it names; the rest is `kwargs...`, which the gpu branch forwards
    by_world, slurp = julia_signature(main, "function render(::Type{F}, world::Vector{<:Hittable}")
    assert by_world == {"width", "height", "camera", "background", "pattern", "gpu"} and slurp == "kwargs"
    forward = re.search(r"draw!\(d_img, world, camera;([^\n]*)\)", main).group(1)
    assert "kwargs..." in forward
    always = set(re.findall(r"(\w+)\s*=", forward))              # background, pattern
    assert always == {"background", "pattern"}

    methods = []
    for F in ("Float32", "Float64"):
        named, sink = julia_signature(device, "function draw!(img::DeviceImage{%s}, world::SceneManager, camera::Camera{%s}" % (F, F))
        assert {"background", "pattern", "seed", "devices", "max_depth"} <= named, named
        methods.append((F, named, sink))

    for call in calls:
        arriving = (call_keywords(call) | scene_kwargs) - by_key          # after the merge of main.jl:92
        at_draw = (arriving - by_world) | always                           # what render(F, world) does not name + what it passes itself
        for F, named, sink in methods:
            missing = at_draw - named
            assert not missing or sink is not None, (call, F, "draw! has no keyword or sink for", missing)
    # ... and the README's GPU example does carry a keyword draw! does not name, so the sink is what makes it work
    named, sink = methods[0][1], methods[0][2]
    assert "spp" not in named and sink is not None


def test_pointers_passed_to_ccall_are_rooted():
    """A `pointer(x)` handed to ccall keeps nothing alive: every array whose raw pointer crosses the boundary in
    CuSceneManager / draw! must be named in the GC.@preserve that encloses the call."""
    for name in ("cuscene_manager.jl", "Ray_Tracing.jl"):
        text = julia_sources()[name]
        for m in re.finditer(r"GC\.@preserve ([^\n]*?) begin\n(.*?)\n    end", text, flags=re.S):
            rooted = set(m.group(1).split())
            for used in re.findall(r"pointer\((\w+)\)", m.group(2)):
                assert used in rooted, (name, used, rooted)
    text = julia_sources()["cuscene_manager.jl"]
    body = text[text.index("function CuSceneManager(bvh::Tuple"):]
    body = body[:body.index("\nend\n")]
    assert "pointer(list)" in body and re.search(r"GC\.@preserve keep list begin", body)
    assert body.index("GC.@preserve") < body.index("pointer(list)")


def test_an_empty_world_is_rendered_by_the_glue():
    """build_bvh of an empty group is a tree without nodes, which is never hit (src/scene_manager.jl:25-29,
    src/BVH.jl:23-25); the library wants objects in every tree, so draw! leaves such trees out and paints the background
    itself when nothing is left -- before anything indexes bvh[1]."""
    text = julia_sources()["Ray_Tracing.jl"]
    for F in ("Float32", "Float64"):
        body = text[text.index("function draw!(img::DeviceImage{%s}" % F):]
        body = body[:body.index("\nend\n")]
        a, b, c = body.index("nonempty_trees(build_bvh(world))"), body.index("isempty(bvh) && return fill_background!"), body.index("CuSceneManager(bvh")
        assert a < b < c
    assert "nonempty_trees(bvh::Tuple) = filter(tree -> !isempty(tree.nodes), bvh)" in text
    flat = julia_sources()["cuscene_manager.jl"]
    assert flat.index("isempty(bvh) && error(") < flat.index("element_type(bvh[1])")

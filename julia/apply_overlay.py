#!/usr/bin/env python3
"""Turns a checkout of rafaellg79/raytracing into one whose `gpu = true` path runs on librt_amd.so.

    python julia/apply_overlay.py /path/to/raytracing [--lib /path/to/librt_amd.so]

What it does (every edit is anchored on the text it expects and stops with a message if the checkout differs):
  * copies librt_amd.jl, cuBVH.jl, cuscene_manager.jl into src/ (the latter two replace the CUDA.jl files of the same
    names) and Ray_Tracing.jl into src/Ray_Tracing_device.jl
  * src/Ray_Tracing.jl: `using CUDA` out, cu_trace / draw_kernel / draw!(::CuArray) out, include of the device file in
  * src/transform.jl, src/constant_medium.jl, src/mesh.jl: the Adapt rules and the CuBVHTree traversal out
  * src/BVH.jl: include("Hittable.jl") -> include("hittable.jl") (the file's real name; case matters on Linux)
  * main.jl: the gpu branch of render() (see julia/main.jl)
  * Project.toml: CUDA and Adapt out, Libdl in
Nothing else of the reference is touched: its CPU path, its scenes and its public API stay as they are.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


class OverlayError(RuntimeError):
    pass


def read(path):
    with open(path) as f:
        return f.read().split("\n")


def write(path, lines):
    with open(path, "w") as f:
        f.write("\n".join(lines))


def find(lines, prefix, path):
    hits = [i for i, l in enumerate(lines) if l.startswith(prefix)]
    if len(hits) != 1:
        raise OverlayError("%s: expected exactly one line starting with %r, found %d" % (path, prefix, len(hits)))
    return hits[0]


def remove_block(lines, prefix, path):
    """Removes the top-level block (function ... end) whose first line starts with `prefix`, and one blank line after it."""
    a = find(lines, prefix, path)
    b = a
    while b < len(lines) and lines[b].rstrip() != "end":
        b += 1
    if b == len(lines):
        raise OverlayError("%s: no closing `end` for the block at %r" % (path, prefix))
    del lines[a:b + 1]
    if a < len(lines) and lines[a].strip() == "":
        del lines[a]


def replace_line(lines, old, new, path):
    i = find(lines, old, path)
    if lines[i].rstrip() != old:
        raise OverlayError("%s: line %d is %r, expected %r" % (path, i + 1, lines[i], old))
    lines[i] = new


def apply(root, lib=None):
    src = os.path.join(root, "src")
    for name in ("Ray_Tracing.jl", "cuBVH.jl", "cuscene_manager.jl", "transform.jl", "constant_medium.jl", "mesh.jl", "BVH.jl"):
        if not os.path.exists(os.path.join(src, name)):
            raise OverlayError("%s is not a checkout of the reference: src/%s is missing" % (root, name))
    # the files of this directory
    shutil.copy(os.path.join(HERE, "librt_amd.jl"), os.path.join(src, "librt_amd.jl"))
    shutil.copy(os.path.join(HERE, "cuBVH.jl"), os.path.join(src, "cuBVH.jl"))
    shutil.copy(os.path.join(HERE, "cuscene_manager.jl"), os.path.join(src, "cuscene_manager.jl"))
    shutil.copy(os.path.join(HERE, "Ray_Tracing.jl"), os.path.join(src, "Ray_Tracing_device.jl"))
    if lib:
        p = os.path.join(src, "librt_amd.jl")
        lines = read(p)
        i = find(lines, "const librt = ", p)
        lines[i] = 'const librt = get(ENV, "RT_AMD_LIB", %s)' % ('"' + os.path.abspath(lib).replace("\\", "\\\\") + '"')
        write(p, lines)

    p = os.path.join(src, "BVH.jl")
    lines = read(p)
    replace_line(lines, 'include("Hittable.jl")', 'include("hittable.jl")', p)
    write(p, lines)

    p = os.path.join(src, "Ray_Tracing.jl")
    lines = read(p)
    replace_line(lines, "using Images, Random, CUDA", "using Images, Random", p)
    remove_block(lines, "function cu_trace(", p)
    remove_block(lines, "function draw_kernel(", p)
    remove_block(lines, "function draw!(img::CuArray{RGB{F},2}", p)
    i = find(lines, 'include("camera.jl")', p)
    lines.insert(i + 1, 'include("Ray_Tracing_device.jl")')
    write(p, lines)

    for name, prefix in (("transform.jl", "function Adapt.adapt_structure(to, transform::Transform)"),
                         ("constant_medium.jl", "function Adapt.adapt_structure(to, constant_medium::ConstantMedium)")):
        p = os.path.join(src, name)
        lines = read(p)
        replace_line(lines, "import Adapt, Base.eltype", "import Base.eltype", p)
        remove_block(lines, prefix, p)
        write(p, lines)

    p = os.path.join(src, "mesh.jl")
    lines = read(p)
    i = find(lines, "import Adapt", p)
    del lines[i]
    remove_block(lines, "function Adapt.adapt_structure(to, mesh::Mesh)", p)
    remove_block(lines, "function Adapt.adapt_structure(to, bvh::CuBVHTree{A, B})", p)
    remove_block(lines, "function cu_hit(next::CuDeviceArray, tree::CuBVHTree{A, <:CuDeviceArray{T}}", p)
    write(p, lines)

    p = os.path.join(root, "main.jl")
    lines = read(p)
    i = find(lines, "        d_img = CUDA.zeros(eltype(img), height, width)", p)
    lines[i] = "        d_img = DeviceImage(F, height, width)"
    j = find(lines, "        draw!(d_img, world, camera; background = background, pattern = pattern)", p)
    lines[j] = "        draw!(d_img, world, camera; background = background, pattern = pattern, kwargs...)"
    write(p, lines)

    p = os.path.join(root, "Project.toml")
    lines = [l for l in read(p) if not l.startswith("Adapt = ") and not l.startswith("CUDA = ")]
    i = find(lines, "Images = ", p)
    lines.insert(i + 1, 'Libdl = "8f399da3-3557-5675-b5ff-fb832c97cbdb"')
    write(p, lines)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    lib = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else None
    try:
        apply(sys.argv[1], lib)
    except OverlayError as e:
        sys.exit("apply_overlay: " + str(e))
    print("overlay applied to", sys.argv[1])

# librt_amd.jl -- the C ABI of librt_amd.so (include/rt_amd.h) as Julia sees it: isbits structs with the field order,
# sizes and offsets of the C structs, the status codes, and `check`.  Included by cuBVH.jl / cuscene_manager.jl /
# Ray_Tracing.jl of this directory, which replace the CUDA.jl halves of the files of the same names in src/.
#
# tests/test_julia_glue.py parses the struct declarations below and holds them against the C header (compiled with gcc)
# and the ctypes mirror, field by field: a change on either side shows up there even though Julia is not installed in
# the build image.
using Libdl

const librt = get(ENV, "RT_AMD_LIB", joinpath(@__DIR__, "..", "raytracing_amd", "librt_amd.so"))

const RT_OK = Cint(0)
const RT_EINVAL = Cint(-1)
const RT_ENODEV = Cint(-2)
const RT_ENOMEM = Cint(-3)
const RT_EHIP = Cint(-4)
const RT_EUNSUPPORTED = Cint(-5)
const RT_ERCCL = Cint(-6)

# object kinds, material types, texture kinds (include/rt_amd.h)
const RT_SPHERE, RT_MOVING_SPHERE, RT_XY_RECT, RT_XZ_RECT, RT_YZ_RECT = Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)
const RT_BOX, RT_TRANSFORM, RT_CONSTANT_MEDIUM, RT_BVH, RT_MESH = Int32(6), Int32(7), Int32(8), Int32(9), Int32(10)
const RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE, RT_TEX_NOISE_MARBLE, RT_TEX_NOISE_SCALAR = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)

struct RtObject
    kind::Int32
    material::Int32
    child::Int32
    aux::Int32
    p::NTuple{12,Float32}
end

struct RtTransform
    xform::NTuple{12,Float32}
    inv_xform::NTuple{12,Float32}
    bbox::NTuple{6,Float32}
end

struct RtMaterial
    type::Int32
    texture::Int32
    fuzz::Float32
    eta::Float32
end

struct RtTexture
    kind::Int32
    a::Int32
    b::Int32
    pad::Int32
    color::NTuple{3,Float32}
    scale::Float32
end

struct RtImage
    width::Int32
    height::Int32
    rgb::Ptr{UInt8}
end

struct RtPerlin
    ranvec::NTuple{768,Float32}
    perm_x::NTuple{256,Int32}
    perm_y::NTuple{256,Int32}
    perm_z::NTuple{256,Int32}
end

struct RtMesh
    n_vertices::Int32
    n_triangles::Int32
    n_materials::Int32
    pad::Int32
    vertices::Ptr{Float32}
    indices::Ptr{Int32}
    materials::Ptr{Int32}
    tri_ids::Ptr{Int32}
end

struct RtBvh
    kind::Int32
    n_objects::Int32
    objects::Ptr{Int32}
    n_nodes::Int32
    pad::Int32
    node_bounds::Ptr{Float32}
    node_left::Ptr{Int32}
    node_right::Ptr{Int32}
    node_leaf::Ptr{UInt8}
end

struct RtSceneDesc
    n_objects::Int32
    objects::Ptr{RtObject}
    n_top::Int32
    top::Ptr{Int32}
    n_transforms::Int32
    transforms::Ptr{RtTransform}
    n_materials::Int32
    materials::Ptr{RtMaterial}
    n_textures::Int32
    textures::Ptr{RtTexture}
    n_images::Int32
    images::Ptr{RtImage}
    n_perlins::Int32
    perlins::Ptr{RtPerlin}
    n_meshes::Int32
    meshes::Ptr{RtMesh}
    n_groups::Int32
    groups::Ptr{RtBvh}
    n_buckets::Int32
    buckets::Ptr{RtBvh}
end

# == Camera{Float32} (src/camera.jl:3-14): a Ref(camera) can be passed where the library wants an rtCamera
struct RtCamera
    origin::NTuple{3,Float32}
    lower_left_corner::NTuple{3,Float32}
    horizontal::NTuple{3,Float32}
    vertical::NTuple{3,Float32}
    u::NTuple{3,Float32}
    v::NTuple{3,Float32}
    w::NTuple{3,Float32}
    lens_radius::Float32
    t0::Float32
    t1::Float32
end

# ---- F = Float64 (render(Float64, ...), main.jl:21,46): the same records with double fields (rtObject64 ... rtCamera64 of
# include/rt_amd.h), taken by the rt_*_f64 entry points

struct RtObject64
    kind::Int32
    material::Int32
    child::Int32
    aux::Int32
    p::NTuple{12,Float64}
end

struct RtTransform64
    xform::NTuple{12,Float64}
    inv_xform::NTuple{12,Float64}
    bbox::NTuple{6,Float64}
end

struct RtMaterial64
    type::Int32
    texture::Int32
    fuzz::Float64
    eta::Float64
end

struct RtTexture64
    kind::Int32
    a::Int32
    b::Int32
    pad::Int32
    color::NTuple{3,Float64}
    scale::Float64
end

struct RtPerlin64
    ranvec::NTuple{768,Float64}
    perm_x::NTuple{256,Int32}
    perm_y::NTuple{256,Int32}
    perm_z::NTuple{256,Int32}
end

struct RtMesh64
    n_vertices::Int32
    n_triangles::Int32
    n_materials::Int32
    pad::Int32
    vertices::Ptr{Float64}
    indices::Ptr{Int32}
    materials::Ptr{Int32}
    tri_ids::Ptr{Int32}
end

struct RtBvh64
    kind::Int32
    n_objects::Int32
    objects::Ptr{Int32}
    n_nodes::Int32
    pad::Int32
    node_bounds::Ptr{Float64}
    node_left::Ptr{Int32}
    node_right::Ptr{Int32}
    node_leaf::Ptr{UInt8}
end

struct RtSceneDesc64
    n_objects::Int32
    objects::Ptr{RtObject64}
    n_top::Int32
    top::Ptr{Int32}
    n_transforms::Int32
    transforms::Ptr{RtTransform64}
    n_materials::Int32
    materials::Ptr{RtMaterial64}
    n_textures::Int32
    textures::Ptr{RtTexture64}
    n_images::Int32
    images::Ptr{RtImage}
    n_perlins::Int32
    perlins::Ptr{RtPerlin64}
    n_meshes::Int32
    meshes::Ptr{RtMesh64}
    n_groups::Int32
    groups::Ptr{RtBvh64}
    n_buckets::Int32
    buckets::Ptr{RtBvh64}
end

# == Camera{Float64}
struct RtCamera64
    origin::NTuple{3,Float64}
    lower_left_corner::NTuple{3,Float64}
    horizontal::NTuple{3,Float64}
    vertical::NTuple{3,Float64}
    u::NTuple{3,Float64}
    v::NTuple{3,Float64}
    w::NTuple{3,Float64}
    lens_radius::Float64
    t0::Float64
    t1::Float64
end

# the record types of an element type
records(::Type{Float32}) = (object = RtObject, transform = RtTransform, material = RtMaterial, texture = RtTexture, perlin = RtPerlin,
                            mesh = RtMesh, bvh = RtBvh, desc = RtSceneDesc)
records(::Type{Float64}) = (object = RtObject64, transform = RtTransform64, material = RtMaterial64, texture = RtTexture64,
                            perlin = RtPerlin64, mesh = RtMesh64, bvh = RtBvh64, desc = RtSceneDesc64)
records(::Type{F}) where F = error("librt_amd: the GPU path renders in Float32 or Float64, not $F")

struct RtStats
    seconds::Float64
    paths::UInt64
    rays::UInt64
    node_visits::UInt64
    prim_tests::UInt64
    node_wave_steps::UInt64
    prim_wave_steps::UInt64
    bounce_wave_steps::UInt64
    extend_seconds::Float64
    shade_seconds::Float64
    extend_launches::UInt64
    shade_launches::UInt64
end

rt_last_error() = unsafe_string(ccall((:rt_last_error, librt), Cstring, ()))

# a non-zero status becomes a Julia error carrying the library's message; the library itself never throws
check(status::Integer) = status == RT_OK ? nothing : error("librt_amd: status $status: $(rt_last_error())")

function rt_device_count()
    n = Ref{Cint}(0)
    check(ccall((:rt_device_count, librt), Cint, (Ref{Cint},), n))
    return Int(n[])
end

# cuscene_manager.jl -- replaces src/cuscene_manager.jl (CuSceneManager, src/cuscene_manager.jl:5-19) and the Adapt rules
# of src/transform.jl:12-15, src/constant_medium.jl:12-16, src/mesh.jl:17-22: CuSceneManager(bvh) no longer adapts the
# tuple of trees to CUDA device arrays; it flattens it into the tables of an RtSceneDesc (include/rt_amd.h) and hands
# them to rt_scene_create, which copies everything to the GPU.  The handle is released by close(::CuSceneManager).
include("cuBVH.jl")
include("scene_manager.jl") # the reference's own file, unchanged (src/scene_manager.jl)

mutable struct CuSceneManager
    handle::Ptr{Cvoid}          # rtScene* (device == -1: rtGroup*, several GPUs); rtScene64* / rtGroup64* for a Float64 scene
    device::Int
    F::DataType                 # element type of the scene: Float32 or Float64
end

# ---- the tables, filled by add!
mutable struct SceneTables{F}              # F: the element type of the scene; the record vectors are records(F)'s
    objects::Vector
    index::IdDict{Any,Int32}            # hittable -> 0-based position in `objects` (mutable and immutable alike)
    transforms::Vector
    materials::Vector
    material_index::Dict{Any,Int32}     # by value: Material is immutable (src/material.jl:11-16)
    textures::Vector
    texture_index::Dict{Any,Int32}
    images::Vector{RtImage}
    image_bytes::Vector{Vector{UInt8}}
    perlins::Vector
    perlin_index::IdDict{Any,Int32}
    meshes::Vector
    mesh_arrays::Vector{Any}            # the vertex / index / material arrays the RtMesh records point into
    groups::Vector{FlatBvh{F}}             # BVHTree objects used as hittables (Transform{BVHTree}, samples/cover_the_next_week.jl:75-82)
end
function SceneTables(::Type{F}) where F
    r = records(F)
    SceneTables{F}(r.object[], IdDict{Any,Int32}(), r.transform[], r.material[], Dict{Any,Int32}(), r.texture[],
                   Dict{Any,Int32}(), RtImage[], Vector{UInt8}[], r.perlin[], IdDict{Any,Int32}(), r.mesh[], Any[], FlatBvh{F}[])
end
ftype(::SceneTables{F}) where F = F
rec(t::SceneTables) = records(ftype(t))

pad12(::Type{F}, v...) where F = ntuple(i -> i <= length(v) ? F(v[i]) : zero(F), 12)
tuple3(::Type{F}, v) where F = (F(v.x), F(v.y), F(v.z))
zero3(::Type{F}) where F = (zero(F), zero(F), zero(F))

# ---- textures (src/texture.jl:8-37)
function add_texture!(t::SceneTables, tex::SolidColor)
    get!(t.texture_index, tex) do
        push!(t.textures, rec(t).texture(RT_TEX_SOLID, 0, 0, 0, tuple3(ftype(t), tex.color), zero(ftype(t))))
        Int32(length(t.textures) - 1)
    end
end
function add_texture!(t::SceneTables, tex::CheckerTexture)
    odd, even = add_texture!(t, tex.odd), add_texture!(t, tex.even)
    get!(t.texture_index, tex) do
        push!(t.textures, rec(t).texture(RT_TEX_CHECKER, odd, even, 0, zero3(ftype(t)), zero(ftype(t))))
        Int32(length(t.textures) - 1)
    end
end
function add_texture!(t::SceneTables, tex::ImageTexture)
    get!(t.texture_index, tex) do
        h, w = size(tex.image)
        bytes = Vector{UInt8}(undef, 3 * h * w)      # rows from the top, 3 bytes per texel: image[i, j] (src/texture.jl:55-76)
        for i in 1:h, j in 1:w
            px = tex.image[i, j]
            k = 3 * ((i - 1) * w + (j - 1))
            bytes[k+1] = round(UInt8, 255 * clamp(Float32(red(px)), 0, 1))
            bytes[k+2] = round(UInt8, 255 * clamp(Float32(green(px)), 0, 1))
            bytes[k+3] = round(UInt8, 255 * clamp(Float32(blue(px)), 0, 1))
        end
        push!(t.image_bytes, bytes)
        push!(t.images, RtImage(Int32(w), Int32(h), pointer(bytes)))
        push!(t.textures, rec(t).texture(RT_TEX_IMAGE, Int32(length(t.images) - 1), 0, 0, zero3(ftype(t)), zero(ftype(t))))
        Int32(length(t.textures) - 1)
    end
end
function add_perlin!(t::SceneTables, noise::Perlin3{T}) where T
    get!(t.perlin_index, noise) do
        length(noise.random) == 256 || error("librt_amd: a Perlin3 table must have 256 points, this one has $(length(noise.random))")
        F = ftype(t)
        ranvec = zeros(F, 768)              # 3 numbers per point: the vector, or the scalar in the first of them
        for (i, r) in enumerate(noise.random)
            if T <: Vec3
                ranvec[3i-2], ranvec[3i-1], ranvec[3i] = F(r.x), F(r.y), F(r.z)
            else
                ranvec[3i-2] = F(r)
            end
        end
        push!(t.perlins, rec(t).perlin(Tuple(ranvec), Tuple(Int32.(noise.perm_x)), Tuple(Int32.(noise.perm_y)), Tuple(Int32.(noise.perm_z))))
        Int32(length(t.perlins) - 1)
    end
end
function add_texture!(t::SceneTables, tex::NoiseTexture{T}) where T
    p = add_perlin!(t, tex.noise)
    kind = T <: Vec3 ? RT_TEX_NOISE_MARBLE : RT_TEX_NOISE_SCALAR    # src/texture.jl:79-81 / :98-100
    # (a NoiseTexture is mutable: every one gets its own record)
    push!(t.textures, rec(t).texture(kind, p, 0, 0, zero3(ftype(t)), ftype(t)(tex.scale)))
    Int32(length(t.textures) - 1)
end

# ---- materials (src/material.jl:11-16)
function add_material!(t::SceneTables, m::Material)
    tex = add_texture!(t, m.albedo)
    get!(t.material_index, (m, tex)) do
        push!(t.materials, rec(t).material(Int32(m.type), tex, ftype(t)(m.fuzz), ftype(t)(m.eta)))
        Int32(length(t.materials) - 1)
    end
end

# ---- hittables: one add! method per concrete type the path supports; each returns the 0-based object index
function push_object!(t::SceneTables, o, kind::Int32; material::Int32 = Int32(-1), child::Int32 = Int32(-1), aux::Int32 = Int32(-1),
                      p = pad12(ftype(t)))
    push!(t.objects, rec(t).object(kind, material, child, aux, p))
    t.index[o] = Int32(length(t.objects) - 1)
end
add!(t::SceneTables, o) = haskey(t.index, o) ? t.index[o] : add_new!(t, o)

add_new!(t::SceneTables, o::Sphere) =
    push_object!(t, o, RT_SPHERE; material = add_material!(t, o.material), p = pad12(ftype(t), o.center.x, o.center.y, o.center.z, o.radius))
add_new!(t::SceneTables, o::MovingSphere) =
    push_object!(t, o, RT_MOVING_SPHERE; material = add_material!(t, o.material),
                 p = pad12(ftype(t), o.center0.x, o.center0.y, o.center0.z, o.t0, o.center1.x, o.center1.y, o.center1.z, o.t1, o.radius))
add_new!(t::SceneTables, o::xy_rect) =
    push_object!(t, o, RT_XY_RECT; material = add_material!(t, o.material), p = pad12(ftype(t), o.x0, o.x1, o.y0, o.y1, o.k))
add_new!(t::SceneTables, o::xz_rect) =
    push_object!(t, o, RT_XZ_RECT; material = add_material!(t, o.material), p = pad12(ftype(t), o.x0, o.x1, o.z0, o.z1, o.k))
add_new!(t::SceneTables, o::yz_rect) =
    push_object!(t, o, RT_YZ_RECT; material = add_material!(t, o.material), p = pad12(ftype(t), o.y0, o.y1, o.z0, o.z1, o.k))
add_new!(t::SceneTables, o::Box) =                     # all six sides carry the one material (src/box.jl:9-24)
    push_object!(t, o, RT_BOX; material = add_material!(t, o.sides[1].material),
                 p = pad12(ftype(t), o.box_min.x, o.box_min.y, o.box_min.z, o.box_max.x, o.box_max.y, o.box_max.z))
function add_new!(t::SceneTables, o::Transform)        # src/transform.jl:3-8
    child = add!(t, o.object)
    b = o.bbox
    F = ftype(t)
    push!(t.transforms, rec(t).transform(F.(o.xform), F.(o.inv_xform), (F(b.min.x), F(b.min.y), F(b.min.z), F(b.max.x), F(b.max.y), F(b.max.z))))
    push_object!(t, o, RT_TRANSFORM; child = child, aux = Int32(length(t.transforms) - 1))
end
function add_new!(t::SceneTables, o::ConstantMedium)   # src/constant_medium.jl:3-8
    child = add!(t, o.boundary)
    push_object!(t, o, RT_CONSTANT_MEDIUM; material = add_material!(t, o.phase_function), child = child, p = pad12(ftype(t), o.neg_inv_density))
end
function add_new!(t::SceneTables, o::BVHTree)          # a tree used as an object (src/BVH.jl:12-16)
    for c in o.objects
        add!(t, c)
    end
    push!(t.groups, flat_bvh(o, c -> t.index[c]))
    push_object!(t, o, RT_BVH; aux = Int32(length(t.groups) - 1))
end
function add_new!(t::SceneTables, o::Mesh)             # src/mesh.jl:5-9
    nv, nt = length(o.vertices), size(o.indices, 2)
    F = ftype(t)
    vertices = Vector{F}(undef, 8nv)                   # position, normal, u, v (src/vertex.jl:3-8)
    for (i, v) in enumerate(o.vertices)
        vertices[8i-7:8i] .= (F(v.position.x), F(v.position.y), F(v.position.z), F(v.normal.x), F(v.normal.y), F(v.normal.z), F(v.u), F(v.v))
    end
    indices = Vector{Int32}(undef, 4nt)                # 4 x n_triangles column-major, 0-based, in the order build_bvh! left them
    for c in 1:nt, r in 1:4
        indices[4(c-1)+r] = Int32(o.indices[r, c] - 1)
    end
    materials = Int32[add_material!(t, m) for m in o.materials]
    push!(t.mesh_arrays, (vertices, indices, materials))
    push!(t.meshes, rec(t).mesh(Int32(nv), Int32(nt), Int32(length(materials)), Int32(0), pointer(vertices), pointer(indices), pointer(materials),
                           Ptr{Int32}(C_NULL)))
    push_object!(t, o, RT_MESH; aux = Int32(length(t.meshes) - 1))
end
add_new!(t::SceneTables, o) = error("librt_amd: hittable of type $(typeof(o)) is not supported on the GPU path")

"""
    flatten(bvh::Tuple) -> (desc::RtSceneDesc or RtSceneDesc64, keep)

`bvh` is what `build_bvh(world::SceneManager)` returns (src/scene_manager.jl:36-38): one BVHTree per type group, in the
order of the groups.  `keep` holds every array the description points into; it must stay alive (GC.@preserve) until
rt_scene_create has returned, which copies all of it.
"""
function flatten(bvh::Tuple)
    isempty(bvh) && error("librt_amd: a scene needs at least one tree with an object (draw! renders an empty world itself)")
    F = element_type(bvh[1])                        # BVHTree{F,T}: the F of render(F, ...) (main.jl:21)
    t = SceneTables(F)
    top = Int32[]                                   # the object vector of render(): the groups one after the other
    for tree in bvh, o in tree.objects
        push!(top, add!(t, o))
    end
    buckets = FlatBvh{F}[flat_bvh(tree, o -> t.index[o]) for tree in bvh]
    group_records = rec(t).bvh[rt_bvh(g) for g in t.groups]
    bucket_records = rec(t).bvh[rt_bvh(b) for b in buckets]
    desc = rec(t).desc(length(t.objects), pointer(t.objects), length(top), pointer(top), length(t.transforms), pointer(t.transforms),
                       length(t.materials), pointer(t.materials), length(t.textures), pointer(t.textures), length(t.images),
                       pointer(t.images), length(t.perlins), pointer(t.perlins), length(t.meshes), pointer(t.meshes),
                       length(group_records), pointer(group_records), length(bucket_records), pointer(bucket_records))
    return desc, (t, top, buckets, group_records, bucket_records)
end

"""
    CuSceneManager(bvh::Tuple; device = 0)
    CuSceneManager(bvh::Tuple; devices = [0, 1, ...])     # several GPUs of this host, `devices = Int[]`: all of them

Replaces `CuSceneManager(scene::SceneManager)` (src/cuscene_manager.jl:16-19).
"""
function CuSceneManager(bvh::Tuple; device::Integer = 0, devices::Union{Nothing,Vector{<:Integer}} = nothing)
    desc, keep = flatten(bvh)
    F = element_type(bvh[1])
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    list = devices === nothing ? Cint[] : Vector{Cint}(devices)     # rooted below: ccall gets a raw pointer into it
    GC.@preserve keep list begin
        members = isempty(list) ? Ptr{Cint}(C_NULL) : pointer(list)
        if F === Float64 && devices === nothing
            check(ccall((:rt_scene_create_f64, librt), Cint, (Ref{RtSceneDesc64}, Cint, Ref{Ptr{Cvoid}}), desc, device, handle))
        elseif F === Float64
            check(ccall((:rt_group_create_f64, librt), Cint, (Ref{RtSceneDesc64}, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}), desc,
                        members, length(list), handle))
        elseif devices === nothing
            check(ccall((:rt_scene_create, librt), Cint, (Ref{RtSceneDesc}, Cint, Ref{Ptr{Cvoid}}), desc, device, handle))
        else
            check(ccall((:rt_group_create, librt), Cint, (Ref{RtSceneDesc}, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}), desc,
                        members, length(list), handle))
        end
    end
    scene = CuSceneManager(handle[], devices === nothing ? Int(device) : -1, F)
    finalizer(close, scene)
    return scene
end

function Base.close(scene::CuSceneManager)
    if scene.handle != C_NULL
        if scene.F === Float64 && scene.device >= 0
            ccall((:rt_scene_destroy_f64, librt), Cint, (Ptr{Cvoid},), scene.handle)
        elseif scene.F === Float64
            ccall((:rt_group_destroy_f64, librt), Cint, (Ptr{Cvoid},), scene.handle)
        elseif scene.device >= 0
            ccall((:rt_scene_destroy, librt), Cint, (Ptr{Cvoid},), scene.handle)
        else
            ccall((:rt_group_destroy, librt), Cint, (Ptr{Cvoid},), scene.handle)
        end
        scene.handle = C_NULL
    end
    return nothing
end

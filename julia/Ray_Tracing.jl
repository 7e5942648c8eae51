# Ray_Tracing.jl (installed as src/Ray_Tracing_device.jl) -- the device half of src/Ray_Tracing.jl: it takes the place of
# cu_trace (src/Ray_Tracing.jl:51-69), draw_kernel (:102-139) and draw!(img::CuArray, ...) (:141-157).  The host half of
# that file (trace, draw!(img::Array, ...), :11-49, :71-100) stays where it is and includes this file after its own
# includes (cuBVH.jl, shapes.jl, scene_manager.jl, cuscene_manager.jl, camera.jl, :3-9); `using CUDA` goes away
# (Project.toml: CUDA and Adapt out, Libdl in).  julia/apply_overlay.py makes those edits to a checkout of the reference.
#
# The device path works in Float32 (the reference's default, main.jl:46,83,96) and in Float64 (render(Float64, ...),
# main.jl:21: the rt_*_f64 entry points).

"Image on the GPU side of render(): what `CUDA.zeros(RGB{F}, height, width)` was (main.jl:35).  The pixels live in host
memory the library writes into; the type only selects the method of draw!."
struct DeviceImage{F}
    pixels::Array{RGB{F},2}
end
DeviceImage(::Type{F}, height::Integer, width::Integer) where F = DeviceImage{F}(zeros(RGB{F}, height, width))
Base.Array(img::DeviceImage) = img.pixels
Base.size(img::DeviceImage) = size(img.pixels)

# A tree without nodes is never hit (src/BVH.jl:23-25, what build_bvh_scene makes of an empty group,
# src/scene_manager.jl:25-29): such trees are left out of what goes to the library, which wants every tree it is given
# to hold an object.
nonempty_trees(bvh::Tuple) = filter(tree -> !isempty(tree.nodes), bvh)

# The picture of a world in which nothing can be hit, with the arithmetic of the sample loop (src/Ray_Tracing.jl:81-92):
# every trace() returns the background (src/Ray_Tracing.jl:16-18).
function fill_background!(img::DeviceImage{F}, background::Vec3{F}, n::Integer) where F
    col = Vec3{F}(0, 0, 0)
    for _ in 1:n
        col += background
    end
    col /= n
    fill!(img.pixels, RGB{F}(sqrt(col.x), sqrt(col.y), sqrt(col.z)))
    return img
end

"""
    draw!(img::DeviceImage{Float32}, world::SceneManager, camera; background, pattern, seed, devices, max_depth, ignored...)

Replaces `draw!(img::CuArray{RGB{F},2}, world::SceneManager, camera; background, pattern)` (src/Ray_Tracing.jl:141-157):
builds the trees as before (`build_bvh(world)`, :144), hands them to the library (`CuSceneManager(bvh)`, :145) and
renders all pattern columns; the library divides by their number and takes the square root (:152-153).  The image
memory order of `Array{RGB{Float32},2}(height, width)` is the library's output order, so it writes into it directly.
`devices = [0, 1, ...]` spreads the pattern columns over several GPUs of this host (`Int[]`: all of them).
The result is reproducible for a given `seed` (the reference's GPU path draws from an unseeded generator).
`ignored...` takes every other keyword render() was called with: render(key::Symbol; ...) merges the caller's keywords
(`spp`, the scene function's own) into its call of render(F, world; ...) (main.jl:90-94), whose `kwargs...` (main.jl:28)
collects what it does not name, and the patched gpu branch hands that collection on so that `seed`, `devices` and
`max_depth` arrive here -- with `spp` and the rest in tow, which the reference's render() drops without a word as well.
"""
function draw!(img::DeviceImage{Float32}, world::SceneManager, camera::Camera{Float32};
               background::Vec3{Float32} = zero(Vec3{Float32}), pattern::Array{Float32,2} = Float32.(MSAA_S16),
               seed::Integer = 0x5EED, devices::Union{Nothing,Vector{<:Integer}} = nothing, max_depth::Integer = 50,
               ignored...)
    height, width = size(img)
    bvh = nonempty_trees(build_bvh(world))
    isempty(bvh) && return fill_background!(img, background, size(pattern, 2))
    scene = CuSceneManager(bvh; devices = devices)
    bg = Float32[background.x, background.y, background.z]
    cam = Ref(camera)                                   # Camera{Float32} has the layout of rtCamera
    pixels = img.pixels
    try
        GC.@preserve pixels pattern bg begin
            out = Ptr{Float32}(pointer(pixels))
            if devices === nothing
                check(ccall((:rt_render, librt), Cint,
                            (Ptr{Cvoid}, Ref{Camera{Float32}}, Ptr{Float32}, Cint, Ptr{Float32}, Cint, Cint, Cint, UInt64, UInt32, Ptr{Float32}, Ptr{Cvoid}),
                            scene.handle, cam, pattern, size(pattern, 2), bg, width, height, max_depth, UInt64(seed), UInt32(0), out, C_NULL))
            else
                check(ccall((:rt_group_render, librt), Cint,
                            (Ptr{Cvoid}, Ref{Camera{Float32}}, Ptr{Float32}, Cint, Ptr{Float32}, Cint, Cint, Cint, UInt64, UInt32, Ptr{Float32}, Ptr{Cvoid}),
                            scene.handle, cam, pattern, size(pattern, 2), bg, width, height, max_depth, UInt64(seed), UInt32(0), out, C_NULL))
            end
        end
    finally
        close(scene)
    end
    return img
end

"""
    draw!(img::DeviceImage{Float64}, world::SceneManager, camera; background, pattern, seed, devices, max_depth, ignored...)

The same for `render(Float64, ...; gpu = true)`: the trees of `build_bvh(world)` go to the library as they are, in
double, and rt_render_f64 (rt_group_render_f64 with `devices`) walks them as the CPU path does (src/BVH.jl:22-53).
"""
function draw!(img::DeviceImage{Float64}, world::SceneManager, camera::Camera{Float64};
               background::Vec3{Float64} = zero(Vec3{Float64}), pattern::Array{Float64,2} = Float64.(MSAA_S16),
               seed::Integer = 0x5EED, devices::Union{Nothing,Vector{<:Integer}} = nothing, max_depth::Integer = 50,
               ignored...)
    height, width = size(img)
    bvh = nonempty_trees(build_bvh(world))
    isempty(bvh) && return fill_background!(img, background, size(pattern, 2))
    scene = CuSceneManager(bvh; devices = devices)
    bg = Float64[background.x, background.y, background.z]
    cam = Ref(camera)                                   # Camera{Float64} has the layout of rtCamera64
    pixels = img.pixels
    try
        GC.@preserve pixels pattern bg begin
            out = Ptr{Float64}(pointer(pixels))
            if devices === nothing
                check(ccall((:rt_render_f64, librt), Cint,
                            (Ptr{Cvoid}, Ref{Camera{Float64}}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint, Cint, UInt64, UInt32, Ptr{Float64}, Ptr{Cvoid}),
                            scene.handle, cam, pattern, size(pattern, 2), bg, width, height, max_depth, UInt64(seed), UInt32(0), out, C_NULL))
            else
                check(ccall((:rt_group_render_f64, librt), Cint,
                            (Ptr{Cvoid}, Ref{Camera{Float64}}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint, Cint, UInt64, UInt32, Ptr{Float64}, Ptr{Cvoid}),
                            scene.handle, cam, pattern, size(pattern, 2), bg, width, height, max_depth, UInt64(seed), UInt32(0), out, C_NULL))
            end
        end
    finally
        close(scene)
    end
    return img
end

draw!(img::DeviceImage{F}, world::SceneManager, camera::Camera{F}; kwargs...) where F =
    error("librt_amd: the GPU path renders in Float32 or Float64, not $F")

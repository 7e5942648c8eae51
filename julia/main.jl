# main.jl -- what becomes of the `gpu` branch of the reference's render() (main.jl:34-38); the rest of that file -- the
# three render families, their keyword arguments, the OBJ path, scenes.jl -- stays as it is.  julia/apply_overlay.py
# makes this edit:
#
#     if gpu
#         d_img = CUDA.zeros(eltype(img), height, width)                                  # main.jl:35
#         draw!(d_img, world, camera; background = background, pattern = pattern)         # main.jl:37
#         img = Array(d_img)                                                              # main.jl:38
#
# becomes
#
#     if gpu
#         d_img = DeviceImage(F, height, width)
#         draw!(d_img, world, camera; background = background, pattern = pattern, kwargs...)
#         img = Array(d_img)
#
# `kwargs...` lets render(:cu_cover2; gpu = true, spp = 1000, seed = 7, devices = [0, 1, 2, 3]) reach the device path's
# own keywords (seed, devices, max_depth); everything else about the call is unchanged, so this is a drop-in.  The
# collection also carries what render(key::Symbol; ...) merged in and this method does not name -- `spp` of the
# README's own render(:cornell_box; spp=10000, gpu=true) -- so draw!(::DeviceImage, ...) ends its keyword list with a
# sink (`ignored...`) and drops them, as this method's unused `kwargs...` always did.
#
# Written out, the patched method (the reference's signature, main.jl:21-29):

function render(::Type{F}, world::Vector{<:Hittable};
                width = 1280,
                height = 720,
                camera = Camera(F),
                background = Vec3{F}(0.70, 0.80, 1.00),
                pattern = F.(MSAA_S16),
                gpu::Bool = false,
                kwargs...
               ) where F<:AbstractFloat
    world = SceneManager(world)
    img = zeros(RGB{F}, height, width)
    if gpu
        d_img = DeviceImage(F, height, width)
        draw!(d_img, world, camera; background = background, pattern = pattern, kwargs...)
        img = Array(d_img)
    else
        draw!(img, world, camera; background = background, pattern = pattern)
    end
    return img
end

"""raytracing_amd: the MI355X (gfx950) backend for the path-tracing loop of rafaellg79/raytracing.

Layout: csrc/ holds the HIP kernels and the C ABI (include/rt_amd.h -> librt_amd.so); the Python modules
mirror the reference's host-side interface for this path (scene vocabulary, SceneManager, build_bvh, render,
scenes_list) and drive the library through ctypes.
"""
from .hittables import *  # noqa: F401,F403
from .hittables import (AABB, Box, Camera, CheckerTexture, ConstantMedium, Dielectric, Emissive, Halton_sequence32,
                        ImageTexture, Isotropic, Lambertian, Material, Mesh, Metal, MovingSphere, NoiseTexture,
                        RotY, SolidColor, Sphere, Transform, Vec3, affine, bounding_box, xy_rect, xz_rect, yz_rect)
from .scene_manager import BVHTree, SceneManager, build_bvh, build_bvh_scene, type_key
from .render import DeviceGroup, DeviceScene, render, save, save_scene, clamp01, to_rgb8
from .scenes import get_scene, scenes, scenes_list

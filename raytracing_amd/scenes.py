"""The scene registry of the reference (scenes.jl, samples/*.jl) mirrored for the host side.
Deterministic scenes are reproduced value for value; where the reference calls the
unseeded rand() (cover1, cover2, cu_cover2) the draws come, in the same order, from a
small seeded generator so that every run and every rank builds the same world.
"""
import math
import os

import numpy as np

from . import hittables as H
from .hittables import (Box, Camera, ConstantMedium, Dielectric, Emissive, Halton_sequence32, ImageTexture,
                        Lambertian, Material, Metal, MovingSphere, NoiseTexture, RotY, Sphere, Transform, Vec3,
                        affine, hfov_from_vfov, xy_rect, xz_rect, yz_rect)

M32 = 0xFFFFFFFF


class SceneRng:
    """Seeded stand-in for the reference's global rand(): a counter fed through the pcg4d hash."""

    def __init__(self, seed=1):
        self.seed, self.ctr = seed & M32, 0

    def _u32(self):
        v = [self.ctr & M32, (self.ctr >> 32) & M32, 0x5CE7E5, self.seed]
        self.ctr += 1
        v = [(x * 1664525 + 1013904223) & M32 for x in v]
        for _ in range(2):
            v[0] = (v[0] + v[1] * v[3]) & M32
            v[1] = (v[1] + v[2] * v[0]) & M32
            v[2] = (v[2] + v[0] * v[1]) & M32
            v[3] = (v[3] + v[1] * v[2]) & M32
            if _ == 0:
                v = [x ^ (x >> 16) for x in v]
        return v[0]

    def rand32(self):  # rand(F): 24 bits for Float32, 53 for Float64
        if H.F is np.float64:
            return self.rand64()
        return np.float32(self._u32() >> 8) * np.float32(1.0 / 16777216.0)

    def rand64(self):  # rand()
        return ((self._u32() >> 5) * 67108864.0 + (self._u32() >> 6)) / 9007199254740992.0

    def rand_vec3(self):  # rand(Vec3{F}), src/vec3.jl:83
        return Vec3(self.rand32(), self.rand32(), self.rand32())

    def rand_vec3_range(self, lo, hi):  # rand(Vec3{F}, min, max), src/vec3.jl:85
        lo, hi = H.F(lo), H.F(hi)
        return Vec3(*[self.rand32() * (hi - lo) + lo for _ in range(3)])

    def shuffle(self, items):
        items = list(items)
        for i in range(len(items) - 1, 0, -1):
            j = self._u32() % (i + 1)
            items[i], items[j] = items[j], items[i]
        return items


ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def _cornell_camera(origin, target, aspect_ratio):
    origin, target = Vec3(origin), Vec3(target)
    fov = hfov_from_vfov(40, aspect_ratio)
    return Camera(origin, origin - target, Vec3(0, 1, 0), fov, aspect_ratio, 0.0, 10.0, 0.0, 1.0)


def cornell_box(spp=1000, **_):
    """samples/cornell_box.jl"""
    width = height = 400
    camera = _cornell_camera((278, 278, -800), (278, 278, 0), 1.0)
    red = Material(Lambertian, Vec3(0.65, 0.05, 0.05))
    white = Material(Lambertian, Vec3(0.73, 0.73, 0.73))
    green = Material(Lambertian, Vec3(0.12, 0.45, 0.15))
    light = Material(Emissive, Vec3(15, 15, 15))
    world = [
        yz_rect(0, 555, 0, 555, 555, green),
        yz_rect(0, 555, 0, 555, 0, red),
        xz_rect(213, 343, 227, 332, 554, light),
        xz_rect(0, 555, 0, 555, 0, white),
        xz_rect(0, 555, 0, 555, 555, white),
        xy_rect(0, 555, 0, 555, 555, white),
        Transform(Box(Vec3(0), Vec3(165, 330, 165), white), affine(RotY(math.radians(0)), [265, 0, 295])),
        Transform(Box(Vec3(0), Vec3(165, 165, 165), white), affine(RotY(math.radians(-18)), [130, 0, 65])),
    ]
    return width, height, camera, world, Vec3(0), Halton_sequence32(spp)


def cornell_smoke_box(spp=200, **_):
    """samples/cornell_smoke_box.jl"""
    width = height = 600
    camera = _cornell_camera((278, 278, -800), (278, 278, 0), 1.0)
    red = Material(Lambertian, Vec3(0.65, 0.05, 0.05))
    white = Material(Lambertian, Vec3(0.73, 0.73, 0.73))
    green = Material(Lambertian, Vec3(0.12, 0.45, 0.15))
    light = Material(Emissive, Vec3(7, 7, 7))
    box1 = Transform(Box(Vec3(0), Vec3(165, 330, 165), white), affine(RotY(math.radians(15)), [265, 0, 295]))
    box2 = Transform(Box(Vec3(0), Vec3(165, 165, 165), white), affine(RotY(math.radians(-18)), [130, 0, 65]))
    world = [
        yz_rect(0, 555, 0, 555, 555, green),
        yz_rect(0, 555, 0, 555, 0, red),
        xz_rect(113, 443, 127, 432, 554, light),
        xz_rect(0, 555, 0, 555, 0, white),
        xz_rect(0, 555, 0, 555, 555, white),
        xy_rect(0, 555, 0, 555, 555, white),
        ConstantMedium(box1, 0.01, Vec3(0.0)),
        ConstantMedium(box2, 0.01, Vec3(1.0)),
    ]
    return width, height, camera, world, Vec3(0), Halton_sequence32(spp)


def cover1(spp=500, seed=1, **_):
    """samples/cover_in_one_weekend.jl"""
    rng = SceneRng(seed)
    aspect_ratio = H.F(3 / 2)
    width = 1200
    height = int(width / aspect_ratio)
    origin, target = Vec3(13, 2, 3), Vec3(0, 0, 0)
    fov = hfov_from_vfov(20, aspect_ratio)
    camera = Camera(origin, origin - target, Vec3(0, 1, 0), fov, aspect_ratio, 0.1, 10.0)
    world = [Sphere(Vec3(0, -1000, -1), 1000, Material(Lambertian, Vec3(0.5, 0.5, 0.5)))]
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = rng.rand32()
            center = Vec3(a + 0.9 * rng.rand64(), 0.2, b + 0.9 * rng.rand64())
            if H.length(center - Vec3(4, 0.2, 0)) > H.F(0.9):
                if choose_mat < 0.8:
                    material = Material(Lambertian, rng.rand_vec3() * rng.rand_vec3())
                elif choose_mat < 0.95:
                    albedo = rng.rand_vec3() * H.F(0.5) + H.F(0.5)
                    material = Material(Metal, albedo, rng.rand32() * H.F(0.5))
                else:
                    material = Material(Dielectric, 1.5)
                world.append(Sphere(center, 0.2, material))
    world.append(Sphere(Vec3(0, 1, 0), 1, Material(Dielectric, 1.5)))
    world.append(Sphere(Vec3(-4, 1, 0), 1, Material(Lambertian, Vec3(0.4, 0.2, 0.1))))
    world.append(Sphere(Vec3(4, 1, 0), 1, Material(Metal, Vec3(0.7, 0.6, 0.5), 0.0)))
    return width, height, camera, world, Vec3(0.70, 0.80, 1.00), Halton_sequence32(spp)


def _next_week(spp, seed, full):
    """samples/cu_cover_the_next_week.jl (full=False) and samples/cover_the_next_week.jl (full=True)"""
    rng = SceneRng(seed)
    width = height = 800
    camera = _cornell_camera((478, 278, -600), (278, 278, 0), 1.0)
    ground = Material(Lambertian, Vec3(0.48, 0.83, 0.53))
    moving_sphere_material = Material(Lambertian, Vec3(0.7, 0.3, 0.1))
    white = Material(Lambertian, Vec3(0.73))
    light = Material(Emissive, Vec3(7, 7, 7))
    missing_texture = Material(Lambertian, Vec3(0, 1, 1))
    boxes_per_side = 20
    boxes = []
    for i in range(1, boxes_per_side + 1):
        for j in range(1, boxes_per_side + 1):
            w = 100
            x0, z0, y0 = -1000 + i * w, -1000 + j * w, 0
            x1, z1 = x0 + w, z0 + w
            y1 = rng.rand32() * H.F(79) + H.F(1)
            boxes.append(Box(Vec3(x0, y0, z0), Vec3(x1, y1, z1), ground))
    objects = []
    center1 = Vec3(400, 400, 200)
    center2 = center1 + Vec3(30, 0, 0)
    objects.append(MovingSphere(center1, center2, 0, 1, 50, moving_sphere_material))
    objects.append(Sphere(Vec3(260, 150, 45), 50, Material(Dielectric, 1.5)))
    objects.append(Sphere(Vec3(0, 150, 145), 50, Material(Metal, Vec3(0.8, 0.8, 0.9), 1.0)))
    boundary = Sphere(Vec3(360, 150, 145), 70, Material(Dielectric, 1.5))
    objects.append(boundary)
    objects.append(ConstantMedium(boundary, 0.2, Vec3(0.2, 0.4, 0.9)))
    boundary = Sphere(Vec3(0), 5000, Material(Dielectric, 1.5))
    objects.append(ConstantMedium(boundary, 0.0001, Vec3(1)))
    if full:
        earth = ImageTexture(os.path.join(ASSETS, "earthmap.rgb8"))
        objects.append(Sphere(Vec3(400, 200, 400), 100, Material(Lambertian, earth, 0.0)))
        objects.append(Sphere(Vec3(220, 280, 300), 80, Material(Lambertian, NoiseTexture(0.1, rng, vector=True), 0.0)))
    else:
        objects.append(Sphere(Vec3(400, 200, 400), 100, missing_texture))
        objects.append(Sphere(Vec3(220, 280, 300), 80, missing_texture))
    ns = 1000
    spheres = [Sphere(rng.rand_vec3_range(0, 165.0), 10, white) for _ in range(ns)]
    xf = affine(RotY(math.radians(15)), [-100, 270, 395])
    if full:
        from .scene_manager import build_bvh
        tail = [Transform(build_bvh(spheres), xf)]
    else:
        tail = [Transform(s, xf) for s in spheres]
    world = [xz_rect(123, 423, 147, 412, 554, light)] + objects + boxes + tail
    return width, height, camera, world, Vec3(0), Halton_sequence32(spp)


def cu_cover2(spp=1000, seed=1, **_):
    return _next_week(spp, seed, False)


def cover2(spp=10000, seed=1, **_):
    return _next_week(spp, seed, True)


def teapot(spp=1000, **_):
    """render("assets/teapot.obj", material) (main.jl:53-81) with the defaults of main.jl:21-29 and a camera
    chosen to match outputs/teapot_100_samples.png (the reference does not record the one it used)."""
    from .obj_loader import load_obj_mesh
    mesh = load_obj_mesh(os.path.join(ASSETS, "teapot.obj"), Material(Lambertian, Vec3(0.73, 0.73, 0.73)))
    origin, target = Vec3(10, 3, 0), Vec3(0, 1.5, 0)
    camera = Camera(origin, origin - target, Vec3(0, 1, 0), 45.0, 16 / 9, 0.0, None)
    return 1280, 720, camera, [mesh], Vec3(0.70, 0.80, 1.00), Halton_sequence32(spp)


scenes = {
    "cornell_box": cornell_box,
    "cornell_smoke_box": cornell_smoke_box,
    "cover1": cover1,
    "cover2": cover2,
    "cu_cover2": cu_cover2,
}
# not in the reference's registry (it is reached through render(filename, material)); kept addressable by name
extra_scenes = {"teapot": teapot}


def get_scene(key, F=None, **kwargs):
    """scenes.jl:8-11; F = the element type the scene function is given (samples/cornell_box.jl:3: F::Type=Float32)"""
    fn = scenes.get(key) or extra_scenes[key]
    with H.element_type(F if F is not None else H.F):
        width, height, camera, world, background, pattern = fn(**kwargs)
    return world, dict(width=width, height=height, camera=camera, background=background, pattern=pattern)


def scenes_list():
    """scenes.jl:12"""
    return scenes.keys()

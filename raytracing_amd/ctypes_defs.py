"""ctypes mirrors of the C structs in include/rt_types.h, rt_host.h and rt_hip.h.

Sizes are asserted at import so a header edit that is not repeated here fails
loudly instead of corrupting memory.
"""
import ctypes as C

import numpy as np

u32, i32, f32 = C.c_uint32, C.c_int32, C.c_float


class BvhNode(C.Structure):
    _fields_ = [("x_min", f32), ("x_max", f32), ("y_min", f32), ("y_max", f32),
                ("z_min", f32), ("z_max", f32), ("right", u32), ("first_obj", u32),
                ("obj_count", u32), ("split_axis", i32)]


class Mesh(C.Structure):
    _fields_ = [("triangle_offset", u32), ("triangle_count", u32), ("bvh_start", u32)]


class Instance(C.Structure):
    _fields_ = [("mesh", u32), ("transform", i32), ("material", i32), ("medium", i32),
                ("light", i32)]


class Vertex(C.Structure):
    _fields_ = [("position_texu", f32 * 4), ("normal_texv", f32 * 4)]


class Triangle(C.Structure):
    _fields_ = [("a", Vertex), ("b", Vertex), ("c", Vertex)]


class Mat4(C.Structure):
    _fields_ = [("m", (f32 * 4) * 4)]


class Material(C.Structure):
    _fields_ = [("albedo", f32 * 3), ("albedo_tex", i32), ("emission", f32 * 3),
                ("emission_tex", i32), ("metallic", f32), ("spec_trans", f32), ("ior", f32),
                ("subsurface", f32), ("roughness", f32), ("specular_tint", f32),
                ("anisotropic", f32), ("sheen", f32), ("sheen_tint", f32), ("clearcoat", f32),
                ("clearcoat_gloss", f32), ("normal_tex", i32), ("metallic_roughness_tex", i32),
                ("_p0", f32 * 3)]


class Medium(C.Structure):
    _fields_ = [("color", f32 * 3), ("density", f32), ("anisotropic", f32), ("type", i32),
                ("_p0", f32 * 2)]


class Light(C.Structure):
    _fields_ = [("intensity", f32 * 3), ("emission_tex", i32), ("direction", f32 * 3),
                ("type", i32), ("mesh", u32), ("transform", i32), ("_p0", f32 * 2)]


class Texture(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("format", u32), ("_pad", u32),
                ("data", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("pixel_delta_u", f32 * 3), ("pixel_delta_v", f32 * 3),
                ("upper_left_pixel", f32 * 3), ("position", f32 * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("tlas", C.POINTER(BvhNode)), ("blas", C.POINTER(BvhNode)),
                ("meshes", C.POINTER(Mesh)), ("instances", C.POINTER(Instance)),
                ("triangles", C.POINTER(Triangle)), ("transforms", C.POINTER(Mat4)),
                ("inverse_transforms", C.POINTER(Mat4)), ("materials", C.POINTER(Material)),
                ("mediums", C.POINTER(Medium)), ("lights", C.POINTER(Light)),
                ("textures", C.POINTER(Texture)),
                ("tlas_count", u32), ("blas_count", u32), ("mesh_count", u32),
                ("instance_count", u32), ("triangle_count", u32), ("transform_count", u32),
                ("material_count", u32), ("medium_count", u32), ("light_count", u32),
                ("texture_count", u32), ("sky_light", i32), ("_pad", u32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("closest_hits", C.c_uint64),
                ("nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("max_stack", C.c_uint64),
                ("shadow_skipped", C.c_uint64), ("seconds", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class HostInfo(C.Structure):
    _fields_ = [("resolution_x", u32), ("resolution_y", u32), ("max_depth", u32),
                ("tile_width", u32), ("tile_height", u32), ("tlas_depth", u32),
                ("blas_depth", u32), ("vertex_count", u32),
                ("cam_position", f32 * 3), ("cam_front", f32 * 3), ("cam_right", f32 * 3),
                ("cam_up", f32 * 3), ("cam_fov", f32)]


_EXPECTED_SIZES = {BvhNode: 40, Mesh: 12, Instance: 20, Vertex: 32, Triangle: 96, Mat4: 64,
                   Material: 96, Medium: 32, Light: 48, Texture: 24, Camera: 48,
                   SceneDesc: 11 * 8 + 12 * 4, Stats: 72}
for _cls, _size in _EXPECTED_SIZES.items():
    assert C.sizeof(_cls) == _size, (_cls.__name__, C.sizeof(_cls), _size)


def as_numpy(ptr, count, ctype):
    """View `count` structs behind a ctypes pointer as a numpy structured array (no copy)."""
    if count == 0:
        return np.zeros(0, dtype=np.dtype(ctype))
    buf = (ctype * count).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=np.dtype(ctype))

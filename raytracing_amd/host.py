"""Python access to the host layer (scene loader + BVH builder) through its C interface.

The library is the C++ restatement of the reference host code that feeds the compute pass
(include/rt_host.h lists the reference entry points each call stands for).  Nothing here
touches the GPU.
"""
import ctypes as C
import os

from . import ctypes_defs as T

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librt_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        l = C.CDLL(_LIB_PATH)
        l.rt_host_last_error.restype = C.c_char_p
        l.rt_host_load_scene.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        l.rt_host_load_snapshot.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        l.rt_host_load_snapshot_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        l.rt_host_save_snapshot.argtypes = [C.c_void_p, C.c_char_p]
        l.rt_host_scene_desc.argtypes = [C.c_void_p, C.POINTER(T.SceneDesc)]
        l.rt_host_scene_info.argtypes = [C.c_void_p, C.POINTER(T.HostInfo)]
        l.rt_host_camera.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(T.Camera)]
        l.rt_host_make_camera.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                          C.c_uint32, C.c_uint32, C.POINTER(T.Camera)]
        l.rt_host_seed_schedule.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64,
                                            C.POINTER(C.c_uint32)]
        l.rt_host_free_scene.argtypes = [C.c_void_p]
        l.rt_host_free_scene.restype = None
        _lib = l
    return _lib


def _check(code):
    if code != 0:
        raise RuntimeError(lib().rt_host_last_error().decode())


class HostScene:
    """A loaded scene with its BVH; owns the host memory `desc` points into."""

    def __init__(self, handle):
        self._handle = C.c_void_p(handle)
        self.desc = T.SceneDesc()
        _check(lib().rt_host_scene_desc(self._handle, C.byref(self.desc)))
        self.info = T.HostInfo()
        _check(lib().rt_host_scene_info(self._handle, C.byref(self.info)))

    @classmethod
    def from_json(cls, path):
        out = C.c_void_p()
        _check(lib().rt_host_load_scene(os.fsencode(path), C.byref(out)))
        return cls(out.value)

    @classmethod
    def from_snapshot(cls, path):
        out = C.c_void_p()
        _check(lib().rt_host_load_snapshot_file(os.fsencode(path), C.byref(out)))
        return cls(out.value)

    def save_snapshot(self, path):
        _check(lib().rt_host_save_snapshot(self._handle, os.fsencode(path)))

    def camera(self, width=None, height=None):
        cam = T.Camera()
        w = width or self.info.resolution_x
        h = height or self.info.resolution_y
        _check(lib().rt_host_camera(self._handle, w, h, C.byref(cam)))
        return cam

    # numpy views (no copies) of the device-visible arrays
    def array(self, name):
        d = self.desc
        table = {
            "tlas": (d.tlas, d.tlas_count, T.BvhNode),
            "blas": (d.blas, d.blas_count, T.BvhNode),
            "meshes": (d.meshes, d.mesh_count, T.Mesh),
            "instances": (d.instances, d.instance_count, T.Instance),
            "triangles": (d.triangles, d.triangle_count, T.Triangle),
            "transforms": (d.transforms, d.transform_count, T.Mat4),
            "inverse_transforms": (d.inverse_transforms, d.transform_count, T.Mat4),
            "materials": (d.materials, d.material_count, T.Material),
            "lights": (d.lights, d.light_count, T.Light),
        }
        ptr, count, ctype = table[name]
        return T.as_numpy(ptr, count, ctype)

    def close(self):
        if self._handle:
            lib().rt_host_free_scene(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_camera(lookfrom, lookat, fov, width, height):
    cam = T.Camera()
    a = (C.c_float * 3)(*lookfrom)
    b = (C.c_float * 3)(*lookat)
    _check(lib().rt_host_make_camera(a, b, fov, width, height, C.byref(cam)))
    return cam


DEFAULT_SEED = 0x5EED


def seed_schedule(first_pass, pass_count, tile_count, seed=DEFAULT_SEED):
    """random_seed words for passes [first_pass, first_pass+pass_count): shape (passes, tiles)."""
    import numpy as np
    out = np.empty(pass_count * tile_count, dtype=np.uint32)
    _check(lib().rt_host_seed_schedule(seed, first_pass * tile_count, out.size,
                                       out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out.reshape(pass_count, tile_count)

"""ctypes access to the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  The product package (raytracing_amd) never imports this module.
"""
import ctypes as C
import os

import numpy as np

from raytracing_amd import ctypes_defs as T

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "liboracle.so")
MISS = 0xFFFFFFFF
_lib = None

fp = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32),
                ("tile_width", C.c_uint32), ("tile_height", C.c_uint32),
                ("first_pass", C.c_uint32), ("pass_count", C.c_uint32), ("preview", C.c_uint32),
                ("seeds", u32p),
                ("rect_x0", C.c_uint32), ("rect_y0", C.c_uint32), ("rect_x1", C.c_uint32),
                ("rect_y1", C.c_uint32), ("threads", C.c_uint32),
                ("sky_importance", C.c_uint32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `make -C oracle`")
        l = C.CDLL(_LIB_PATH)
        l.oracle_render.argtypes = [C.POINTER(T.SceneDesc), C.POINTER(T.Camera),
                                    C.POINTER(Params), fp, C.POINTER(T.Stats)]
        l.oracle_primary_ids.argtypes = [C.POINTER(T.SceneDesc), C.POINTER(T.Camera),
                                         C.POINTER(Params), u32p, u32p, fp, fp]
        l.oracle_set_ulp_noise.argtypes = [C.c_int]
        l.oracle_set_ulp_noise.restype = None
        l.oracle_rand01_sequence.argtypes = [C.c_uint32, C.c_uint32, fp, u32p]
        l.oracle_rand01_sequence.restype = None
        l.oracle_hash_combine.argtypes = [C.c_uint32, C.c_uint32]
        l.oracle_hash_combine.restype = C.c_uint32
        l.oracle_rand_uint.argtypes = [u32p, C.c_uint32, C.c_uint32]
        l.oracle_rand_uint.restype = C.c_uint32
        l.oracle_hit_aabb.argtypes = [fp, fp, fp, C.c_float, C.c_float]
        l.oracle_hit_triangle.argtypes = [fp, fp, fp, C.c_float, C.c_float, fp]
        l.oracle_hit_triangle_quick.argtypes = [fp, fp, fp, C.c_float, C.c_float]
        l.oracle_eval_disney.argtypes = [C.POINTER(T.Material), C.c_int, fp, fp, fp, fp]
        l.oracle_eval_disney.restype = None
        l.oracle_sample_disney.argtypes = [C.POINTER(T.Material), C.c_int, fp, fp, u32p, fp]
        l.oracle_sample_disney.restype = None
        l.oracle_sample_texture.argtypes = [C.POINTER(T.Texture), C.c_float, C.c_float, fp]
        l.oracle_sample_texture.restype = None
        l.oracle_eval_sky_light.argtypes = [C.POINTER(T.SceneDesc), fp, fp]
        l.oracle_eval_sky_light.restype = None
        l.oracle_eval_sky_light_importance.argtypes = [C.POINTER(T.SceneDesc), fp, fp]
        l.oracle_eval_sky_light_importance.restype = None
        l.oracle_eval_sky_light_many.argtypes = [C.POINTER(T.SceneDesc), C.c_int, C.c_uint32, fp,
                                                 fp]
        l.oracle_eval_sky_light_many.restype = None
        l.oracle_sample_sky_importance.argtypes = [C.POINTER(T.SceneDesc), u32p, fp]
        l.oracle_sample_sky_importance.restype = None
        l.oracle_sample_sky_importance_many.argtypes = [C.POINTER(T.SceneDesc), C.c_uint32,
                                                        C.c_uint32, fp, fp]
        l.oracle_sample_sky_importance_many.restype = None
        _lib = l
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(fp)


def _params(width, height, max_depth, tile, seeds, first_pass=0, preview=False, rect=None,
            threads=0, sky_importance=False):
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    tiles = (width // tile[0]) * (height // tile[1])
    assert seeds.ndim == 2 and seeds.shape[1] == tiles, (seeds.shape, tiles)
    p = Params()
    p.width, p.height, p.max_depth = width, height, max_depth
    p.tile_width, p.tile_height = tile
    p.first_pass, p.pass_count = first_pass, seeds.shape[0]
    p.preview = 1 if preview else 0
    p.seeds = seeds.ctypes.data_as(u32p)
    if rect is not None:
        p.rect_x0, p.rect_y0, p.rect_x1, p.rect_y1 = rect
    p.threads = threads
    p.sky_importance = 1 if sky_importance else 0
    return p, seeds


def render(desc, camera, width, height, max_depth, tile, seeds, accum=None, first_pass=0,
           preview=False, rect=None, threads=0, sky_importance=False):
    """Adds seeds.shape[0] passes to `accum` (H, W, 4 float32; created when None).

    Returns (accum, stats_dict).
    """
    p, seeds = _params(width, height, max_depth, tile, seeds, first_pass, preview, rect, threads,
                       sky_importance)
    if accum is None:
        accum = np.zeros((height, width, 4), dtype=np.float32)
    assert accum.dtype == np.float32 and accum.shape == (height, width, 4)
    assert accum.flags["C_CONTIGUOUS"]
    stats = T.Stats()
    rc = lib().oracle_render(C.byref(desc), C.byref(camera), C.byref(p), _fp(accum),
                             C.byref(stats))
    if rc != 0:
        raise RuntimeError("oracle_render: invalid arguments")
    return accum, stats.as_dict()


def primary_ids(desc, camera, width, height, tile, seeds, threads=0):
    """closest_hit of each pixel's pass-0 primary ray: (instance, triangle, t, bary)."""
    p, seeds = _params(width, height, 0, tile, seeds, threads=threads)
    inst = np.empty((height, width), dtype=np.uint32)
    tri = np.empty((height, width), dtype=np.uint32)
    t = np.empty((height, width), dtype=np.float32)
    bary = np.empty((height, width, 3), dtype=np.float32)
    rc = lib().oracle_primary_ids(C.byref(desc), C.byref(camera), C.byref(p),
                                  inst.ctypes.data_as(u32p), tri.ctypes.data_as(u32p), _fp(t),
                                  _fp(bary))
    if rc != 0:
        raise RuntimeError("oracle_primary_ids: invalid arguments")
    return inst, tri, t, bary


class ulp_noise:
    """Context manager: renders inside it see transcendental results perturbed in the last bit
    (oracle.h: oracle_set_ulp_noise), for measuring how sensitive an image is to the math
    library."""

    def __enter__(self):
        lib().oracle_set_ulp_noise(1)

    def __exit__(self, *exc):
        lib().oracle_set_ulp_noise(0)


def rand01_sequence(seed, n):
    values = np.empty(n, dtype=np.float32)
    states = np.empty(n, dtype=np.uint32)
    lib().oracle_rand01_sequence(seed, n, _fp(values), states.ctypes.data_as(u32p))
    return values, states


def hash_combine(seed, v):
    return lib().oracle_hash_combine(seed, v)


def rand_uint(seed, lo, hi):
    s = C.c_uint32(seed)
    value = lib().oracle_rand_uint(C.byref(s), lo, hi)
    return value, s.value


def hit_aabb(box, origin, direction, t_min, t_max):
    box, origin, direction = _f(box), _f(origin), _f(direction)
    return bool(lib().oracle_hit_aabb(_fp(box), _fp(origin), _fp(direction), t_min, t_max))


def hit_triangle(tri, origin, direction, t_min, t_max):
    tri, origin, direction = _f(tri).reshape(9), _f(origin), _f(direction)
    out = np.zeros(4, dtype=np.float32)
    hit = lib().oracle_hit_triangle(_fp(tri), _fp(origin), _fp(direction), t_min, t_max, _fp(out))
    return bool(hit), out


def hit_triangle_quick(tri, origin, direction, t_min, t_max):
    tri, origin, direction = _f(tri).reshape(9), _f(origin), _f(direction)
    return bool(lib().oracle_hit_triangle_quick(_fp(tri), _fp(origin), _fp(direction), t_min,
                                                t_max))


def eval_disney(material, front_face, normal, v_in, l):
    normal, v_in, l = _f(normal), _f(v_in), _f(l)
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_eval_disney(C.byref(material), int(front_face), _fp(normal), _fp(v_in), _fp(l),
                             _fp(out))
    return out


def sample_disney(material, front_face, normal, v_in, seed):
    normal, v_in = _f(normal), _f(v_in)
    out = np.zeros(3, dtype=np.float32)
    s = C.c_uint32(seed)
    lib().oracle_sample_disney(C.byref(material), int(front_face), _fp(normal), _fp(v_in),
                               C.byref(s), _fp(out))
    return out, s.value


def sample_texture(texture, u, v):
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_sample_texture(C.byref(texture), u, v, _fp(out))
    return out


def eval_sky_light(desc, direction, importance=False):
    direction = _f(direction)
    out = np.zeros(4, dtype=np.float32)
    call = lib().oracle_eval_sky_light_importance if importance else lib().oracle_eval_sky_light
    call(C.byref(desc), _fp(direction), _fp(out))
    return out


def sample_sky_importance(desc, seed):
    out = np.zeros(3, dtype=np.float32)
    s = C.c_uint32(seed)
    lib().oracle_sample_sky_importance(C.byref(desc), C.byref(s), _fp(out))
    return out, s.value


def sample_sky_importance_many(desc, seed, n):
    directions = np.zeros((n, 3), dtype=np.float32)
    pdfs = np.zeros(n, dtype=np.float32)
    lib().oracle_sample_sky_importance_many(C.byref(desc), seed, n, _fp(directions), _fp(pdfs))
    return directions, pdfs


def eval_sky_light_many(desc, directions, importance=False):
    """eval_sky_light for an [n, 3] array of directions: [n, 4] (radiance rgb, pdf)."""
    directions = _f(directions).reshape(-1, 3)
    out = np.zeros((len(directions), 4), dtype=np.float32)
    lib().oracle_eval_sky_light_many(C.byref(desc), 1 if importance else 0, len(directions),
                                     _fp(directions), _fp(out))
    return out

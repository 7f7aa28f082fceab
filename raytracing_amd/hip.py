"""Python access to librt_hip.so (the gfx950 path tracer) through its C ABI.

There is no CPU fallback: if the library is missing, or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import ctypes_defs as T

# RT_HIP_LIB selects another build of the same library (kernel A/B experiments).
_LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "lib", "librt_hip.so")
_lib = None

ACCEL_REFERENCE = 0
ACCEL_REBUILT = 1
FLAG_COUNT_NODES = 1
FLAG_FUSED = 2
FLAG_TIME_KERNELS = 4
FLAG_BIN_RAYS = 8
FLAG_NO_BIN_RAYS = 16
FLAG_SKY_IMPORTANCE = 32
FLAG_TRACE_REFILL = 64
FLAG_TRACE_CHUNKS = 128
BINNING_UNDECIDED, BINNING_OFF, BINNING_ON = 0, 1, 2
REFILL_UNDECIDED, REFILL_OFF, REFILL_ON = 0, 1, 2
KERNEL_RENDER, KERNEL_EXTEND, KERNEL_SHADE = 0, 1, 2
MISS = 0xFFFFFFFF
NOALLOC_MAX_PATHS = 1 << 18
NOALLOC_MAX_DEPTH = 63

EXPORTS = [
    "rt_hip_create", "rt_hip_upload_scene", "rt_hip_upload_seeds", "rt_hip_clear",
    "rt_hip_render", "rt_hip_reserve", "rt_hip_tune", "rt_hip_shade_features", "rt_hip_render_noalloc", "rt_hip_render_tile", "rt_hip_render_preview", "rt_hip_primary_ids",
    "rt_hip_probe", "rt_hip_triangle_never_hit", "rt_hip_read_accum", "rt_hip_read_preview", "rt_hip_accum_ptr", "rt_hip_passes",
    "rt_hip_set_passes", "rt_hip_reduce",
    "rt_hip_stats", "rt_hip_kernel_time", "rt_hip_kernel_time_of", "rt_hip_kernel_busy_of", "rt_hip_trip_counts",
    "rt_hip_trace_work", "rt_hip_trace_refill", "rt_hip_ray_binning", "rt_hip_destroy", "rt_hip_last_error",
]


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32),
                ("tile_width", C.c_uint32), ("tile_height", C.c_uint32), ("device", C.c_int32),
                ("accel", C.c_uint32), ("flags", C.c_uint32)]


class TraceWork(C.Structure):
    _fields_ = [(name, C.c_uint64) for name in (
        "nodes", "tris", "nodes_from_lds", "instance_records", "instance_records_from_lds",
        "finish_nodes", "finish_tris", "node_bytes", "node_bytes_cached", "triangle_bytes",
        "instance_bytes", "node_wave_steps", "triangle_wave_steps", "instance_wave_steps")]


PROBE_RAND01, PROBE_HASH_COMBINE, PROBE_RAND_UINT, PROBE_HIT_AABB, PROBE_HIT_TRIANGLE, \
    PROBE_EVAL_DISNEY, PROBE_SAMPLE_DISNEY, PROBE_SAMPLE_TEXTURE, PROBE_SKY, \
    PROBE_SKY_SAMPLE, PROBE_TRIANGLE_DEGENERATE = range(11)


class ProbeIn(C.Structure):
    _fields_ = [("u", C.c_uint32 * 4), ("f", C.c_float * 16), ("material", T.Material)]


class ProbeOut(C.Structure):
    _fields_ = [("u", C.c_uint32 * 16), ("f", C.c_float * 16)]


fp = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class _Missing:
    """Stands for a symbol that an older build of the library (RT_HIP_LIB, A/B runs) lacks."""

    def __init__(self, name):
        self.name = name
        self.argtypes = self.restype = None

    def __call__(self, *args):
        raise RuntimeError(f"{_LIB_PATH} does not export {self.name}")


class _Bound:
    """The loaded library; with `tolerant`, a missing symbol fails when called, not when bound."""

    def __init__(self, dll, tolerant):
        self._dll, self._tolerant, self._missing = dll, tolerant, {}

    def __getattr__(self, name):
        try:
            return getattr(self._dll, name)
        except AttributeError:
            if not self._tolerant:
                raise
            return self._missing.setdefault(name, _Missing(name))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        try:
            # PyTorch ships its own libamdhip64 under the same SONAME as /opt/rocm's.  Loading
            # torch first makes the process use ONE HIP runtime, so torch streams and tensors
            # can be handed to this library.
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(_LIB_PATH)
        l = _Bound(l, tolerant=bool(os.environ.get("RT_HIP_LIB")))
        l.rt_hip_last_error.restype = C.c_char_p
        l.rt_hip_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        l.rt_hip_upload_scene.argtypes = [C.c_void_p, C.POINTER(T.SceneDesc)]
        l.rt_hip_upload_seeds.argtypes = [C.c_void_p, u32p, C.c_uint32, C.c_uint32]
        l.rt_hip_clear.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.rt_hip_render.argtypes = [C.c_void_p, C.POINTER(T.Camera), C.c_uint32, C.c_uint32,
                                    u32p, C.c_void_p, C.c_void_p]
        l.rt_hip_render_noalloc.argtypes = l.rt_hip_render.argtypes
        l.rt_hip_reserve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        l.rt_hip_shade_features.argtypes = [C.c_void_p, u32p, u32p]
        l.rt_hip_tune.argtypes = [C.c_void_p, C.POINTER(T.Camera), C.c_uint32, C.c_void_p]
        l.rt_hip_render_tile.argtypes = [C.c_void_p, C.POINTER(T.Camera), C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_void_p]
        l.rt_hip_render_preview.argtypes = [C.c_void_p, C.POINTER(T.Camera), C.c_uint32,
                                            C.c_void_p]
        l.rt_hip_primary_ids.argtypes = [C.c_void_p, C.POINTER(T.Camera), C.c_uint32, u32p, u32p,
                                         fp, fp]
        l.rt_hip_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(ProbeIn),
                                   C.POINTER(ProbeOut)]
        l.rt_hip_triangle_never_hit.argtypes = [fp, C.c_uint32, u32p]
        l.rt_hip_read_accum.argtypes = [C.c_void_p, fp, u32p, C.c_void_p]
        l.rt_hip_read_preview.argtypes = [C.c_void_p, fp, C.c_void_p]
        l.rt_hip_accum_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        l.rt_hip_passes.argtypes = [C.c_void_p, u32p]
        l.rt_hip_set_passes.argtypes = [C.c_void_p, C.c_uint32]
        l.rt_hip_reduce.argtypes = [C.POINTER(C.c_void_p), C.c_uint32]
        l.rt_hip_stats.argtypes = [C.c_void_p, C.POINTER(T.Stats), C.c_int]
        l.rt_hip_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), u32p]
        l.rt_hip_kernel_time_of.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), u32p]
        l.rt_hip_kernel_busy_of.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
        l.rt_hip_trip_counts.argtypes = [C.c_void_p, C.c_uint32, u32p]
        l.rt_hip_trace_work.argtypes = [C.c_void_p, C.POINTER(TraceWork)]
        l.rt_hip_ray_binning.argtypes = [C.c_void_p, u32p, C.POINTER(C.c_double)]
        l.rt_hip_trace_refill.argtypes = [C.c_void_p, u32p, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), u32p]
        l.rt_hip_destroy.argtypes = [C.c_void_p]
        _lib = l
    return _lib


def _check(code):
    if code != 0:
        raise RuntimeError(lib().rt_hip_last_error().decode())


class HipRenderer:
    """One renderer handle on one GPU.  Mirrors the C ABI one to one."""

    def __init__(self, width, height, max_depth, tile, device=-1, accel=ACCEL_REBUILT, flags=0):
        cfg = Config(width, height, max_depth, tile[0], tile[1], device, accel, flags)
        self.width, self.height, self.max_depth, self.tile = width, height, max_depth, tuple(tile)
        self.tile_count = (width // tile[0]) * (height // tile[1])
        self._h = C.c_void_p()
        _check(lib().rt_hip_create(C.byref(cfg), C.byref(self._h)))

    def upload_scene(self, desc):
        _check(lib().rt_hip_upload_scene(self._h, C.byref(desc)))

    def upload_seeds(self, seeds, first_pass=0):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        assert seeds.ndim == 2 and seeds.shape[1] == self.tile_count, seeds.shape
        _check(lib().rt_hip_upload_seeds(self._h, seeds.ctypes.data_as(u32p), first_pass,
                                         seeds.shape[0]))

    def clear(self, accum_dev=None, stream=None):
        _check(lib().rt_hip_clear(self._h, accum_dev, stream))

    def render(self, camera, first_pass, pass_count, rect=None, accum_dev=None, stream=None):
        r = (C.c_uint32 * 4)(*rect) if rect is not None else None
        _check(lib().rt_hip_render(self._h, C.byref(camera), first_pass, pass_count, r,
                                   accum_dev, stream))

    def reserve(self, width, height, pass_count):
        """Allocate the pipeline's path state for calls of that size now (rt_hip_reserve)."""
        _check(lib().rt_hip_reserve(self._h, width, height, pass_count))

    def shade_features(self):
        """(feature bits of the uploaded scene, feature set of the shading kernels chosen)."""
        scene, kernel = C.c_uint32(), C.c_uint32()
        _check(lib().rt_hip_shade_features(self._h, C.byref(scene), C.byref(kernel)))
        return scene.value, kernel.value

    def tune(self, camera, pass_count, stream=None):
        """Settle the per-scene trace policies now (rt_hip_tune): two untimed-by-the-caller
        calls of pass_count passes into a scratch image."""
        _check(lib().rt_hip_tune(self._h, C.byref(camera), pass_count, stream))

    def render_noalloc(self, camera, first_pass, pass_count, rect=None, accum_dev=None,
                       stream=None):
        """rt_hip_render without any device allocation: state in the library's fixed array."""
        r = (C.c_uint32 * 4)(*rect) if rect is not None else None
        _check(lib().rt_hip_render_noalloc(self._h, C.byref(camera), first_pass, pass_count, r,
                                           accum_dev, stream))

    def render_tile(self, camera, random_seed, tile_x, tile_y, stream=None):
        _check(lib().rt_hip_render_tile(self._h, C.byref(camera), random_seed, tile_x, tile_y,
                                        stream))

    def render_preview(self, preview_camera, random_seed, stream=None):
        _check(lib().rt_hip_render_preview(self._h, C.byref(preview_camera), random_seed, stream))

    def primary_ids(self, camera, pass_index=0):
        n = (self.height, self.width)
        inst = np.empty(n, dtype=np.uint32)
        tri = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float32)
        bary = np.empty(n + (3,), dtype=np.float32)
        _check(lib().rt_hip_primary_ids(self._h, C.byref(camera), pass_index,
                                        inst.ctypes.data_as(u32p), tri.ctypes.data_as(u32p),
                                        t.ctypes.data_as(fp), bary.ctypes.data_as(fp)))
        return inst, tri, t, bary

    def probe(self, kind, records):
        """Evaluates one device function per record (rt_hip_probe).  records: list of
        (u, f, material-or-None); returns (u [n, 16] uint32, f [n, 16] float32)."""
        n = len(records)
        arr = (ProbeIn * n)()
        for i, (u, f, material) in enumerate(records):
            for k, v in enumerate(u):
                arr[i].u[k] = int(v)
            for k, v in enumerate(f):
                arr[i].f[k] = float(v)
            if material is not None:
                arr[i].material = material
        out = (ProbeOut * n)()
        _check(lib().rt_hip_probe(self._h, kind, n, arr, out))
        flat = np.frombuffer(out, dtype=np.uint32).reshape(n, 32)
        return flat[:, :16].copy(), flat[:, 16:].copy().view(np.float32)

    def read_accum(self, stream=None):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        passes = C.c_uint32()
        _check(lib().rt_hip_read_accum(self._h, out.ctypes.data_as(fp), C.byref(passes), stream))
        return out, passes.value

    def read_preview(self, stream=None):
        out = np.empty((max(1, self.height // 10), max(1, self.width // 10), 4), dtype=np.float32)
        _check(lib().rt_hip_read_preview(self._h, out.ctypes.data_as(fp), stream))
        return out

    def accum_ptr(self):
        p = C.c_void_p()
        _check(lib().rt_hip_accum_ptr(self._h, C.byref(p)))
        return p.value

    def passes(self):
        n = C.c_uint32()
        _check(lib().rt_hip_passes(self._h, C.byref(n)))
        return n.value

    def set_passes(self, passes):
        _check(lib().rt_hip_set_passes(self._h, passes))

    def stats(self, reset=False):
        s = T.Stats()
        _check(lib().rt_hip_stats(self._h, C.byref(s), 1 if reset else 0))
        return s.as_dict()

    def kernel_time(self, kernel=KERNEL_RENDER):
        ms = C.c_double()
        n = C.c_uint32()
        _check(lib().rt_hip_kernel_time_of(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_busy(self, kernel):
        """Milliseconds during which at least one launch of that kind was running."""
        ms = C.c_double()
        _check(lib().rt_hip_kernel_busy_of(self._h, kernel, C.byref(ms)))
        return ms.value

    def ray_binning(self):
        """(BINNING_* state for the uploaded scene, measured ns per third-bounce ray or 0.0)."""
        state = C.c_uint32()
        ns = C.c_double()
        _check(lib().rt_hip_ray_binning(self._h, C.byref(state), C.byref(ns)))
        return state.value, ns.value

    def trace_work(self):
        """What the traversal fetched and how it was served (needs FLAG_COUNT_NODES)."""
        out = TraceWork()
        _check(lib().rt_hip_trace_work(self._h, C.byref(out)))
        return {name: getattr(out, name) for name, _ in out._fields_}

    def trace_refill(self):
        """(REFILL_* state, summed trace ms of the call timed without / with refilling, bit mask of
        the bounces whose trace launch refills)."""
        state, mask = C.c_uint32(), C.c_uint32()
        off, on = C.c_double(), C.c_double()
        _check(lib().rt_hip_trace_refill(self._h, C.byref(state), C.byref(off), C.byref(on),
                                         C.byref(mask)))
        return state.value, off.value, on.value, mask.value

    def trip_counts(self, trip):
        """(live paths, closest-hit rays, occlusion rays) of bounce `trip` in the last batch."""
        out = (C.c_uint32 * 3)()
        _check(lib().rt_hip_trip_counts(self._h, trip, out))
        return tuple(out)

    def close(self):
        if self._h:
            lib().rt_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def triangle_never_hit(corners):
    """rt_hip_triangle_never_hit for an [n, 9] float32 array of corners: [n] bool."""
    corners = np.ascontiguousarray(corners, dtype=np.float32).reshape(-1, 9)
    out = np.zeros(len(corners), dtype=np.uint32)
    _check(lib().rt_hip_triangle_never_hit(corners.ctypes.data_as(fp), len(corners),
                                           out.ctypes.data_as(u32p)))
    return out != 0


def reduce(renderers):
    """rt_hip_reduce: sums the accumulators of `renderers` onto renderers[0] (RCCL between
    GPUs, a kernel within one) and clears the others."""
    handles = (C.c_void_p * len(renderers))(*[r._h.value for r in renderers])
    _check(lib().rt_hip_reduce(handles, len(renderers)))

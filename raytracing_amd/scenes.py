"""Scene sources for tests and benchmarks.

* Snapshots of the reference's shipped scenes (tests/golden/scenes/*.rtsnap), made from the
  reference's asset files by tools/make_fixtures.py; they hold exactly what the scene loader
  returns, so the BVH build, the upload and the render run unchanged without the assets.
* Generated scenes written as JSON + OBJ (+ textures) in the reference's own scene format and
  then read back through the normal loader: the instanced-sphere scene of the 4K configuration
  and a small textured scene with an environment map.
"""
import json
import math
import os
import struct

import numpy as np

from . import host

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAPSHOT_DIR = os.path.join(_ROOT, "tests", "golden", "scenes")


def snapshot_path(name):
    return os.path.join(SNAPSHOT_DIR, name + ".rtsnap")


def load_snapshot(name):
    path = snapshot_path(name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"no snapshot for scene '{name}' at {path}")
    return host.HostScene.from_snapshot(path)


GENERATED = ("synthetic_4k",)


def load(name):
    """A scene by name: a snapshot of one of the reference's scenes, or a generated one
    (`synthetic_4k`: 10 000 instances of one 15 872-triangle sphere on a jittered 100 x 100
    grid, three area lights, constant sky; written in the reference's scene format with a fixed
    seed and read back through the normal loader)."""
    if name == "synthetic_4k":
        import shutil
        import tempfile
        directory = tempfile.mkdtemp(prefix="rt_synthetic_4k_")
        try:
            return host.HostScene.from_json(write_instanced_scene(directory))
        finally:
            shutil.rmtree(directory, ignore_errors=True)
    return load_snapshot(name)


def describe(name):
    if name in GENERATED:
        return ("synthetic: generated scene (fixed seed, reference scene format, loaded through "
                "the host loader), seeds and accumulator resident")
    return ("synthetic camera samples on a snapshot of the reference scene's loader output "
            "(tests/golden/scenes), resident in HBM")


# ---------------------------------------------------------------------------------------------
# generated geometry


def write_uv_sphere_obj(path, segments=32, rings=16, radius=1.0):
    """Quad/triangle UV sphere with per-vertex normals and no texture coordinates.

    segments x (rings - 2) quads plus two triangle fans; the loader fan-triangulates quads.
    """
    lines = ["o sphere"]
    verts = [(0.0, radius, 0.0)]
    for r in range(1, rings):
        theta = math.pi * r / rings
        for s in range(segments):
            phi = 2.0 * math.pi * s / segments
            verts.append((radius * math.sin(theta) * math.cos(phi), radius * math.cos(theta),
                          radius * math.sin(theta) * math.sin(phi)))
    verts.append((0.0, -radius, 0.0))
    for v in verts:
        lines.append("v %.6f %.6f %.6f" % v)
    for v in verts:
        n = np.array(v) / radius
        lines.append("vn %.4f %.4f %.4f" % tuple(n))

    def vid(r, s):  # r in 1..rings-1
        return 2 + (r - 1) * segments + (s % segments)

    last = len(verts)
    for s in range(segments):
        a, b = vid(1, s), vid(1, s + 1)
        lines.append(f"f 1//1 {b}//{b} {a}//{a}")
    for r in range(1, rings - 1):
        for s in range(segments):
            a, b, c, d = vid(r, s), vid(r, s + 1), vid(r + 1, s + 1), vid(r + 1, s)
            lines.append(f"f {a}//{a} {b}//{b} {c}//{c} {d}//{d}")
    for s in range(segments):
        a, b = vid(rings - 1, s), vid(rings - 1, s + 1)
        lines.append(f"f {last}//{last} {a}//{a} {b}//{b}")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def write_quad_obj(path, size=1.0, uv_scale=1.0):
    """Horizontal quad in the y = 0 plane, facing +y, with texture coordinates."""
    s = size
    text = f"""o quad
v {-s} 0 {s}
v {s} 0 {s}
v {-s} 0 {-s}
v {s} 0 {-s}
vt 0 0
vt {uv_scale} 0
vt {uv_scale} {uv_scale}
vt 0 {uv_scale}
vn 0 1 0
f 1/1/1 2/2/1 4/3/1 3/4/1
"""
    with open(path, "w") as f:
        f.write(text)


def write_light_quad_obj(path, half=0.5):
    """Small quad facing -y (an overhead area light), no texture coordinates."""
    text = f"""o light
v {-half} 0 {-half}
v {-half} 0 {half}
v {half} 0 {-half}
v {half} 0 {half}
vn 0 -1 0
f 1//1 2//1 4//1 3//1
"""
    with open(path, "w") as f:
        f.write(text)


def write_flat_hdr(path, rgb):
    """Radiance .hdr (uncompressed RGBE scanlines) from a float array [H, W, 3]."""
    rgb = np.asarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    m = rgb.max(axis=2)
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)
    nz = m > 1e-32
    mant, expo = np.frexp(m[nz])
    scale = mant * 256.0 / m[nz]
    rgbe[nz, :3] = np.clip(rgb[nz] * scale[:, None], 0, 255).astype(np.uint8)
    rgbe[nz, 3] = (expo + 128).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        f.write(rgbe.tobytes())


def write_png(path, rgba):
    from PIL import Image
    Image.fromarray(np.asarray(rgba, dtype=np.uint8)).save(path)


# ---------------------------------------------------------------------------------------------
# generated scenes

# The material set of assets/mustang_red.json (reference), reused by the instanced scene.
_MUSTANG_MATERIALS = {
    "black": {"albedo": [0.008, 0.008, 0.008], "roughness": 1.0},
    "chrome": {"albedo": [0.9, 0.9, 0.9], "roughness": 0.0, "metallic": 1.0},
    "floor": {"albedo": [0.01, 0.01, 0.01], "roughness": 1.0},
    "glass": {"albedo": [1.0, 1.0, 1.0], "roughness": 0.0, "spec_trans": 1.0, "ior": 1.45},
    "headlight": {"albedo": [1.0, 1.0, 1.0], "roughness": 0.1, "spec_trans": 1.0, "ior": 1.45},
    "paint": {"albedo": [0.82, 0.0, 0.0], "roughness": 0.0, "metallic": 0.2, "clearcoat": 1.0},
    "red_interior": {"albedo": [0.2, 0.02, 0.02], "roughness": 0.5, "ior": 1.1},
    "red_taillight": {"albedo": [0.581, 0.009, 0.009], "roughness": 0.0, "metallic": 0.738},
    "tyres": {"albedo": [0.008, 0.008, 0.008], "roughness": 0.5},
    "yellow_blinkers": {"albedo": [1.0, 0.28, 0.122], "roughness": 0.156, "spec_trans": 1.0,
                        "ior": 1.45},
}


def write_instanced_scene(directory, grid=100, sphere_segments=128, sphere_rings=63,
                          resolution=(3840, 2160), tile=(128, 72), max_depth=7, seed=12345):
    """grid x grid instances of ONE sphere mesh on a jittered grid, three area lights and a
    constant sky: the synthetic 4K configuration (10 000 instances of a 15 872-triangle BLAS at
    the defaults).  Returns the JSON path."""
    os.makedirs(os.path.join(directory, "meshes"), exist_ok=True)
    write_uv_sphere_obj(os.path.join(directory, "meshes", "sphere.obj"), sphere_segments,
                        sphere_rings)
    write_quad_obj(os.path.join(directory, "meshes", "floor.obj"), size=1.0)
    write_light_quad_obj(os.path.join(directory, "meshes", "light.obj"))
    rng = np.random.RandomState(seed)
    names = sorted(_MUSTANG_MATERIALS)
    extent = float(grid)
    primitives = {
        "floor": {"mesh": "meshes/floor.obj", "material": "floor",
                  "position": [0.0, -0.6, 0.0], "scale": [extent, 1.0, extent]},
    }
    index = 0
    for gz in range(grid):
        for gx in range(grid):
            jitter = rng.uniform(-0.25, 0.25, size=2)
            scale = float(rng.uniform(0.25, 0.45))
            angle = float(rng.uniform(0, 2 * math.pi))
            axis = rng.normal(size=3)
            axis /= np.linalg.norm(axis)
            quat = [math.cos(angle / 2)] + list(math.sin(angle / 2) * axis)
            primitives["s%05d" % index] = {
                "mesh": "meshes/sphere.obj",
                "material": names[index % len(names)],
                "position": [float(gx - grid / 2 + 0.5 + jitter[0]), 0.0,
                             float(gz - grid / 2 + 0.5 + jitter[1])],
                "rotation": [float(q) for q in quat],
                "scale": [scale, scale, scale],
            }
            index += 1
    lights = {}
    for i, (lx, lz) in enumerate([(-0.25, -0.25), (0.25, 0.0), (0.0, 0.3)]):
        lights["light%d" % i] = {
            "intensity": [40.0, 38.0, 34.0], "mesh": "meshes/light.obj", "two_sided": False,
            "position": [lx * extent, 6.0 + i, lz * extent], "scale": [8.0, 1.0, 8.0],
        }
    doc = {
        "renderer": {"resolution": list(resolution), "max_depth": max_depth, "tile": list(tile)},
        "camera": {"lookfrom": [0.0, 0.18 * extent + 4.0, -0.55 * extent],
                   "lookat": [0.0, 0.0, 0.0], "fov": 45.0},
        "sky_light": {"intensity": [0.6, 0.7, 0.9]},
        "material": _MUSTANG_MATERIALS,
        "primitive": primitives,
        "light": {"area": lights},
    }
    path = os.path.join(directory, "instanced.json")
    with open(path, "w") as f:
        json.dump(doc, f)
    return path


def write_textured_scene(directory, resolution=(160, 120), tile=(40, 30), max_depth=4,
                         with_area_light=True):
    """A checker-textured floor, a rough metal and a glass sphere under an HDR environment map
    (plus an optional textured area light): exercises albedo / normal / metallic-roughness /
    emission textures (the normal map on an untransformed and on a rotated, scaled surface) and
    the environment-map sky path.  Returns the JSON path."""
    os.makedirs(os.path.join(directory, "meshes"), exist_ok=True)
    os.makedirs(os.path.join(directory, "tex"), exist_ok=True)
    write_uv_sphere_obj(os.path.join(directory, "meshes", "sphere.obj"), 24, 12)
    write_quad_obj(os.path.join(directory, "meshes", "floor.obj"), size=6.0, uv_scale=3.7)
    write_light_quad_obj(os.path.join(directory, "meshes", "light.obj"), half=1.0)
    # light.obj has no vt records; the loader synthesises (0,0),(0,1),(1,1) per triangle.
    yy, xx = np.mgrid[0:64, 0:64]
    checker = (((xx // 8) + (yy // 8)) % 2).astype(np.float64)
    rgba = np.zeros((64, 64, 4), dtype=np.uint8)
    rgba[..., 0] = 40 + 200 * checker
    rgba[..., 1] = 60 + 150 * checker
    rgba[..., 2] = 200 - 120 * checker
    rgba[..., 3] = 255
    write_png(os.path.join(directory, "tex", "checker.png"), rgba)
    mr = np.zeros((32, 32, 4), dtype=np.uint8)
    mr[..., 1] = np.linspace(40, 230, 32)[None, :]   # green -> roughness
    mr[..., 2] = np.linspace(255, 120, 32)[:, None]  # blue -> metallic
    mr[..., 3] = 255
    write_png(os.path.join(directory, "tex", "mr.png"), mr)
    glow = np.zeros((16, 16, 4), dtype=np.uint8)
    glow[..., 0] = np.linspace(80, 255, 16)[None, :]
    glow[..., 1] = np.linspace(255, 90, 16)[:, None]
    glow[..., 2] = 160
    glow[..., 3] = 255
    write_png(os.path.join(directory, "tex", "glow.png"), glow)
    by, bx = np.mgrid[0:32, 0:32]
    bumps = np.zeros((32, 32, 4), dtype=np.uint8)   # a normal map: tilts of up to ~25 degrees
    bumps[..., 0] = np.round(128 + 55 * np.sin(bx * (2 * np.pi / 8)))
    bumps[..., 1] = np.round(128 + 55 * np.cos(by * (2 * np.pi / 16)))
    bumps[..., 2] = 235
    bumps[..., 3] = 255
    write_png(os.path.join(directory, "tex", "bumps.png"), bumps)
    h, w = 32, 64
    vv, uu = np.mgrid[0:h, 0:w]
    sky = np.zeros((h, w, 3))
    sky[..., 0] = 0.3 + 0.5 * (uu / w)
    sky[..., 1] = 0.4 + 0.3 * (1 - vv / h)
    sky[..., 2] = 0.9 * (1 - vv / h) + 0.1
    sun = np.exp(-(((uu - 20) / 3.0) ** 2 + ((vv - 8) / 2.0) ** 2))
    sky += 30.0 * sun[..., None]
    write_flat_hdr(os.path.join(directory, "tex", "sky.hdr"), sky)
    doc = {
        "renderer": {"resolution": list(resolution), "max_depth": max_depth, "tile": list(tile)},
        "camera": {"lookfrom": [0.0, 2.5, -7.0], "lookat": [0.0, 0.6, 0.0], "fov": 40.0},
        "sky_light": {"intensity": [1.0, 0.9, 0.8], "environment_tex": "tex/sky.hdr"},
        "material": {
            "floor": {"albedo": [1.0, 1.0, 1.0], "albedo_tex": "tex/checker.png",
                      "normal_tex": "tex/bumps.png", "roughness": 0.6},
            "metal": {"albedo": [0.9, 0.8, 0.6], "metallic": 1.0, "roughness": 0.3,
                      "anisotropic": 0.4, "metallic_roughness_tex": "tex/mr.png"},
            "glass": {"albedo": [1.0, 1.0, 1.0], "spec_trans": 1.0, "ior": 1.45,
                      "roughness": 0.05},
            "coat": {"albedo": [0.1, 0.5, 0.2], "roughness": 0.2, "clearcoat": 1.0,
                     "clearcoat_gloss": 0.7, "sheen": 0.5, "subsurface": 0.3,
                     "normal_tex": "tex/bumps.png"},
        },
        "primitive": {
            "floor": {"mesh": "meshes/floor.obj", "material": "floor"},
            "ball_a": {"mesh": "meshes/sphere.obj", "material": "metal",
                       "position": [-1.6, 1.0, 0.0]},
            "ball_b": {"mesh": "meshes/sphere.obj", "material": "glass",
                       "position": [1.6, 1.0, 0.5], "scale": [1.0, 1.2, 1.0]},
            "ball_c": {"mesh": "meshes/sphere.obj", "material": "coat",
                       "position": [0.0, 0.6, 1.8], "scale": [0.6, 0.6, 0.6],
                       "rotation": [0.9238795, 0.0, 0.3826834, 0.0]},
        },
    }
    if with_area_light:
        doc["light"] = {"area": {"panel": {
            "intensity": [6.0, 6.0, 6.0], "mesh": "meshes/light.obj", "two_sided": True,
            "emission_tex": "tex/glow.png", "position": [0.0, 4.0, 1.0]}}}
    path = os.path.join(directory, "textured.json")
    with open(path, "w") as f:
        json.dump(doc, f)
    return path

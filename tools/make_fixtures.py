"""Regenerates everything under tests/golden/ (run in the build container, needs the
reference checkout at /root/reference for the asset files and its screenshot).

  scenes/<name>.rtsnap        loader output for the reference's shipped scenes (mustang_red and
                              staircase reduced to the meshes that ship, teapot with a stand-in
                              for its missing meshes: see reduced_mustang, reduced_staircase,
                              teapot_standin)
  oracle_<name>.npz           small fixed-seed oracle renders + primary-hit id maps
  ref_hyperion_thumb.png      down-scaled copies of the reference's own screenshots
  ref_staircase_thumb.png     (outputs/hyperion_rect_light.png, outputs/staircase.png), the only
                              outputs of the reference itself that a render can be compared with
  workload_constants.json     oracle counters on the reference BVH at the bench resolution:
                              nodes and triangles per ray, rays per sample (roofline inputs)

The oracle images are produced by oracle/oracle.c itself: they pin the oracle (and through it
the HIP kernels) against regressions, they are NOT outputs of the reference.
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from raytracing_amd import host  # noqa: E402

REF = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden")

# name -> (width, height, tile, spp, max_depth) of the small golden render
SMALL = {
    "cornell_box": (96, 96, (24, 24), 4, 3),
    "hyperion_rect_light": (160, 90, (32, 18), 4, 5),
    "mustang_red_reduced": (160, 90, (32, 18), 4, 7),
    "staircase_reduced": (160, 200, (32, 40), 4, 7),
    "teapot_standin": (160, 90, (32, 18), 4, 2),
}


def link_assets(directory, *subdirs):
    for sub in subdirs:
        os.symlink(os.path.join(REF, "assets", sub), os.path.join(directory, sub))


def write_scene(directory, name, scene):
    path = os.path.join(directory, name + ".json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def shipped_primitives(scene):
    return {k: v for k, v in scene["primitive"].items()
            if os.path.exists(os.path.join(REF, "assets", v["mesh"]))}


def reduced_mustang(directory):
    """assets/mustang_red.json as far as it can be loaded: the five primitives whose meshes
    ship with the reference (floor plane, glass, headlights, blinkers, tail lights: spec_trans
    glass, rough transmission and metal materials under an environment map) and
    textures/computer_room.hdr in place of the missing HDR/leadenhall_market.hdr.  Camera,
    materials, renderer block and sky intensity are the file's own."""
    with open(os.path.join(REF, "assets", "mustang_red.json")) as f:
        scene = json.load(f)
    scene["primitive"] = shipped_primitives(scene)
    scene["sky_light"]["environment_tex"] = "textures/computer_room.hdr"
    link_assets(directory, "Mustang", "textures")
    return write_scene(directory, "mustang_red_reduced", scene)


def reduced_staircase(directory):
    """assets/staircase.json without the one primitive whose mesh does not ship (brass:
    staircase/Brass.obj): 19 primitives, 88 k triangles, ten 1024x1024 albedo textures, one
    rectangular area light.  Everything else is the file's own (640x800, max_depth 7)."""
    with open(os.path.join(REF, "assets", "staircase.json")) as f:
        scene = json.load(f)
    scene["primitive"] = shipped_primitives(scene)
    link_assets(directory, "staircase")
    return write_scene(directory, "staircase_reduced", scene)


def teapot_standin(directory):
    """assets/teapot.json as far as it can be run: the textured floor (teapot/Mesh050.obj with
    teapot/checker.png) is the file's own; the two teapot meshes and HDR/aft_lounge.hdr do not
    ship, so freeform_objs/unit_sphere.obj (15 872 triangles), scaled to the teapot's size at the
    camera's look-at point and given the teapot's material, stands in for the single mesh, and
    textures/computer_room.hdr for the environment map."""
    with open(os.path.join(REF, "assets", "teapot.json")) as f:
        scene = json.load(f)
    scene["primitive"] = shipped_primitives(scene)
    scene["primitive"]["teapot_standin"] = {
        "mesh": "freeform_objs/unit_sphere.obj", "material": "blue",
        "position": [-0.953633, 2.2, -0.0972613], "scale": [2.2, 2.2, 2.2]}
    scene["sky_light"]["environment_tex"] = "textures/computer_room.hdr"
    link_assets(directory, "teapot", "freeform_objs", "textures")
    return write_scene(directory, "teapot_standin", scene)


DERIVED = {"mustang_red_reduced": reduced_mustang, "staircase_reduced": reduced_staircase,
           "teapot_standin": teapot_standin}


def main():
    os.makedirs(os.path.join(GOLDEN, "scenes"), exist_ok=True)
    constants = {}
    for name, (w, h, tile, spp, md) in SMALL.items():
        if name in DERIVED:
            with tempfile.TemporaryDirectory() as scratch:
                sc = host.HostScene.from_json(DERIVED[name](scratch))
        else:
            sc = host.HostScene.from_json(os.path.join(REF, "assets", name + ".json"))
        sc.save_snapshot(os.path.join(GOLDEN, "scenes", name + ".rtsnap"))
        cam = sc.camera(w, h)
        tiles = (w // tile[0]) * (h // tile[1])
        seeds = host.seed_schedule(0, spp, tiles)
        accum, stats = oracle.render(sc.desc, cam, w, h, md, tile, seeds)
        inst, tri, t, bary = oracle.primary_ids(sc.desc, cam, w, h, tile, seeds[:1])
        np.savez_compressed(os.path.join(GOLDEN, f"oracle_{name}.npz"), accum=accum,
                            instance=inst, triangle=tri, t=t, bary=bary,
                            params=np.array([w, h, tile[0], tile[1], spp, md], dtype=np.uint32),
                            closest_rays=np.uint64(stats["closest_rays"]),
                            shadow_rays=np.uint64(stats["shadow_rays"]))
        print(name, "golden", accum.shape, stats)

        # Roofline inputs at the bench configuration: 1280x720, max_depth 7, 2 passes.
        bw, bh, btile, bmd = 1280, 720, (128, 72), 7
        bseeds = host.seed_schedule(0, 2, (bw // btile[0]) * (bh // btile[1]))
        _, bstats = oracle.render(sc.desc, sc.camera(bw, bh), bw, bh, bmd, btile, bseeds)
        rays = bstats["closest_rays"] + bstats["shadow_rays"]
        constants[name] = {
            "config": {"width": bw, "height": bh, "max_depth": bmd, "passes": 2},
            "closest_rays_per_sample": bstats["closest_rays"] / bstats["samples"],
            "shadow_rays_per_sample": bstats["shadow_rays"] / bstats["samples"],
            "nodes_per_ray": bstats["nodes_visited"] / rays,
            "tris_per_ray": bstats["tris_tested"] / rays,
            "max_stack": bstats["max_stack"],
            "tlas_depth": sc.info.tlas_depth, "blas_depth": sc.info.blas_depth,
        }
        print(name, "constants", constants[name])
    with open(os.path.join(GOLDEN, "workload_constants.json"), "w") as f:
        json.dump(constants, f, indent=2, sort_keys=True)

    from PIL import Image
    shot = Image.open(os.path.join(REF, "outputs", "hyperion_rect_light.png")).convert("RGB")
    shot.resize((160, 90), Image.BOX).save(os.path.join(GOLDEN, "ref_hyperion_thumb.png"))
    shot = Image.open(os.path.join(REF, "outputs", "staircase.png")).convert("RGB")
    shot.resize((160, 200), Image.BOX).save(os.path.join(GOLDEN, "ref_staircase_thumb.png"))


if __name__ == "__main__":
    main()

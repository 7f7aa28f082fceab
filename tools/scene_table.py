"""One line per snapshot scene (tests/golden/scenes): throughput of the HIP path at 1280x720,
max_depth 7, and the traversal work per ray on the rebuilt acceleration structure next to the
oracle's counts on the reference BVH (tests/golden/workload_constants.json).  Not a test and
not bench.py: the table in the design notes comes from here.

usage: python tools/scene_table.py [spp] [scene ...]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from raytracing_amd import hip, host, scenes  # noqa: E402

W, H, MD, TILE = 1280, 720, 7, (128, 72)


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    with open(os.path.join(scenes.SNAPSHOT_DIR, "..", "workload_constants.json")) as f:
        constants = json.load(f)
    names = sys.argv[2:] or sorted(constants)
    for name in names:
        sc = scenes.load_snapshot(name)
        cam = sc.camera(W, H)
        r = hip.HipRenderer(W, H, MD, TILE, accel=hip.ACCEL_REBUILT, flags=hip.FLAG_TIME_KERNELS)
        r.upload_scene(sc.desc)
        r.upload_seeds(host.seed_schedule(0, 3 * spp, r.tile_count))
        r.clear()
        r.render(cam, 0, spp)
        r.stats(reset=True)
        r.render(cam, spp, spp)
        r.render(cam, 2 * spp, spp)
        st = r.stats()
        ms, n = r.kernel_time()
        e_ms, e_n = r.kernel_time(hip.KERNEL_EXTEND)
        s_ms, s_n = r.kernel_time(hip.KERNEL_SHADE)
        r.close()
        rays = st["closest_rays"] + st["shadow_rays"]
        c = hip.HipRenderer(W, H, MD, TILE, accel=hip.ACCEL_REBUILT, flags=hip.FLAG_COUNT_NODES)
        c.upload_scene(sc.desc)
        c.upload_seeds(host.seed_schedule(0, 1, c.tile_count))
        c.clear()
        c.render(cam, 0, 1)
        cs = c.stats()
        c.close()
        crays = cs["closest_rays"] + cs["shadow_rays"]
        k = constants[name]
        print(f"{name}: {W * H * spp / ms / 1e3:.0f} Msamples/s, {rays / n / ms / 1e3:.0f} Mrays/s, "
              f"{rays / st['samples']:.2f} rays/sample, {ms:.2f} ms per {spp}-pass call "
              f"(trace {e_ms * e_n / n:.2f}, shade {s_ms * s_n / n:.2f}); per ray "
              f"{cs['nodes_visited'] / crays:.1f} node fetches, {cs['tris_tested'] / crays:.2f} "
              f"triangle tests, deepest stack {cs['max_stack']} "
              f"(reference BVH: {k['nodes_per_ray']:.1f} nodes, {k['tris_per_ray']:.2f} triangles, "
              f"stack {k['max_stack']})", flush=True)


if __name__ == "__main__":
    main()

"""Throughput of the HIP path on the generated configurations that have no snapshot in
tests/golden (not a test, not bench.py): the synthetic instanced scene (default: 10 000
instances of a 15 872-triangle sphere, as in the 4K configuration, rendered at 1280x720) and
the textured scene.

usage: python tools/bench_generated.py [instanced|textured] [grid] [width height spp]
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from raytracing_amd import hip, host, scenes  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "instanced"
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    W, H, spp = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (1280, 720, 16)
    tile = (W // 10, H // 10)
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        if kind == "instanced":
            path = scenes.write_instanced_scene(d, grid=grid, resolution=(W, H), tile=tile)
        else:
            path = scenes.write_textured_scene(d, resolution=(W, H), tile=tile, max_depth=7)
        sc = host.HostScene.from_json(path)
        print(f"{kind}: scene written and loaded in {time.time() - t0:.1f} s; "
              f"{sc.desc.instance_count} instances, {sc.desc.triangle_count} triangles, "
              f"{sc.desc.texture_count} textures", flush=True)
    cam = sc.camera(W, H)
    md = 7
    images = {}
    for name, flags in (("pipeline", hip.FLAG_TIME_KERNELS), ("fused", hip.FLAG_FUSED)):
        r = hip.HipRenderer(W, H, md, tile, accel=hip.ACCEL_REBUILT, flags=flags)
        t0 = time.time()
        r.upload_scene(sc.desc)
        upload = time.time() - t0
        r.upload_seeds(host.seed_schedule(0, 3 * spp, r.tile_count))
        r.clear()
        r.render(cam, 0, spp)          # warm-up (allocations, block order)
        r.stats(reset=True)
        r.clear()
        r.render(cam, spp, spp)
        r.render(cam, 2 * spp, spp)
        img, _ = r.read_accum()
        st = r.stats()
        ms, n = r.kernel_time()
        rays = st["closest_rays"] + st["shadow_rays"]
        line = (f"  {name}: upload {upload:.2f} s, {ms:.2f} ms per call, "
                f"{W * H * spp / ms / 1e3:.0f} Msamples/s, {rays / n / ms / 1e3:.0f} Mrays/s, "
                f"{rays / st['samples']:.2f} rays/sample")
        if name == "pipeline":
            e_ms, e_n = r.kernel_time(hip.KERNEL_EXTEND)
            s_ms, s_n = r.kernel_time(hip.KERNEL_SHADE)
            state, probe_ns = r.ray_binning()
            line += (f", ray binning {['undecided', 'off', 'on'][state]} "
                     f"(probe {probe_ns:.2f} ns per ray)")
            line += (f", trace {e_ms * e_n / n:.2f} ms, shade {s_ms * s_n / n:.2f} ms per call; "
                     f"rays per bounce {[sum(r.trip_counts(t)[1:]) for t in range(md + 1)]}")
        print(line, flush=True)
        images[name] = img
        r.close()
    print("  pipeline image bit-identical to fused:",
          np.array_equal(images["pipeline"], images["fused"], equal_nan=True))
    r = hip.HipRenderer(W, H, md, tile, accel=hip.ACCEL_REBUILT, flags=hip.FLAG_COUNT_NODES)
    r.upload_scene(sc.desc)
    r.upload_seeds(host.seed_schedule(0, 1, r.tile_count))
    r.clear()
    r.render(cam, 0, 1)
    st = r.stats()
    rays = st["closest_rays"] + st["shadow_rays"]
    print(f"  per ray: {st['nodes_visited'] / rays:.1f} node fetches, "
          f"{st['tris_tested'] / rays:.2f} triangle tests; deepest stack {st['max_stack']}")
    r.close()


if __name__ == "__main__":
    main()

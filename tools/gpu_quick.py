"""Quick GPU-vs-oracle comparison used while developing kernels (not a test).

usage: python tools/gpu_quick.py [scene] [width height spp max_depth]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import oracle  # noqa: E402
from raytracing_amd import hip, host  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests",
                      "golden", "scenes")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
    W, H, spp, md = (int(a) for a in sys.argv[2:6]) if len(sys.argv) > 5 else (320, 180, 4, 5)
    tile = (W // 5, H // 5)
    sc = host.HostScene.from_snapshot(os.path.join(GOLDEN, name + ".rtsnap"))
    cam = sc.camera(W, H)
    tiles = (W // tile[0]) * (H // tile[1])
    seeds = host.seed_schedule(0, spp, tiles)
    t0 = time.time()
    o_inst, o_tri, o_t, o_b = oracle.primary_ids(sc.desc, cam, W, H, tile, seeds[:1])
    o_acc, o_st = oracle.render(sc.desc, cam, W, H, md, tile, seeds)
    print(f"oracle: {time.time() - t0:.2f}s", o_st)
    fused_image = {}
    for accel, extra in ((hip.ACCEL_REFERENCE, hip.FLAG_FUSED), (hip.ACCEL_REBUILT, hip.FLAG_FUSED),
                         (hip.ACCEL_REFERENCE, 0), (hip.ACCEL_REBUILT, 0)):
        r = hip.HipRenderer(W, H, md, tile, accel=accel, flags=hip.FLAG_COUNT_NODES | extra)
        t0 = time.time()
        r.upload_scene(sc.desc)
        print(f"accel {accel} {'fused' if extra else 'wavefront'}: upload {time.time() - t0:.3f}s")
        r.upload_seeds(seeds)
        g_inst, g_tri, g_t, g_b = r.primary_ids(cam, 0)
        bad = (g_inst != o_inst) | (g_tri != o_tri)
        print(f"  id mismatches: {bad.sum()} of {bad.size};"
              f" t bit-equal: {(g_t.view(np.uint32) == o_t.view(np.uint32)).mean():.6f};"
              f" bary bit-equal: {(g_b.view(np.uint32) == o_b.view(np.uint32)).mean():.6f}")
        r.clear()
        r.render(cam, 0, spp)
        g_acc, passes = r.read_accum()
        st = r.stats()
        ms, n = r.kernel_time()
        diff = np.abs(g_acc[..., :3] - o_acc[..., :3])
        rel = diff / (np.abs(o_acc[..., :3]) + 1e-3)
        print(f"  passes {passes} kernel {ms:.3f} ms x{n}  Msamples/s {W * H * spp / ms / 1e3:.1f}")
        print("  stats", st)
        print(f"  image: max abs {diff.max():.4g} mean abs {diff.mean():.4g} "
              f"frac(rel>1e-3) {(rel > 1e-3).mean():.5f} frac(exact) "
              f"{(g_acc == o_acc).mean():.5f} nan {np.isnan(g_acc).sum()}")
        if extra:
            fused_image[accel] = g_acc
        else:
            same = np.array_equal(g_acc.view(np.uint32), fused_image[accel].view(np.uint32))
            print(f"  wavefront image bit-identical to fused: {same}")
        r.close()


if __name__ == "__main__":
    main()

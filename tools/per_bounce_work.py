"""Traversal work per bounce: renders the bench workload at max_depth 0, 1, 2 ... with the work
counters on and prints the differences (rays, wide-node fetches and triangle tests that each
further bounce adds).  usage: python tools/per_bounce_work.py [scene] [max_depth]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from raytracing_amd import hip, host, scenes  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "hyperion_rect_light"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w, h, spp = (3840, 2160, 8) if scene == "synthetic_4k" else (1280, 720, 64)
sc = scenes.load(scene)
cam = sc.camera(w, h)
tile = (w // 10, h // 10)
prev = (0, 0, 0)
for md in range(top + 1):
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT,
                        flags=hip.FLAG_COUNT_NODES | hip.FLAG_NO_BIN_RAYS | hip.FLAG_TRACE_REFILL)
    r.upload_scene(sc.desc)
    r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
    r.clear()
    r.render(cam, 0, spp)
    st, wk = r.stats(), r.trace_work()
    r.close()
    now = (st["closest_rays"] + st["shadow_rays"], wk["nodes"], wk["tris"])
    d = tuple(a - b for a, b in zip(now, prev))
    print(f"bounce {md}: rays {d[0] / 1e6:8.2f} M  node fetches per ray {d[1] / max(d[0], 1):6.2f}  "
          f"triangle tests per ray {d[2] / max(d[0], 1):6.2f}")
    prev = now

"""bench.py pieces that can be checked without a GPU: the default workload is the one the
committed PMC traffic measurement and the oracle's per-ray constants belong to, so the
roofline object of the default run is complete."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_default_workload_has_traffic_and_constants(monkeypatch):
    bench = load_bench(monkeypatch)
    args = bench.parse_args()
    assert (args.gpus, args.scene, args.width, args.height, args.max_depth) == \
        (1, "hyperion_rect_light", 1280, 720, 7)
    assert args.pipeline == "wavefront" and args.accel == "rebuilt"
    traffic = bench.measured_traffic("wf_extend_kernel", args)
    assert traffic is not None and 1e8 < traffic < 1e10
    with open(os.path.join(ROOT, "profiles", "r01", "traffic.json")) as f:
        on_file = json.load(f)
    assert any(t["scene"] == args.scene and t["spp_per_step"] == args.spp_per_step
               for t in on_file)
    # another workload has no measurement on file: traffic must be null, not a stale number
    args.spp_per_step += 1
    assert bench.measured_traffic("wf_extend_kernel", args) is None


def test_workload_constants_cover_the_snapshot_scenes(monkeypatch):
    bench = load_bench(monkeypatch)
    for scene in ("cornell_box", "hyperion_rect_light", "mustang_red_reduced",
                  "staircase_reduced", "teapot_standin"):
        k = bench.load_constants(scene)
        assert k["config"] == {"width": 1280, "height": 720, "max_depth": 7, "passes": 2}
        assert k["nodes_per_ray"] > 1.0 and k["tris_per_ray"] > 0.1
    assert bench.load_constants("no_such_scene") is None

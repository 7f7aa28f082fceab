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
    measured = bench.committed_counters(args)
    assert measured is not None
    traffic = measured["bytes_per_launch"]["wf_extend_kernel"]
    assert 1e8 < traffic < 1e10
    # another workload has no measurement on file: traffic must be null, not a stale number
    args.spp_per_step += 1
    assert bench.committed_counters(args) is None


def test_synthetic_4k_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--scene", "synthetic_4k"])
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args()
    assert (args.width, args.height, args.max_depth) == (3840, 2160, 7)
    # a step is two batches of the pipeline, each of at most 2^28 paths
    assert args.spp_per_step % 2 == 0
    assert args.width * args.height * (args.spp_per_step // 2) <= 1 << 28


def test_cpu_threads_respects_the_jobs_share(monkeypatch):
    bench = load_bench(monkeypatch)
    monkeypatch.setenv("OMP_NUM_THREADS", "1")
    threads, host_cores, why = bench.cpu_threads()
    assert threads == 1 and host_cores == os.cpu_count() and "OMP_NUM_THREADS" in why
    monkeypatch.delenv("OMP_NUM_THREADS")
    threads, host_cores, _ = bench.cpu_threads()
    assert 1 <= threads <= host_cores


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself(monkeypatch, tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: two child processes with the rank
    environment, started before the parent touches the GPU; the parent's exit code is the worst
    child's.  Here (no GPU) both children stop at the GPU check, which proves they got that far
    with the right environment."""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["RT_BENCH_ECHO_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                         text=True, timeout=300)
    ranks = sorted(line for line in out.stderr.splitlines() if line.startswith("bench rank "))
    assert ranks == ["bench rank 0 of 2 (local 0)", "bench rank 1 of 2 (local 1)"], out.stderr
    assert out.returncode != 0  # no GPU here: the children refuse to run on the CPU


def test_workload_constants_cover_the_snapshot_scenes(monkeypatch):
    bench = load_bench(monkeypatch)
    for scene in ("cornell_box", "hyperion_rect_light", "mustang_red_reduced",
                  "staircase_reduced", "teapot_standin"):
        k = bench.load_reference_tree_constants(scene)
        assert k["config"] == {"width": 1280, "height": 720, "max_depth": 7, "passes": 2}
        assert k["nodes_per_ray"] > 1.0 and k["tris_per_ray"] > 0.1
    assert bench.load_reference_tree_constants("no_such_scene") is None

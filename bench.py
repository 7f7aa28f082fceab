#!/usr/bin/env python3
"""Throughput benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one launch of the path-tracing kernel over the whole 1280x720 frame at
max_depth 7, tracing --spp-per-step passes of every pixel (synthetic = scene snapshot already
resident in HBM, seeds resident, accumulator resident).  The reported value is
pixel-samples per second over all GPUs, in Msamples/s; Mrays/s comes from the device ray
counters of the same timed region.

N > 1: one process per GPU (torchrun); each rank traces its own pass range of the same frame
(global pass index in the seed schedule, scene replicated) and the accumulators are summed
onto rank 0 with one RCCL reduce inside the timed region.  Work per GPU is fixed: weak
scaling.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracing_amd import hip, host, scenes  # noqa: E402
from raytracing_amd.distributed import new_accumulator, reduce_accumulator  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec)

# Algorithmic bytes (roofline numerator), per unit of work:
NODE_BYTES = 32       # 6-float box + 2 topology words: the minimum for a binary BVH node
TRI_BYTES = 36        # 9 position floats
HIT_BYTES = 300       # shading fetch per closest hit: full triangle 96 + material 96 +
#                       instance 20 + transform 64 + descriptors/light 24
FRAME_BYTES = 32      # accumulator read-modify-write per pixel per launch


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="hyperion_rect_light")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--max-depth", type=int, default=7)
    ap.add_argument("--spp-per-step", type=int, default=64,
                    help="passes per render call = one batch of the pipeline (59 M paths, 8.3 GB of "
                         "path state at 1280x720); 16 per call is 14 %% slower per sample")
    ap.add_argument("--accel", choices=["rebuilt", "reference"], default="rebuilt")
    ap.add_argument("--pipeline", choices=["wavefront", "fused"], default="wavefront",
                    help="trace/shade kernels per bounce (default) or the single fused kernel")
    ap.add_argument("--binning", choices=["auto", "on", "off"], default="auto",
                    help="regroup the ray queues before each trace launch: decided per scene by "
                         "the library during warm-up (auto), or forced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target duration of the CPU baseline sample")
    return ap.parse_args()


def load_constants(scene_name):
    path = os.path.join(ROOT, "tests", "golden", "workload_constants.json")
    with open(path) as f:
        return json.load(f).get(scene_name)


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/pmc_passes.sh:
    FETCH_SIZE doubled as the gfx950 counter requires, plus WRITE_SIZE), or None when no
    measurement of this exact workload is on file.  bench.py cannot collect counters itself."""
    path = os.path.join(ROOT, "profiles", "r01", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        measured = json.load(f)
    for t in measured if isinstance(measured, list) else [measured]:
        if (t.get("scene") == args.scene and t.get("width") == args.width
                and t.get("height") == args.height and t.get("max_depth") == args.max_depth
                and t.get("spp_per_step") == args.spp_per_step
                and t.get("pipeline") == args.pipeline and t.get("accel") == args.accel):
            return t.get("bytes_per_launch", {}).get(kernel)
    return None


def cpu_baseline(sc, cam, args, tile, target_seconds):
    """Times the CPU restatement (the oracle) on a bounded sample of the same workload."""
    import oracle
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    tiles = (args.width // tile[0]) * (args.height // tile[1])
    seeds = host.seed_schedule(0, 1, tiles)
    t0 = time.perf_counter()
    oracle.render(sc.desc, cam, args.width, args.height, args.max_depth, tile, seeds,
                  threads=threads)
    one = time.perf_counter() - t0
    passes = max(1, min(64, int(target_seconds / max(one, 1e-3))))
    seeds = host.seed_schedule(0, passes, tiles)
    t0 = time.perf_counter()
    _, st = oracle.render(sc.desc, cam, args.width, args.height, args.max_depth, tile, seeds,
                          threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": args.width * args.height * passes / dt / 1e6,
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{passes} passes of the full {args.width}x{args.height} frame, same scene, "
                  f"seed schedule and max_depth; {dt:.1f} s; OpenMP over image rows",
        "mrays_per_s": (st["closest_rays"] + st["shadow_rays"]) / dt / 1e6,
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torchrun")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    tile = (args.width // 10, args.height // 10)  # 128x72 at 1280x720, as mustang_red.json
    sc = scenes.load_snapshot(args.scene)
    cam = sc.camera(args.width, args.height)
    accel = hip.ACCEL_REBUILT if args.accel == "rebuilt" else hip.ACCEL_REFERENCE
    flags = hip.FLAG_FUSED if args.pipeline == "fused" else hip.FLAG_TIME_KERNELS
    flags |= {"auto": 0, "on": hip.FLAG_BIN_RAYS, "off": hip.FLAG_NO_BIN_RAYS}[args.binning]
    r = hip.HipRenderer(args.width, args.height, args.max_depth, tile, device=local_rank,
                        accel=accel, flags=flags)
    r.upload_scene(sc.desc)

    spp = args.spp_per_step
    total_steps = args.warmup + args.steps
    passes_per_rank = total_steps * spp
    first_pass = rank * passes_per_rank  # global pass index: ranks trace disjoint passes
    seeds = host.seed_schedule(first_pass, passes_per_rank, r.tile_count)
    r.upload_seeds(seeds, first_pass)

    accum = new_accumulator(args.height, args.width, device)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        r.render(cam, first_pass + i * spp, spp, accum_dev=accum.data_ptr(), stream=stream)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    r.stats(reset=True)
    accum.zero_()
    accum[..., 3] = 1.0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    reduce_accumulator(accum, dst=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    st = r.stats()
    kernel_ms, launches = r.kernel_time()
    times = torch.tensor([elapsed], dtype=torch.float64, device=device)
    counts = torch.tensor([st["samples"], st["closest_rays"], st["shadow_rays"],
                           st["closest_hits"]], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elapsed = float(times.item())
    samples, closest, shadow, hits = (float(v) for v in counts.tolist())

    if rank == 0:
        nan_pixels = int(torch.isnan(accum).any(dim=2).sum().item())
        k = load_constants(args.scene)
        roofline = None
        if k is not None and launches > 0:
            # rays counted on the device x the reference-BVH per-ray node/triangle counts
            # measured by the oracle at this configuration
            trace_bytes = ((st["closest_rays"] + st["shadow_rays"])
                           * (k["nodes_per_ray"] * NODE_BYTES + k["tris_per_ray"] * TRI_BYTES))
            shade_bytes = st["closest_hits"] * HIT_BYTES
            frame_bytes = args.width * args.height * FRAME_BYTES * launches
            if args.pipeline == "fused":
                name, ms, n = "path_trace_kernel", kernel_ms, launches
                bytes_l = (trace_bytes + shade_bytes + frame_bytes) / n
                extra = {}
            else:
                # the dominant kernel of the pipeline is the trace launch (one per bounce);
                # late bounces are handed to a finishing kernel, whose rays do not count here
                ms, n = r.kernel_time(hip.KERNEL_EXTEND)
                shade_ms, shade_n = r.kernel_time(hip.KERNEL_SHADE)
                name = "wf_extend_kernel"
                per_call = n // max(launches, 1)
                per_bounce = [sum(r.trip_counts(t)[1:]) for t in range(args.max_depth + 1)]
                traced = sum(per_bounce[:per_call]) * launches  # every call traces the same counts +- noise
                per_ray = k["nodes_per_ray"] * NODE_BYTES + k["tris_per_ray"] * TRI_BYTES
                bytes_l = traced * per_ray / max(n, 1)
                binning_state, probe_ns = r.ray_binning()
                extra = {"ray_binning": ["undecided", "off", "on"][binning_state],
                         "ray_binning_probe_ns_per_ray": probe_ns,
                         "rays_per_bounce_last_call": per_bounce,
                         "trace_launches_per_call": per_call,
                         "render_call_avg_ms": kernel_ms, "render_calls": launches,
                         "shade_kernel_avg_ms": shade_ms, "shade_launches": shade_n,
                         "extend_ms_per_call": ms * n / launches,
                         "shade_ms_per_call": shade_ms * shade_n / launches,
                         "whole_call_algorithmic_gbs":
                             (trace_bytes + shade_bytes + frame_bytes) / launches
                             / (kernel_ms * 1e-3) / 1e9}
            achieved = bytes_l / (ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": measured_traffic(name, args),
                        "kernel": name, "kernel_avg_ms": ms, "launches": n,
                        "algorithmic_bytes_per_launch": bytes_l, **extra}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sc, cam, args, tile, args.cpu_seconds)
        line = {
            "metric": "Msamples/s at 1280x720 max_depth 7",
            "value": samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.scene} (reference scene, snapshot of its loader output), "
                            f"{args.width}x{args.height}, max_depth {args.max_depth}, "
                            f"{spp} spp per step, accel={args.accel}, pipeline={args.pipeline}",
                "samples_per_step_per_gpu": args.width * args.height * spp,
                "parallelism": f"sample-range split x{world}, scene replicated, "
                               "one RCCL sum-reduce" if world > 1 else "single GPU",
            },
            "mrays_per_s": (closest + shadow) / elapsed / 1e6,
            "rays_per_sample": (closest + shadow) / max(samples, 1.0),
            "nan_pixels": nan_pixels,  # the shader itself yields NaN for edge-on light samples
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Throughput benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scene NAME]

One step = one rt_hip_render call over the whole frame at max_depth 7, tracing --spp-per-step
passes of every pixel (scene, seeds and accumulator already resident in HBM).  The reported
value is pixel-samples per second over all GPUs, in Msamples/s; Mrays/s comes from the device
ray counters of the same timed region.

Scenes: snapshots of the reference's scenes (tests/golden/scenes; default
hyperion_rect_light at 1280x720) and `synthetic_4k`, the generated 10 000-instance scene at
3840x2160 (raytracing_amd/scenes.py).

N > 1: one process per GPU.  Under torchrun (WORLD_SIZE set) this process is one rank; without
it, `--gpus N` starts the N ranks itself as child processes before touching the GPU.  Each
rank traces its own pass range of the same frame (global pass index in the seed schedule,
scene replicated) and the accumulators are summed onto rank 0 with one RCCL reduce inside the
timed region.  Work per GPU is fixed: weak scaling.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (spec)
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth of the 8 XCDs (measured figure of the microarchitecture guide)
L2_BYTES = 32 << 20     # 8 x 4 MiB

# Algorithmic bytes of the trace kernel, per unit of ITS OWN work (the rebuilt 4-wide tree):
NODE_BYTES = 112        # one wide node = boxes of four children (96 B) + four child words
TRI_BYTES = 36          # nine position floats per triangle test
INSTANCE_BYTES = 64     # one instance record per TLAS leaf entered (reported, not in `achieved`)
# per-ray state the trace launch streams: pid 4 + ray 32 in, hit record 24 out (closest) /
# pid 4 + ray 32 in, flag 4 out (occlusion) -- reported, not in `achieved`
CLOSEST_STATE_BYTES = 60
SHADOW_STATE_BYTES = 40
# the previous definition, on the REFERENCE's binary tree (kept under its own key):
REF_NODE_BYTES = 32
REF_TRI_BYTES = 36

DEFAULTS = {  # scene -> (width, height, spp per step)
    "synthetic_4k": (3840, 2160, 8),   # 8 passes = 66.4 M paths, one batch of the pipeline
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="hyperion_rect_light")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=7)
    ap.add_argument("--spp-per-step", type=int, default=None,
                    help="passes per render call = one batch of the pipeline (default 64: 59 M "
                         "paths, 8.3 GB of path state at 1280x720; 16 per call is 14 %% slower "
                         "per sample)")
    ap.add_argument("--accel", choices=["rebuilt", "reference"], default="rebuilt")
    ap.add_argument("--pipeline", choices=["wavefront", "fused"], default="wavefront",
                    help="trace/shade kernels per bounce (default) or the single fused kernel")
    ap.add_argument("--binning", choices=["auto", "on", "off"], default="auto",
                    help="regroup the ray queues before each trace launch: decided per scene by "
                         "the library from one untimed call before warm-up (auto), or forced")
    ap.add_argument("--refill", choices=["auto", "on", "off"], default="auto",
                    help="trace launches refill finished lanes / take fixed chunks (auto: measured)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-work-counters", action="store_true",
                    help="skip the untimed counting call (roofline becomes null)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target duration of the CPU baseline sample")
    args = ap.parse_args()
    width, height, spp = DEFAULTS.get(args.scene, (1280, 720, 64))
    args.width = args.width or width
    args.height = args.height or height
    args.spp_per_step = args.spp_per_step or spp
    return args


def spawn_ranks(n):
    """Starts the n ranks as fresh child processes (this process has not touched the GPU and
    never will) and returns the worst exit code.  Rank 0's JSON line passes through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                         env=env))
    codes = [c.wait() for c in children]
    return max((abs(c) for c in codes), default=0)


def cpu_threads():
    """Threads for the CPU baseline: every core this process may use.  A cgroup CPU quota or an
    affinity mask narrower than the machine counts; so does OMP_NUM_THREADS when the box sets it
    to this job's share of a larger host (its os.cpu_count() shows the whole machine)."""
    host = os.cpu_count() or 1
    limit, why = host, "os.cpu_count()"
    try:
        affinity = len(os.sched_getaffinity(0))
        if affinity < limit:
            limit, why = affinity, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            cores = max(1, math.ceil(int(quota) / int(period)))
            if cores < limit:
                limit, why = cores, "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    omp = os.environ.get("OMP_NUM_THREADS")
    if omp and omp.isdigit() and 0 < int(omp) < limit:
        limit, why = int(omp), "OMP_NUM_THREADS (the job's CPU share on this host)"
    return limit, host, why


def cpu_baseline(sc, cam, args, tile, target_seconds):
    """Times the CPU restatement (the oracle) on a bounded sample of the same workload."""
    import oracle
    from raytracing_amd import host
    threads, host_cores, why = cpu_threads()
    tiles = (args.width // tile[0]) * (args.height // tile[1])
    # calibrate on a strip of the frame (a whole 4K pass of a heavy scene can take minutes)
    rows = max(tile[1], (args.height // 8) // tile[1] * tile[1])
    seeds = host.seed_schedule(0, 1, tiles)
    t0 = time.perf_counter()
    oracle.render(sc.desc, cam, args.width, args.height, args.max_depth, tile, seeds,
                  threads=threads, rect=(0, (args.height - rows) // 2 // tile[1] * tile[1],
                                         args.width,
                                         (args.height - rows) // 2 // tile[1] * tile[1] + rows))
    strip = time.perf_counter() - t0
    per_pass = strip * args.height / rows
    if per_pass <= target_seconds:
        passes = max(1, min(64, int(target_seconds / max(per_pass, 1e-3))))
        rect = (0, 0, args.width, args.height)
        what = f"{passes} passes of the full {args.width}x{args.height} frame"
    else:
        passes = 1
        keep = max(tile[1], int(args.height * target_seconds / per_pass) // tile[1] * tile[1])
        y0 = (args.height - keep) // 2 // tile[1] * tile[1]
        rect = (0, y0, args.width, y0 + keep)
        what = (f"1 pass of rows {y0}..{y0 + keep} (the middle {keep} of {args.height}) of the "
                f"{args.width}x{args.height} frame")
    seeds = host.seed_schedule(0, passes, tiles)
    t0 = time.perf_counter()
    _, st = oracle.render(sc.desc, cam, args.width, args.height, args.max_depth, tile, seeds,
                          threads=threads, rect=rect)
    dt = time.perf_counter() - t0
    return {
        "value": st["samples"] / dt / 1e6,
        "unit": "Msamples/s",
        "cores": threads,
        "host_cores": host_cores,
        "cores_limited_by": why,
        "kind": "port",
        "sample": f"{what}, same scene, seed schedule and max_depth; {dt:.1f} s; one OpenMP "
                  f"thread per core over image rows",
        "mrays_per_s": (st["closest_rays"] + st["shadow_rays"]) / dt / 1e6,
    }


def load_reference_tree_constants(scene_name):
    path = os.path.join(ROOT, "tests", "golden", "workload_constants.json")
    with open(path) as f:
        return json.load(f).get(scene_name)


def committed_counters(args):
    """PMC measurements of this exact workload committed under profiles/ (bench.py cannot
    collect counters itself): HBM bytes per launch and the issue-side fractions."""
    for rnd in ("r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, "traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            measured = json.load(f)
        for t in measured if isinstance(measured, list) else [measured]:
            if (t.get("scene") == args.scene and t.get("width") == args.width
                    and t.get("height") == args.height and t.get("max_depth") == args.max_depth
                    and t.get("spp_per_step") == args.spp_per_step
                    and t.get("pipeline") == args.pipeline and t.get("accel") == args.accel):
                return t
    return None


def count_work(hip, host, sc, cam, args, tile, accel, device_index, first_pass, seeds_row,
               refill):
    """One untimed render call of the same step with the node / triangle counters on: what the
    trace launches of this build fetch per ray, how much of it their LDS copy serves and how many
    lanes their steps keep in use (`refill`: hand out rays the way the timed steps did)."""
    flags = hip.FLAG_COUNT_NODES | hip.FLAG_NO_BIN_RAYS  # the order of the queues changes no count
    flags |= hip.FLAG_TRACE_REFILL if refill else hip.FLAG_TRACE_CHUNKS
    r = hip.HipRenderer(args.width, args.height, args.max_depth, tile, device=device_index,
                        accel=accel, flags=flags)
    try:
        r.upload_scene(sc.desc)
        r.upload_seeds(seeds_row, first_pass)
        r.clear()
        r.render(cam, first_pass, args.spp_per_step)
        st = r.stats()
        work = r.trace_work()
        per_bounce = [r.trip_counts(t) for t in range(args.max_depth + 1)]
    finally:
        r.close()
    return st, work, per_bounce


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    from raytracing_amd import hip, host, scenes
    from raytracing_amd.distributed import new_accumulator, reduce_accumulator

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RT_BENCH_ECHO_RANK"):
        print(f"bench rank {rank} of {world} (local {local_rank})", file=sys.stderr, flush=True)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node "
                         f"({torch.cuda.device_count()} visible)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    tile = (args.width // 10, args.height // 10)  # 128x72 at 1280x720, as mustang_red.json
    sc = scenes.load(args.scene)
    cam = sc.camera(args.width, args.height)
    accel = hip.ACCEL_REBUILT if args.accel == "rebuilt" else hip.ACCEL_REFERENCE
    flags = hip.FLAG_FUSED if args.pipeline == "fused" else hip.FLAG_TIME_KERNELS
    flags |= {"auto": 0, "on": hip.FLAG_BIN_RAYS, "off": hip.FLAG_NO_BIN_RAYS}[args.binning]
    flags |= {"auto": 0, "on": hip.FLAG_TRACE_REFILL, "off": hip.FLAG_TRACE_CHUNKS}[args.refill]
    r = hip.HipRenderer(args.width, args.height, args.max_depth, tile, device=local_rank,
                        accel=accel, flags=flags)
    r.upload_scene(sc.desc)

    spp = args.spp_per_step
    setup_calls = 3                                     # the untimed set-up calls below
    total_steps = args.warmup + args.steps + setup_calls
    passes_per_rank = total_steps * spp
    first_pass = rank * passes_per_rank  # global pass index: ranks trace disjoint passes
    seeds = host.seed_schedule(first_pass, passes_per_rank, r.tile_count)
    r.upload_seeds(seeds, first_pass)

    accum = new_accumulator(args.height, args.width, device)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        r.render(cam, first_pass + i * spp, spp, accum_dev=accum.data_ptr(), stream=stream)

    # Set-up calls: the first allocates the pipeline state and, with --binning auto, is the call
    # the library measures for the ray binning; the next two are the ones it times with fixed
    # chunks and with refilled lanes in the trace launches.  Waiting for each and asking for the
    # verdict settles both policies, so every warm-up and timed step below runs the same
    # configuration.
    binning_state, probe_ns, refill = 1, 0.0, (0, -1.0, -1.0, 0)
    for i in range(setup_calls):
        step(i)
        torch.cuda.synchronize()
        if args.pipeline != "fused":
            binning_state, probe_ns = r.ray_binning()
            refill = r.trace_refill()
    for i in range(args.warmup):
        step(setup_calls + i)
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
        step(setup_calls + args.warmup + i)
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
        roofline = None
        if launches > 0 and not args.no_work_counters:
            if args.pipeline == "fused":
                name, ms, n = "path_trace_kernel", kernel_ms, launches
                extra = {}
            else:
                name = "wf_extend_kernel"
                ms, n = r.kernel_time(hip.KERNEL_EXTEND)
                shade_ms, shade_n = r.kernel_time(hip.KERNEL_SHADE)
                per_call = n // max(launches, 1)
                extra = {"ray_binning": ["undecided", "off", "on"][binning_state],
                         "ray_binning_probe_ns_per_ray": probe_ns,
                         "trace_refill": ["undecided", "off", "on"][refill[0]],
                         "trace_refill_probe_ms": [refill[1], refill[2]],
                         "trace_refill_bounces": f"{refill[3]:#x}",
                         "trace_launches_per_call": per_call,
                         "render_call_avg_ms": kernel_ms, "render_calls": launches,
                         "shade_kernel_avg_ms": shade_ms, "shade_launches": shade_n,
                         "extend_ms_per_call": ms * n / launches,
                         "shade_ms_per_call": shade_ms * shade_n / launches}
            # the same step once more, untimed, with the work counters on
            cst, work, per_bounce = count_work(
                hip, host, sc, cam, args, tile, accel, local_rank, first_pass, seeds[:spp],
                refill[0] != hip.REFILL_OFF)
            if args.pipeline == "fused":
                rays_l = cst["closest_rays"] + cst["shadow_rays"]
                closest_l, shadow_l = cst["closest_rays"], cst["shadow_rays"]
                nodes_l, tris_l = work["nodes"], work["tris"]
                launches_per_call = 1
            else:
                # the trace launches of one call: every bounce that was not handed to the
                # finishing kernel (whose own fetches are subtracted); launches of later bounces
                # return at once and still count as launches, as rocprofv3 counts them
                closest_l = sum(c[1] for c in per_bounce)
                shadow_l = sum(c[2] for c in per_bounce)
                rays_l = closest_l + shadow_l
                nodes_l = work["nodes"] - work["finish_nodes"]
                tris_l = work["tris"] - work["finish_tris"]
                launches_per_call = max(per_call, 1)
                extra["rays_per_bounce"] = [c[1] + c[2] for c in per_bounce]
            bytes_call = nodes_l * NODE_BYTES + tris_l * TRI_BYTES
            bytes_l = bytes_call / launches_per_call
            achieved = bytes_l / (ms * 1e-3) / 1e9
            structure = work["node_bytes"] + work["triangle_bytes"] + work["instance_bytes"]
            in_l2 = structure <= L2_BYTES
            peak = L2_PEAK_GBS if in_l2 else HBM_PEAK_GBS
            lds_share = work["nodes_from_lds"] / max(nodes_l, 1)
            memory_bytes_l = ((nodes_l - work["nodes_from_lds"]) * NODE_BYTES
                              + tris_l * TRI_BYTES) / launches_per_call
            measured = committed_counters(args)
            traffic = (measured or {}).get("bytes_per_launch", {}).get(name)
            k = load_reference_tree_constants(args.scene)
            reference_tree = None
            if k is not None:
                reference_tree = (rays_l * (k["nodes_per_ray"] * REF_NODE_BYTES
                                            + k["tris_per_ray"] * REF_TRI_BYTES)
                                  / launches_per_call / (ms * 1e-3) / 1e9)
            roofline = {
                "bound": "l2" if in_l2 else "hbm",
                "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                "traffic": traffic,
                "kernel": name, "kernel_avg_ms": ms, "launches": n,
                "algorithmic_bytes_per_launch": bytes_l,
                "definition": "(wide-node fetches x 112 B + triangle tests x 36 B) of this "
                              "build's own tree, counted on the device in an untimed repeat of "
                              "one step, / trace launches per call / average launch time; peak "
                              "= aggregate L2 bandwidth when nodes + triangles + instance "
                              "records fit the 32 MiB of L2, HBM otherwise",
                "per_ray": {"node_fetches": nodes_l / max(rays_l, 1),
                            "triangle_tests": tris_l / max(rays_l, 1),
                            "instance_records": work["instance_records"] / max(rays_l, 1),
                            "bytes": bytes_call / max(rays_l, 1)},
                "lanes_in_use": {
                    "node_steps": ((work["nodes"] - work["finish_nodes"])
                                   / (64.0 * work["node_wave_steps"])
                                   if work.get("node_wave_steps") else None),
                    "triangle_tests": ((work["tris"] - work["finish_tris"])
                                       / (64.0 * work["triangle_wave_steps"])
                                       if work.get("triangle_wave_steps") else None),
                    "instance_entries": (work["instance_records"]
                                         / (64.0 * work["instance_wave_steps"])
                                         if work.get("instance_wave_steps") else None),
                    "wave_steps_per_ray": {
                        "node": work.get("node_wave_steps", 0) * 64.0 / max(rays_l, 1),
                        "triangle": work.get("triangle_wave_steps", 0) * 64.0 / max(rays_l, 1),
                        "instance": work.get("instance_wave_steps", 0) * 64.0 / max(rays_l, 1)}},
                "rays_per_launch": rays_l / launches_per_call,
                "deepest_traversal_stack": cst["max_stack"],
                "node_fetches_served_from_lds": lds_share,
                "memory_served_gbs": memory_bytes_l / (ms * 1e-3) / 1e9,
                "lds_served_gbs": work["nodes_from_lds"] * NODE_BYTES / launches_per_call
                                  / (ms * 1e-3) / 1e9,
                "streamed_ray_state_gbs": (closest_l * CLOSEST_STATE_BYTES
                                           + shadow_l * SHADOW_STATE_BYTES)
                                          / launches_per_call / (ms * 1e-3) / 1e9,
                "structure_bytes": {"nodes": work["node_bytes"],
                                    "nodes_in_lds": work["node_bytes_cached"],
                                    "triangles": work["triangle_bytes"],
                                    "instance_records": work["instance_bytes"]},
                "traffic_frac_of_hbm_peak": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                             if traffic else None),
                "issue": (measured or {}).get("issue"),
                "source": (measured or {}).get("source"),
                "reference_tree_equivalent_gbs": reference_tree,
                **extra}
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
            "data": scenes.describe(args.scene),
            "config": {
                "workload": f"{args.scene}, {args.width}x{args.height}, max_depth "
                            f"{args.max_depth}, {spp} spp per step, accel={args.accel}, "
                            f"pipeline={args.pipeline}",
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

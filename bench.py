#!/usr/bin/env python3
"""Throughput benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scene NAME] [--total-spp S]

One step = one rt_hip_render call over the whole frame at max_depth 7, tracing --spp-per-step
passes of every pixel (scene, seeds and accumulator already resident in HBM; default 512 passes,
which the library traces as two batches of 256 side by side on two streams: 2 x 236 M paths,
76 GB of path state of the 288 GB).  The reported
value is pixel-samples per second over all GPUs, in Msamples/s; Mrays/s comes from the device
ray counters of the same timed region.

Scenes: snapshots of the reference's scenes (tests/golden/scenes; default
hyperion_rect_light at 1280x720) and `synthetic_4k`, the generated 10 000-instance scene at
3840x2160 (raytracing_amd/scenes.py).  After the headline measurement a default single-GPU run
adds a short leg for each of the other configurations (`other_scenes` in the JSON line).

N > 1: one process per GPU.  Under torchrun (WORLD_SIZE set) this process is one rank; without
it, `--gpus N` starts the N ranks itself as child processes before touching the GPU.  Each
rank traces its own pass range of the same frame (global pass index in the seed schedule,
scene replicated) and the accumulators are summed onto rank 0 with one RCCL reduce inside the
timed region.
  * default: work per GPU is fixed (--spp-per-step passes per rank and step): weak scaling.
  * --total-spp S: one step is S passes of the frame in all, shared by the ranks with
    distributed.pass_range (rank r traces global passes first_r .. first_r + count_r of the
    step, in render calls of at most --spp-per-step passes): strong scaling.
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

# Peaks of the memory levels a traversal's node / triangle fetches can be served from
# (MI355X microarchitecture guide): HBM3E; the 256 MiB Infinity Cache read by a chip-wide
# random gather; the eight XCD-private 4 MiB L2 slices together.
HBM_PEAK_GBS = 8000.0
INFINITY_CACHE_GATHER_GBS = 8600.0
L2_PEAK_GBS = 34500.0
L2_SLICE_BYTES = 4 << 20           # per XCD: every XCD walks the whole structure
INFINITY_CACHE_BYTES = 256 << 20

# Algorithmic bytes of the trace kernel, per unit of ITS OWN work (the rebuilt 4-wide tree):
NODE_BYTES = 112        # one wide node = boxes of four children (96 B) + four child words
TRI_BYTES = 36          # nine position floats per triangle test
INSTANCE_BYTES = 64     # one instance record per TLAS leaf entered (reported, not in `achieved`)
# per-ray state the trace launch streams: pid 4 + ray 32 in, hit record 24 out (closest) /
# pid 4 + ray 32 + radiance 16 + pending light sample 16 in, radiance 16 out for the rays that
# get through, about half (occlusion) -- reported, not in `achieved`
CLOSEST_STATE_BYTES = 60
SHADOW_STATE_BYTES = 76
# path state the two shading launches stream (rt_wavefront.h): a live-list entry, the hit
# record's first quad and the raw slot for every live path; for a path that goes on, also
# radiance, throughput, ray origin and direction and the rest of the hit record in, and radiance,
# throughput, light sample, two rays and the raw slot out
SHADE_LIVE_BYTES = 4 + 16 + 4
SHADE_GOING_READ_BYTES = 8 + 4 * 16 + 8
SHADE_GOING_WRITE_BYTES = 7 * 16 + 4
PATH_STATE_BYTES = 136
# the previous definition, on the REFERENCE's binary tree (kept under its own key):
REF_NODE_BYTES = 32
REF_TRI_BYTES = 36

DEFAULT_SCENE = "hyperion_rect_light"
DEFAULT_SPP = 512       # per render call: two batches of 256 passes = 236 M paths each
DEFAULTS = {  # scene -> (width, height, spp per step)
    "synthetic_4k": (3840, 2160, 64),  # two batches of 32 passes = 265 M paths each
}
# the other configurations of the baseline: (scene, seconds of CPU baseline)
OTHER_SCENES = (("mustang_red_reduced", 6.0), ("cornell_box", 3.0), ("staircase_reduced", 3.0),
                ("teapot_standin", 3.0), ("synthetic_4k", 3.0))
SETUP_CALLS = 3  # untimed: fixed-chunk base line, refilled lanes, settled (see measure())


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default=DEFAULT_SCENE)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=7)
    ap.add_argument("--spp-per-step", type=int, default=None,
                    help="passes per render call (default 512 = two batches of 256 passes in "
                         "flight side by side: 2 x 236 M paths, 76 GB of path state at 1280x720; "
                         "128 per call is 7 %% slower per sample, 64 per call 18 %%)")
    ap.add_argument("--total-spp", type=int, default=None,
                    help="strong scaling: one step = this many passes of the frame in all, "
                         "shared by the ranks (pass_range); default: --spp-per-step passes per "
                         "rank and step (weak scaling)")
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
    ap.add_argument("--no-other-scenes", action="store_true",
                    help="skip the short legs on the other configurations")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target duration of the CPU baseline sample")
    args = ap.parse_args(argv)
    width, height, spp = DEFAULTS.get(args.scene, (1280, 720, DEFAULT_SPP))
    args.width = args.width or width
    args.height = args.height or height
    args.spp_per_step = args.spp_per_step or spp
    if args.total_spp is not None and args.total_spp < 1:
        ap.error("--total-spp must be at least 1")
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


def cpu_baseline(sc, cam, w, tile, target_seconds):
    """Times the CPU restatement (the oracle) on a bounded sample of the same workload."""
    import oracle
    from raytracing_amd import host
    threads, host_cores, why = cpu_threads()
    tiles = (w.width // tile[0]) * (w.height // tile[1])
    # calibrate on a strip of the frame (a whole 4K pass of a heavy scene can take minutes)
    rows = max(tile[1], (w.height // 8) // tile[1] * tile[1])
    seeds = host.seed_schedule(0, 1, tiles)
    t0 = time.perf_counter()
    oracle.render(sc.desc, cam, w.width, w.height, w.max_depth, tile, seeds,
                  threads=threads, rect=(0, (w.height - rows) // 2 // tile[1] * tile[1],
                                         w.width,
                                         (w.height - rows) // 2 // tile[1] * tile[1] + rows))
    strip = time.perf_counter() - t0
    per_pass = strip * w.height / rows
    if per_pass <= target_seconds:
        passes = max(1, min(64, int(target_seconds / max(per_pass, 1e-3))))
        rect = (0, 0, w.width, w.height)
        what = f"{passes} passes of the full {w.width}x{w.height} frame"
    else:
        passes = 1
        keep = max(tile[1], int(w.height * target_seconds / per_pass) // tile[1] * tile[1])
        y0 = (w.height - keep) // 2 // tile[1] * tile[1]
        rect = (0, y0, w.width, y0 + keep)
        what = (f"1 pass of rows {y0}..{y0 + keep} (the middle {keep} of {w.height}) of the "
                f"{w.width}x{w.height} frame")
    seeds = host.seed_schedule(0, passes, tiles)
    t0 = time.perf_counter()
    _, st = oracle.render(sc.desc, cam, w.width, w.height, w.max_depth, tile, seeds,
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


def committed_counters(w):
    """PMC measurements of this exact workload committed under profiles/ (bench.py cannot
    collect counters itself): HBM bytes per launch and the issue-side fractions."""
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, "traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            measured = json.load(f)
        for t in measured if isinstance(measured, list) else [measured]:
            if (t.get("scene") == w.scene and t.get("width") == w.width
                    and t.get("height") == w.height and t.get("max_depth") == w.max_depth
                    and t.get("spp_per_step") == w.spp_per_step
                    and t.get("pipeline") == w.pipeline and t.get("accel") == w.accel):
                return t
    return None


def call_schedule(w, world, rank, steps, warmup):
    """The render calls of one rank, in order: (global first pass, pass count) each, and the
    index of the first call of every step (set-up and warm-up steps included).

    Weak scaling: every step is one call of spp_per_step passes; rank r owns the contiguous
    global passes [r * P, (r + 1) * P), P = all its calls together.
    Strong scaling: step s covers the global passes [s * total, (s + 1) * total); rank r traces
    pass_range(total, world, r) of them in calls of at most spp_per_step passes.  The set-up
    calls (which settle the trace policies and must be large) stay spp_per_step passes each and
    use passes beyond every step's."""
    from raytracing_amd.distributed import pass_range
    spp = w.spp_per_step
    all_steps = SETUP_CALLS + warmup + steps
    calls, starts = [], []
    if w.total_spp is None:
        first = rank * all_steps * spp
        for s in range(all_steps):
            starts.append(len(calls))
            calls.append((first + s * spp, spp))
        return calls, starts
    first_r, count_r = pass_range(w.total_spp, world, rank)
    beyond = (warmup + steps) * w.total_spp + rank * SETUP_CALLS * spp
    for s in range(SETUP_CALLS):
        starts.append(len(calls))
        calls.append((beyond + s * spp, spp))
    for s in range(warmup + steps):
        starts.append(len(calls))
        done = 0
        while done < count_r:
            n = min(spp, count_r - done)
            calls.append((s * w.total_spp + first_r + done, n))
            done += n
    return calls, starts


class Workload:
    """What one measurement runs: scene, frame, depth, passes per call and the trace options."""

    def __init__(self, args, scene=None, total_spp="args"):
        self.scene = scene or args.scene
        if scene is None:
            self.width, self.height, self.spp_per_step = args.width, args.height, args.spp_per_step
        else:
            self.width, self.height, self.spp_per_step = DEFAULTS.get(scene,
                                                                      (1280, 720, DEFAULT_SPP))
        self.max_depth = args.max_depth
        self.accel, self.pipeline = args.accel, args.pipeline
        self.binning, self.refill = args.binning, args.refill
        self.total_spp = args.total_spp if total_spp == "args" else total_spp
        self.tile = (self.width // 10, self.height // 10)  # 128x72 at 1280x720, as mustang_red.json

    def describe(self):
        return (f"{self.scene}, {self.width}x{self.height}, max_depth {self.max_depth}, "
                f"{self.spp_per_step} spp per render call, accel={self.accel}, "
                f"pipeline={self.pipeline}")


def measure(env, w, steps, warmup, work_counters):
    """Runs `steps` timed steps of workload `w` on this rank (after the set-up and warm-up
    steps) and returns the raw measurements; rank 0 also gets the counted work."""
    torch, dist, hip, host, scenes = env.torch, env.dist, env.hip, env.host, env.scenes
    from raytracing_amd.distributed import new_accumulator, reduce_accumulator
    sc = scenes.load(w.scene)
    cam = sc.camera(w.width, w.height)
    accel = hip.ACCEL_REBUILT if w.accel == "rebuilt" else hip.ACCEL_REFERENCE
    flags = hip.FLAG_FUSED if w.pipeline == "fused" else hip.FLAG_TIME_KERNELS
    flags |= {"auto": 0, "on": hip.FLAG_BIN_RAYS, "off": hip.FLAG_NO_BIN_RAYS}[w.binning]
    flags |= {"auto": 0, "on": hip.FLAG_TRACE_REFILL, "off": hip.FLAG_TRACE_CHUNKS}[w.refill]
    r = hip.HipRenderer(w.width, w.height, w.max_depth, w.tile, device=env.local_rank,
                        accel=accel, flags=flags)
    r.upload_scene(sc.desc)

    calls, starts = call_schedule(w, env.world, env.rank, steps, warmup)
    starts.append(len(calls))
    # the seed table holds this rank's calls back to back; call k starts at local pass local[k]
    # and carries the seeds of ITS global passes, so the union over the ranks is the seed
    # schedule of one GPU rendering every pass
    import numpy as np
    local, at = [], 0
    rows = []
    for first, count in calls:
        local.append(at)
        rows.append(host.seed_schedule(first, count, r.tile_count))
        at += count
    seeds = np.concatenate(rows, axis=0) if rows else np.zeros((0, r.tile_count), np.uint32)
    if len(seeds):
        r.upload_seeds(seeds, 0)

    accum = new_accumulator(w.height, w.width, env.device)
    stream = torch.cuda.current_stream().cuda_stream

    def step(s):
        for k in range(starts[s], starts[s + 1]):
            r.render(cam, local[k], calls[k][1], accum_dev=accum.data_ptr(), stream=stream)

    # Set-up calls: the first allocates the pipeline state and runs the trace launches in fixed
    # chunks, each timed on the device (the base line of both policy decisions); the second
    # refills finished lanes and is timed likewise.  Waiting for each and asking for the verdict
    # settles the policies per bounce, so the third call and every warm-up and timed step run
    # the same configuration.
    binning_state, probe_ns, refill = 1, 0.0, (0, -1.0, -1.0, 0)
    for s in range(SETUP_CALLS):
        step(s)
        torch.cuda.synchronize()
        if w.pipeline != "fused":
            binning_state, probe_ns = r.ray_binning()
            refill = r.trace_refill()
    for s in range(warmup):
        step(SETUP_CALLS + s)
    torch.cuda.synchronize()
    r.stats(reset=True)
    accum.zero_()
    accum[..., 3] = 1.0
    torch.cuda.synchronize()
    if env.world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        step(SETUP_CALLS + warmup + s)
    reduce_accumulator(accum, dst=0)
    torch.cuda.synchronize()
    if env.world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    st = r.stats()
    times = torch.tensor([elapsed], dtype=torch.float64, device=env.device)
    counts = torch.tensor([st["samples"], st["closest_rays"], st["shadow_rays"],
                           st["closest_hits"]], dtype=torch.float64, device=env.device)
    if env.world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    m = {"elapsed": float(times.item()), "steps": steps, "sc": sc, "cam": cam}
    m["samples"], m["closest"], m["shadow"], m["hits"] = (float(v) for v in counts.tolist())
    m["passes_this_rank_per_step"] = sum(c[1] for c in calls[starts[SETUP_CALLS + warmup]:
                                                          starts[SETUP_CALLS + warmup + 1]]) \
        if steps > 0 else 0
    if env.rank == 0:
        m["nan_pixels"] = int(torch.isnan(accum).any(dim=2).sum().item())
        m["render_ms"], m["render_calls"] = r.kernel_time()
        m["binning_state"], m["probe_ns"], m["refill"] = binning_state, probe_ns, refill
        if w.pipeline != "fused":
            m["extend_ms"], m["extend_n"] = r.kernel_time(hip.KERNEL_EXTEND)
            m["shade_ms"], m["shade_n"] = r.kernel_time(hip.KERNEL_SHADE)
            # two batches of a call run side by side, so launches of one kind overlap: the time
            # during which at least one of them ran
            m["extend_busy_ms"] = r.kernel_busy(hip.KERNEL_EXTEND)
            m["shade_busy_ms"] = r.kernel_busy(hip.KERNEL_SHADE)
    r.close()
    del accum
    if env.rank == 0 and work_counters and m["render_calls"] > 0:
        m["work"] = count_work(env, w, sc, cam, accel, host.seed_schedule(0, w.spp_per_step,
                                                                           (w.width // w.tile[0]) *
                                                                           (w.height // w.tile[1])),
                               refill[0] != hip.REFILL_OFF)
    return m


def count_work(env, w, sc, cam, accel, seeds_row, refill):
    """One untimed render call of the same size with the node / triangle counters on: what the
    trace launches of this build fetch per ray, how much of it their LDS copy serves and how many
    lanes their steps keep in use (`refill`: hand out rays the way the timed steps did)."""
    hip = env.hip
    flags = hip.FLAG_COUNT_NODES | hip.FLAG_NO_BIN_RAYS  # the order of the queues changes no count
    flags |= hip.FLAG_TRACE_REFILL if refill else hip.FLAG_TRACE_CHUNKS
    if w.pipeline == "fused":
        flags |= hip.FLAG_FUSED
    r = hip.HipRenderer(w.width, w.height, w.max_depth, w.tile, device=env.local_rank,
                        accel=accel, flags=flags)
    try:
        r.upload_scene(sc.desc)
        r.upload_seeds(seeds_row, 0)
        r.clear()
        r.render(cam, 0, w.spp_per_step)
        st = r.stats()
        work = r.trace_work()
        per_bounce = ([] if w.pipeline == "fused"   # (the fused kernel keeps no lists)
                      else [r.trip_counts(t) for t in range(w.max_depth + 1)])
    finally:
        r.close()
    return st, work, per_bounce


def memory_level(structure_bytes):
    """Where the traversal's structure can stay resident, and the peak it is then read at.
    Each of the eight XCDs walks the whole structure through its own 4 MiB L2 slice."""
    if structure_bytes <= L2_SLICE_BYTES:
        return "l2", L2_PEAK_GBS
    if structure_bytes <= INFINITY_CACHE_BYTES:
        return "infinity-cache", INFINITY_CACHE_GATHER_GBS
    return "hbm", HBM_PEAK_GBS


def roofline_of(w, m):
    """The roofline object of the dominant kernel (the trace kernel) from the timed launches'
    average duration and the counted work of one call."""
    if "work" not in m:
        return None
    cst, work, per_bounce = m["work"]
    calls = max(m["render_calls"], 1)
    extra = {}
    if w.pipeline == "fused":
        name, ms, n = "path_trace_kernel", m["render_ms"], m["render_calls"]
        ms_wall = ms
        closest_l, shadow_l = cst["closest_rays"], cst["shadow_rays"]
        nodes_l, tris_l = cst["nodes_visited"], cst["tris_tested"]  # (its own counters)
        launches_per_call = 1
    else:
        name, ms_wall, n = "wf_extend_kernel", m["extend_ms"], m["extend_n"]
        # net of overlap: the time during which at least one trace launch ran / launches
        ms = m["extend_busy_ms"] / max(n, 1)
        shade_net = m["shade_busy_ms"] / max(m["shade_n"], 1)
        per_call = n // calls
        refill = m["refill"]
        extra = {"ray_binning": ["undecided", "off", "on"][m["binning_state"]],
                 "ray_binning_probe_ns_per_ray": m["probe_ns"],
                 "trace_refill": ["undecided", "off", "on"][refill[0]],
                 "trace_refill_probe_ms": [refill[1], refill[2]],
                 "trace_refill_bounces": f"{refill[3]:#x}",
                 "trace_launches_per_call": per_call,
                 "render_call_avg_ms": m["render_ms"], "render_calls": m["render_calls"],
                 "shade_kernel_avg_ms": shade_net,
                 "shade_kernel_avg_ms_wall": m["shade_ms"], "shade_launches": m["shade_n"],
                 "extend_ms_per_call": m["extend_busy_ms"] / calls,
                 "shade_ms_per_call": m["shade_busy_ms"] / calls,
                 "extend_ms_per_call_sum_of_launches": ms_wall * n / calls,
                 "shade_ms_per_call_sum_of_launches": m["shade_ms"] * m["shade_n"] / calls}
        # the trace launches of one call: every bounce that was not handed to the finishing
        # kernel (whose own fetches are subtracted); launches of later bounces return at once
        # and still count as launches, as rocprofv3 counts them
        closest_l = sum(c[1] for c in per_bounce)
        shadow_l = sum(c[2] for c in per_bounce)
        nodes_l = work["nodes"] - work["finish_nodes"]
        tris_l = work["tris"] - work["finish_tris"]
        launches_per_call = max(per_call, 1)
        extra["rays_per_bounce"] = [c[1] + c[2] for c in per_bounce]
    if ms <= 0.0:
        return None
    rays_l = closest_l + shadow_l
    seconds = ms * 1e-3
    bytes_call = nodes_l * NODE_BYTES + tris_l * TRI_BYTES
    bytes_l = bytes_call / launches_per_call
    lds_bytes_l = work["nodes_from_lds"] * NODE_BYTES / launches_per_call
    memory_bytes_l = bytes_l - lds_bytes_l
    structure = work["node_bytes"] + work["triangle_bytes"] + work["instance_bytes"]
    bound, peak = memory_level(structure)
    achieved = memory_bytes_l / seconds / 1e9
    measured = committed_counters(w) or {}
    traffic = measured.get("bytes_per_launch", {}).get(name)
    issue = measured.get("issue") or {}
    own = issue.get(name) or {}
    issue_frac = None
    if own.get("valu_busy_of_kernel_time") is not None and own.get("valu_lanes_in_use") is not None:
        issue_frac = own["valu_busy_of_kernel_time"] * own["valu_lanes_in_use"]
    k = load_reference_tree_constants(w.scene)
    reference_tree = None
    if k is not None:
        reference_tree = (rays_l * (k["nodes_per_ray"] * REF_NODE_BYTES
                                    + k["tris_per_ray"] * REF_TRI_BYTES)
                          / launches_per_call / seconds / 1e9)
    roofline = {
        "bound": bound,
        "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
        "frac_of_each_level": {"l2": achieved / L2_PEAK_GBS,
                               "infinity-cache": achieved / INFINITY_CACHE_GATHER_GBS,
                               "hbm": achieved / HBM_PEAK_GBS},
        "traffic": traffic,
        "kernel": name, "kernel_avg_ms": ms, "kernel_avg_ms_wall": ms_wall, "launches": n,
        "kernel_avg_ms_definition": "two batches of a render call are in flight side by side, "
                                    "so two trace launches often share the GPU: kernel_avg_ms = "
                                    "(time during which at least one trace launch was running) / "
                                    "launches, from the HIP events of every timed launch; "
                                    "kernel_avg_ms_wall = the launches' own start-to-end times "
                                    "averaged, which is what rocprofv3 --stats reports",
        "algorithmic_bytes_per_launch": bytes_l,
        "algorithmic_bytes_per_launch_served_by_memory": memory_bytes_l,
        "algorithmic_bytes_per_launch_served_by_lds": lds_bytes_l,
        "definition": "(wide-node fetches x 112 B + triangle tests x 36 B) of this build's own "
                      "tree, counted on the device in an untimed repeat of one render call, / "
                      "trace launches per call / average launch time.  `achieved` and `frac` "
                      "count only the fetches the memory system served (the workgroup's LDS copy "
                      "of the hottest nodes serves the rest: lds_served_*).  bound/peak: where "
                      "nodes + triangles + instance records stay resident - one XCD's 4 MiB L2 "
                      "slice (every XCD walks the whole structure; 34.5 TB/s over the eight), "
                      "else the 256 MiB Infinity Cache (8.6 TB/s gather), else HBM (8 TB/s).  A "
                      "structure a few times the size of an L2 slice is still served mostly by "
                      "the L2s and the CUs' own caches (the top of the tree is hot), so `frac` "
                      "against the Infinity Cache figure can reach or pass 1 (the staircase): "
                      "frac_of_each_level gives the same rate against all three peaks",
        "binding": "valu-issue",
        "issue_frac": issue_frac,
        "issue_frac_definition": "VALU busy share of the kernel's time x share of the 64 lanes "
                                 "doing useful work per VALU instruction (PMC, `source`): the "
                                 "roof this kernel runs into before any memory level",
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
        "lds_served_share_of_node_fetches": work["nodes_from_lds"] / max(nodes_l, 1),
        "lds_served_gbs": lds_bytes_l / seconds / 1e9,
        "algorithmic_gbs_lds_and_memory": bytes_l / seconds / 1e9,
        "streamed_ray_state_gbs": (closest_l * CLOSEST_STATE_BYTES
                                   + shadow_l * SHADOW_STATE_BYTES)
                                  / launches_per_call / seconds / 1e9,
        "structure_bytes": {"nodes": work["node_bytes"],
                            "nodes_in_lds": work["node_bytes_cached"],
                            "triangles": work["triangle_bytes"],
                            "instance_records": work["instance_bytes"],
                            "total": structure},
        "traffic_frac_of_hbm_peak": (traffic / seconds / 1e9 / HBM_PEAK_GBS
                                     if traffic else None),
        "issue": issue or None,
        "source": measured.get("source"),
        "reference_tree_equivalent_gbs": reference_tree,
        **extra}
    if w.pipeline != "fused":
        roofline["shade"] = shade_block(m, per_bounce, measured, calls)
    return roofline


def shade_block(m, per_bounce, measured, calls):
    """The two shading launches per bounce (wf_settle_kernel, wf_shade_kernel) - the other
    half of a frame: time, streamed path state (algorithmic, from the counted paths) and HBM
    traffic and issue fractions from the committed PMC passes."""
    live = [c[0] for c in per_bounce]
    # a path that goes on at bounce t is alive at bounce t + 1 (all but the few whose sampled
    # direction had no density)
    going = live[1:] + [0]
    state_bytes = (sum(live) * SHADE_LIVE_BYTES
                   + sum(going) * (SHADE_GOING_READ_BYTES + SHADE_GOING_WRITE_BYTES))
    per_launch = measured.get("bytes_per_launch", {})
    launches_per_call = m["shade_n"] // max(calls, 1)  # timed pairs of launches
    traffic = None
    if "wf_shade_kernel" in per_launch:
        traffic = (per_launch["wf_shade_kernel"] + per_launch.get("wf_settle_kernel", 0.0)) \
            * launches_per_call
    ms_call = m["shade_busy_ms"] / max(calls, 1)
    issue = measured.get("issue") or {}
    return {
        "kernels": ["wf_settle_kernel", "wf_shade_kernel"],
        "ms_per_call": ms_call,
        "launch_pairs_per_call": launches_per_call,
        "paths_shaded_per_call": sum(live), "paths_that_go_on_per_call": sum(going),
        "state_bytes_per_path": PATH_STATE_BYTES,
        "algorithmic_state_bytes_per_call": state_bytes,
        "algorithmic_state_gbs": state_bytes / (ms_call * 1e-3) / 1e9 if ms_call > 0 else None,
        "traffic_bytes_per_call": traffic,
        "traffic_frac_of_hbm_peak": (traffic / (ms_call * 1e-3) / 1e9 / HBM_PEAK_GBS
                                     if traffic and ms_call > 0 else None),
        "bound": "valu-issue",
        "issue": {k: issue[k] for k in ("wf_shade_kernel", "wf_settle_kernel") if k in issue}
                 or None,
    }


def other_scene_line(env, args, scene, cpu_seconds):
    """A short leg on another configuration: 3 warm-up + 5 timed steps, one counting call."""
    w = Workload(args, scene=scene, total_spp=None)
    m = measure(env, w, steps=5, warmup=3, work_counters=True)
    roofline = roofline_of(w, m)
    line = {
        "scene": scene,
        "workload": w.describe(),
        "value": m["samples"] / m["elapsed"] / 1e6,
        "unit": "Msamples/s",
        "ms_per_step": m["elapsed"] / m["steps"] * 1e3,
        "mrays_per_s": (m["closest"] + m["shadow"]) / m["elapsed"] / 1e6,
        "rays_per_sample": (m["closest"] + m["shadow"]) / max(m["samples"], 1.0),
        "nan_pixels": m["nan_pixels"],
        "roofline_bound": roofline and roofline["bound"],
        "roofline_frac": roofline and roofline["frac"],
        "lds_served_share_of_node_fetches": roofline and
        roofline["lds_served_share_of_node_fetches"],
        "extend_ms_per_call": roofline and roofline.get("extend_ms_per_call"),
        "shade_ms_per_call": roofline and roofline.get("shade_ms_per_call"),
        "lanes_in_use_node_steps": roofline and roofline["lanes_in_use"]["node_steps"],
        "per_ray": roofline and roofline["per_ray"],
        "deepest_traversal_stack": roofline and roofline["deepest_traversal_stack"],
        "structure_bytes": roofline and roofline["structure_bytes"],
    }
    if not args.no_cpu_baseline and cpu_seconds > 0:
        cpu = cpu_baseline(m["sc"], m["cam"], w, w.tile, cpu_seconds)
        line["cpu_value"] = cpu["value"]
        line["cpu_baseline"] = cpu
        line["gpu_over_cpu"] = line["value"] / cpu["value"] if cpu["value"] > 0 else None
    return line


class Env:
    pass


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    from raytracing_amd import hip, host, scenes

    env = Env()
    env.torch, env.dist, env.hip, env.host, env.scenes = torch, dist, hip, host, scenes
    env.world = world
    env.rank = rank = int(os.environ.get("RANK", "0"))
    env.local_rank = local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    env.device = torch.device("cuda", local_rank)
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 1
    devices = torch.tensor([local_rank], dtype=torch.int64, device=env.device)
    if world > 1:
        gathered = [torch.zeros_like(devices) for _ in range(world)]
        dist.all_gather(gathered, devices)
        devices = torch.cat(gathered)
    rank_devices = [int(d) for d in devices.tolist()]

    w = Workload(args)
    m = measure(env, w, args.steps, args.warmup, work_counters=not args.no_work_counters)

    if rank == 0:
        roofline = roofline_of(w, m)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(m["sc"], m["cam"], w, w.tile, args.cpu_seconds)
        strong = args.total_spp is not None
        if strong:
            parallelism = (f"strong scaling: {args.total_spp} passes per step shared by {world} "
                           f"rank(s) (pass_range), scene replicated"
                           + (", one RCCL sum-reduce" if world > 1 else ""))
        elif world > 1:
            parallelism = (f"sample-range split x{world}, scene replicated, one RCCL sum-reduce")
        else:
            parallelism = "single GPU"
        line = {
            "metric": f"Msamples/s at {w.width}x{w.height} max_depth {w.max_depth}",
            "value": m["samples"] / m["elapsed"] / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": scenes.describe(w.scene),
            "config": {
                "workload": w.describe(),
                "samples_per_step": (w.width * w.height
                                     * (args.total_spp if strong else w.spp_per_step * world)),
                "samples_per_step_per_gpu": w.width * w.height * m["passes_this_rank_per_step"],
                "total_spp_per_step": args.total_spp,
                "parallelism": parallelism,
            },
            "rccl_ranks": rccl_ranks,
            "rank_devices": rank_devices,
            "mrays_per_s": (m["closest"] + m["shadow"]) / m["elapsed"] / 1e6,
            "rays_per_sample": (m["closest"] + m["shadow"]) / max(m["samples"], 1.0),
            "nan_pixels": m["nan_pixels"],  # the shader itself yields NaN for edge-on light samples
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        m = None
        if (world == 1 and not args.no_other_scenes and w.scene == DEFAULT_SCENE
                and args.total_spp is None):
            line["other_scenes"] = [other_scene_line(env, args, scene, seconds)
                                    for scene, seconds in OTHER_SCENES]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Copies the profile sets that tools/profile_round.sh left under gpurun_out/ into
profiles/<round>/ and rebuilds profiles/<round>/traffic.json from their PMC passes.

usage: python tools/collect_profiles.py r03        (expects gpurun_out/r03{h,k,s,m}_*)"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = {  # letter -> bench workload
    "h": dict(scene="hyperion_rect_light", width=1280, height=720, spp_per_step=512),
    "k": dict(scene="synthetic_4k", width=3840, height=2160, spp_per_step=64),
    "s": dict(scene="staircase_reduced", width=1280, height=720, spp_per_step=512),
    "m": dict(scene="mustang_red_reduced", width=1280, height=720, spp_per_step=512),
}
NOTE = ("profiles/{rnd}/{letter}_pmc.txt (tools/pmc_passes.sh: one rocprofv3 --pmc pass per counter "
        "group over `bench.py --steps 2 --warmup 1 --no-work-counters`, i.e. 6 render calls, the "
        "first of them the base-line call of the policy measurement; FETCH_SIZE in KB doubled "
        "for gfx950, WRITE_SIZE as reported; averages over every launch, including those that "
        "return at once after the hand-over)")


def main():
    rnd = sys.argv[1]
    out = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(out, exist_ok=True)
    entries = []
    for letter, workload in SETS.items():
        prefix = os.path.join(ROOT, "gpurun_out", rnd + letter)
        if not os.path.exists(prefix + "_pmc.txt"):
            print("no profile set", prefix)
            continue
        for name in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "pmc.txt"):
            shutil.copy(f"{prefix}_{name}", os.path.join(out, f"{letter}_{name}"))
        args = [sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), prefix + "_pmc",
                NOTE.format(rnd=rnd, letter=letter)]
        args += [f"{k}={v}" for k, v in workload.items()]
        args += ["max_depth=7", "pipeline=wavefront", "accel=rebuilt"]
        entries.append(json.loads(subprocess.check_output(args)))
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump(entries, f, indent=1)
        f.write("\n")
    print("wrote", len(entries), "entries")


if __name__ == "__main__":
    main()

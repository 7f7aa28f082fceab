"""Turns the PMC passes of tools/pmc_passes.sh into one entry of profiles/rNN/traffic.json:
HBM bytes per launch of every kernel (FETCH_SIZE is reported in kilobytes and counts 64 B per
128-B request on gfx950, so it is doubled; WRITE_SIZE as reported) and the issue-side view of
the trace and shade kernels (share of wave cycles with a VALU instruction issuing, waiting for
anything, waiting for instruction issue).

usage: python tools/pmc_summary.py <gpurun_out prefix of the passes> <source note> key=value...
       (keys: scene width height max_depth spp_per_step pipeline accel)"""
import collections
import csv
import glob
import json
import re
import sys


def main():
    prefix, source = sys.argv[1], sys.argv[2]
    entry = {}
    for kv in sys.argv[3:]:
        k, v = kv.split("=", 1)
        entry[k] = int(v) if v.isdigit() else v
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in sorted(glob.glob(prefix + "_p*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(path_trace_kernel|wf_[a-z_]+_kernel)", row["Kernel_Name"])
            if m:
                acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[m.group(1)][row["Counter_Name"]] += 1
    bytes_per_launch, issue = {}, {}
    for k, c in sorted(acc.items()):
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch = c["FETCH_SIZE"] * 1024.0 * 2.0 / launches[k]["FETCH_SIZE"]
            write = c["WRITE_SIZE"] * 1024.0 / launches[k]["WRITE_SIZE"]
            bytes_per_launch[k] = fetch + write
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0 and k in (
                "wf_extend_kernel", "wf_shade_kernel", "wf_settle_kernel", "path_trace_kernel"):
            # per-launch averages (some counters are collected in two passes)
            a = {name: value / launches[k][name] for name, value in c.items()}
            wc = a["SQ_WAVE_CYCLES"]
            simds = 1024.0  # 256 CUs x 4
            issue[k] = {
                # share of a wave's resident time during which one of its VALU instructions
                # executes; times the resident waves per SIMD = how busy the SIMD's VALU is
                "valu_active_of_wave_cycles": a.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                "waiting_any_of_wave_cycles": a.get("SQ_WAIT_ANY", 0.0) / wc,
                "waiting_for_issue_of_wave_cycles": a.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                "lds_active_of_wave_cycles": a.get("SQ_ACTIVE_INST_LDS", 0.0) / wc,
                # SQ_ACTIVE_INST_VALU counts 4-cycle units summed over the SIMDs;
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                "valu_busy_of_kernel_time": (a.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / simds)
                                            / (a["GRBM_GUI_ACTIVE"] / 8.0)
                                            if a.get("GRBM_GUI_ACTIVE") else None,
                "valu_lanes_in_use": a.get("SQ_THREAD_CYCLES_VALU", 0.0)
                                     / (a["SQ_ACTIVE_INST_VALU"] * 64.0)
                                     if a.get("SQ_ACTIVE_INST_VALU") else None,
                "valu_instructions_per_launch": a.get("SQ_INSTS_VALU", 0.0),
                "salu_instructions_per_launch": a.get("SQ_INSTS_SALU", 0.0),
            }
    entry["source"] = source
    entry["bytes_per_launch"] = bytes_per_launch
    entry["issue"] = issue
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()

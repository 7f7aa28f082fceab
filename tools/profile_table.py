"""Prints the figures of profiles/<round>/ that DESIGN.md's measurement table quotes.
usage: python tools/profile_table.py r03"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = os.path.join(ROOT, "profiles", sys.argv[1])
    for x in "hmskct":
        d = json.load(open(os.path.join(rnd, f"{x}_bench.json")))
        r = d["roofline"]
        lanes, per_ray = r["lanes_in_use"], r["per_ray"]
        print(x, "value %.1f Mrays %.0f ms %.1f trace/shade %.1f / %.1f" % (
            d["value"], d["mrays_per_s"], d["ms_per_step"], r["extend_ms_per_call"],
            r["shade_ms_per_call"]))
        print("   per ray %.2f %.2f %.2f lanes %.2f %.2f %.2f" % (
            per_ray["node_fetches"], per_ray["triangle_tests"], per_ray["instance_records"],
            lanes["node_steps"], lanes["triangle_tests"], lanes["instance_entries"]))
        print("   %s achieved %.0f frac %.3f (l2 %.3f) lds %.3f %.0f GB/s avg %.3f / %.3f issue %s traffic %s" % (
            r["bound"], r["achieved"], r["frac"], r["frac_of_each_level"]["l2"],
            r["lds_served_share_of_node_fetches"], r["lds_served_gbs"], r["kernel_avg_ms"],
            r["kernel_avg_ms_wall"], r.get("issue_frac"), r.get("traffic")))
        shade = r.get("shade") or {}
        print("   cpu %.2f shade traffic %s frac %s other %s" % (
            d["cpu_baseline"]["value"], shade.get("traffic_bytes_per_call"),
            shade.get("traffic_frac_of_hbm_peak"),
            [(o["scene"], round(o["value"])) for o in d.get("other_scenes", [])]))
        stats = os.path.join(rnd, f"{x}_kernel_stats.csv")
        if os.path.exists(stats):
            rows = [row for row in csv.DictReader(open(stats)) if "wf_extend_kernel" in row["Name"]]
            total = sum(float(row["TotalDurationNs"]) for row in rows)
            calls = sum(int(row["Calls"]) for row in rows)
            u = json.load(open(os.path.join(rnd, f"{x}_bench_under_rocprof.json")))
            print("   rocprofv3 avg %.3f ms over %d launches; events of that run %.3f; value %.1f" % (
                total / calls / 1e6, calls, u["roofline"]["kernel_avg_ms_wall"], u["value"]))
    for e in json.load(open(os.path.join(rnd, "traffic.json"))):
        i = e["issue"]
        print(e["scene"], " ".join("%s %.2f x %.2f" % (
            k.split("_")[1], i[k]["valu_busy_of_kernel_time"], i[k]["valu_lanes_in_use"])
            for k in ("wf_extend_kernel", "wf_shade_kernel", "wf_settle_kernel") if k in i))


if __name__ == "__main__":
    main()

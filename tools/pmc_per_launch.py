"""Per-launch view of one PMC pass (tools/pmc_one.sh): one line per dispatch of the kernels whose
name contains the pattern, in dispatch order, with every counter of the pass.

usage: python tools/pmc_per_launch.py <gpurun_out/prefix> <pattern> [first] [count]"""
import collections
import csv
import glob
import sys


def main():
    prefix, pattern = sys.argv[1], sys.argv[2]
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
    rows = collections.OrderedDict()
    for f in glob.glob(prefix + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pattern in row["Kernel_Name"]:
                key = int(row["Dispatch_Id"])
                rows.setdefault(key, {"name": row["Kernel_Name"].split("(")[0][-40:]})
                rows[key][row["Counter_Name"]] = float(row["Counter_Value"])
    for n, (key, r) in enumerate(sorted(rows.items())):
        if first <= n < first + count:
            print(key, r.pop("name"), " ".join("%s=%.4g" % kv for kv in sorted(r.items())))


if __name__ == "__main__":
    main()

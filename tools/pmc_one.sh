#!/bin/bash
# One rocprofv3 PMC pass over a short bench run, summed per kernel: tools/pmc_one.sh <out-prefix> "<counters>" [bench args]
OUT=$1; CTRS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/${OUT} -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-work-counters --no-other-scenes --steps 2 --warmup 1 "$@" > $ROOT/gpurun_out/${OUT}.log 2>&1 || { echo "pass failed"; tail -5 $ROOT/gpurun_out/${OUT}.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$ROOT/gpurun_out/${OUT}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(path_trace_kernel|wf_[a-z_]+_kernel)(<[^>]*>)?", row["Kernel_Name"])
        if not m: continue
        acc[m.group(0)][row["Counter_Name"]] += float(row["Counter_Value"]); n[(m.group(0), row["Counter_Name"])] += 1
for k in sorted(acc):
    print(k, " ".join("%s=%.4g(n=%d)" % (c, v, n[(k, c)]) for c, v in sorted(acc[k].items())))
PY

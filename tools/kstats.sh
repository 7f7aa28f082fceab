#!/bin/bash
# rocprofv3 kernel statistics of one short bench run: tools/kstats.sh <out-prefix> [bench args]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${OUT}_stats -o s -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-scenes --no-work-counters --steps 6 "$@" > $ROOT/gpurun_out/${OUT}.json 2> $ROOT/gpurun_out/${OUT}.err || exit 1
python3 - <<PY
import csv,glob
for f in glob.glob("$ROOT/gpurun_out/${OUT}_stats/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        n=row["Name"]; 
        if float(row["Percentage"])<0.3: continue
        print("%-70s calls %6s total %9.2f ms avg %8.1f us  %5s%%" % (n[:70], row["Calls"], float(row["TotalDurationNs"])/1e6, float(row["AverageNs"])/1e3, row["Percentage"]))
PY

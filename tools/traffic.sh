#!/bin/bash
# HBM traffic per kernel (rocprofv3 PMC FETCH_SIZE / WRITE_SIZE, one pass each): tools/traffic.sh <out-prefix> [bench args]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/${OUT}_$C -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-scenes --no-work-counters --steps 2 --warmup 1 "$@" > $ROOT/gpurun_out/${OUT}_$C.log 2>&1 || { echo "pass $C failed"; tail -5 $ROOT/gpurun_out/${OUT}_$C.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$ROOT/gpurun_out/${OUT}_%s/*counter_collection.csv" % c):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(path_trace_kernel|wf_[a-z_]+_kernel)(<[^>]*>)?", row["Kernel_Name"])
            if not m or row["Counter_Name"] != c: continue
            k = m.group(0)
            # rocprofv3 reports kilobytes; gfx950 FETCH_SIZE counts 64 B per 128-B request
            tot[k][c] += float(row["Counter_Value"]) * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
            n[(k, c)] += 1
calls = 6.0  # render calls in the profiled run: 3 set-up + 1 warm-up + 2 timed
for k in sorted(tot):
    print("%-40s launches %4d  read %8.3f GB/call  written %8.3f GB/call" % (k, n[(k, "FETCH_SIZE")], tot[k]["FETCH_SIZE"] / calls / 1e9, tot[k]["WRITE_SIZE"] / calls / 1e9))
PY

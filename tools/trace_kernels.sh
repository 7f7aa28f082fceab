#!/bin/bash
# Per-launch kernel durations of one bench step (rocprofv3 --kernel-trace), in launch order.
# usage: tools/trace_kernels.sh <out-name> [bench args]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/$OUT -o t -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-scenes --steps 1 --warmup 1 "$@" > $ROOT/gpurun_out/$OUT.log 2>&1 || { tail -5 $ROOT/gpurun_out/$OUT.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$ROOT/gpurun_out/$OUT/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "wf_" in r["Kernel_Name"] or "path_trace" in r["Kernel_Name"]]
half = rows[len(rows) // 2:]
t0 = int(half[0]["Start_Timestamp"])
for r in half:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %9.1f us  dur %9.1f us  grid %s" % (r["Kernel_Name"].split("(")[0][-28:], (s - t0) / 1e3, (e - s) / 1e3, r["Grid_Size_X"]))
PY

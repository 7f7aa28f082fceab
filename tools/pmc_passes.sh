#!/bin/bash
# Collects rocprofv3 PMC counters for the bench kernel, one pass per run (the hardware cannot
# split counters over passes by itself).  usage: tools/pmc_passes.sh <out-prefix> [bench args]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32" \
  "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/${OUT}_p$i -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-scenes --no-work-counters --steps 2 --warmup 1 "$@" > $ROOT/gpurun_out/${OUT}_p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $ROOT/gpurun_out/${OUT}_p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, re
# per kernel (short name): counter -> sum over the launches of the last profiled step
for d in sorted(glob.glob("$ROOT/gpurun_out/${OUT}_p*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for row in csv.DictReader(open(f)):
            m = re.search(r"(path_trace_kernel|wf_[a-z_]+_kernel)", row["Kernel_Name"])
            if not m:
                continue
            acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
            n[(m.group(1), row["Counter_Name"])] += 1
        for k in sorted(acc):
            print(k, " ".join("%s=%.4g(n=%d)" % (c, v, n[(k, c)]) for c, v in sorted(acc[k].items())))
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                if c in acc[k]:
                    # rocprofv3 reports kilobytes; gfx950 FETCH_SIZE counts 64 B per 128-B request
                    per_launch = acc[k][c] * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0) / n[(k, c)]
                    print("  %s %s bytes per launch (corrected) = %.4g" % (k, c, per_launch))
PY

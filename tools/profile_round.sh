#!/bin/bash
# The profile set of one build: bench line, rocprofv3 kernel statistics of the same command,
# and the PMC passes (tools/pmc_passes.sh).  usage: tools/profile_round.sh <prefix> [bench args]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python3 $ROOT/bench.py "$@" > $ROOT/gpurun_out/${OUT}_bench.json 2> $ROOT/gpurun_out/${OUT}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
# (40 timed steps: the first set-up call traces in fixed chunks for the policy measurement and
# would otherwise weigh 1/15 in the kernel's average)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${OUT}_stats -o s -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-scenes --steps 40 "$@" > $ROOT/gpurun_out/${OUT}_bench_under_rocprof.json 2> $ROOT/gpurun_out/${OUT}_stats.err || exit 1
cp $ROOT/gpurun_out/${OUT}_stats/*kernel_stats.csv $ROOT/gpurun_out/${OUT}_kernel_stats.csv || exit 1
cd $ROOT && bash tools/pmc_passes.sh ${OUT}_pmc "$@" > $ROOT/gpurun_out/${OUT}_pmc.txt 2>&1

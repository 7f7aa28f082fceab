#!/bin/bash
# A/B of library builds on the bench, timing only, with one environment: tools/ab3.sh "<ENV=..>" "<bench args>" lib1.so lib2.so ...
ENVS=$1; ARGS=$2; shift; shift
for l in "$@"; do
  env $ENVS RT_HIP_LIB=$PWD/raytracing_amd/lib/$l timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-scenes --no-work-counters --steps 6 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$l', '$ARGS', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms')" || exit 1
done

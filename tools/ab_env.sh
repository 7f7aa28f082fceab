#!/bin/bash
# A/B of environment settings on the bench: tools/ab_env.sh "<bench args>" "A=1" "A=2 B=3" ...
ARGS=$1; shift
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-scenes --no-work-counters --steps 6 $ARGS 2>>gpurun_out/ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$e', '$ARGS', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms')" || exit 1
done

#!/bin/bash
# A/B of environment settings on the bench: tools/ab_env.sh "<scene args>" "VAR=1" "VAR=0" "" ...
ARGS=$1; shift
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-scenes --steps 6 $ARGS 2>>gpurun_out/ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('[$e]', '$ARGS', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms',
      'refill', r.get('trace_refill'), r.get('trace_refill_probe_ms'), r.get('trace_refill_bounces'), 'binning', r.get('ray_binning'))" || exit 1
done

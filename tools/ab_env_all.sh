#!/bin/bash
# A/B of environment settings on the four heavy bench scenes: tools/ab_env_all.sh "A=1" "A=2 B=3" ...
for s in "" "--scene mustang_red_reduced" "--scene staircase_reduced" "--scene synthetic_4k"; do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-scenes --steps 6 $s 2>>gpurun_out/ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('$e', '$s', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms; trace', round(r.get('extend_ms_per_call',0),2), 'shade', round(r.get('shade_ms_per_call',0),2), 'deepest', r.get('deepest_traversal_stack'), 'lds share', round(r.get('lds_served_share_of_node_fetches',0),3), 'nodes in lds', r['structure_bytes']['nodes_in_lds']//128, 'of', r['structure_bytes']['nodes']//128)" || exit 1
  done
done

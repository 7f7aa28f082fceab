import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1], round(d['value'],1), json.dumps(r['lanes_in_use']), json.dumps(r['per_ray']))

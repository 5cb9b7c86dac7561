#!/bin/bash
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if k!="workload" and not isinstance(v,(dict,list))})'
for i in 1 2 3; do
  echo -n "new "; python tools/mirror_bench.py 400 2>/dev/null | python -c "$P"
  echo -n "old "; (cd _old && python tools/mirror_bench.py 400 2>/dev/null | python -c "$P")
done

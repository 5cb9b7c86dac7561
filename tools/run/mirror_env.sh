#!/bin/bash
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["us_per_verify_step"],1))'
for i in 1 2 3; do
  echo -n "plain "; python tools/mirror_bench.py 300 2>/dev/null | python -c "$P"
  echo -n "hwq8+poll "; GPU_MAX_HW_QUEUES=8 HSA_ENABLE_INTERRUPT=0 python tools/mirror_bench.py 300 2>/dev/null | python -c "$P"
  echo -n "poll "; HSA_ENABLE_INTERRUPT=0 python tools/mirror_bench.py 300 2>/dev/null | python -c "$P"
  echo -n "hwq8 "; GPU_MAX_HW_QUEUES=8 python tools/mirror_bench.py 300 2>/dev/null | python -c "$P"
done

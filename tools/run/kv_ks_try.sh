#!/bin/bash
# the KV / hidden commit with KS slabs per workgroup (LANTERN_KV_KS; 0 = the per-slab kernel): parity tests, then the step at each setting
O=gpurun_out/kvks
mkdir -p $O
export LANTERN_STEP_TURNS=0
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_loop.py tests/test_gpu_fullsize_properties.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for ks in 0 1 2 4 8 0 4; do
  timeout -k 10 400 python3 bench.py --tuning kv_ks=$ks --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/ks$ks.json 2> $O/ks$ks.err || { tail -5 $O/ks$ks.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$O/ks$ks.json").read().strip().splitlines()[-1])
print("KS $ks:", round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,3), "M tok/s", round(d["roofline"]["avg_launch_ms"]*1e3,1), {k: round(v["avg_launch_ms"]*1e3,1) for k,v in d["kernels"].items()})
PY
done

#!/bin/bash
# full GPU suite, then the saturating-batch profiles (rocprofv3 stats + PMC at 4096 and 512), then the driver's bench invocation
O=gpurun_out/r04a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for B in 4096 512; do
  mkdir -p gpurun_out/r04_sw$B && cp gpurun_out/commit.txt gpurun_out/r04_sw$B/commit.txt
  bash tools/run/ep_sweep_prof.sh r04_sw$B $B 5 > $O/sw$B.txt 2>&1 || { tail -20 $O/sw$B.txt; exit 1; }
  head -4 gpurun_out/r04_sw$B/summary.txt
done
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("saturating"))
print({k: (v.get("ms_per_step"), v.get("value")) for k, v in d.get("configs", {}).items() if isinstance(v, dict)})
PY

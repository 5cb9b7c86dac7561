#!/bin/bash
O=gpurun_out/turns
mkdir -p $O
for cfg in "0 63 3" "2 63 3" "2 64 4" "0 63 3" "2 63 3"; do
  set -- $cfg
  LANTERN_STEP_TURNS=$1 timeout -k 10 400 python3 bench.py --gpus 1 --steps 200 --warmup 20 --seqs-per-gpu $2 --groups $3 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/t$1s$2g$3.json 2> $O/t$1s$2g$3.err || { tail -5 $O/t$1s$2g$3.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$O/t$1s$2g$3.json").read().strip().splitlines()[-1])
print("turns $1 seqs $2 groups $3:", round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,3), "M tok/s")
PY
done
cd /tmp && export TMPDIR=/tmp
LANTERN_STEP_TURNS=2 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 200 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras > $GRAFT_REPO_ROOT/$O/prof2.json 2> $GRAFT_REPO_ROOT/$O/prof2.err
echo done

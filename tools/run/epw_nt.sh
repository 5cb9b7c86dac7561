#!/bin/bash
# chain kernel at 256 / 512 / 1024 threads per sequence (probability rows, one group, 64 sequences): the evaluate_posterior launch time
cd $GRAFT_REPO_ROOT
O=gpurun_out/nt
mkdir -p $O
for nt in 512 256 1024; do
  LANTERN_EPW_NT=$nt timeout -k 10 300 python3 bench.py --ep chain --no-fuse-o7 --groups 1 --steps 60 --warmup 10 --no-extras --cpu-seconds 0 --ep-sweep "" > $O/b_$nt.json 2> $O/b_$nt.err || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/nt/b_*.json")):
    d=json.loads(open(f).read().strip().split("\n")[-1])
    print(f, round(d["value"]), d["ms_per_step"], json.dumps(d.get("kernels"))[:600])
PY

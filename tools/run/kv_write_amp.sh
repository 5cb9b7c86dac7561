#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of update_inputs_kernel with non-temporal (variant 20, default) and plain (22) stores, and with the hidden copy's share
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kvamp
mkdir -p $O
for v in 20 22 21; do
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/v${v}_$c -o p -- python3 bench.py --tuning kv_variant=$v --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/v${v}_$c.json 2> $O/v${v}_$c.err || { tail -5 $O/v${v}_$c.err; exit 1; }
    echo "variant $v $c"; python3 tools/pmc_sum.py $O/v${v}_$c update_inputs_kernel
  done
  python3 - <<PY
import json
d=json.loads(open("$O/v${v}_WRITE_SIZE.json").read().strip().splitlines()[-1])
k=d["kernels"]["kv_gather"]
print("variant $v moved bytes/launch", k["algorithmic_bytes_per_launch"], "launch us", round(k["avg_launch_ms"]*1e3,1))
PY
done

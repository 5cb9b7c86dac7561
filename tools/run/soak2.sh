#!/bin/bash
# wider randomized soaks (static / dynamic kernel sets, node kernels) + long step loops of the optional launch paths against the CPU oracle
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2soak2}
mkdir -p $OUT
timeout -k 10 400 python tests/fuzz_soak.py 60 > $OUT/static_soak.txt 2>&1; tail -1 $OUT/static_soak.txt
timeout -k 10 400 python tests/fuzz_soak.py 60 dynamic > $OUT/dynamic_soak.txt 2>&1; tail -1 $OUT/dynamic_soak.txt
timeout -k 10 400 python tests/fuzz_soak.py 80 nodes > $OUT/nodes_soak.txt 2>&1; tail -1 $OUT/nodes_soak.txt
# 1000-step loops with KV slabs (2560 rows: a whole 768x768 image fits), 32 sequences: single-launch accept, worker-thread launches
python - <<PY
import json
for n in ("fused_accept","launcher"):
    try:
        d=json.loads(open("$OUT/soak_%s.json"%n).read().strip().splitlines()[-1]); c=d["cpu_baseline"]
        print(n, round(d["value"]), d["steps"], c["matches_gpu_token_stream"], c["sample"])
    except Exception as e: print(n, "ERR", e)
PY

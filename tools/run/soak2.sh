#!/bin/bash
# wider randomized soaks (static / dynamic kernel sets, node kernels) + long step loops of the optional launch paths against the CPU oracle
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2soak2}
mkdir -p $OUT
timeout -k 10 400 python tests/fuzz_soak.py 60 > $OUT/static_soak.txt 2>&1; tail -1 $OUT/static_soak.txt
timeout -k 10 400 python tests/fuzz_soak.py 60 dynamic > $OUT/dynamic_soak.txt 2>&1; tail -1 $OUT/dynamic_soak.txt
timeout -k 10 400 python tests/fuzz_soak.py 80 nodes > $OUT/nodes_soak.txt 2>&1; tail -1 $OUT/nodes_soak.txt
# 1000-step loops with KV slabs (2560 rows: a whole 768x768 image fits), 32 sequences: single-launch accept, worker-thread launches
timeout -k 10 400 python bench.py --steps 1000 --warmup 20 --seqs-per-gpu 32 --kv-smax 2560 --pool-steps 8 --cpu-seconds 100 --ep-sweep "" --no-extras --no-events --groups 2 --fused-accept > $OUT/soak_fused_accept.json 2> $OUT/soak_fused_accept.err || tail -3 $OUT/soak_fused_accept.err
timeout -k 10 400 python bench.py --steps 1000 --warmup 20 --seqs-per-gpu 32 --kv-smax 2560 --pool-steps 8 --cpu-seconds 100 --ep-sweep "" --no-extras --no-events --groups 4 --launch-threads 4 > $OUT/soak_launcher.json 2> $OUT/soak_launcher.err || tail -3 $OUT/soak_launcher.err
python - <<PY
import json
for n in ("fused_accept","launcher"):
    try:
        d=json.loads(open("$OUT/soak_%s.json"%n).read().strip().splitlines()[-1]); c=d["cpu_baseline"]
        print(n, round(d["value"]), d["steps"], c["matches_gpu_token_stream"], c["sample"])
    except Exception as e: print(n, "ERR", e)
PY

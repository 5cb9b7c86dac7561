#!/bin/bash
# randomized / long-run parity soaks of the round-2 paths (output under gpurun_out/<tag>/)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2soak}
mkdir -p $OUT
timeout -k 10 500 python tests/fuzz_soak.py ${2:-30} nodes > $OUT/nodes_soak.txt 2>&1; tail -3 $OUT/nodes_soak.txt
# the default step loop (raw rows, 3 groups) over 1400 steps of 48 sequences against the CPU oracle's loop
timeout -k 10 400 python bench.py --no-kv --steps 1400 --warmup 20 --seqs-per-gpu 48 --cpu-seconds 120 --ep-sweep "" --no-extras --no-events > $OUT/soak_default.json 2> $OUT/soak_default.err || tail -3 $OUT/soak_default.err
timeout -k 10 400 python bench.py --no-kv --steps 1400 --warmup 20 --seqs-per-gpu 48 --cpu-seconds 120 --ep-sweep "" --no-extras --no-events --ep nodes --no-fuse-o7 --spec-rows 0 --groups 2 > $OUT/soak_nodes.json 2> $OUT/soak_nodes.err || tail -3 $OUT/soak_nodes.err
python - <<PY
import json
for n in ("default","nodes"):
    try:
        d=json.loads(open("$OUT/soak_%s.json"%n).read().strip().splitlines()[-1]); c=d["cpu_baseline"]
        print(n, round(d["value"]), d["steps"], c["matches_gpu_token_stream"], c["sample"])
    except Exception as e: print(n, "ERR", e)
PY

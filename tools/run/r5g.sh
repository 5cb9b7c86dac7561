#!/bin/bash
O=gpurun_out/${1:-r5g}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_loop.py tests/test_gpu_window.py tests/test_gpu_fuzz.py tests/test_gpu_nodes.py -m gpu -q -x > $O/test.log 2>&1; echo "tests rc=$?"; tail -3 $O/test.log
timeout -k 10 300 python tools/ep_sweep.py 512,4096 24 chain > $O/sweep.json 2> $O/sweep.err || tail -5 $O/sweep.err
python - <<PY
import json
d=json.loads(open("$O/sweep.json").read().strip().splitlines()[-1])
for r in d["sweep"]:
    c=r["chain"]; print(r["sequences_per_launch"], "launch_ms", round(c["launch_ms"],4), "needed MB", round(c["hbm_bytes_needed_per_launch"]/1e6,1), "frac", round(c["frac"],3), "per-rejection MB", round(c["bytes_if_every_rejection_read_its_row_from_hbm"]/1e6,1))
PY
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out "" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"

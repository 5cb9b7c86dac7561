#!/bin/bash
O=gpurun_out/${1:-r5k}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_loop.py tests/test_gpu_configs.py tests/test_gpu_window.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "throughput or two_workgroups or fullsize or harness_loop or grouping" > $O/test.log 2>&1; echo "tests rc=$?"; tail -5 $O/test.log
for v in 1 0; do
LANTERN_EPW_WALKER=$v timeout -k 10 300 python tools/ep_sweep.py 1024,4096 24 chain > $O/sweep_$v.json 2> $O/sweep_$v.err || tail -5 $O/sweep_$v.err
python - <<PY
import json
d=json.loads(open("$O/sweep_$v.json").read().strip().splitlines()[-1])
for r in d["sweep"]:
    c=r["chain"]; print("WALKER=$v", r["sequences_per_launch"], "launch_us", round(1e3*c["launch_ms"],1), "b2b", round(1e3*c["back_to_back_ms"],1), "needed MB", round(c["hbm_bytes_needed_per_launch"]/1e6,1), "frac", round(c["frac"],3))
PY
done

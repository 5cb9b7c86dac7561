#!/bin/bash
# throughput forms of the chain kernel at the saturating batch (rotating inputs): LANTERN_EPW_TP 0 (generic two-per-CU) / 1 / 2 / 3
O=gpurun_out/tp_try
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_window.py tests/test_gpu_configs.py tests/test_gpu_loop.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for tp in 1 2 3; do
  LANTERN_EPW_TP=$tp python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "two_workgroups" > $O/test_tp$tp.txt 2>&1 || { tail -30 $O/test_tp$tp.txt; exit 1; }
  tail -1 $O/test_tp$tp.txt
  LANTERN_EPW_TP=$tp timeout -k 10 400 python3 tools/ep_sweep.py 64,512,4096 24 chain > $O/tp$tp.json 2> $O/tp$tp.err || { tail -20 $O/tp$tp.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$O/tp$tp.json"))
for r in d["sweep"]:
    c=r.get("chain")
    print("tp=$tp", r["sequences_per_launch"], r.get("rotation_sets"), c and (round(c["launch_ms"]*1e3,1), round(c["back_to_back_ms"]*1e3,1), round(c["frac"],3)), r.get("skipped"))
PY
done
LANTERN_EPW_SPEC=0 timeout -k 10 400 python3 tools/ep_sweep.py 512,4096 24 chain > $O/tp0.json 2> $O/tp0.err
python3 - <<PY
import json
d=json.load(open("$O/tp0.json"))
for r in d["sweep"]:
    c=r.get("chain")
    print("generic", r["sequences_per_launch"], c and (round(c["launch_ms"]*1e3,1), round(c["back_to_back_ms"]*1e3,1), round(c["frac"],3)), r.get("skipped"))
PY

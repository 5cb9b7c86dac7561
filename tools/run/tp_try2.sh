#!/bin/bash
O=gpurun_out/tp_try2
mkdir -p $O
for tp in $@; do
  LANTERN_EPW_TP=$tp python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "two_workgroups" > $O/test_tp$tp.txt 2>&1 || { tail -30 $O/test_tp$tp.txt; exit 1; }
  tail -1 $O/test_tp$tp.txt
  LANTERN_EPW_TP=$tp timeout -k 10 400 python3 tools/ep_sweep.py 512,4096 24 chain > $O/tp$tp.json 2> $O/tp$tp.err || { tail -20 $O/tp$tp.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$O/tp$tp.json"))
for r in d["sweep"]:
    c=r.get("chain")
    print("tp=$tp", r["sequences_per_launch"], r.get("rotation_sets"), c and (round(c["launch_ms"]*1e3,1), round(c["back_to_back_ms"]*1e3,1), round(c["frac"],3)), r.get("skipped"))
PY
done

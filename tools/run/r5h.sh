#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
LANTERN_EPW_OCC2=1 timeout -k 10 500 python tools/ep_sweep.py 64,256,512,768,1536,3072 24 chain > $O/sweep_occ.json 2> $O/sweep.err || tail -5 $O/sweep.err
python - <<PY
import json
d=json.loads(open("$O/sweep_occ.json").read().strip().splitlines()[-1])
for r in d["sweep"]:
    c=r["chain"]; print(r["sequences_per_launch"], "launch_us", round(1e3*c["launch_ms"],1), "b2b_us", round(1e3*c["back_to_back_ms"],1), "frac", round(c["frac"],3))
PY

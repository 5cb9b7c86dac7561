#!/bin/bash
O=gpurun_out/${1:-r5l}; mkdir -p $O
LANTERN_EPW_WALKER=0 timeout -k 10 300 python tools/ep_sweep.py 1024,4096 24 chain > $O/sweep.json 2> $O/sweep.err || tail -5 $O/sweep.err
python - <<PY
import json
d=json.loads(open("$O/sweep.json").read().strip().splitlines()[-1])
for r in d["sweep"]:
    c=r["chain"]; print(r["sequences_per_launch"], "launch_us", round(1e3*c["launch_ms"],1), "b2b", round(1e3*c["back_to_back_ms"],1), "needed MB", round(c["hbm_bytes_needed_per_launch"]/1e6,1), "frac", round(c["frac"],3))
PY

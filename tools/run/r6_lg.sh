#!/bin/bash
# LlamaGen's two-per-CU throughput instance against the generic one-per-CU instance (tools/lg_sweep.py)
O=gpurun_out/r6_lg
mkdir -p $O
timeout -k 10 500 python3 tools/lg_sweep.py ${1:-512,2048} ${2:-10} > $O/lg.json 2> $O/lg.err || { tail -20 $O/lg.err; exit 1; }
python3 - <<PY
import json
for l in open("$O/lg.json"):
    d=json.loads(l)
    if "lg_sweep" in d: continue
    print(d["sequences_per_launch"], {k:[(round(x["launch_us"],1), round(x["frac"],3)) for x in v] for k,v in d["variants"].items()}, "o7", round(d["variants"]["1"][0]["cfg_mask_topk_us"],1), round(d["variants"]["1"][0]["cfg_mask_topk_frac"],3))
PY

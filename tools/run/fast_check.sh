#!/bin/bash
# fast-walk kernel: phase traces + bench comparisons against the chain kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/fast
mkdir -p $O
EPW_B=64 EPW_MODE=chain timeout -k 10 200 python3 tools/epf_trace.py > $O/trace_probs64.txt 2>&1 &&
EPW_B=21 EPW_MODE=raw timeout -k 10 200 python3 tools/epf_trace.py > $O/trace_raw21.txt 2>&1 &&
for ep in chain fast; do
  timeout -k 10 300 python3 bench.py --ep $ep --no-fuse-o7 --groups 1 --steps 60 --warmup 10 --no-extras --cpu-seconds 0 --ep-sweep "" > $O/b_unfused_g1_$ep.json 2> $O/b_unfused_g1_$ep.err || exit 1
  timeout -k 10 300 python3 bench.py --ep $ep --steps 100 --warmup 20 --no-extras --cpu-seconds 0 --ep-sweep "" > $O/b_default_$ep.json 2> $O/b_default_$ep.err || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/fast/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1])
        ks=d.get("kernels",{})
        print(f, d["value"], d["ms_per_step"], {k:(round(v.get("avg_us",0),1) if isinstance(v,dict) else v) for k,v in ks.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -12 $O/trace_probs64.txt

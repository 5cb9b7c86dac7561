#!/bin/bash
# the driver's invocation of bench.py + a longer one (scratch output under gpurun_out/)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2s}
mkdir -p $OUT
timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err || tail -5 $OUT/bench_20_5.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_20_5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["seqs_per_gpu"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], {a:(round(b["avg_launch_ms"]*1e3,1), round(b["frac"],3)) for a,b in d["kernels"].items()})
for k,v in d["per_kernel_single_group"].items(): print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["avg_launch_ms"], {a:(round(b["avg_launch_ms"]*1e3,1), round(b["frac"],3)) for a,b in v["kernels"].items()})
print(d["other_groupings"]); print(d["lambda_mode"]); print(d["dynamic_tree"]["value"], d["dynamic_tree"]["ms_per_step"]); print(d["step_latency_us"]); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["matches_gpu_token_stream"])
for r in d.get("ep_batch_sweep", []): print(r["sequences_per_launch"], {k: (round(v["launch_ms"]*1e3,1), round(v["frac"],3)) for k,v in r.items() if isinstance(v, dict)})
PY

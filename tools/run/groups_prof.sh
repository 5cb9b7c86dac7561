#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command at 1 / 2 / 4 stream groups: the walk kernel's mean launch time by sequences per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 1 2 4; do
  O=gpurun_out/gprof/g$g; mkdir -p $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" --groups $g > $O/line.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  rm -f $O/s_kernel_trace.csv $O/s_agent_info.csv
  python3 - <<PY
import csv, json
d = json.loads(open("$O/line.json").read().strip().splitlines()[-1]); rl = d["roofline"]
rows = [r for r in csv.DictReader(open("$O/s_kernel_stats.csv")) if "epw_kernel" in r["Name"]]
us = float(rows[0]["AverageNs"]) / 1e3
print(json.dumps({"groups": $g, "sequences_per_launch": rl["sequences_per_launch"], "value_under_rocprof": round(d["value"]), "contract_MB_per_launch": round(rl["algorithmic_bytes_per_launch"] / 1e6, 2),
                  "epw_live_us": round(1e3 * rl["avg_launch_ms"], 1), "epw_rocprof_mean_us": round(us, 1), "frac_live": round(rl["frac"], 4),
                  "frac_rocprof": round(rl["algorithmic_bytes_per_launch"] / (us * 1e-6) / 8e12, 4)}))
PY
done

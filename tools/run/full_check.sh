#!/bin/bash
# the round's end-to-end check on one GPU: the GPU test suite, then the driver's bench invocation with every extra object
cd $GRAFT_REPO_ROOT
O=gpurun_out/full
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/t_all.txt 2>&1 || { tail -40 $O/t_all.txt; exit 1; }
tail -2 $O/t_all.txt
timeout -k 10 1000 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -5 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/full/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], d["config"]["seqs_per_gpu"])
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","avg_launch_ms","frac","frac_needed","traffic","traffic_source")})
cb=d.get("cpu_baseline",{})
ac = cb.get("all_cores") or {}
print("cpu", cb.get("value"), cb.get("cores"), "1/seq:", cb.get("sequences_over_threads",{}).get("value"), cb.get("sequences_over_threads",{}).get("ms_per_seq_step"), "all:", ac.get("value"), cb.get("usable_cores"), cb.get("host_cores"), "single:", cb.get("single_thread",{}).get("value"), cb.get("single_thread",{}).get("ms_per_seq_step"), cb.get("matches_gpu_token_stream"))
c=d.get("configs",{})
print("C2", {k:c.get("C2",{}).get(k) for k in ("value","ms_per_step","mean_accept_length","kernel_ms","evaluate_posterior")})
for x in c.get("C4",[]): print("C4", x["lantern_delta"], x["lantern_k"], x["value"], x["ms_per_step"], x["mean_accept_length"], x["evaluate_posterior"])
print("dyn", {k:d.get("dynamic_tree",{}).get(k) for k in ("value","ms_per_step")})
PY

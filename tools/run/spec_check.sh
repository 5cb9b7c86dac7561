#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/spec
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_window.py tests/test_gpu_loop.py tests/test_gpu_parity.py tests/test_gpu_nodes.py tests/test_gpu_fuzz.py tests/test_gpu_generate_ref.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -40 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for sp in 2 1 0 2 1; do
  timeout -k 10 300 python3 bench.py --tuning epw_spec=$sp --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/b_$sp.json 2> $O/b_$sp.err || tail -3 $O/b_$sp.err
  python3 - <<PY
import json
d=json.loads(open("$O/b_$sp.json").read().strip().splitlines()[-1])
print("spec=$sp", round(d["value"]), round(1e3*d["ms_per_step"],2), "us/step; epw", round(1e3*d["roofline"]["avg_launch_ms"],2), "us;", {k:round(1e3*v["avg_launch_ms"],1) for k,v in d.get("kernels",{}).items() if isinstance(v,dict) and "avg_launch_ms" in v})
PY
done
for sp in 2 1 0; do
  timeout -k 10 300 python3 bench.py --tuning epw_spec=$sp --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --no-fuse-o7 --spec-rows 0 --groups 1 --seqs-per-gpu 64 > $O/c_$sp.json 2> $O/c_$sp.err || tail -3 $O/c_$sp.err
  python3 - <<PY
import json
d=json.loads(open("$O/c_$sp.json").read().strip().splitlines()[-1])
print("chain64 spec=$sp", round(d["value"]), round(1e3*d["ms_per_step"],2), "us/step; epw", round(1e3*d["roofline"]["avg_launch_ms"],2), "us")
PY
done

#!/bin/bash
# round 6's profile set: rocprofv3 stats + FETCH / WRITE passes of the bench command (default configuration: 64 sequences in 4 stream groups), the
# saturating-batch passes at 4096 and 512 sequences per launch (compact throughput instance), and the driver's invocation with its full report
O=gpurun_out/r06prof
mkdir -p $O
git rev-parse --short HEAD > tools/run/.commit 2>/dev/null || true
bash tools/run/prof_default.sh r06p > $O/prof_default.txt 2>&1 || { tail -20 $O/prof_default.txt; exit 1; }
tail -14 $O/prof_default.txt
python3 tools/prune_prof.py gpurun_out/r06p; rm -f gpurun_out/r06p/prof/*kernel_trace.csv
# the same command free-running (--commit-window 0): with turn-taking the walk kernel's duration includes its wait for the group's turn
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06p_free -o default -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" --commit-window 0 > $O/b_free.json 2> $O/b_free.err ) || tail -5 $O/b_free.err
rm -f gpurun_out/r06p_free/*kernel_trace.csv
for B in 4096 512; do
  mkdir -p gpurun_out/r06_sw$B
  bash tools/run/ep_sweep_prof.sh r06_sw$B $B 5 > $O/sw$B.txt 2>&1 || { tail -20 $O/sw$B.txt; exit 1; }
  head -3 gpurun_out/r06_sw$B/summary.txt
  python3 tools/prune_prof.py gpurun_out/r06_sw$B; rm -f gpurun_out/r06_sw$B/stats/*kernel_trace.csv
done
# LlamaGen's 16384-id window at the saturating batch: the two-per-CU instance against the generic one (events), then rocprofv3 stats and a FETCH_SIZE pass of the instance
timeout -k 10 400 python3 tools/lg_sweep.py 512,4096 10 1,0 2 > $O/lg_sweep.json 2> $O/lg_sweep.err || { tail -20 $O/lg_sweep.err; exit 1; }
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_lg/stats -o s -- python3 tools/lg_sweep.py 4096 10 1 1 > $O/lg_stats.json 2> $O/lg_stats.err &&
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r06_lg/fetch -o p -- python3 tools/lg_sweep.py 4096 6 1 1 > $O/lg_fetch.json 2> $O/lg_fetch.err ) || { tail -5 $O/lg_stats.err $O/lg_fetch.err; exit 1; }
python3 tools/pmc_sum.py gpurun_out/r06_lg/fetch "epw_kernel" > gpurun_out/r06_lg/summary.txt; cat gpurun_out/r06_lg/summary.txt
python3 tools/prune_prof.py gpurun_out/r06_lg; rm -f gpurun_out/r06_lg/stats/*kernel_trace.csv
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras-out $O/bench_full.json > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(len(open("$O/bench.json").read()), d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("saturating",{}).get("frac"), d.get("extras"))
PY
du -sh gpurun_out

#!/bin/bash
# round 6's soaks against the CPU oracle on the final library: round 5's randomized soaks (tests/fuzz_soak.py), the compact throughput instance of
# evaluate_posterior over 1056 sequences x 6 steps in delta and lambda mode (every third sequence replayed by the oracle), and the timed loop with commit
# turn-taking over 400 steps with the oracle's replay (bench.py exits 3 on a mismatch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-soak_r06}; mkdir -p $O
{
timeout -k 10 500 python tests/fuzz_soak.py 60 2>&1 | tail -3
timeout -k 10 400 python tests/fuzz_soak.py 40 dynamic 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 800 o7 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 1500 o3 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 15 top_p 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 800 draws 2>&1 | tail -1
timeout -k 10 900 python3 - <<'PY'
import sys, time
sys.path[:0] = ["tests", "tests/golden", "."]
import test_gpu_loop as T
for delta in (0.1, 5.0):
    t0 = time.time()
    T._lumina_static_loop_big("mc_sim_7b_63", 1056, 6, 3, lantern_delta=delta)
    print(f"compact throughput instance, lantern_delta {delta}: 1056 sequences x 6 steps, 352 sequences replayed by the oracle: identical ({time.time() - t0:.0f}s)", flush=True)
PY
timeout -k 10 600 python3 - <<'PY'
import sys, time
sys.path[:0] = ["tests", "tests/golden", "."]
import test_gpu_loop as T
t0 = time.time()
T._llamagen_dynamic_loop(False, 1, 0, 1.0, n_seq=1056, steps=5, every=6)
print(f"LlamaGen two-per-CU instance: 1056 sequences x 5 steps, 176 sequences replayed by the oracle: identical ({time.time() - t0:.0f}s)", flush=True)
PY
timeout -k 10 600 python3 bench.py --steps 400 --warmup 20 --cpu-seconds 40 --ep-sweep "" --no-extras --commit-window 1 > $O/turn400.json 2> $O/turn400.err; echo "bench exit $?"
python3 -c "import json; d=json.load(open('$O/turn400.json')); c=d['cpu_baseline']; print('turn-taking, 400 steps: us/step %.2f; oracle replay: %s, mismatches %d (%s)' % (1e3*d['ms_per_step'], c['matches_gpu_token_stream'], c['mismatches'], c['sample']))"
} | tee $O/soak.txt

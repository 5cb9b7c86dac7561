"""The LlamaGen / Anole mirrors' generate() on the scripted target models and drafters of the reference-recorded runs (tests/golden/gen_fakes_lg.py:
forwards that cost almost nothing), static and EAGLE-2 trees: microseconds per verify step with the step through ONE lantern_verify_step call against a ctypes call
per kernel -- the host + kernel cost of the loop body (models/ea_model_llamagen.py:1109-1169) at the reference's batch of one.
usage: mirror_lg_bench.py [repeats=5]"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import gen_fakes_lg as G
import test_gpu_generate_lg as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
out = []
for case in G.CASES:
    row = dict(case=case["name"], model=case["model"], tree=case["tree"])
    for native in (True, False):
        parts = T.build(case, "window")
        parts[0].uniform_window = 4096
        T.run_case(case, "window", parts, native=native)          # warm-up: tree buffers, packed table, KV cache
        torch.cuda.synchronize()
        n_steps, t = 0, 0.0
        for _ in range(reps):
            parts[2].calls.clear(); parts[1].model.calls.clear()
            t0 = time.perf_counter()
            mdl = T.run_case(case, "window", parts, native=native)[0]
            torch.cuda.synchronize()
            t += time.perf_counter() - t0
            n_steps += len(mdl.last_steps)
        row["us_per_step_one_call" if native else "us_per_step_per_kernel"] = round(1e6 * t / n_steps, 1)
        row["steps"] = n_steps // reps
    out.append(row)
    print(row, flush=True)
print(json.dumps(out))

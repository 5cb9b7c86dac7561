"""Extended soak of tests/test_gpu_fuzz.py::test_static_batches_vs_oracle over many more seeds (development aid, not part of the suite):
    python tools/fuzz_soak.py <seeds>   ->  8 parameter sets x <seeds> batches of 32 sequences, both kernel sets vs the oracle."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import test_gpu_fuzz as F
params = [("lumina", "mc_sim_7b_63", True, 100, 0.1, 1.0), ("lumina", "mc_sim_7b_63", True, 300, 5.0, 2.0),
          ("lumina", "naive_extend_57", True, 10, 0.3, 0.5), ("llamagen", "naive_extend_57", True, 50, 0.1, 1.0),
          ("llamagen", "mc_sim_7b_63", True, 200, 10.0, 2.0), ("anole", "naive_extend_57", True, 10, 5.0, 1.0),
          ("anole", "naive_extend_57", True, 5, 20.0, 3.0), ("anole", "mc_sim_7b_63", False, 1, 0.1, 0.5)]
t0 = time.time(); n = 0; fails = 0
for seed in range(100, 100 + int(sys.argv[1])):
    for p in params:
        try:
            F.test_static_batches_vs_oracle(*p, seed)
        except AssertionError as e:
            fails += 1
            print("FAIL", p, seed, str(e)[:300], flush=True)
        n += 1
print(f"cases={n} batches x 32 sequences, fails={fails}, {time.time() - t0:.0f}s")

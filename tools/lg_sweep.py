"""evaluate_posterior alone for LlamaGen's standard verify at batches beyond the CU count (bench.lg_batch_sweep, stand-alone): the two-per-CU throughput
instance against its variants and the generic one-per-CU instance, alternating inside one process; the target of the rocprofv3 passes of tools/run/r06_profiles.sh.

usage: lg_sweep.py <batches, e.g. 512,4096> [steps per variant] [variants, e.g. 1,3,2,4,0] [repetitions]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,4096").split(",") if x]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
variants = tuple(int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,3,2,4,0").split(","))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for B in batches:
    print(json.dumps(bench.lg_batch_sweep([B], dev, iters=iters, variants=variants, reps=reps)[0]), flush=True)

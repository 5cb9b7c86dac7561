"""Dynamic-tree step loop as G independent workloads on G streams (diagnostic): does the stream-group overlap of the static harness carry over?
Usage: python tools/dyn_groups.py [groups] [n_seq] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 63
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda")
n -= n % G
wls = [HN.DynamicVerifyWorkload(HN.DynamicConfig(n_seq=n // G, max_steps=2 * steps + 32, fuse_o7=True, seed=3700 + g), dev) for g in range(G)]
sts = [torch.cuda.Stream(device=dev) for _ in range(G)]
def run(k):
    for _ in range(k):
        for wl, st in zip(wls, sts):
            with torch.cuda.stream(st):
                wl.step()
run(10); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
toks = sum(wl.accepted_tokens(10, 10 + steps) for wl in wls)
for wl in wls: wl.check_status(0, steps + 10)
print(f"G={G} n={n}: host loop {1e6*(t1-t0)/steps:.1f} us/step, GPU done {1e6*(t2-t0)/steps:.1f} us/step, {toks/(t2-t0):.0f} tokens/s", flush=True)

"""What a join() (the current stream waits on one event per group stream) in front of the closing synchronise costs a timed loop: the headline configuration three times
in one process, 100 timed steps each, closed by torch.cuda.synchronize() alone (default) or by join() + synchronize() (JOIN=1): 65.1-65.8 against 87.6-92.3 us per step.
(`keep`: the freed slabs stay in torch's caching allocator between the builds; SPIN: seconds of prime(); KV_SMAX: rows per slab -- none of them matters.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from lantern_amd import harness as HN
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
res = []
KEEP = len(sys.argv) > 1 and sys.argv[1] == "keep"          # keep the freed slabs in torch's caching allocator (no hipFree / hipMalloc between the workloads)
if KEEP:
    torch.cuda.empty_cache = lambda: None
    _mgi = torch.cuda.mem_get_info
    torch.cuda.mem_get_info = lambda *a: (lambda f, t: (f + torch.cuda.memory_reserved() - torch.cuda.memory_allocated(), t))(*_mgi(*a))          # cached blocks count as free
for rep in range(3):
    cfg = HN.WorkloadConfig(n_seq=64, n_groups=4, ep_kernel="chain", fuse_o7=True, spec_rows=1, fused_prepare=True, commit_window=1, max_steps=200, sigma=5.0, kv_smax=int(os.environ.get("KV_SMAX", "4096")))
    wl = HN.LuminaVerifyWorkload(cfg, dev)
    wl.prime(float(os.environ.get("SPIN", "0.5")))
    for _ in range(10):
        wl.step()
    (wl.join() if os.environ.get("JOIN") == "1" else None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        wl.step()
    (wl.join() if os.environ.get("JOIN") == "1" else None); torch.cuda.synchronize()
    res.append(round(1e6 * (time.perf_counter() - t0) / 100, 2))
    print(rep, res[-1], torch.cuda.memory_reserved() >> 30, flush=True)
    wl.release_kv(); del wl; torch.cuda.empty_cache()
print(json.dumps(res))

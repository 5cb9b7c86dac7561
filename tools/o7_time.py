"""cfg_mask_topk_window alone over B sequences x 26 rows (probability rows, top-k 2000, the bench's pools rotating over `pool_steps` inputs):
microseconds per launch between HIP events, bytes = rows x W x (2 x 2 + 4), fraction of the 8 TB/s peak.  usage: o7_time.py [B=63] [launches=200]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 63
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
PS = 8
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=PS, with_kv=False, max_steps=8), torch.device("cuda"))
N, V, W = wl.N, HN.V, 8192


def launch(i):
    s = i % PS
    ops.cfg_mask_topk_window(wl.cond[s].view(B * N, V), wl.uncond[s].view(B * N, V), 3.0, 4, W, model=ops.MODEL_LUMINA, pos_ids=wl.d_pos_ids,
                             pos_base=67, top_k=2000, seq_len=wl.lens[0], rows_per_seq=N, out=wl.proc, row_hot=wl.row_hot, probs=True)


for i in range(20):
    launch(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    launch(i)
    ev[i + 1].record()
torch.cuda.synchronize()
t = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
hot = int((wl.row_hot >= 0).sum())
rows = B * N - hot
byts = rows * W * 8
med = t[n // 2]
print(json.dumps({"rows": B * N, "grid_rows": rows, "us_median": med, "us_min": t[0], "bytes": byts, "TBps": byts / med / 1e6, "frac_of_8TBps": byts / med / 8e6}))

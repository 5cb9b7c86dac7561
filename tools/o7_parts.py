"""cfg_mask_topk_window under ablations (diagnostic): which part of the row post-process costs what.  Usage: python tools/o7_parts.py [n_seq]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN, ops

def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = HN.WorkloadConfig(n_seq=B, pool_steps=2, with_kv=False, max_steps=64, ep_kernel="chain")
wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
wl.step(); torch.cuda.synchronize()
N, V = wl.N, HN.V
lens = wl.lens[0]
def o7(top_k, probs):
    return lambda: ops.cfg_mask_topk_window(wl.cond[0].view(B * N, V), wl.uncond[0].view(B * N, V), 3.0, 4, 8192, model=ops.MODEL_LUMINA,
                                            pos_ids=wl.d_pos_ids, pos_base=67, top_k=top_k, seq_len=lens, rows_per_seq=N, out=wl.proc, row_hot=wl.row_hot, probs=probs)
for tk, pr in ((2000, True), (0, True), (2000, False), (0, False)):
    print(f"B={B} rows={B*N} top_k={tk} probs={pr}: {timeit(o7(tk, pr)):.1f} us per launch (back to back, incl. dispatch gap)", flush=True)

"""cfg_mask_topk_window alone (probability rows), N identical launches: target of rocprofv3 --pmc passes.  Diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
probs = (sys.argv[2] != "logits") if len(sys.argv) > 2 else True
topk = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=1, with_kv=False, max_steps=8), torch.device("cuda"))
N, V = wl.N, HN.V
for _ in range(20):
    ops.cfg_mask_topk_window(wl.cond[0].view(B * N, V), wl.uncond[0].view(B * N, V), 3.0, 4, 8192, model=ops.MODEL_LUMINA, pos_ids=wl.d_pos_ids,
                             pos_base=67, top_k=topk, seq_len=wl.lens[0], rows_per_seq=N, out=wl.proc, row_hot=wl.row_hot, probs=probs)
torch.cuda.synchronize()

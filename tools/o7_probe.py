"""GPU probe (diagnostic): cfg_mask_topk_window / kv_gather / accept_gather timings under ablations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN, ops

def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SM = int(os.environ.get("KV_SMAX", "4096"))
cfg = HN.WorkloadConfig(n_seq=B, pool_steps=2, with_kv=True, max_steps=64, kv_smax=SM, use_graph=False)
wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
wl.step(); torch.cuda.synchronize()
N, V = wl.N, HN.V
lens = wl.lens[0]
def o7(top_k, model=ops.MODEL_LUMINA):
    return lambda: ops.cfg_mask_topk_window(wl.cond[0].view(B * N, V), wl.uncond[0].view(B * N, V), 3.0, 4, 8192, model=model,
                                            pos_ids=wl.d_pos_ids, pos_base=67, top_k=top_k, seq_len=lens, rows_per_seq=N, out=wl.proc, row_hot=wl.row_hot)
print("O7w top_k=2000 us", timeit(o7(2000)))
print("O7w top_k=0    us", timeit(o7(0)))
print("O7w anole(no topk) us", timeit(o7(0, ops.MODEL_ANOLE)))
best, alen = wl.st_best, wl.st_alen
slab_prev = wl.lens[0]
def kv():
    ops.kv_gather(wl.slabs, wl.slab_seq, slab_prev, wl.d_retrieve, best, alen, slab_ptrs=wl.slab_ptrs)
print("kv_gather us", timeit(kv), "moved MB", float((alen.float() + 1).sum()) * 2 * 2 * 64 * 32 * 128 * 2 / 1e6)
def ag():
    ops.accept_gather(wl.hidden[0], wl.d_retrieve, wl.cand, best, alen)
print("accept_gather(hidden only) us", timeit(ag))
x = torch.empty(64 << 20, device="cuda"); y = torch.empty_like(x)
print("torch copy 256MB us", timeit(lambda: y.copy_(x), 20), "-> GB/s", 2 * 256e6 * 1.048576 / (timeit(lambda: y.copy_(x), 20) * 1e-6) / 1e9)

"""update_inference_inputs alone, N identical launches on fixed verdicts (the target of FETCH_SIZE / WRITE_SIZE passes): every launch moves exactly the
same rows, so the PMC bytes can be held to the exact byte count of the launch -- KV rows read and written (rows already in place are skipped), the
accepted-hidden rows read, the [B, 2, D, H] output written (zero rows behind the accepted ones included), the accepted tokens.
usage: kv_only.py [sequences=21] [launches=40]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN
from lantern_amd._lib import check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda")
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=2, max_steps=16, ep_kernel="chain", fuse_o7=True, spec_rows=3), dev)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
best, alen = wl.log_best[3].clone(), wl.log_alen[3].clone()
c = wl.cfg
vp = C.c_void_p
A = wl._group_args(0, 0, 0)
st = vp(torch.cuda.current_stream().cuda_stream)
L = wl._L
for _ in range(n):
    check(L.lantern_update_inference_inputs(A["slab_ptrs"], A["slab_seq"], A["cur"], 2 * B, 2, C.c_int64(2 * c.kv_layers * c.kv_heads), C.c_int64(c.kv_smax + c.kv_pad_rows),
                                            C.c_int64(c.kv_dim), vp(wl.d_retrieve.data_ptr()), 0, wl.P, wl.D, vp(best.data_ptr()), vp(alen.data_ptr()), A["nxt"],
                                            A["hidden"], 2, B, 2, wl.N, HN.HIDDEN, A["cand"], A["out_hidden"], A["acc_tokens"], st), "update")
torch.cuda.synchronize()
ret = wl.d_retrieve.reshape(wl.P, wl.D)[best.long()].cpu()                      # [B, D]
t = torch.arange(wl.D)
live = t[None] <= alen.cpu().long()[:, None]
moved_rows = int(((ret != t) & live).sum()) * 2                                # two slabs per sequence
row_bytes = 2 * c.kv_layers * c.kv_heads * c.kv_dim * 2                        # one position of one slab: K and V of every layer / head
kv = moved_rows * row_bytes
hid_read = int(live.sum()) * 2 * HN.HIDDEN * 2
hid_write = B * 2 * wl.D * HN.HIDDEN * 2
print(json.dumps({"sequences": B, "launches": n, "kv_bytes_read": kv, "kv_bytes_written": kv, "hidden_bytes_read": hid_read, "hidden_bytes_written": hid_write,
                  "expected_read": kv + hid_read, "expected_written": kv + hid_write + B * wl.D * 8}))

"""update_inference_inputs alone at several batch sizes / slab lengths: microseconds per launch between HIP events (back-to-back launches on fixed verdicts) --
what of a launch's time is fixed and what scales with the rows moved.
usage: kv_lat.py [launches=50]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN
from lantern_amd._lib import check
from lantern_amd import _lib as _L
_KNOBS = _L.tuning_from_env()          # LANTERN_<NAME>=<int> of this tool's environment -> explicit lantern_tuning_set calls

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cases = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "1:4096,4:4096,21:4096,63:4096").split(",")]
dev = torch.device("cuda")
vp = C.c_void_p
out = []
for B, smax in cases:
    wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=2, max_steps=16, ep_kernel="chain", fuse_o7=True, spec_rows=3, kv_smax=smax), dev)
    for _ in range(12):
        wl.step()
    wl.sync()
    torch.cuda.synchronize()
    step = int(os.environ.get("KV_LAT_STEP", "3"))
    best, alen = wl.log_best[step].clone(), wl.log_alen[step].clone()
    print("moved MB (r+w) per logged step:", [round(wl.kv_moved_bytes(i, i + 1) / 1e6, 1) for i in range(12)], flush=True)
    c = wl.cfg
    A = wl._group_args(0, 0, 0)
    st = vp(torch.cuda.current_stream().cuda_stream)
    L = wl._L

    # a fresh block of rows per launch (previous length + 24 * i): nothing a launch reads was touched by an earlier one, as in the loop, where
    # the rows a step moves were written by the target forward long before
    cur0 = torch.full((2 * B,), 600, dtype=torch.int64, device=dev) if smax >= 4096 else torch.full((2 * B,), 8, dtype=torch.int64, device=dev)
    stride = 24 if smax >= 4096 else 0
    curs = [(cur0 + stride * i).contiguous() for i in range(n + 5)]
    it = [0]

    def launch():
        cur = vp(curs[it[0] % len(curs)].data_ptr())
        it[0] += 1
        check(L.lantern_update_inference_inputs(A["slab_ptrs"], A["slab_seq"], cur, 2 * B, 2, C.c_int64(2 * c.kv_layers * c.kv_heads), C.c_int64(c.kv_smax + c.kv_pad_rows),
                                                C.c_int64(c.kv_dim), vp(wl.d_retrieve.data_ptr()), 0, wl.P, wl.D, vp(best.data_ptr()), vp(alen.data_ptr()), A["nxt"],
                                                A["hidden"], 2, B, 2, wl.N, HN.HIDDEN, A["cand"], A["out_hidden"], A["acc_tokens"], st), "update")
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ret = wl.d_retrieve.reshape(wl.P, wl.D)[best.long()].cpu()
    t = torch.arange(wl.D)
    live = t[None] <= alen.cpu().long()[:, None]
    moved = int(((ret != t) & live).sum()) * 2 * (2 * c.kv_layers * c.kv_heads * c.kv_dim * 2)
    us = 1e3 * e0.elapsed_time(e1) / n
    out.append({"sequences": B, "S_max": smax, "us_per_launch": round(us, 2), "kv_MB_moved_rw": round(2 * moved / 1e6, 2), "GBps": round(2 * moved / us / 1e3, 1)})
    print(out[-1], flush=True)
    wl.release_kv()
    del wl
    torch.cuda.empty_cache()
print(json.dumps({"knobs": {k: os.environ.get(k) for k in ("LANTERN_KV_VARIANT", "LANTERN_KV_GX")}, "cases": out}))

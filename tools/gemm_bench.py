"""The drafter layer's four weight-streaming GEMMs alone (20 bf16 rows at Lumina-mGPT-7B size), each timed between HIP events over rotating
weight copies (so that a 256 MB last-level cache cannot hold them), GB/s of weight bytes against the 8 TB/s HBM figure.
Usage: python tools/gemm_bench.py [rows=20] [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev, bf = torch.device("cuda"), torch.bfloat16
H, I, COPIES = 4096, 11008, 4
torch.manual_seed(0)
mk = lambda n, k: [torch.randn(n, k, device=dev, dtype=bf) / k ** 0.5 for _ in range(COPIES)]
x = torch.randn(M, H, device=dev, dtype=bf)
xi = torch.randn(M, I, device=dev, dtype=bf)
res = torch.randn(M, H, device=dev, dtype=bf)
FORM = os.environ.get("GEMM_FORM", "packed")
if FORM in ("streamk", "packed"):          # packed: what the decoder layer runs (the weights in the kernel's brick layout)
    pk = (lambda ws, pair=0: [ops.pack_linear_weight(w, pair) for w in ws]) if FORM == "packed" else (lambda ws, pair=0: ws)
    tag = "stream-K, packed" if FORM == "packed" else "stream-K, row-major"
    shapes = {
        f"qkv_proj ({tag}, 12288 x 4096)": (pk(mk(3 * H, H)), lambda w: ops.linear_rows_streamk(x, w)),
        f"o_proj + residual ({tag}, 4096 x 4096)": (pk(mk(H, H)), lambda w: ops.linear_rows_streamk(x, w, ops.EPI_RESIDUAL, residual=res)),
        f"gate_up + silu*mul ({tag}, 2 x 11008 x 4096)": (pk(mk(2 * I, H), I), lambda w: ops.linear_rows_streamk(x, w, ops.EPI_SILU_MUL, pair_rows=I)),
        f"down_proj + residual ({tag}, 4096 x 11008)": (pk(mk(H, I)), lambda w: ops.linear_rows_streamk(xi, w, ops.EPI_RESIDUAL, residual=res)),
    }
else:                          # the per-tile kernels of round 2
    shapes = {
        "qkv_proj (linear_rows, 12288 x 4096)": (mk(3 * H, H), lambda w: ops.linear_rows(x, w, 0, w.shape[0])),
        "o_proj + residual (split-K, 4096 x 4096)": (mk(H, H), lambda w: ops.linear_rows_splitk(x, w, residual=res)),
        "gate_up + silu*mul (epilogue, 2 x 11008 x 4096)": (mk(2 * I, H), lambda w: ops.linear_rows_epilogue(x, w, ops.EPI_SILU_MUL, pair_rows=I)),
        "down_proj + residual (split-K, 4096 x 11008)": (mk(H, I), lambda w: ops.linear_rows_splitk(xi, w, residual=res)),
    }
out = {"rows": M, "form": FORM, "kernels": {}}
tot_b = tot_t = 0.0
for name, (ws, fn) in shapes.items():
    for w in ws:
        fn(w)
    torch.cuda.synchronize()
    K = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(K):
        fn(ws[i % COPIES])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    nbytes = (ws[0].data if hasattr(ws[0], 'data') and not torch.is_tensor(ws[0]) else ws[0]).numel() * 2
    out["kernels"][name] = {"us": us, "weight_MB": nbytes / 1e6, "GBps": nbytes / us / 1e3, "frac_of_8TBps": nbytes / us / 1e3 / 8000.0}
    tot_b += nbytes; tot_t += us
    print(f"{name:60s} {us:7.1f} us  {nbytes / 1e6:6.1f} MB  {nbytes / us / 1e3:7.0f} GB/s  {nbytes / us / 1e3 / 8000.0:.3f}", flush=True)
out["all"] = {"us": tot_t, "weight_MB": tot_b / 1e6, "GBps": tot_b / tot_t / 1e3, "frac_of_8TBps": tot_b / tot_t / 1e3 / 8000.0}
print(f"{'all four':52s} {tot_t:7.1f} us  {tot_b / 1e6:6.1f} MB  {tot_b / tot_t / 1e3:7.0f} GB/s  {tot_b / tot_t / 1e3 / 8000.0:.3f}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)

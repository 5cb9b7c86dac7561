"""lantern_linear_rows_packed (the drafter layer's GEMMs at prefill row counts) against torch's F.linear (hipBLASLt) on the same shapes: TFLOP/s of
each, the 7B layer's four products.  usage: gemm_bench.py [rows, e.g. 1200,4096] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lantern_amd import ops
from lantern_amd import _lib as _L
_KNOBS = _L.tuning_from_env()          # LANTERN_<NAME>=<int> of this tool's environment -> explicit lantern_tuning_set calls

rows = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1200,4096").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev, bf = torch.device("cuda"), torch.bfloat16
shapes = [("qkv", 4096, 12288, 0), ("o_proj", 4096, 4096, 1), ("gate_up", 4096, 11008, 2), ("down", 11008, 4096, 1)]


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for M in rows:
    for name, K, N, epi in shapes:
        x = torch.randn(M, K, device=dev, dtype=bf)
        w = (torch.randn((2 * N if epi == 2 else N), K, device=dev) / K ** 0.5).to(bf)
        res = torch.randn(M, N, device=dev, dtype=bf)
        pk = ops.pack_linear_weight(w, N if epi == 2 else 0)
        kw = dict(residual=res) if epi == 1 else {}
        ms_hip = timed(lambda: ops.linear_rows_packed(x, pk, epi, **kw))
        if epi == 2:
            ms_t = timed(lambda: F.silu(F.linear(x, w[:N])) * F.linear(x, w[N:]))
        elif epi == 1:
            ms_t = timed(lambda: res + F.linear(x, w))
        else:
            ms_t = timed(lambda: F.linear(x, w))
        fl = 2.0 * M * K * w.shape[0]
        out.append(dict(rows=M, gemm=name, K=K, N=w.shape[0], ms_hip=round(ms_hip, 4), ms_torch=round(ms_t, 4), TFLOPs_hip=round(fl / ms_hip / 1e9, 1),
                        TFLOPs_torch=round(fl / ms_t / 1e9, 1)))
        print(out[-1], flush=True)
        del x, w, res, pk
print(json.dumps(out))

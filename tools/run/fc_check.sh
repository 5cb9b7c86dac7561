#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/fc
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_more.py tests/test_gpu_drafter.py tests/test_drafter_layer.py tests/test_gpu_generate.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -40 $O/t.txt; exit 1; }
tail -2 $O/t.txt
timeout -k 10 300 python3 - <<'PY' 2>&1 | grep -v amdgpu
import torch, sys
sys.path.insert(0, ".")
from lantern_amd import ops
dev, bf = torch.device("cuda"), torch.bfloat16
H, V, M = 4096, 65536, 20
torch.manual_seed(0)
def timeit(fn, n=40):
    for _ in range(4): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ids = torch.randint(0, 8000, (M,), device=dev)
hid = torch.randn(M, H, device=dev, dtype=bf)
emb = torch.randn(8200, H, device=dev, dtype=bf)
Ws = [torch.randn(H, 2 * H, device=dev, dtype=bf) / 90 for _ in range(4)]
pks = [ops.pack_linear_weight(w) for w in Ws]
big = torch.randn(40, H, device=dev, dtype=bf); idb = torch.randint(0, 8000, (40,), device=dev)
print("drafter_fc 20 rows, 67 MB: per-tile (40 rows path) %.1f us | stream-K row-major %.1f us | stream-K packed %.1f us" % (
    timeit(lambda i: ops.drafter_fc(idb, big, emb, Ws[i % 4])), timeit(lambda i: ops.drafter_fc(ids, hid, emb, Ws[i % 4])),
    timeit(lambda i: ops.drafter_fc(ids, hid, emb, Ws[i % 4], packed=pks[i % 4]))))
heads = [torch.randn(V, H, device=dev, dtype=bf) / 64 for _ in range(3)]
hp = [ops.pack_linear_weight(w[4:8196].contiguous()) for w in heads]
A = torch.randn(20, H, device=dev, dtype=bf)
pos = 5 + torch.arange(10, device=dev)
sc = torch.randn(10, device=dev)
kw = dict(model=ops.MODEL_LUMINA, pos_ids=pos, pos_base=2, top_k_filter=2000, scores_in=sc, top_k=10)
print("head_expand 2x10 rows, 67 MB window: per-tile %.1f us | stream-K row-major %.1f us | stream-K packed %.1f us" % (
    timeit(lambda i: ops.head_expand(A, heads[i % 3], 4, 8192, 3.0, streamk=False, **kw)), timeit(lambda i: ops.head_expand(A, heads[i % 3], 4, 8192, 3.0, **kw)),
    timeit(lambda i: ops.head_expand(A, heads[i % 3], 4, 8192, 3.0, packed=hp[i % 3], **kw))))
PY

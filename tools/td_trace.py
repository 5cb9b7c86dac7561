"""Phase stamps of tree_dynamic_finalize_kernel (needs a build with -DTD_TRACE: make -C lantern_amd/csrc clean; make -C lantern_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DTD_TRACE")."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import ops, _lib
B, V = 64, 65536
g = torch.Generator(device="cuda").manual_seed(1)
scores = torch.randn((B, 410), generator=g, device="cuda")
tokens = torch.randint(4, 8196, (B, 410), generator=g, device="cuda")
par = [torch.zeros((B, 1), dtype=torch.int64, device="cuda")]
cs = torch.arange(10, device="cuda").expand(B, 10)
for d in range(4):
    par.append(cs + 1 + 100 * max(0, d - 1) + (10 if d > 0 else 0))
parents = torch.cat(par, 1).contiguous()
sample = torch.randint(4, 8196, (B,), generator=g, device="cuda")
for _ in range(3):
    ops.tree_dynamic_finalize(scores, tokens, parents, sample, 10, 58)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
assert _lib.lib().lantern_debug_td_trace(buf) == 0
t = list(buf)[:10]
print("stamps (cycles from start):", [x - t[0] for x in t])
print("phase cycles:", [t[i + 1] - t[i] for i in range(9)])

"""evaluate_posterior alone, N identical launches at a given batch size -- the target of the rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE need their own runs, MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Prints the algorithmic bytes."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN
from lantern_amd._lib import check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
path = sys.argv[3] if len(sys.argv) > 3 else "window"
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=1, with_kv=False, max_steps=8, path=path), torch.device("cuda"))
wl.step(); torch.cuda.synchronize()
L = wl._L
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
buf = wl.ep_buffers(0, 0)
win = wl.ep_window(0) if wl.windowed else None
for _ in range(iters):
    wl.cursor.zero_()
    if wl.windowed:
        check(L.lantern_evaluate_posterior_window(C.byref(wl._ep_prm), C.byref(buf), C.byref(win), st), "ep")
    else:
        check(L.lantern_evaluate_posterior(C.byref(wl._ep_prm), C.byref(buf), st), "ep")
torch.cuda.synchronize()
print(json.dumps({"sequences": B, "iters": iters, "path": path,
                  "window_bytes_per_launch": wl.ep_window_bytes_from(wl.st_cnt) if wl.windowed else None,
                  "dense_contract_bytes_per_launch": wl.ep_algorithmic_bytes_from(wl.st_cnt)}))

"""GPU probe: where does epw_kernel's time go?  Times the windowed evaluate_posterior alone on one pool slot
under ablations (not a benchmark; diagnostic only)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN
from lantern_amd._lib import check

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cfg = HN.WorkloadConfig(n_seq=B, pool_steps=2, with_kv=False, max_steps=64, use_graph=False)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    wl.step()   # fills cand/proc/row_hot for slot 0
    torch.cuda.synchronize()
    L = wl._L
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(prm, uni=None, bonus=True):
        buf = wl.ep_buffers(0, 0)
        win = wl.ep_window(0)
        if uni is not None: buf.uniforms = uni.data_ptr()
        if not bonus: win.u_bonus = None; win.token = None
        def f():
            wl.cursor.zero_()
            check(L.lantern_evaluate_posterior_window(C.byref(prm), C.byref(buf), C.byref(win), st), "ep")
        return f
    base = wl._ep_prm
    print("B =", B)
    print("cursor.zero_ only      us", timeit(lambda: wl.cursor.zero_()))
    print("full                    us", timeit(run(base)), wl.st_cnt.float().mean(0).tolist())
    zeros = torch.zeros_like(wl.uniforms)
    print("accept-all (u=0)        us", timeit(run(base, zeros)), wl.st_cnt.float().mean(0).tolist())
    ones = torch.full_like(wl.uniforms, 0.9999999)
    print("reject-mostly (u~1)     us", timeit(run(base, ones)), wl.st_cnt.float().mean(0).tolist())
    import copy
    p2 = copy.copy(base); p2.lantern = 0
    print("lantern off             us", timeit(run(p2)), wl.st_cnt.float().mean(0).tolist())
    print("no bonus draw           us", timeit(run(base, bonus=False)))
    p3 = copy.copy(base); p3.D = 1
    print("D=1 (final softmax only) us", timeit(run(p3)))

main()

"""Probe: is a kernel's store into torch-pinned host memory visible to the CPU while LATER kernels of the same stream are still running?
(The mirrors' verdict record: written by evaluate_posterior straight into pinned memory and polled by the host, instead of a hipMemcpy D2H + sync.)
A small library kernel (lantern_pack_vq_table) writes into a pinned buffer, ~20 ms of matmuls follow on the same stream, the host polls the buffer."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from lantern_amd import _lib

dev = torch.device("cuda")
L = _lib.lib()
src = torch.full((4, 16), 7, dtype=torch.int16, device=dev)
pin = torch.zeros((4, 16), dtype=torch.int16).pin_memory()
view = pin.numpy()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
res = []
for trial in range(5):
    pin.zero_()
    torch.cuda.synchronize()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    for _ in range(2):
        b = a @ a                       # work in front of the store
    _lib.check(L.lantern_pack_vq_table(C.c_void_p(src.data_ptr()), 4, 16, C.c_void_p(pin.data_ptr()), 16, st), "pack")
    for _ in range(40):
        b = a @ a                       # work behind it
    t_enq = time.perf_counter() - t0
    seen = None
    while time.perf_counter() - t0 < 2.0:
        if view[3, 15] == 7:
            seen = time.perf_counter() - t0
            break
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    res.append({"enqueued_ms": 1e3 * t_enq, "store_seen_ms": None if seen is None else 1e3 * seen, "stream_done_ms": 1e3 * t_all})
print(json.dumps({"probe": "kernel store into pinned host memory, polled by the CPU while the stream goes on", "trials": res}))

#!/usr/bin/env python3
"""Times lantern_tree_attention at the baseline shapes against the reference's eager formulation in torch
(bf16 matmul + additive f32 mask + f32 softmax + bf16 matmul) on the same inputs.  Kernel-only time from HIP events recorded
by the launch itself; achieved bytes = K and V rows read once + Q + out.  Usage: python tools/ta_bench.py [--json out.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import ops  # noqa: E402


def eager(q, k, v, mask, scale):
    w = torch.matmul(q.transpose(1, 2), k.transpose(2, 3)) * scale + mask
    p = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, v).transpose(1, 2).reshape(q.shape[0], q.shape[1], -1)


def time_it(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    rows = []
    # (label, B rows, Hq, Hkv, N, d, S)
    shapes = [("lumina 1 seq static (cond+uncond)", 2, 32, 32, 26, 128, 2400), ("lumina 1 seq dynamic", 2, 32, 32, 59, 128, 2400),
              ("lumina 8 seq static", 16, 32, 32, 26, 128, 2400), ("lumina 48 seq static", 96, 32, 32, 26, 128, 2400),
              ("lumina 48 seq dynamic", 96, 32, 32, 59, 128, 2400), ("llamagen 1 seq dynamic", 2, 20, 20, 59, 64, 400),
              ("llamagen 48 seq dynamic", 96, 20, 20, 59, 64, 400)]
    for label, B, Hq, Hkv, N, d, S in shapes:
        g = torch.Generator(device="cuda").manual_seed(0)
        q = torch.randn(B, N, Hq, d, generator=g, device="cuda").to(torch.bfloat16)
        k = torch.randn(B, Hkv, S, d, generator=g, device="cuda").to(torch.bfloat16)
        v = torch.randn(B, Hkv, S, d, generator=g, device="cuda").to(torch.bfloat16)
        tm = torch.tril(torch.ones(N, N, device="cuda"))
        bits = ops.tree_mask_bits(tm)
        lens = torch.full((B,), S, dtype=torch.int64, device="cuda")
        out = torch.empty(B, N, Hq * d, dtype=torch.bfloat16, device="cuda")
        us = time_it(lambda: ops.tree_attention(q, k, v, bits, kv_len=lens, max_kv_len=S, out=out))
        mask = torch.zeros(B, 1, N, S, device="cuda")
        mask[:, :, :, S - N:][:, :, tm == 0] = torch.finfo(torch.float32).min
        qe = q
        us_eager = time_it(lambda: eager(qe, k, v, mask, d ** -0.5), iters=5, warmup=2) if B * Hq * N * S * 4 < 8e9 else None
        nbytes = 2 * B * Hkv * S * d * 2 + 2 * B * N * Hq * d * 2
        flops = 4.0 * B * Hq * N * S * d
        rows.append({"shape": label, "B": B, "Hq": Hq, "N": N, "d": d, "S": S, "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1),
                     "frac_hbm": round(nbytes / us / 1e3 / 8000, 3), "TFLOPs": round(flops / us / 1e6, 1),
                     "eager_us": None if us_eager is None else round(us_eager, 1)})
        print(rows[-1], flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()

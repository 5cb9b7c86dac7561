#!/usr/bin/env python3
"""Kernel-only durations of lantern_tree_attention (+ merge) per shape via lantern_profile_next_launch-free plain HIP events
around a single launch would include the dispatch gap, so this runs each shape 30 times under rocprofv3 and the summary is
read from the kernel trace:  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/ta_profile.py
Without rocprofv3 it just runs the launches.  Prints the launch order (one line per shape) so trace rows can be matched."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import ops  # noqa: E402

SHAPES = [("lumina_1seq_N26", 2, 32, 32, 26, 128, 2400), ("lumina_1seq_N59", 2, 32, 32, 59, 128, 2400),
          ("lumina_8seq_N26", 16, 32, 32, 26, 128, 2400), ("lumina_48seq_N26", 96, 32, 32, 26, 128, 2400),
          ("lumina_48seq_N59", 96, 32, 32, 59, 128, 2400), ("llamagen_1seq_N59", 2, 20, 20, 59, 64, 400),
          ("llamagen_48seq_N59", 96, 20, 20, 59, 64, 400)]
ITERS = 30

for label, B, Hq, Hkv, N, d, S in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(B, N, Hq, d, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, Hkv, S, d, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, Hkv, S, d, generator=g, device="cuda").to(torch.bfloat16)
    bits = ops.tree_mask_bits(torch.tril(torch.ones(N, N, device="cuda")))
    lens = torch.full((B,), S, dtype=torch.int64, device="cuda")
    out = torch.empty(B, N, Hq * d, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    for _ in range(ITERS):
        ops.tree_attention(q, k, v, bits, kv_len=lens, max_kv_len=S, out=out)
    torch.cuda.synchronize()
    nbytes = 2 * B * Hkv * S * d * 2 + 2 * B * N * Hq * d * 2
    print(f"SHAPE {label} launches={ITERS} bytes={nbytes} flops={4.0 * B * Hq * N * S * d:.0f}", flush=True)

#!/usr/bin/env python3
"""Every device entry point once over at Lumina-7B sizes (B = 1 and B = 64 sequences), 20 launches each, to be run under
`rocprofv3 --kernel-trace --stats`: a table of kernel-only times to spot outliers beside the benchmarked path."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import ops  # noqa: E402
from lantern_amd.drafters.choices import mc_sim_7b_63  # noqa: E402
from lantern_amd.verify import generate_tree_buffers  # noqa: E402

V, LO, HI, H = 65536, 4, 8196, 4096
W = HI - LO
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
tb = generate_tree_buffers(mc_sim_7b_63, device="cuda")
hip = tb["_hip"]
N, (P, D) = hip["N"], tb["retrieve_indices"].shape
R = 11
ITERS = 20


def rep(fn):
    for _ in range(ITERS):
        fn()
    torch.cuda.synchronize()


for B in (1, 64):
    print(f"B={B}", flush=True)
    ss_token = torch.randint(LO, HI, (B, R, 10), generator=g, device=dev)
    ss_prob = torch.rand((B, R, 10), generator=g, device=dev)
    sample = torch.randint(LO, HI, (B,), generator=g, device=dev)
    rep(lambda: ops.gather_candidates(ss_token, ss_prob, sample, tb["tree_indices"], tb["retrieve_indices"]))
    probs = torch.softmax(torch.randn((B * R, W), generator=g, device=dev), -1)
    idx = torch.multinomial(probs, 10)
    rep(lambda: ops.sample_static(probs, idx))
    # drafter input stage and head
    M = 2 * 10 * B if B == 1 else 2 * 10 * 6           # M <= 128 rows per call
    ids = torch.randint(0, V, (M,), generator=g, device=dev)
    hid = torch.randn((M, H), generator=g, device=dev).to(torch.bfloat16)
    emb = torch.randn((V, H), generator=g, device=dev).to(torch.bfloat16)
    Wt = (torch.randn((H, 2 * H), generator=g, device=dev) / 90).to(torch.bfloat16)
    bias = torch.zeros(H, device=dev, dtype=torch.bfloat16)
    rep(lambda: ops.drafter_fc(ids, hid, emb, Wt, bias))
    head = (torch.randn((V, H), generator=g, device=dev) / 64).to(torch.bfloat16)
    rep(lambda: ops.linear_rows(hid, head, LO, W))
    am = torch.ones((2, 300), dtype=torch.bool, device=dev)
    tm = torch.ones((1, 1, 10, 40), device=dev)
    rep(lambda: ops.drafter_attention_mask(am, tm, 2, 10, 290))
    # O3 / O4
    rows = torch.randn((B, 10, V), generator=g, device=dev)
    sc = torch.randn((B, 10), generator=g, device=dev)
    rep(lambda: ops.expand_dynamic(rows, sc, 10))
    scores = torch.randn((B, 410), generator=g, device=dev)
    toks = torch.randint(LO, HI, (B, 410), generator=g, device=dev)
    par = [torch.zeros((B, 1), dtype=torch.int64, device=dev)]
    cs = torch.arange(10, device=dev).expand(B, 10)
    for d in range(4):
        par.append(cs + 1 + 100 * max(0, d - 1) + (10 if d > 0 else 0))
    parents = torch.cat(par, 1).contiguous()
    rep(lambda: ops.tree_dynamic_finalize(scores, toks, parents, sample, 10, 58))
    # greedy accept (LlamaGen-style dense rows are [N, V = 16384]; here Lumina-size window rows)
    nl = torch.randn((B, N, 16384), generator=g, device=dev)
    cand = torch.randint(0, 16384, (B, P, D), generator=g, device=dev)
    ri = torch.from_numpy(np.maximum(tb["retrieve_indices"].cpu().numpy(), 0).astype(np.int32)).cuda()
    table = torch.randint(0, 16384, (16384, 16), generator=g, device=dev).to(torch.int16)
    rep(lambda: ops.evaluate_posterior_greedy(nl, ri, cand, lantern=True, k=10, delta=0.2, tok_offset=0, table=table))
    win = torch.softmax(torch.randn((B, W), generator=g, device=dev), -1)
    ot = torch.full((B,), -1, dtype=torch.int32, device=dev)
    om = torch.zeros(B, device=dev)
    rep(lambda: ops.window_to_dense(win, ot, om, V, LO) if hasattr(ops, "window_to_dense") else None)
print("done")

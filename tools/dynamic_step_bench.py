#!/usr/bin/env python3
"""The verify step with the EAGLE-2 dynamic tree (eagle_version 2: N = 59 nodes, a different tree per sequence and step) at the
C3 sizes, through the same entry points the model mirrors use:
    O4 lantern_tree_dynamic_finalize -> candidates = draft_tokens[retrieve] -> O7 lantern_cfg_mask_topk_window (59 rows / sequence,
    per-node positions) -> O8 lantern_evaluate_posterior_window (MODE_DYNAMIC, per-sequence row maps, ragged paths) ->
    O9 + O10 lantern_update_inference_inputs (per-sequence retrieve rows).
bench.py times the static tree (the reference's default for Lumina); this script puts the dynamic half of C3 on the clock.  Pools
are synthetic (drafter scores from the O3 kernel on random logits; target rows random with the drafted tokens made plausible);
parity of every kernel in this mode is in the test-suite (test_c3_lumina_dynamic_tree_full_size, tests/fuzz_soak.py dynamic).
Usage: python tools/dynamic_step_bench.py [--seqs 64] [--steps 100]      (kernel-only times: run it under rocprofv3 --kernel-trace --stats)"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import ops  # noqa: E402
from lantern_amd.harness import build_neighbour_table  # noqa: E402

V, LO, HI, K = 65536, 4, 8196, 8192
W = HI - LO
TOPK, DEPTH, TOTAL = 10, 4, 58
N = TOTAL + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--kv-rows", type=int, default=640)
    ap.add_argument("--lantern-k", type=int, default=1000)
    ap.add_argument("--lantern-delta", type=float, default=0.1)
    a = ap.parse_args()
    dev = torch.device("cuda")
    B, S = a.seqs, a.slots
    g = torch.Generator(device=dev).manual_seed(7)

    def img_logits(*shape):
        x = torch.full((*shape, V), float("-inf"), device=dev)
        x[..., LO:HI] = 4.0 * torch.randn((*shape, W), generator=g, device=dev)
        kth = torch.topk(x, 2000, dim=-1).values[..., -1:]
        return x.masked_fill(x < kth, float("-inf"))

    table = ops.pack_vq_table(build_neighbour_table(dev, 0), -(-(a.lantern_k + 1) // 8) * 8)
    pools = []
    for s in range(S):                               # drafter side of a step: O3 per depth on random rows (setup, untimed)
        ti, cu, ci, sc = ops.expand_dynamic(img_logits(B, 1), None, TOPK)
        sl, tl, pl = [cu.reshape(B, -1)], [ti.reshape(B, -1)], [torch.zeros((B, 1), dtype=torch.int64, device=dev)]
        cs = torch.arange(TOPK, device=dev).expand(B, TOPK)
        for d in range(DEPTH):
            pl.append(cs + 1 + TOPK * TOPK * max(0, d - 1) + (TOPK if d > 0 else 0))
            ti, cu, ci, sc = ops.expand_dynamic(img_logits(B, TOPK), sc, TOPK)
            cs = ci
            sl.append(cu.reshape(B, -1)); tl.append(ti.reshape(B, -1))
        scores, tokens, parents = torch.cat(sl, 1).contiguous(), torch.cat(tl, 1).contiguous(), torch.cat(pl, 1).contiguous()
        sample = torch.randint(LO, HI, (B,), generator=g, device=dev)
        draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(scores, tokens, parents, sample, TOPK, TOTAL)
        cond = (2.0 * torch.randn((B, N, V), generator=g, device=dev)).to(torch.bfloat16)
        unc = torch.randn((B, N, V), generator=g, device=dev).to(torch.bfloat16)
        r6 = ret[:, :, :DEPTH + 2]
        par, ch = r6[:, :, :-1], r6[:, :, 1:]
        ok = ch >= 0
        bi = torch.arange(B, device=dev)[:, None, None].expand_as(ch)[ok]
        tok = draft.gather(1, ch.clamp(min=0).reshape(B, -1)).reshape(ch.shape)[ok]
        cond[bi, par[ok], tok] = (8.0 - 2.0 * torch.rand(bi.shape, generator=g, device=dev)).to(torch.bfloat16)   # drafted tokens plausible under the target
        hidden = torch.randn((B, 2, N, 4096), generator=g, device=dev).to(torch.bfloat16)
        pools.append(dict(scores=scores, tokens=tokens, parents=parents, cond=cond, unc=unc, hidden=hidden))
    slabs = [torch.zeros((64, 1, 32, a.kv_rows, 128), dtype=torch.bfloat16, device=dev) for _ in range(2 * B)]
    slab_ptrs = torch.tensor([s.data_ptr() for s in slabs], dtype=torch.int64, device=dev)
    slab_seq = torch.arange(B, dtype=torch.int32, device=dev).repeat(2)
    uniforms = torch.rand((B, 64 * (a.steps + 8)), generator=g, device=dev, dtype=torch.float64)
    cursor = torch.zeros(B, dtype=torch.int32, device=dev)
    u_bonus = torch.rand((a.steps + 8, B), generator=g, device=dev, dtype=torch.float64)
    cfg = ops.EpConfig.lumina(False, lantern=True, k=a.lantern_k, delta=a.lantern_delta)
    win = torch.empty((B, N, W), dtype=torch.float32, device=dev)
    hot = torch.empty((B, N), dtype=torch.int32, device=dev)
    prompt = 64

    def step(i, st):
        p = pools[i % S]
        draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(p["scores"], p["tokens"], p["parents"], st["token"], TOPK, TOTAL)
        r6 = ret[:, :N, :DEPTH + 2].contiguous()
        cand = torch.cat((draft, draft.new_full((B, 1), -1)), 1).gather(1, torch.where(r6 < 0, N, r6).reshape(B, -1)).reshape(r6.shape)
        ri = torch.where(r6 < 0, N - 1, r6).to(torch.int32)
        pos_abs = (pos + 1 + st["lens"][:B, None]).reshape(-1)
        ops.cfg_mask_topk_window(p["cond"].view(B * N, V), p["unc"].view(B * N, V), 3.0, LO, W, model=ops.MODEL_LUMINA, pos_ids=pos_abs,
                                 pos_base=prompt + 3, top_k=2000, probs=True, out=win.view(B * N, W), row_hot=hot.view(B * N))
        out = ops.evaluate_posterior_window(cfg, V, win, LO, ri, cand, uniforms, row_hot=hot, table=table, cursor=cursor, u_bonus=u_bonus[i],
                                            n_paths=nl, n_depth=md, want_window=False, rows_probs=True)
        new_len, _, _ = ops.update_inference_inputs(slabs, slab_seq, st["lens"], r6, out["best"], out["accept_len"], p["hidden"], cand,
                                                    slab_ptrs=slab_ptrs)
        over = (new_len[B:] - 3) >= 400                    # keep the short slabs of this script from filling up: wrap early
        st["lens"] = torch.where(over.repeat(2), st["base"], new_len)
        st["token"] = out["token"]
        st["acc"] += out["accept_len"].sum() + B
        st["cnt"] += out["counters"][:, :3].sum(0)
        st["bad"] += (out["counters"][:, 5] != 0).sum()

    base = torch.cat([torch.full((B,), prompt + 3, dtype=torch.int64), torch.full((B,), 3, dtype=torch.int64)]).to(dev)
    st = dict(lens=base.clone(), base=base, token=torch.randint(LO, HI, (B,), generator=g, device=dev), acc=torch.zeros((), dtype=torch.int64, device=dev),
              cnt=torch.zeros(3, dtype=torch.int64, device=dev), bad=torch.zeros((), dtype=torch.int64, device=dev))
    for i in range(5):
        step(i, st)
    torch.cuda.synchronize()
    st["acc"].zero_(); st["cnt"].zero_()
    t0 = time.perf_counter()
    for i in range(5, 5 + a.steps):
        step(i, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc, cnt = int(st["acc"]), st["cnt"].tolist()
    print(json.dumps({"workload": f"C3 dynamic tree N={N}, {B} sequences, k={a.lantern_k}, delta={a.lantern_delta}", "steps": a.steps,
                      "ms_per_step": 1e3 * dt / a.steps, "accepted_tokens_per_s": acc / dt, "mean_accept_length": acc / (a.steps * B),
                      "levels_tried_rejected_per_step": [c / (a.steps * B) for c in cnt], "status_errors": int(st["bad"]),
                      "note": "wall clock of the Python loop over the op wrappers (allocations and torch glue included); kernel-only times: rocprofv3"}))


if __name__ == "__main__":
    main()

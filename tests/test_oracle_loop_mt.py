"""CPU: lo_verify_loop_mt (the oracle's whole verify loop in one C call on N pthreads -- bench.py's CPU baseline, C leg) against
the same loop driven step by step through the oracle's per-function wrappers.  Test infrastructure checking test infrastructure:
the two must agree exactly, for any thread count."""
import numpy as np
import pytest

import cases as CS
import oracle


@pytest.mark.parametrize("threads", [1, 3])
def test_threaded_c_loop_equals_the_stepwise_oracle_loop(threads):
    m = CS.MODELS["lumina"]
    V, lo, hi = m["V"], m["img_lo"], m["img_hi"]
    W = hi - lo
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["tree_position_ids"], tree_attn_mask=tb["tree_attn_mask"],
                retrieve_indices=tb["retrieve_indices"])
    S, B, steps, Hd = 2, 4, 7, 16
    rs = np.random.RandomState(3)
    gs = [[CS.gen_static(500 + 10 * s + b, "lumina", bufs, sigma=2.0) for b in range(B)] for s in range(S)]
    R = gs[0][0]["orig_prob"].shape[0]
    op_off = gs[0][0]["op_off"]
    sst = np.stack([[g["ss_token"] for g in row] for row in gs])
    ssp = np.stack([[CS.ss_prob_from(g["orig_prob"], g["ss_token"]) for g in row] for row in gs]).astype(np.float32)
    f2b = lambda x: (np.ascontiguousarray(x, np.float32).view(np.uint32) >> 16).astype(np.uint16)      # truncation is fine: both loops read the same bits
    tgt = np.stack([[np.where(np.isfinite(g["node_logits"]), g["node_logits"], -40.0) for g in row] for row in gs]).astype(np.float32)
    cond, unc = f2b(tgt), f2b(np.zeros_like(tgt))
    orig = np.stack([[g["orig_prob"] for g in row] for row in gs]).astype(np.float32)
    hid = rs.randint(0, 60000, size=(S, B, 2, N, Hd)).astype(np.uint16)
    uni = rs.random_sample((B, 64 * steps + 64))
    ub = rs.random_sample((steps, B))
    first = rs.randint(lo, hi, size=B).astype(np.int64)
    table = CS.build_table(m["K"])
    cfg = oracle.EpConfig(mode=oracle.MODE_STATIC_LUMINA, syntax_shortcut=True, tok_offset=m["off"], img_lo=lo, img_hi=hi, syntax=m["syntax"],
                          lantern=True, k=40, delta=0.2)
    prompt, tpi, w, h = 5, 10 ** 6, 6, 6
    kv_shape = (4, 1, 2, 96, 8)
    slabs_c = [rs.randint(0, 60000, size=kv_shape).astype(np.uint16) for _ in range(2 * B)]
    slabs_py = [x.copy() for x in slabs_c]
    best, alen, tok = oracle.verify_loop_mt(cfg, tb, op_off, dict(ss_token=sst, ss_prob=ssp, cond=cond, uncond=unc, orig_win=np.ascontiguousarray(orig[..., lo:hi]),
                                                                   hidden=hid), uni, ub, first, table, steps, threads, 1.0, prompt, tpi, 100, w_latent=w,
                                            h_latent=h, newline_id=m["syntax"][2], eos_id=m["syntax"][0], win_lo=lo, slabs=slabs_c)
    ri = tb["retrieve_indices"].copy()
    ri[ri < 0] += N
    ri = ri.astype(np.int32)
    n_acc = 0
    for b in range(B):
        lens, cursor, t = [prompt + 3, 3], 0, int(first[b])
        for i in range(steps):
            s = i % S
            cand, cp, tc = oracle.gather_candidates(sst[s, b], ssp[s, b], t, tb["tree_indices"], tb["retrieve_indices"])
            proc = oracle.cfg_mask_topk(cond[s, b], unc[s, b], 1.0, model=oracle.MODEL_LUMINA, pos_ids=tb["tree_position_ids"] + 1 + lens[0],
                                        pos_base=prompt + 3, w=w, h=h, img_lo=lo, img_hi=hi, newline_id=m["syntax"][2], eos_id=m["syntax"][0],
                                        top_k=100, bf16=True)
            aux = oracle.StaticAux(cart_prob=cp, orig_prob=orig[s, b], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"],
                                   tree_cand=tc)
            bb, aa, sp, cnt = oracle.evaluate_posterior(cfg, proc, ri, cand, uni[b, cursor:cursor + 64], table=table, aux=aux)
            cursor += int(cnt[3])
            row = tb["retrieve_indices"][bb]
            for j in range(2):
                oracle.kv_gather(slabs_py[2 * b + j], row, aa + 1, lens[j])
            t = oracle.sample_inverse_cdf(sp, float(ub[i, b]))
            lens = [l + aa + 1 for l in lens]
            assert (int(best[i, b]), int(alen[i, b]), int(tok[i, b])) == (bb, aa, t), (b, i)
            n_acc += aa
    assert n_acc > 0
    for a, c in zip(slabs_c, slabs_py):
        assert np.array_equal(a, c)

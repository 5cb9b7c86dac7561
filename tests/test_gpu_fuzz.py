"""GPU (-m gpu): randomized differential test of evaluate_posterior -- both kernel sets against the oracle over a few hundred
seeded static-tree verify steps per model (reduced vocabularies of tests/golden/cases.py), batched 32 per launch the way the
serving loop batches sequences: every sequence has its own rows, candidates, drafter distributions and uniform stream.
Parameters per batch: model x tree x (lantern off | delta | lambda) x k x drafter noise."""
import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle
from lantern_amd import ops
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-5
B = 32


def cfgs(model, static, **kw):
    m = CS.MODELS[model]
    if model == "lumina":
        mk = lambda E: E.lumina(static, **kw)
    elif model == "llamagen":
        mk = lambda E: E.llamagen(static, **kw)
    else:
        mk = lambda E: E.anole(static, **kw)
    co, ch = mk(oracle.EpConfig), mk(ops.EpConfig)
    for c in (co, ch):          # reduced-vocabulary constants
        c.img_lo, c.img_hi, c.tok_offset = m["img_lo"], m["img_hi"], m["off"]
        if model == "lumina":
            c.syntax = tuple(m["syntax"])
    return co, ch


@pytest.mark.parametrize("model,tree,lantern,k,delta,sigma,seed", [
    ("lumina", "mc_sim_7b_63", True, 100, 0.1, 1.0, 1), ("lumina", "mc_sim_7b_63", True, 300, 5.0, 2.0, 2),
    ("lumina", "naive_extend_57", True, 10, 0.3, 0.5, 3), ("lumina", "mc_sim_7b_63", False, 1, 0.1, 3.0, 4),
    ("llamagen", "naive_extend_57", True, 50, 0.1, 1.0, 5), ("llamagen", "mc_sim_7b_63", True, 200, 10.0, 2.0, 6),
    ("anole", "naive_extend_57", True, 10, 5.0, 1.0, 7), ("anole", "naive_extend_57", True, 5, 20.0, 3.0, 8),
    ("anole", "mc_sim_7b_63", False, 1, 0.1, 0.5, 9), ("llamagen", "naive_extend_57", True, 1000, 0.05, 1.5, 10)])
def test_static_batches_vs_oracle(model, tree, lantern, k, delta, sigma, seed):
    _static_batches(model, tree, lantern, k, delta, sigma, seed, 1.0, 150)


@pytest.mark.parametrize("model,tree,lantern,k,delta,sigma,seed,top_p,top_k", [
    ("llamagen", "naive_extend_57", True, 50, 0.1, 1.0, 21, 0.9, 150), ("llamagen", "mc_sim_7b_63", False, 1, 0.1, 2.0, 22, 0.6, 0),
    ("anole", "naive_extend_57", True, 10, 5.0, 1.0, 23, 0.95, 150), ("anole", "mc_sim_7b_63", True, 20, 0.2, 3.0, 24, 0.3, 40),
    ("llamagen", "mc_sim_7b_63", True, 200, 10.0, 0.5, 25, 0.99, 0)])
def test_static_batches_with_top_p_vs_oracle(model, tree, lantern, k, delta, sigma, seed, top_p, top_k):
    """TopPLogitsWarper between the temperature and the top-k, per visited row inside the dense kernel AND inside the windowed chain kernel on
    logit rows (drafters/utils.py:36-52 applied at ea_model_llamagen.py:725,785), 32 sequences per launch against the oracle's sort-based restatement."""
    _static_batches(model, tree, lantern, k, delta, sigma, seed, top_p, top_k)


def _static_batches(model, tree, lantern, k, delta, sigma, seed, top_p, top_k):
    m = CS.MODELS[model]
    V, lo, W = m["V"], (m["img_lo"] if model != "llamagen" else 0), (m["img_hi"] - m["img_lo"] if model != "llamagen" else m["V"])
    tb = oracle.tree_static_build(H.tree_choices(tree))
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["tree_position_ids"], tree_attn_mask=tb["tree_attn_mask"],
                retrieve_indices=tb["retrieve_indices"])
    table = CS.build_table(m["K"])
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    gs = [CS.gen_static(100000 * seed + b, model, bufs, sigma=sigma) for b in range(B)]
    cands, cps, tcs = [], [], []
    for g in gs:
        c, cp, tc = oracle.gather_candidates(g["ss_token"], CS.ss_prob_from(g["orig_prob"], g["ss_token"]), g["sample_token"],
                                             tb["tree_indices"], tb["retrieve_indices"])
        cands.append(c); cps.append(cp); tcs.append(tc)
    co, ch = cfgs(model, True, lantern=lantern, k=k, delta=delta)
    if model != "lumina":       # LlamaGen / Anole: HF processors inside evaluate_posterior
        for c in (co, ch):
            c.temperature, c.top_k, c.top_p = 0.9, top_k, top_p
    nl = np.stack([g["node_logits"] for g in gs])
    aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack([g["orig_prob"] for g in gs])), op_off=dev(gs[0]["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)),
                        tree_cand=dev(np.stack(tcs)))
    uni = np.stack([g["uniforms"] for g in gs])
    tab = dev(table.view(np.int16))
    dense = ops.evaluate_posterior(ch, dev(nl), dev(ri), dev(np.stack(cands)), dev(uni), table=tab if lantern else None, aux=aux)
    # (round 5: logit-row windows apply TopPLogitsWarper inside the chain kernel too -- LANTERN_ROWS_LOGITS with prm.top_p)
    win = ops.evaluate_posterior_window(ch, V, dev(np.ascontiguousarray(nl[:, :, lo:lo + W])), lo, dev(ri), dev(np.stack(cands)), dev(uni),
                                        table=tab if lantern else None, aux=aux, want_dense=True)
    n_rej = n_acc = 0
    for b, g in enumerate(gs):
        a = oracle.StaticAux(cart_prob=cps[b], orig_prob=g["orig_prob"], op_off=g["op_off"], p_idx=tb["p_indices"], b_off=tb["b_off"],
                             b_idx=tb["b_idx"], tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(co, g["node_logits"], ri, cands[b], g["uniforms"], table=table if lantern else None, aux=a)
        n_rej += int(ocnt[2]); n_acc += oa
        forms = [("dense", dense[0], dense[1], dense[2], dense[3])]
        if win is not None:
            forms.append(("window", win["best"], win["accept_len"], win["sample_p"], win["counters"]))
        for name, best, alen, sp, cnt in forms:
            st = int(cnt[b, 5])
            if name == "window" and st == 6 and k >= m["K"] - 24:
                continue            # residual vanished (`gtp.sum()==0 -> ones`): only the dense set represents it
            assert st == 0, (name, b, st)
            assert (int(best[b]), int(alen[b])) == (ob, oa), (name, b, int(best[b]), int(alen[b]), ob, oa)
            assert np.array_equal(cnt[b, :5].cpu().numpy(), ocnt[:5]), (name, b)
            np.testing.assert_allclose(sp[b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
    assert n_rej > 0 and n_acc > 0      # the batch really exercises both outcomes

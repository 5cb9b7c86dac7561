"""CPU: the byte model behind `roofline.saturating` (bench.py / tools/ep_sweep.py).  The drafter row of a static-tree walk is priced once per
tree LEVEL with a rejection (all candidates of a level share their parent's original_prob row: ea_model_lumina_mgpt.py:697), which needs the
number of such levels per sequence: harness.static_rejection_levels replays the walk's control flow from the step's inputs and verdict.
Here it is held to the oracle (the checker): the replay's rejection total must equal the oracle's counter for every sequence, and the level
count must equal an instrumented scalar replay of the reference's loop."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def _levels_with_rejection_scalar(cand, cart, best, alen, levels):
    """The reference's loop order (ea_model_lumina_mgpt.py:640-700) in plain Python, control flow only."""
    P, D = cand.shape
    acc = cand[best]
    n_lv, n_rej = 0, 0
    for i in range(1, levels + 1):
        tried, rej = [], 0
        for j in range(P):
            if not all(cand[j, t] == acc[t] for t in range(i)):
                continue
            x = cand[j, i]
            if x == -1 or x in tried:
                continue
            tried.append(x)
            if cart[j, i] <= 0:
                continue
            if i <= alen and x == acc[i]:
                break
            rej += 1
        n_rej += rej
        n_lv += rej > 0
    return n_lv, n_rej


def test_static_rejection_levels_against_the_oracle():
    import cases as CS
    import oracle
    from lantern_amd import harness as HN
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["tree_position_ids"], tree_attn_mask=tb["tree_attn_mask"],
                retrieve_indices=tb["retrieve_indices"])
    m = CS.MODELS["lumina"]
    N = len(tb["tree_indices"])
    ri = tb["retrieve_indices"].copy()
    ri[ri < 0] += N
    cfg = oracle.EpConfig(mode=oracle.MODE_STATIC_LUMINA, syntax_shortcut=True, tok_offset=4, img_lo=4, img_hi=m["img_hi"],
                          syntax=m["syntax"], lantern=True, k=50, delta=0.1)
    table = CS.build_table(m["K"])
    cands, carts, bests, alens, cnts, want = [], [], [], [], [], []
    for seed in range(48):
        g = CS.gen_static(7000 + seed, "lumina", bufs, sigma=(0.5, 1.5, 4.0)[seed % 3])
        ssp = CS.ss_prob_from(g["orig_prob"], g["ss_token"])
        cand, cp, tc = oracle.gather_candidates(g["ss_token"], ssp, g["sample_token"], tb["tree_indices"], tb["retrieve_indices"])
        aux = oracle.StaticAux(cart_prob=cp, orig_prob=g["orig_prob"], op_off=g["op_off"], p_idx=tb["p_indices"], b_off=tb["b_off"],
                               b_idx=tb["b_idx"], tree_cand=tc)
        best, alen, _, cnt = oracle.evaluate_posterior(cfg, g["node_logits"], ri.astype(np.int32), cand, g["uniforms"], table=table, aux=aux)
        lv, rj = _levels_with_rejection_scalar(cand, cp, best, alen, int(cnt[0]))
        assert rj == int(cnt[2])
        cands.append(cand); carts.append(cp); bests.append(best); alens.append(alen); cnts.append(np.asarray(cnt)); want.append(lv)
    got = HN.static_rejection_levels(torch.from_numpy(np.stack(cands)), torch.from_numpy(np.stack(carts)), torch.tensor(bests), torch.tensor(alens),
                                     torch.from_numpy(np.stack(cnts)))
    assert got.tolist() == want
    assert 0 < sum(want) < int(np.stack(cnts)[:, 2].sum())          # several rejections share a level somewhere: the two models differ

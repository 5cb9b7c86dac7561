"""CPU: the oracle (oracle/lantern_oracle.c) against the golden vectors captured from the
reference's own functions (tests/golden/make_golden.py).  Integers bit-exact; probabilities
within 1e-6 (the reference's torch-CPU expf/sum differ from libm in the last ulp)."""
import numpy as np
import pytest

import cases as CS
import helpers as H
import oracle

SPECS = H.ep_specs()
TREES = [str(x) for x in H.load("trees.npz")["names"]] + [str(x) for x in H.load("trees_random.npz")["names"]]


@pytest.mark.parametrize("name", TREES)
def test_static_tree_buffers(name):
    g = H.tree_buffers(name)
    o = oracle.tree_static_build(H.tree_choices(name))
    assert np.array_equal(o["tree_attn_mask"], g["mask"])
    assert np.array_equal(o["tree_indices"], g["tree_indices"])
    assert np.array_equal(o["tree_position_ids"], g["pos"])
    assert np.array_equal(o["retrieve_indices"], g["retrieve"])
    assert np.array_equal(o["p_indices"], g["p_indices"])
    assert np.array_equal(o["b_off"], g["b_off"])
    assert np.array_equal(o["b_idx"], g["b_idx"])


@pytest.mark.parametrize("name", TREES)
def test_drafter_tree_buffers(name):
    g = H.tree_buffers(name)
    o = oracle.tree_drafter_build(H.tree_choices(name))
    L = int(g["d_levels"][0])
    assert len(o["tree_indices"]) == L
    for l in range(L):
        assert np.array_equal(o["attn_mask"][l], g[f"d_mask{l}"])
        assert np.array_equal(o["tree_indices"][l], g[f"d_ti{l}"])
        assert o["repeat_nums"][l] == g[f"d_rep{l}"].tolist()


def _check_ep(best, alen, sp, cnt, case):
    assert best == int(case["best"])
    assert alen == int(case["accept_len"])
    assert cnt[3] == int(case["n_draws"])
    np.testing.assert_allclose(sp, case["sample_p"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "static"])
def test_evaluate_posterior_static(i):
    spec, case = SPECS[i], H.ep_case(i)
    tb, g = H.static_inputs(spec, case)
    # O6 first: candidate assembly must reproduce what the reference fed to evaluate_posterior
    cand, cprob, tcand = oracle.gather_candidates(case["ss_token"], case["ss_prob"], int(case["sample_token"]),
                                                  tb["tree_indices"], tb["retrieve"])
    assert np.array_equal(cand, case["cand"])
    assert np.array_equal(tcand, case["tree_cand"])
    assert np.array_equal(cprob, case["cart_prob"])
    np.testing.assert_allclose(CS.ss_prob_from(g["orig_prob"], g["ss_token"]), case["ss_prob"], atol=0)
    N = len(tb["tree_indices"])
    m = CS.MODELS[spec["model"]]
    best, alen, sp, cnt = oracle.evaluate_posterior(
        H.ep_config(spec), g["node_logits"], H.row_index_from_retrieve(tb["retrieve"], N), case["cand"],
        case["uniforms"], table=H.table(m["K"]), aux=H.static_aux(tb, g, case))
    _check_ep(best, alen, sp, cnt, case)


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "dynamic"])
def test_evaluate_posterior_dynamic(i):
    spec, case = SPECS[i], H.ep_case(i)
    nl, uniforms = H.dynamic_node_logits(spec, case)
    N = len(case["draft_tokens"])
    m = CS.MODELS[spec["model"]]
    best, alen, sp, cnt = oracle.evaluate_posterior(
        H.ep_config(spec), nl, H.row_index_from_retrieve(case["retrieve"], N), case["cand"], uniforms,
        table=H.table(m["K"]))
    _check_ep(best, alen, sp, cnt, case)


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "greedy"])
def test_evaluate_posterior_greedy(i):
    spec, case = SPECS[i], H.ep_case(i)
    nl, _ = H.dynamic_node_logits(spec, case, greedy=True)
    N = len(case["draft_tokens"])
    m = CS.MODELS[spec["model"]]
    best, alen, row = oracle.evaluate_posterior_greedy(
        nl, H.row_index_from_retrieve(case["retrieve"], N), case["cand"], lantern=spec["lantern"], k=spec["k"],
        delta=spec["delta"], tok_offset=m["off"], table=H.table(m["K"]))
    assert (best, alen) == (int(case["best"]), int(case["accept_len"]))
    assert np.array_equal(row, case["out_row"])


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] in ("dynamic", "greedy")])
def test_dynamic_tree_expand_and_finalize(i):
    """O3 + O4: replay the scripted drafter logits through expand_dynamic / tree_dynamic_finalize
    and compare with what the reference's topK_genrate returned."""
    spec, case = SPECS[i], H.ep_case(i)
    depth = int(case["depth"])
    script = H.dynamic_script(spec["seed"], spec["model"], depth)
    assert abs(sum(CS.checksum(s) for s in script) - float(case["chk_script"])) < 1e-6
    k = CS.TOPK
    ti, cu, ci, scores = oracle.expand_dynamic(H.hf_process_rows(script[0][None], H.DYN_TOP_K), None, k)
    scores_list, tokens_list, parents_list = [cu.reshape(-1)], [ti.reshape(-1)], [np.zeros(1, np.int64)]
    topk_cs_index = np.arange(k)
    for d in range(depth):
        bias = 1 + k * k * max(0, d - 1) + (k if d > 0 else 0)
        parents_list.append(topk_cs_index + bias)
        ti, cu, ci, scores = oracle.expand_dynamic(H.hf_process_rows(script[d + 1], H.DYN_TOP_K), scores, k)
        topk_cs_index = ci
        scores_list.append(cu.reshape(-1))
        tokens_list.append(ti.reshape(-1))
    draft, retrieve, mask, pos = oracle.tree_dynamic_finalize(
        np.concatenate(scores_list), np.concatenate(tokens_list), np.concatenate(parents_list), k,
        int(case["total_tokens"]), int(case["sample_token"]), sort_rows=True)
    assert np.array_equal(draft, case["draft_tokens"])
    assert np.array_equal(retrieve, case["retrieve"])
    assert np.array_equal(mask, case["mask"])
    assert np.array_equal(pos, case["pos"])


def _bf16_bits(x):
    import torch
    return torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_cfg_mask_topk(tag):
    g = H.load("o7.npz")
    m = CS.MODELS["lumina"]
    cond, unc = g["cond"], g["uncond"]
    bf = tag == "bf16"
    if bf:
        cond, unc = _bf16_bits(cond), _bf16_bits(unc)
    kw = dict(w=int(g["w"]), h=int(g["h"]), img_lo=m["img_lo"], img_hi=m["img_hi"], newline_id=m["syntax"][2],
              eos_id=m["syntax"][0], bf16=bf)
    out = oracle.cfg_mask_topk(cond, unc, 3.0, model=oracle.MODEL_LUMINA, pos_ids=g["pos"],
                               pos_base=int(g["img_start"]) + 3, top_k=100, **kw)
    assert np.array_equal(out, g[f"lumina_{tag}"])
    out = oracle.cfg_mask_topk(cond, unc, 3.0, model=oracle.MODEL_ANOLE, **kw)
    assert np.array_equal(out, g[f"anole_{tag}"])
    out = oracle.cfg_mask_topk(cond, unc, 3.0, model=oracle.MODEL_PLAIN, **kw)
    assert np.array_equal(out, g[f"plain_{tag}"])


def test_kv_and_hidden_gather():
    g = H.load("kv.npz")
    slab = g["before"].copy()
    best, alen, prev = int(g["best"]), int(g["accept_len"]), int(g["prev"])
    row = g["retrieve"][best]
    oracle.kv_gather(slab, row, alen + 1, prev)
    assert np.array_equal(slab, g["after"])
    assert np.all(g["current_length"] == prev + alen + 1)
    out = oracle.hidden_gather(g["hidden"], row, alen + 1)
    assert np.array_equal(out, g["accept_hidden"])
    # bonus token: the reference drew from a one-hot sample_p -> token 7 regardless of RNG
    p = np.zeros(20, np.float32)
    p[7] = 1.0
    assert oracle.sample_inverse_cdf(p, 0.3) == int(g["token"].reshape(-1)[0]) == 7


def test_sample_static():
    g = H.load("sample.npz")
    out = oracle.sample_static(g["full"], g["idx"])
    np.testing.assert_allclose(out, g["prob"], rtol=0, atol=1e-7)


def test_codebook_table():
    """The reference sorts float32 cdist values; two codes whose float64 distances differ by
    < 1e-6 relative are a tie in float32 and may come out in either order.  Everything else
    must be identical."""
    g = H.load("codebook.npz")
    t = oracle.build_vq_table(g["codebook"])
    r = g["table"]
    cb = g["codebook"].astype(np.float64)
    d = np.sqrt(((cb[:, None] - cb[None]) ** 2).sum(-1))
    bad = np.argwhere(t != r)
    assert len(bad) <= 0.001 * t.size
    for a, c in bad:
        assert abs(d[a, t[a, c]] - d[a, r[a, c]]) <= 1e-6 * d[a, t[a, c]]
    assert np.array_equal(np.sort(t, axis=1), np.sort(r.astype(np.uint16), axis=1))


def test_inverse_cdf_distribution():
    rs = np.random.RandomState(0)
    p = rs.random_sample(16).astype(np.float32)
    p[3] = 0
    p /= p.sum()
    us = (np.arange(20000) + 0.5) / 20000
    toks = np.array([oracle.sample_inverse_cdf(p, u) for u in us])
    freq = np.bincount(toks, minlength=16) / len(us)
    np.testing.assert_allclose(freq, p, atol=2e-4)
    assert freq[3] == 0

"""GPU (-m gpu): the HIP kernel sets -- dense, windowed chain, node-parallel, greedy, O7 dense + windowed, the table
builder -- against REFERENCE runs at BASELINE's real sizes (tests/golden/make_golden_fullsize.py: V = 65536 / K = 8192 /
k = 1000 for Lumina and Anole, V = K = 16384 for LlamaGen; the neighbour table by the generate_codebook.py recipe at its
real shape).  Integers bit-exact, probabilities within 1e-5 (north_star)."""
import hashlib

import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle
from lantern_amd import ops
from test_gpu_parity import dev, hip_cfg
from test_oracle_fullsize import o7_full_check, o7_full_inputs

pytestmark = pytest.mark.gpu
SPECS = H.full_specs()
PROB_TOL = 1e-5
_tab = {}


def _ids(kind, pred=lambda s: True):
    return [i for i, s in enumerate(SPECS) if s["kind"] == kind and pred(s)]


def table_dev(spec, packed_for_k=None):
    m = CS.model_dims(spec)
    key = (m["K"], m["C"], packed_for_k)
    if key not in _tab:
        if packed_for_k is None:
            _tab[key] = dev(H.table_for(spec).view(np.int16))
        else:
            _tab[key] = ops.pack_vq_table(table_dev(spec), -(-(packed_for_k + 1) // 8) * 8)
    return _tab[key]


def window_of(spec):
    m = CS.model_dims(spec)
    return (0, m["V"]) if spec["model"] == "llamagen" else (m["img_lo"], m["img_hi"] - m["img_lo"])


def static_aux(tb, g, case, orig=None):
    return ops.StaticAux(cart_prob=dev(case["cart_prob"])[None], orig_prob=dev(g["orig_prob"] if orig is None else orig)[None],
                         op_off=dev(g["op_off"]), p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]),
                         b_idx=dev(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)), tree_cand=dev(case["tree_cand"])[None])


def check(best, alen, cnt, sp, case):
    assert int(cnt[0, 5]) == 0, f"status {int(cnt[0, 5])}"
    assert (int(best[0]), int(alen[0])) == (int(case["best"]), int(case["accept_len"]))
    assert int(cnt[0, 3]) == int(case["n_draws"])
    sp = sp[0].cpu().numpy()
    np.testing.assert_allclose(sp, case["sample_p"], rtol=0, atol=PROB_TOL)
    assert abs(float(sp.astype(np.float64).sum()) - float(case["sample_p_sum"])) < 1e-4
    assert np.array_equal(np.flatnonzero(sp > 1e-6), np.flatnonzero(case["sample_p"] > 1e-6))


# ------------------------------------------------------------------------------------------------ dense kernel set

@pytest.mark.parametrize("i", _ids("static"))
def test_dense_static_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    tb, g = H.static_inputs(spec, case)
    N = len(tb["tree_indices"])
    best, alen, sp, cnt = ops.evaluate_posterior(hip_cfg(spec), dev(g["node_logits"])[None], dev(H.row_index_from_retrieve(tb["retrieve"], N)),
                                                 dev(case["cand"])[None], dev(case["uniforms"])[None], table=table_dev(spec),
                                                 aux=static_aux(tb, g, case))
    check(best, alen, cnt, sp, case)


@pytest.mark.parametrize("i", _ids("dynamic"))
def test_dense_dynamic_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    nl, uniforms = H.dynamic_node_logits(spec, case)
    N = len(case["draft_tokens"])
    best, alen, sp, cnt = ops.evaluate_posterior(hip_cfg(spec), dev(nl)[None], dev(H.row_index_from_retrieve(case["retrieve"], N)),
                                                 dev(case["cand"])[None], dev(uniforms)[None], table=table_dev(spec))
    check(best, alen, cnt, sp, case)


# ------------------------------------------------------------------------------------------ windowed chain kernels

def _window_rows(spec, nl):
    lo, W = window_of(spec)
    if spec["model"] != "llamagen":
        # Lumina: -inf outside the image range by construction; Anole: the same rows (the reference masks with finfo.min, zero mass)
        assert not np.isfinite(np.delete(nl, np.s_[lo:lo + W], axis=1)).any()
    return lo, W, np.ascontiguousarray(nl[:, lo:lo + W])


@pytest.mark.parametrize("i", _ids("static"))
def test_window_static_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    tb, g = H.static_inputs(spec, case)
    m = CS.model_dims(spec)
    lo, W, rows = _window_rows(spec, g["node_logits"])
    N = len(tb["tree_indices"])
    u = 0.1 + 0.8 * ((i * 37) % 100) / 100.0
    ri = dev(H.row_index_from_retrieve(tb["retrieve"], N))
    out = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(rows)[None], lo, ri, dev(case["cand"])[None], dev(case["uniforms"])[None],
                                        table=table_dev(spec), aux=static_aux(tb, g, case), u_bonus=dev(np.array([u])), want_dense=True)
    check(out["best"], out["accept_len"], out["counters"], out["sample_p"], case)
    assert int(out["token"][0]) == oracle.sample_inverse_cdf(out["sample_p"][0].cpu().numpy(), u)
    # the hot-path layout: windowed drafter pool + the packed table [K, ceil8(k+1)] -> the same bits
    if spec["lantern"]:
        aux = static_aux(tb, g, case, orig=np.ascontiguousarray(g["orig_prob"][:, lo:lo + W]))
        out2 = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(rows)[None], lo, ri, dev(case["cand"])[None], dev(case["uniforms"])[None],
                                             table=table_dev(spec, spec["k"]), aux=aux, orig_windowed=True, u_bonus=dev(np.array([u])),
                                             want_dense=True)
        for key in ("best", "accept_len", "counters", "sample_p", "token"):
            assert torch.equal(out2[key], out[key]), key


@pytest.mark.parametrize("i", _ids("dynamic"))
def test_window_dynamic_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    nl, uniforms = H.dynamic_node_logits(spec, case)
    m = CS.model_dims(spec)
    lo, W, rows = _window_rows(spec, nl)
    N = len(case["draft_tokens"])
    u = 0.05 + 0.9 * ((i * 53) % 100) / 100.0
    out = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(rows)[None], lo, dev(H.row_index_from_retrieve(case["retrieve"], N)),
                                        dev(case["cand"])[None], dev(uniforms)[None], table=table_dev(spec), u_bonus=dev(np.array([u])),
                                        want_dense=True)
    check(out["best"], out["accept_len"], out["counters"], out["sample_p"], case)
    assert int(out["token"][0]) == oracle.sample_inverse_cdf(out["sample_p"][0].cpu().numpy(), u)


# ---------------------------------------------------------------- probability rows (what the timed loop feeds) + node kernels

def _prob_rows(spec, rows, lo, W):
    T, tk = (1.0, 0) if spec["model"] == "lumina" else (spec.get("temperature", 1.0), spec.get("top_k", 0))
    full = np.full((rows.shape[0], CS.model_dims(spec)["V"]), -np.inf, np.float32)
    full[:, lo:lo + W] = rows
    pr, _ = ops.cfg_mask_topk_window(dev(full), None, 1.0, lo, W, model=ops.MODEL_PLAIN if spec["model"] == "llamagen" else ops.MODEL_ANOLE,
                                     img_lo=lo, img_hi=lo + W, top_k=tk, temperature=T, probs=True)
    return pr


@pytest.mark.parametrize("i", _ids("static"))
def test_probability_rows_chain_and_nodes_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    tb, g = H.static_inputs(spec, case)
    m = CS.model_dims(spec)
    lo, W, rows = _window_rows(spec, g["node_logits"])
    N = len(tb["tree_indices"])
    pr = _prob_rows(spec, rows, lo, W)
    cfg = hip_cfg(spec)
    cfg.temperature, cfg.top_k, cfg.top_p = 1.0, 0, 1.0
    nt = ops.tree_node_tables(tb["retrieve"], N, tb["p_indices"], tb["b_off"], g["op_off"], device="cuda")
    u = 0.1 + 0.8 * ((i * 41) % 100) / 100.0
    args = (cfg, m["V"], pr[None], lo, dev(H.row_index_from_retrieve(tb["retrieve"], N)), dev(case["cand"])[None], dev(case["uniforms"])[None])
    kw = dict(table=table_dev(spec, spec["k"]) if spec["lantern"] else None, aux=static_aux(tb, g, case), u_bonus=dev(np.array([u])),
              want_dense=True, rows_probs=True)
    chain = ops.evaluate_posterior_window(*args, **kw)
    check(chain["best"], chain["accept_len"], chain["counters"], chain["sample_p"], case)
    node = ops.evaluate_posterior_window(*args, nodes=nt, **kw)
    if int(node["counters"][0, 5]) == 8:          # duplicate sibling tokens: the node view does not hold and the kernel says so
        return
    check(node["best"], node["accept_len"], node["counters"], node["sample_p"], case)
    for key in ("best", "accept_len", "counters", "token", "sample_p"):
        assert torch.equal(node[key], chain[key]), key


# -------------------------------------------------------------------------------------------------------- greedy (a9)

@pytest.mark.parametrize("i", _ids("greedy"))
def test_greedy_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    nl, _ = H.dynamic_node_logits(spec, case, greedy=True)
    m = CS.model_dims(spec)
    lo, W = window_of(spec)
    N = len(case["draft_tokens"])
    best, alen, row = ops.evaluate_posterior_greedy(dev(nl)[None], dev(H.row_index_from_retrieve(case["retrieve"], N)), dev(case["cand"])[None],
                                                    lantern=spec["lantern"], k=spec["k"], delta=spec["delta"], tok_offset=m["off"],
                                                    table=table_dev(spec), win_lo=lo, win_len=W)
    assert (int(best[0]), int(alen[0])) == (int(case["best"]), int(case["accept_len"]))
    assert hashlib.sha256(row[0].cpu().numpy().tobytes()).hexdigest() == str(case["out_row_sha"])


# ------------------------------------------------------------------------------------------------------------ O7

@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_cfg_mask_topk_full_size_reference(tag):
    d, cond, unc = o7_full_inputs()
    m = CS.FULL["lumina"]
    dt = torch.float32 if tag == "f32" else torch.bfloat16
    c, u = torch.from_numpy(cond).to(dt).cuda(), torch.from_numpy(unc).to(dt).cuda()
    kw = dict(model=ops.MODEL_LUMINA, pos_ids=dev(d["o7.pos"]), pos_base=int(d["o7.img_start"]) + 3, top_k=2000, w=48, h=48,
              img_lo=m["img_lo"], img_hi=m["img_hi"], newline_id=m["syntax"][2], eos_id=m["syntax"][0])
    out = ops.cfg_mask_topk(c, u, 3.0, **kw)
    o7_full_check(out.cpu().numpy(), d, tag)
    # windowed form: the window of grid rows, row_hot for the forced rows
    lo, W = m["img_lo"], m["img_hi"] - m["img_lo"]
    win, hot = ops.cfg_mask_topk_window(c, u, 3.0, lo, W, **kw)
    dense = np.full((12, m["V"]), -np.inf, np.float32)
    win, hot = win.cpu().numpy(), hot.cpu().numpy()
    for r in range(12):
        if hot[r] >= 0:
            dense[r, hot[r]] = out[r, hot[r]].item()
        else:
            dense[r, lo:lo + W] = win[r]
    o7_full_check(dense, d, tag)


# ----------------------------------------------------------------------------------------------- table builder (8f-1)

@pytest.mark.parametrize("model", ["lumina", "llamagen"])
def test_vq_table_builder_at_the_real_codebook_sizes(model):
    spec = dict(model=model, size="full")
    m = CS.FULL[model]
    want = H.table_for(spec)                       # == the reference recipe in float64 (recorded in the fixture)
    got = ops.build_vq_table(dev(CS.full_codebook(m["K"], m["C"]))).cpu().numpy().view(np.uint16)
    same = float((got == want).mean())
    assert same > 0.999999, same                   # exact distance ties may come out in either order
    bad = np.flatnonzero((got != want).any(1))
    assert all(np.array_equal(np.sort(got[r]), np.sort(want[r])) for r in bad)


# ------------------------------------------------------------------------------------------------ tree building + commit
@pytest.mark.parametrize("i", _ids("dynamic"))
def test_dynamic_tree_kernels_at_full_vocabulary(i):
    """O3 + O4: expand_dynamic / tree_dynamic_finalize on [10, 65536] / [10, 16384] drafter rows against the tree the reference's
    topK_genrate built at that vocabulary (cnets_lumina_mgpt.py:803-933)."""
    spec, case = SPECS[i], H.full_case(i)
    depth, k = int(case["depth"]), CS.TOPK
    script = H.dynamic_script(spec["seed"], spec["model"], depth, m=CS.model_dims(spec))
    ti, cu, ci, scores = ops.expand_dynamic(dev(H.hf_process_rows(script[0][None], H.DYN_TOP_K))[None], None, k)
    scores_list, tokens_list = [cu.reshape(-1)], [ti.reshape(-1)]
    parents_list = [torch.zeros(1, dtype=torch.int64, device="cuda")]
    topk_cs_index = torch.arange(k, device="cuda")
    for d in range(depth):
        parents_list.append(topk_cs_index + (1 + k * k * max(0, d - 1) + (k if d > 0 else 0)))
        ti, cu, ci, scores = ops.expand_dynamic(dev(H.hf_process_rows(script[d + 1], H.DYN_TOP_K))[None], scores, k)
        topk_cs_index = ci[0]
        scores_list.append(cu.reshape(-1))
        tokens_list.append(ti.reshape(-1))
    draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(
        torch.cat(scores_list)[None], torch.cat(tokens_list)[None], torch.cat(parents_list)[None],
        torch.tensor([int(case["sample_token"])], device="cuda"), k, int(case["total_tokens"]), sort_rows=True)
    nl, md = int(nl[0]), int(md[0])
    assert np.array_equal(draft[0].cpu().numpy(), case["draft_tokens"])
    assert np.array_equal(ret[0, :nl, :md].cpu().numpy(), case["retrieve"])
    assert torch.all(ret[0, nl:] == -1) and torch.all(ret[0, :, md:] == -1)
    assert np.array_equal(mask[0].cpu().numpy(), case["mask"]) and np.array_equal(pos[0].cpu().numpy(), case["pos"])


@pytest.mark.parametrize("i", _ids("static")[:3])
def test_gather_candidates_full_size(i):
    """O2 with the reference's real-size token ids (offset + 4, ids up to 8195 / 16383)."""
    spec, case = SPECS[i], H.full_case(i)
    tb = H.tree_buffers(spec["tree"])
    cand, cp, tc = ops.gather_candidates(dev(case["ss_token"])[None], dev(case["ss_prob"])[None],
                                         dev(np.array([int(case["sample_token"])])), dev(tb["tree_indices"]), dev(tb["retrieve"]))
    assert np.array_equal(cand[0].cpu().numpy(), case["cand"])
    assert np.array_equal(cp[0].cpu().numpy(), case["cart_prob"]) and np.array_equal(tc[0].cpu().numpy(), case["tree_cand"])


@pytest.mark.parametrize("fused", [False, True])
def test_kv_commit_at_the_7b_slab_geometry(fused):
    """O9 (+ O10 when fused): the reference's update_inference_inputs on its own [64, 2, 32, 96, 128] slab (kv_cache.py:101-122),
    compared through the SHA-256 of the whole moved slab."""
    from test_oracle_fullsize import kv_full_data
    d, slab_np = kv_full_data()
    best_i, a, prev = int(d["kv.best"]), int(d["kv.accept_len"]), int(d["kv.prev"])
    slab = dev(slab_np)
    best, alen = dev(np.array([best_i], np.int32)), dev(np.array([a], np.int32))
    seq, prv, ret = dev(np.zeros(1, np.int32)), dev(np.array([prev], np.int64)), dev(d["kv.retrieve"])
    if fused:
        hid = dev(np.stack([d["kv.hidden"][0], d["kv.hidden"][0]])[None])          # [B=1, G=2, N, H]
        new_len, out_h, acc = ops.update_inference_inputs([slab], seq, prv, ret, best, alen, hid, dev(d["kv.cand"])[None])
        n = a + 1
        assert np.array_equal(out_h.cpu().numpy()[0, 0, :n], d["kv.accept_hidden"][0])
        assert np.array_equal(out_h.cpu().numpy()[0, 1, :n], d["kv.accept_hidden"][0])
        assert np.array_equal(acc.cpu().numpy()[0, :n], d["kv.new_ids_tail"]) and np.all(acc.cpu().numpy()[0, n:] == -1)
    else:
        new_len = ops.kv_gather([slab], seq, prv, ret, best, alen)
    assert int(new_len[0]) == int(d["kv.current_length"][0]) == prev + a + 1
    assert hashlib.sha256(slab.cpu().numpy().tobytes()).hexdigest() == str(d["kv.after_sha256"])
    # the bonus token (ea_model_lumina_mgpt.py:779-790): inverse CDF of the one-hot residual the reference sampled from
    p = np.zeros((1, 65536), np.float32)
    p[0, 4321] = 1.0
    hid1 = dev(d["kv.hidden"])[None]
    out_h, acc, tok = ops.accept_gather(hid1, ret, dev(d["kv.cand"])[None], best, alen, sample_p=dev(p), u=dev(np.array([0.77])))
    assert int(tok[0]) == int(d["kv.token"].reshape(-1)[0]) == 4321
    assert np.array_equal(out_h.cpu().numpy()[0, 0, :a + 1], d["kv.accept_hidden"][0])

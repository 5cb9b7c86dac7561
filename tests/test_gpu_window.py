"""GPU (-m gpu): the windowed (v2) kernels -- O7w, O8w (+ fused bonus-token draw), window->dense -- against the
golden vectors and the oracle.  Same bar as the dense kernels: integers bit-exact, probabilities <= 1e-5."""
import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle
from lantern_amd import ops
from test_gpu_parity import dev, hip_cfg, table_dev, _bf16


def _supported(spec):
    """Every reference case runs on the windowed kernels (round 5: logit-row windows apply TopPLogitsWarper inside the chain kernel --
    LANTERN_ROWS_LOGITS with prm.top_p; before, those cases ran on the dense kernel only)."""
    return True


pytestmark = pytest.mark.gpu
SPECS = H.ep_specs()
PROB_TOL = 1e-5


def window_of(model):
    m = CS.MODELS[model]
    return (m["img_lo"], m["img_hi"] - m["img_lo"]) if model != "llamagen" else (0, m["V"])


def check_out(out, case, V, lo, u=None, spec=None):
    if int(out["counters"][0, 5]) == 6:
        # LANTERN_ST_NEEDS_DENSE: the residual was zeroed completely (`gtp.sum()==0 -> ones`, uniform over all V).
        # Only reachable when k+1 neighbours cover the whole codebook (the k=1022 / K=1024 reduced-vocabulary cases);
        # the dense kernel represents it (test_gpu_parity covers the same case), the windowed one reports it.
        assert spec is not None and spec["k"] >= CS.MODELS[spec["model"]]["K"] - 2
        return
    assert int(out["counters"][0, 5]) == 0, int(out["counters"][0, 5])
    assert int(out["best"][0]) == int(case["best"])
    assert int(out["accept_len"][0]) == int(case["accept_len"])
    assert int(out["counters"][0, 3]) == int(case["n_draws"])
    np.testing.assert_allclose(out["sample_p"][0].cpu().numpy(), case["sample_p"], rtol=0, atol=PROB_TOL)
    d2 = ops.window_to_dense(out["sample_win"], out["out_tok"], out["out_mass"], V, lo)
    assert torch.equal(d2, out["sample_p"])
    if u is not None:
        assert int(out["token"][0]) == oracle.sample_inverse_cdf(out["sample_p"][0].cpu().numpy(), u)


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "static" and _supported(s)])
def test_window_static_golden(i):
    spec, case = SPECS[i], H.ep_case(i)
    tb, g = H.static_inputs(spec, case)
    m = CS.MODELS[spec["model"]]
    lo, W = window_of(spec["model"])
    N = len(tb["tree_indices"])
    nl = g["node_logits"]
    assert not np.isfinite(np.delete(nl, np.s_[lo:lo + W], axis=1)).any()          # rows really are window rows
    aux = ops.StaticAux(cart_prob=dev(case["cart_prob"])[None], orig_prob=dev(g["orig_prob"])[None], op_off=dev(g["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]),
                        b_idx=dev(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)), tree_cand=dev(case["tree_cand"])[None])
    u = 0.1 + 0.8 * ((i * 37) % 100) / 100.0
    out = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(nl[:, lo:lo + W])[None], lo, dev(H.row_index_from_retrieve(tb["retrieve"], N)),
                                        dev(case["cand"])[None], dev(case["uniforms"])[None], table=table_dev(m["K"]), aux=aux,
                                        u_bonus=dev(np.array([u])), want_dense=True)
    check_out(out, case, m["V"], lo, u, spec)
    # windowed drafter pool gives the same answer
    aux.orig_prob = dev(np.ascontiguousarray(g["orig_prob"][:, lo:lo + W]))[None]
    out2 = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(nl[:, lo:lo + W])[None], lo, dev(H.row_index_from_retrieve(tb["retrieve"], N)),
                                         dev(case["cand"])[None], dev(case["uniforms"])[None], table=table_dev(m["K"]), aux=aux,
                                         orig_windowed=True, want_dense=True)
    assert torch.equal(out2["sample_p"], out["sample_p"]) and int(out2["best"][0]) == int(out["best"][0])
    # packed neighbour table (16-byte aligned rows of ceil8(k+1) ids): same answer through the 16-byte staging path
    if spec["lantern"] and spec["k"] + 1 <= 1016:
        packed = ops.pack_vq_table(table_dev(m["K"]), -(-(spec["k"] + 1) // 8) * 8)
        out3 = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(nl[:, lo:lo + W])[None], lo, dev(H.row_index_from_retrieve(tb["retrieve"], N)),
                                             dev(case["cand"])[None], dev(case["uniforms"])[None], table=packed, aux=aux, orig_windowed=True,
                                             want_dense=True)
        for key in ("best", "accept_len", "counters", "sample_p"):
            assert torch.equal(out3[key], out2[key]), key


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "dynamic" and _supported(s)])
def test_window_dynamic_golden(i):
    spec, case = SPECS[i], H.ep_case(i)
    nl, uniforms = H.dynamic_node_logits(spec, case)
    m = CS.MODELS[spec["model"]]
    lo, W = window_of(spec["model"])
    N = len(case["draft_tokens"])
    u = 0.05 + 0.9 * ((i * 53) % 100) / 100.0
    out = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(np.ascontiguousarray(nl[:, lo:lo + W]))[None], lo,
                                        dev(H.row_index_from_retrieve(case["retrieve"], N)), dev(case["cand"])[None], dev(uniforms)[None],
                                        table=table_dev(m["K"]), u_bonus=dev(np.array([u])), want_dense=True)
    check_out(out, case, m["V"], lo, u, spec)


@pytest.mark.parametrize("model,with_counts", [("lumina", True), ("lumina", False), ("anole", True), ("llamagen", True)])
def test_window_dynamic_ragged_batch(model, with_counts):
    """One launch over dynamic trees of DIFFERENT shapes (leaf count, depth): candidates / row maps padded with -1 to the
    widest tree, per-sequence row_index [B,P,D], optional n_paths / n_depth.  Every sequence must land on its own golden."""
    idx = [i for i, s in enumerate(SPECS) if s["kind"] == "dynamic" and s["model"] == model and _supported(s) and s["lantern"]
           and s["k"] == 10 and s["delta"] == 0.1 and not s.get("special") and s.get("top_p", 1.0) >= 1.0
           and s.get("temperature", 1.0) == 1.0 and s.get("top_k", 0) == 0]
    idx += [i for i, s in enumerate(SPECS) if s["kind"] == "dynamic" and s["model"] == model and _supported(s) and s["lantern"]
            and s["k"] == 300 and s["delta"] == 0.1 and not s.get("special")][:1]
    cases = [(SPECS[i], H.ep_case(i)) for i in idx]
    assert len(cases) >= 2
    m = CS.MODELS[model]
    lo, W = window_of(model)
    Pm = max(c["cand"].shape[0] for _, c in cases)
    Dm = max(c["cand"].shape[1] for _, c in cases)
    N = max(len(c["draft_tokens"]) for _, c in cases)
    assert len({c["cand"].shape for _, c in cases}) > 1          # really ragged
    B = len(cases)
    cand = np.full((B, Pm, Dm), -1, np.int64)
    ri = np.zeros((B, Pm, Dm), np.int32)
    rows = np.full((B, N, W), -np.inf, np.float32)
    uni = np.zeros((B, 64))
    # the cases differ in k: run them as two launches by k but keep the ragged shapes inside each launch
    for kk in sorted({s["k"] for s, _ in cases}):
        sel = [b for b, (s, _) in enumerate(cases) if s["k"] == kk]
        for b in sel:
            s, c = cases[b]
            nl, u = H.dynamic_node_logits(s, c)
            P, D = c["cand"].shape
            cand[b, :P, :D] = c["cand"]
            ri[b, :P, :D] = H.row_index_from_retrieve(c["retrieve"], len(c["draft_tokens"]))
            rows[b, :nl.shape[0]] = nl[:, lo:lo + W]
            uni[b] = u
        kw = {}
        if with_counts:
            kw = dict(n_paths=dev(np.array([cases[b][1]["cand"].shape[0] for b in sel], np.int32)),
                      n_depth=dev(np.array([cases[b][1]["cand"].shape[1] for b in sel], np.int32)))
        out = ops.evaluate_posterior_window(hip_cfg(cases[sel[0]][0]), m["V"], dev(rows[sel]), lo, dev(ri[sel]), dev(cand[sel]), dev(uni[sel]),
                                            table=table_dev(m["K"]), want_dense=True, **kw)
        for j, b in enumerate(sel):
            c = cases[b][1]
            assert int(out["counters"][j, 5]) == 0
            assert (int(out["best"][j]), int(out["accept_len"][j])) == (int(c["best"]), int(c["accept_len"])), (b, j)
            np.testing.assert_allclose(out["sample_p"][j].cpu().numpy(), c["sample_p"], rtol=0, atol=PROB_TOL)
            assert int(out["counters"][j, 3]) == int(c["n_draws"])


@pytest.mark.parametrize("static", [True, False])
def test_window_one_hot_rows_vs_oracle(static):
    """Newline / end-of-image rows (MultiModalLogitsProcessor) are one-hot OUTSIDE the image window: the mass travels
    as (out_tok, out_mass); image-token candidates under such a row are rejected, the residual stays one-hot."""
    m = CS.MODELS["lumina"]
    V, lo, W = m["V"], m["img_lo"], m["img_hi"] - m["img_lo"]
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["tree_position_ids"], tree_attn_mask=tb["tree_attn_mask"],
                retrieve_indices=tb["retrieve_indices"])
    N = len(tb["tree_indices"])
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    newline, eos = m["syntax"][2], m["syntax"][0]
    for seed, hot_nodes in [(1, {0: newline}), (2, {1: newline, 2: eos}), (3, {0: eos}), (4, {5: newline, 12: newline})]:
        g = CS.gen_static(700 + seed, "lumina", bufs, sigma=0.7)
        nl = g["node_logits"].copy()
        hot = np.full(N, -1, np.int32)
        for n, t in hot_nodes.items():
            nl[n, :] = -np.inf
            nl[n, t] = 0.0
            hot[n] = t
        ssp = CS.ss_prob_from(g["orig_prob"], g["ss_token"])
        cand, cp, tc = oracle.gather_candidates(g["ss_token"], ssp, g["sample_token"], tb["tree_indices"], tb["retrieve_indices"])
        cfg_o = oracle.EpConfig(mode=oracle.MODE_STATIC_LUMINA if static else oracle.MODE_DYNAMIC, syntax_shortcut=True, tok_offset=4,
                                img_lo=lo, img_hi=lo + W, syntax=m["syntax"], lantern=True, k=100, delta=0.2)
        cfg_h = ops.EpConfig(mode=cfg_o.mode, syntax_shortcut=True, tok_offset=4, img_lo=lo, img_hi=lo + W, syntax=m["syntax"], lantern=True,
                             k=100, delta=0.2)
        aux_o = aux_h = None
        if static:
            aux_o = oracle.StaticAux(cart_prob=cp, orig_prob=g["orig_prob"], op_off=g["op_off"], p_idx=tb["p_indices"], b_off=tb["b_off"],
                                     b_idx=tb["b_idx"], tree_cand=tc)
            aux_h = ops.StaticAux(cart_prob=dev(cp)[None], orig_prob=dev(g["orig_prob"])[None], op_off=dev(g["op_off"]), p_idx=dev(tb["p_indices"]),
                                  b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(tc)[None])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, nl, ri, cand, g["uniforms"], table=H.table(m["K"]), aux=aux_o)
        win_rows = np.ascontiguousarray(nl[:, lo:lo + W])
        out = ops.evaluate_posterior_window(cfg_h, V, dev(win_rows)[None], lo, dev(ri), dev(cand)[None], dev(g["uniforms"])[None],
                                            row_hot=dev(hot)[None], table=table_dev(m["K"]), aux=aux_h, u_bonus=dev(np.array([0.37])),
                                            want_dense=True)
        assert (int(out["best"][0]), int(out["accept_len"][0])) == (ob, oa), (seed, static)
        assert np.array_equal(out["counters"][0, :5].cpu().numpy(), ocnt[:5])
        np.testing.assert_allclose(out["sample_p"][0].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
        assert int(out["token"][0]) == oracle.sample_inverse_cdf(osp, 0.37)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_cfg_window_golden(tag):
    g = H.load("o7.npz")
    m = CS.MODELS["lumina"]
    lo, W = m["img_lo"], m["img_hi"] - m["img_lo"]
    cond, unc = (dev(g["cond"]), dev(g["uncond"])) if tag == "f32" else (_bf16(g["cond"]), _bf16(g["uncond"]))
    kw = dict(w=int(g["w"]), h=int(g["h"]), img_lo=m["img_lo"], img_hi=m["img_hi"], newline_id=m["syntax"][2], eos_id=m["syntax"][0])
    out, hot = ops.cfg_mask_topk_window(cond, unc, 3.0, lo, W, model=ops.MODEL_LUMINA, pos_ids=dev(g["pos"]),
                                        pos_base=int(g["img_start"]) + 3, top_k=100, **kw)
    exp = g[f"lumina_{tag}"]
    hot = hot.cpu().numpy()
    for n in range(exp.shape[0]):
        fin = np.nonzero(np.isfinite(exp[n]))[0]
        if len(fin) == 1 and not (lo <= fin[0] < lo + W):
            assert hot[n] == fin[0]                                   # forced newline / eos row
        else:
            assert hot[n] == -1
            assert np.array_equal(out[n].cpu().numpy(), exp[n, lo:lo + W])
            assert not np.isfinite(np.delete(exp[n], np.s_[lo:lo + W])).any()
    out, hot = ops.cfg_mask_topk_window(cond, unc, 3.0, lo, W, model=ops.MODEL_ANOLE, **kw)
    assert np.array_equal(out.cpu().numpy(), g[f"anole_{tag}"][:, lo:lo + W]) and torch.all(hot == -1)
    out, hot = ops.cfg_mask_topk_window(cond, unc, 3.0, 0, m["V"], model=ops.MODEL_PLAIN, **kw)
    assert np.array_equal(out.cpu().numpy(), g[f"plain_{tag}"])


def test_window_full_size_lumina_pipeline_vs_oracle():
    """V=65536, window [4,8196), k=1000: O7w -> O8w (batched, seq_len positions incl. a newline row) vs oracle O7 -> O8."""
    V, K, lo, W = 65536, 8192, 4, 8192
    rs = np.random.RandomState(5)
    full = np.stack([rs.permutation(K - 1) for _ in range(32)]).astype(np.int64)
    tab = np.zeros((K, K - 1), np.uint16)
    for c in range(K):
        row = full[c % 32]
        tab[c] = np.where(row >= c, row + 1, row).astype(np.uint16)
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    B, R = 4, 11
    cond = torch.from_numpy((rs.standard_normal((B, N, V)) * 2).astype(np.float32)).to(torch.bfloat16)
    unc = torch.from_numpy(rs.standard_normal((B, N, V)).astype(np.float32)).to(torch.bfloat16)
    prompt = 20
    seq_len = np.array([prompt + 3 + 10, prompt + 3 + 47, prompt + 3 + 48, prompt + 3 + 100], np.int64)   # 2nd/3rd hit newline rows
    pos1 = tb["tree_position_ids"] + 1
    win, hot = ops.cfg_mask_topk_window(cond.cuda().reshape(B * N, V), unc.cuda().reshape(B * N, V), 3.0, lo, W, model=ops.MODEL_LUMINA,
                                        pos_ids=dev(pos1), pos_base=prompt + 3, top_k=2000, seq_len=dev(seq_len), rows_per_seq=N)
    win, hot = win.reshape(B, N, W), hot.reshape(B, N)
    cb, ub = cond.view(torch.int16).numpy().view(np.uint16), unc.view(torch.int16).numpy().view(np.uint16)
    ti, pos = tb["tree_indices"], tb["tree_position_ids"]
    par = CS.node_parents(tb["tree_attn_mask"], pos)
    par_row = np.zeros(R, np.int64)
    for n in range(1, N):
        par_row[(ti[n] - 1) // 10] = par[n]
    depth_of_row = pos[par_row]
    op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    cfg_o = oracle.EpConfig.lumina(True, lantern=True, k=1000, delta=0.1)
    cfg_h = ops.EpConfig.lumina(True, lantern=True, k=1000, delta=0.1)
    procs, ops_, cands, cps, tcs = [], [], [], [], []
    for b in range(B):
        proc = oracle.cfg_mask_topk(cb[b], ub[b], 3.0, model=oracle.MODEL_LUMINA, pos_ids=pos1 + seq_len[b], pos_base=prompt + 3,
                                    top_k=2000, bf16=True)
        procs.append(proc)
        dr = np.where(np.isfinite(proc[par_row]), proc[par_row], -30.0) + 2.0 * rs.standard_normal((R, V)).astype(np.float32)
        dr[:, :lo] = -np.inf
        dr[:, lo + W:] = -np.inf
        op = CS.softmax64(CS.topk_filter(dr.astype(np.float32), 2000)).astype(np.float32)
        sst = np.stack([rs.choice(V, 10, replace=False, p=op[r].astype(np.float64) / op[r].astype(np.float64).sum()) for r in range(R)])
        c, cp, tc = oracle.gather_candidates(sst, CS.ss_prob_from(op, sst), 50 + b, ti, tb["retrieve_indices"])
        ops_.append(op); cands.append(c); cps.append(cp); tcs.append(tc)
    for b in range(B):      # O7w == oracle O7 restricted to the window / hot rows
        for n in range(N):
            fin = np.nonzero(np.isfinite(procs[b][n]))[0]
            if int(hot[b, n]) >= 0:
                assert len(fin) == 1 and fin[0] == int(hot[b, n])
            else:
                assert np.array_equal(win[b, n].cpu().numpy(), procs[b][n, lo:lo + W])
    assert int((hot >= 0).sum()) > 0
    uni = rs.random_sample((B, 64))
    ub_ = rs.random_sample(B)
    aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack(ops_)), op_off=dev(op_off), p_idx=dev(tb["p_indices"]),
                        b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(np.stack(tcs)))
    out = ops.evaluate_posterior_window(cfg_h, V, win, lo, dev(ri), dev(np.stack(cands)), dev(uni), row_hot=hot, table=dev(tab.view(np.int16)),
                                        aux=aux, u_bonus=dev(ub_), want_dense=True)
    for b in range(B):
        a = oracle.StaticAux(cart_prob=cps[b], orig_prob=ops_[b], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"],
                             tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, procs[b], ri, cands[b], uni[b], table=tab, aux=a)
        assert (int(out["best"][b]), int(out["accept_len"][b])) == (ob, oa), b
        assert np.array_equal(out["counters"][b, :5].cpu().numpy(), ocnt[:5])
        np.testing.assert_allclose(out["sample_p"][b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
        assert int(out["token"][b]) == oracle.sample_inverse_cdf(osp, ub_[b])
    # probability rows: O7w applies the softmax to every row, O8w copies; identical results, bit for bit
    pw, hot2 = ops.cfg_mask_topk_window(cond.cuda().reshape(B * N, V), unc.cuda().reshape(B * N, V), 3.0, lo, W, model=ops.MODEL_LUMINA,
                                        pos_ids=dev(pos1), pos_base=prompt + 3, top_k=2000, seq_len=dev(seq_len), rows_per_seq=N, probs=True)
    pw = pw.reshape(B, N, W)
    assert torch.equal(hot2.reshape(B, N), hot)
    for b in range(B):
        for n in range(N):
            if int(hot[b, n]) < 0:
                ref = CS.softmax64(procs[b][n, lo:lo + W][None])[0]
                np.testing.assert_allclose(pw[b, n].cpu().numpy(), ref, rtol=0, atol=1e-7)
    out_p = ops.evaluate_posterior_window(cfg_h, V, pw, lo, dev(ri), dev(np.stack(cands)), dev(uni), row_hot=hot, table=dev(tab.view(np.int16)),
                                          aux=aux, u_bonus=dev(ub_), want_dense=True, rows_probs=True)
    for key in ("best", "accept_len", "counters", "token", "sample_p", "sample_win", "out_tok", "out_mass"):
        assert torch.equal(out_p[key], out[key]), key


@pytest.mark.parametrize("V", [2048, 4096, 16384])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cfg_window_probs_and_temperature(dtype, V):
    """out_kind = PROBS and the temperature argument of O7w against torch: softmax(topk(cfg(c,u)/T)); every window width
    class (its own workgroup shape in the f32 and in the 16-byte-load bf16 kernel)."""
    lo, W, k, T = 0, V, 500, 0.7
    g = torch.Generator().manual_seed(3)
    c = (3 * torch.randn(5, V, generator=g)).to(dtype)
    u = torch.randn(5, V, generator=g).to(dtype)
    for temp in (1.0, T):
        lg, _ = ops.cfg_mask_topk_window(c.cuda(), u.cuda(), 2.0, lo, W, model=ops.MODEL_PLAIN, top_k=k, temperature=temp)
        pr, _ = ops.cfg_mask_topk_window(c.cuda(), u.cuda(), 2.0, lo, W, model=ops.MODEL_PLAIN, top_k=k, temperature=temp, probs=True)
        if dtype == torch.bfloat16:
            x = (u + (2.0 * (c - u)).to(dtype)).float()      # bf16 rounding after every op, as torch does on bf16 tensors
        else:
            x = u + 2.0 * (c - u)
        x = x / temp
        kth = torch.topk(x, k, dim=-1).values[..., -1:]
        x = x.masked_fill(x < kth, float("-inf"))
        assert torch.equal(lg.cpu(), x)
        assert int(torch.isfinite(lg).sum(-1).min()) >= k
        np.testing.assert_allclose(pr.cpu().numpy(), torch.softmax(x.double(), -1).float().numpy(), rtol=0, atol=1e-7)


def _top_p_ref(x, top_p):
    """TopPLogitsWarper with the oracle's tie rule (stable ascending sort) and torch.cumsum's f32 outputs."""
    out = x.copy()
    for r in range(x.shape[0]):
        order = np.lexsort((np.arange(x.shape[1]), x[r]))
        sv = x[r][order].astype(np.float64)
        e = np.exp((sv - sv.max()).astype(np.float32)).astype(np.float32)
        p = (e / np.float32(e.astype(np.float64).sum())).astype(np.float32)
        cum = np.cumsum(p.astype(np.float64)).astype(np.float32)
        rm = cum <= np.float32(1.0 - top_p)
        rm[-1] = False
        out[r, order[rm]] = -np.inf
    return out


@pytest.mark.parametrize("top_p", [0.9, 0.5, 0.05, 1e-8])
@pytest.mark.parametrize("ties", [False, True])
def test_cfg_window_top_p(top_p, ties):
    """TopPLogitsWarper inside O7w (Temperature -> TopP -> TopK), rows with and without equal values at the boundary."""
    V, T, k = 4096, 0.8, 300
    rs = np.random.RandomState(11 + int(ties))
    x = (3 * rs.standard_normal((6, V))).astype(np.float32)
    if ties:
        x = np.round(x * 4) / 4            # heavy ties: groups of equal logits everywhere
    c = torch.from_numpy(x).cuda()
    got, _ = ops.cfg_mask_topk_window(c, None, 1.0, 0, V, model=ops.MODEL_PLAIN, top_k=0, temperature=T, top_p=top_p)
    ref = _top_p_ref((x / np.float32(T)).astype(np.float32), top_p)
    g = got.cpu().numpy()
    kept_g, kept_r = np.isfinite(g), np.isfinite(ref)
    assert (kept_g.sum(1) >= 1).all()
    mism = (kept_g != kept_r)
    # the block-parallel mass sums differ from the sequential cumsum in the last f64 bits: at most the boundary entry may flip
    assert mism.sum(1).max() <= 1, mism.sum(1)
    if top_p in (0.9, 0.5):
        assert mism.sum() == 0
    assert np.array_equal(g[kept_g & kept_r], ref[kept_g & kept_r])
    # and with top-k behind it + probabilities out
    pr, _ = ops.cfg_mask_topk_window(c, None, 1.0, 0, V, model=ops.MODEL_PLAIN, top_k=k, temperature=T, top_p=top_p, probs=True)
    y = torch.from_numpy(ref)
    kk = min(k, int(np.isfinite(ref).sum(1).min()))
    kth = torch.topk(y, k, dim=-1).values[..., -1:]
    y = y.masked_fill(y < kth, float("-inf"))
    if mism.sum() == 0:
        np.testing.assert_allclose(pr.cpu().numpy(), torch.softmax(y.double(), -1).float().numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("model", ["lumina", "anole", "llamagen"])
@pytest.mark.parametrize("accept_last", [True, False])
def test_window_more_candidates_than_prefetch_slots(model, accept_last):
    """A level with ten distinct children of which the first nine (or all ten) are rejected: the kernel prefetches neighbour ids
    for six candidates per level, so candidates 7..10 take the refill path (a finished slot is restaged mid-level) and the
    residual is rebuilt up to ten times.  Both row kinds, packed and reference table layouts, against the oracle."""
    m = CS.MODELS[model]
    V, K, off = m["V"], m["K"], m["off"]
    lo, W = window_of(model)
    rs = np.random.RandomState(5 + len(model))
    k = 40
    # tree: root -> 10 children (depth 1), the last child has one child of its own (depth 2): paths [root, c_j] and [root, c_10, g]
    P, D, N = 10, 3, 12
    toks = rs.choice(np.arange(lo + 50, lo + W - 50), size=11, replace=False)
    cand = np.full((P, D), -1, np.int64)
    ri = np.zeros((P, D), np.int32)
    root_tok = int(lo + 7)
    for j in range(10):
        cand[j, 0], cand[j, 1] = root_tok, toks[j]
        ri[j, 0], ri[j, 1] = 0, 1 + j
    cand[9, 2] = toks[10]
    ri[9, 2] = 11
    rows = (1.5 * rs.standard_normal((N, V))).astype(np.float32)
    rows[:, :lo] = -np.inf
    rows[:, lo + W:] = -np.inf
    rows[0, toks[:9]] = -25.0                                  # p ~ 0 for the first nine children: rejected for any uniform > 1e-9
    rows[0, toks[9]] = 12.0 if accept_last else -25.0          # the tenth is (almost surely) accepted, or rejected as well
    table = CS.build_table(K)
    for x in toks[:10]:                                        # their neighbourhoods carry no mass either
        nb = table[x - off, :k].astype(np.int64) + off
        rows[0, nb[(nb != toks[9]) | (not accept_last)]] = -25.0
    uni = np.concatenate([np.full(9, 0.5), [0.3 if accept_last else 0.5], rs.random_sample(54)])
    if model == "lumina":
        mk = lambda E: E.lumina(False, lantern=True, k=k, delta=0.2)
    elif model == "anole":
        mk = lambda E: E.anole(False, lantern=True, k=k, delta=0.2)
    else:
        mk = lambda E: E.llamagen(False, lantern=True, k=k, delta=0.2)
    co, ch = mk(oracle.EpConfig), mk(ops.EpConfig)
    for c in (co, ch):
        c.img_lo, c.img_hi, c.tok_offset = m["img_lo"], m["img_hi"], off
        if model == "lumina":
            c.syntax = tuple(m["syntax"])
    ob, oa, osp, ocnt = oracle.evaluate_posterior(co, rows, ri, cand, uni, table=table)
    assert int(ocnt[1]) >= 10 and int(ocnt[2]) >= 9            # ten candidates tried at level 1, at least nine rejections
    assert oa == (1 if accept_last else 0) or oa == 2
    win = dev(rows[None, :, lo:lo + W])
    tabs = {"reference layout": dev(table.view(np.int16)), "packed": ops.pack_vq_table(dev(table.view(np.int16)), 48)}
    outs = []
    for name, tab in tabs.items():
        out = ops.evaluate_posterior_window(ch, V, win, lo, dev(ri), dev(cand[None]), dev(uni[None]), table=tab, want_dense=True)
        assert int(out["counters"][0, 5]) == 0, name
        assert (int(out["best"][0]), int(out["accept_len"][0])) == (ob, oa), name
        assert np.array_equal(out["counters"][0, :5].cpu().numpy(), ocnt[:5]), name
        np.testing.assert_allclose(out["sample_p"][0].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
        outs.append(out)
    assert torch.equal(outs[0]["sample_p"], outs[1]["sample_p"])
    d = ops.evaluate_posterior(ch, dev(rows[None]), dev(ri), dev(cand[None]), dev(uni[None]), table=dev(table.view(np.int16)))
    assert (int(d[0][0]), int(d[1][0])) == (ob, oa)


@pytest.mark.parametrize("n_children,expect_limit", [(17, False), (18, True)])
def test_window_static_sibling_list_limit_is_reported(n_children, expect_limit):
    """A hand-built static tree whose root has 17 / 18 children, every one rejected in turn: the last rejection zeroes 16 / 17
    earlier siblings in the drafter row.  Sixteen is what the windowed kernel stages; seventeen must come back as
    LANTERN_ST_TREE_LIMIT (never a silently shorter list), while the dense kernel follows the oracle for both."""
    m = CS.MODELS["llamagen"]
    V = m["V"]
    rs = np.random.RandomState(n_children)
    P, D, N = n_children, 2, n_children + 1
    toks = rs.choice(V, size=n_children, replace=False).astype(np.int64)
    cand = np.stack([np.full(P, 3, np.int64), toks], axis=1)
    ri = np.stack([np.zeros(P, np.int32), np.arange(1, P + 1, dtype=np.int32)], axis=1)
    rows = (1.0 * rs.standard_normal((N, V))).astype(np.float32)
    rows[0, toks] = -30.0                                       # target mass ~ 0 on every child: all rejected
    q = np.exp(rs.standard_normal((1, V))).astype(np.float32)
    q[0, toks] += 50.0                                          # the drafter liked them
    q /= q.sum()
    cart = np.stack([np.ones(P, np.float32), q[0, toks]], axis=1)
    b_off = np.zeros(P * D + 1, np.int32)
    b_idx = []
    for j in range(P):                                          # cell (j,1): the earlier children's node ids 1..j
        b_off[j * D + 1] = len(b_idx)
        b_idx += list(range(1, j + 1))
        b_off[j * D + 2] = len(b_idx)
    for j in range(P):
        b_off[j * D] = b_off[j * D + 1] if j == 0 else b_off[j * D]   # keep the CSR monotone: cell (j,0) is empty
    b_off = np.maximum.accumulate(b_off)
    tree_cand = np.concatenate([[3], toks]).astype(np.int64)
    aux_o = oracle.StaticAux(cart_prob=cart, orig_prob=q, op_off=np.zeros(1, np.int32), p_idx=np.zeros((P, D), np.int32), b_off=b_off,
                             b_idx=np.array(b_idx, np.int32), tree_cand=tree_cand)
    co, ch = oracle.EpConfig.llamagen(True, lantern=False), ops.EpConfig.llamagen(True, lantern=False)
    uni = np.full(64, 0.5)
    ob, oa, osp, ocnt = oracle.evaluate_posterior(co, rows, ri, cand, uni, aux=aux_o)
    assert oa == 0 and int(ocnt[2]) == n_children              # every child tried and rejected
    aux_h = ops.StaticAux(cart_prob=dev(cart[None]), orig_prob=dev(q[None]), op_off=dev(np.zeros(1, np.int32)), p_idx=dev(np.zeros((P, D), np.int32)),
                          b_off=dev(b_off), b_idx=dev(np.array(b_idx, np.int32)), tree_cand=dev(tree_cand[None]))
    d = ops.evaluate_posterior(ch, dev(rows[None]), dev(ri), dev(cand[None]), dev(uni[None]), aux=aux_h)
    assert (int(d[0][0]), int(d[1][0])) == (ob, oa)
    np.testing.assert_allclose(d[2][0].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
    w = ops.evaluate_posterior_window(ch, V, dev(rows[None]), 0, dev(ri), dev(cand[None]), dev(uni[None]), aux=aux_h, want_dense=True)
    if expect_limit:
        assert int(w["counters"][0, 5]) == 7
        with pytest.raises(Exception, match="staging limits"):
            ops.raise_on_status(w["counters"])
    else:
        assert int(w["counters"][0, 5]) == 0
        assert (int(w["best"][0]), int(w["accept_len"][0])) == (ob, oa)
        np.testing.assert_allclose(w["sample_p"][0].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)


@pytest.mark.parametrize("model", ["llamagen", "anole"])
def test_window_maximum_tree_shape(model):
    """The largest tree the windowed kernel accepts: 64 paths x 16 depths (1024 cells, the LDS tables at their capacity), 64
    uniforms per step, ~900 node rows (row_hot read from HBM: more than 128 rows per sequence).  One trunk that the target likes
    (acceptance runs deep) with 63 branches leaving it at different depths; batched over three sequences with different
    uniforms; dense kernel and oracle must agree with it."""
    m = CS.MODELS[model]
    V, K, off = m["V"], m["K"], m["off"]
    lo, W = window_of(model)
    rs = np.random.RandomState(99)
    P, D, B, k = 64, 16, 3, 10
    pool = rs.permutation(np.arange(lo + 8, lo + W - 8))
    take = iter(pool)
    trunk = [int(next(take)) for _ in range(D)]
    cand = np.zeros((P, D), np.int64)
    ri = np.zeros((P, D), np.int32)
    cand[0], ri[0] = trunk, np.arange(D)
    n_rows = D
    for p in range(1, P):
        a = 1 + (p % (D - 2))                                  # shares the trunk up to depth a (inclusive)
        cand[p, :a + 1], ri[p, :a + 1] = trunk[:a + 1], np.arange(a + 1)
        for d in range(a + 1, D):
            cand[p, d] = int(next(take)) if d < D - (p % 3) else -1      # some branches end early (-1 padding)
            ri[p, d] = n_rows
            n_rows += 1
    assert n_rows > 128
    rows = (1.0 * rs.standard_normal((n_rows, V))).astype(np.float32)
    rows[:, :lo] = -np.inf
    rows[:, lo + W:] = -np.inf
    for d in range(D - 1):                                      # row d predicts depth d + 1: the trunk token gets most of the mass
        rows[d, trunk[d + 1]] = 7.5
    table = CS.build_table(K)
    mk = (lambda E: E.llamagen(False, lantern=True, k=k, delta=0.3)) if model == "llamagen" else (lambda E: E.anole(False, lantern=True, k=k, delta=0.3))
    co, ch = mk(oracle.EpConfig), mk(ops.EpConfig)
    for c in (co, ch):
        c.img_lo, c.img_hi, c.tok_offset = m["img_lo"], m["img_hi"], off
    uni = rs.random_sample((B, 64))
    uni[1] *= 0.2                                               # an easy-going sequence: accepts further
    ref = [oracle.evaluate_posterior(co, rows, ri, cand, uni[b], table=table) for b in range(B)]
    assert max(r[1] for r in ref) >= 6                          # the deep levels are really walked
    win = dev(np.broadcast_to(rows[None, :, lo:lo + W], (B, n_rows, W)).copy())
    out = ops.evaluate_posterior_window(ch, V, win, lo, dev(ri), dev(np.broadcast_to(cand[None], (B, P, D)).copy()), dev(uni),
                                        table=dev(table.view(np.int16)), want_dense=True)
    ops.raise_on_status(out["counters"])
    dn = ops.evaluate_posterior(ch, dev(np.broadcast_to(rows[None], (B, n_rows, V)).copy()), dev(ri),
                                dev(np.broadcast_to(cand[None], (B, P, D)).copy()), dev(uni), table=dev(table.view(np.int16)))
    for b in range(B):
        ob, oa, osp, ocnt = ref[b]
        assert (int(out["best"][b]), int(out["accept_len"][b])) == (ob, oa), b
        assert np.array_equal(out["counters"][b, :5].cpu().numpy(), ocnt[:5]), b
        np.testing.assert_allclose(out["sample_p"][b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
        assert (int(dn[0][b]), int(dn[1][b])) == (ob, oa), b
    # one cell more than the tables hold is refused on the host
    with pytest.raises(Exception):
        ops.evaluate_posterior_window(ch, V, win, lo, dev(np.zeros((65, D), np.int32)), dev(np.zeros((B, 65, D), np.int64)), dev(uni),
                                      table=dev(table.view(np.int16)))

"""Node-parallel evaluate_posterior (lantern_evaluate_posterior_nodes): the per-node tables on the CPU, and on the GPU the
same golden vectors / oracle comparisons as the chain kernels -- integers bit-exact, probabilities <= 1e-5 -- plus
bit-equality with the chain kernel on the same probability rows (both run the same arithmetic)."""
import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle
from lantern_amd import ops

SPECS = H.ep_specs()
PROB_TOL = 1e-5
TREES = ["mc_sim_7b_63", "naive_extend_57"] + [f"rand{i:02d}" for i in range(0, 40, 3)]


def _tables_py(ret, N):
    """Plain restatement of the node view: children in order of their first path, uniforms consumed on the way down."""
    P, D = ret.shape
    kids, parent, depth, cell = {n: [] for n in range(N)}, {}, {0: 0}, {0: 0}
    for j in range(P):
        for i in range(1, D):
            c = int(ret[j, i])
            if c < 0:
                break
            if c not in parent:
                parent[c], depth[c], cell[c] = int(ret[j, i - 1]), i, j * D + i
                kids[parent[c]].append(c)
    uoff = {0: 0}
    todo = [0]
    while todo:
        n = todo.pop(0)
        for t, c in enumerate(kids[n]):
            uoff[c] = uoff[n] + t + 1
            todo.append(c)
    return kids, parent, depth, cell, uoff


@pytest.mark.parametrize("tree", TREES)
def test_node_tables_host(tree):
    tb = oracle.tree_static_build(H.tree_choices(tree))
    N = len(tb["tree_indices"])
    ret = tb["retrieve_indices"]
    P, D = ret.shape
    pos = tb["tree_position_ids"]
    # op_off as the harness / cases.py derive it: first drafter row of every depth
    g_op = CS.op_off_of(tb) if hasattr(CS, "op_off_of") else None
    op_off = g_op if g_op is not None else _op_off(tb)
    nt = ops.tree_node_tables(ret, N, tb["p_indices"], tb["b_off"], op_off)
    t = nt.host
    kids, parent, depth, cell, uoff = _tables_py(ret, N)
    internal = [n for n in range(N) if kids[n]]
    assert (nt.n_nodes, nt.n_internal, nt.n_children, nt.max_children) == (N, len(internal), N - 1, max(len(kids[n]) for n in internal))
    info = t[8:8 + 16 * nt.n_internal].reshape(-1, 16)
    child = t[8 + 16 * nt.n_internal:8 + 16 * nt.n_internal + 4 * nt.n_children].reshape(-1, 4)
    node = t[8 + 16 * nt.n_internal + 4 * nt.n_children:][:4 * N].reshape(N, 4)
    order = t[8 + 16 * nt.n_internal + 4 * nt.n_children + 4 * N:][:N]
    assert t[6] == 1 and order[:nt.n_internal].tolist() == info[:, 0].tolist() and sorted(order.tolist()) == list(range(N))
    assert sorted(info[:, 0].tolist()) == internal
    assert all(info[r, 2] >= info[r + 1, 2] for r in range(len(info) - 1))          # longest child lists first
    for r, (n, c0, nch, d, uo, qrow, fp, _, *first4) in enumerate(info.tolist()):
        assert first4[:min(nch, 4)] == kids[n][:4] and first4[4:4 + min(nch, 4)] == [cell[c] for c in kids[n][:4]]
        assert child[c0:c0 + nch, 0].tolist() == kids[n] and d == depth[n] and uo == uoff[n] and fp == cell[n] // D
        assert node[n, 2] == r
        for s, (c, cc, b0, nsib) in enumerate(child[c0:c0 + nch].tolist()):
            assert cc == cell[c] and ret.reshape(-1)[cc] == c
            # earlier siblings of the reference's b_indices == the children tried before this one
            assert tb["b_idx"][b0:b0 + nsib].tolist() == kids[n][:s]
            assert qrow == op_off[d] + tb["p_indices"].reshape(-1)[cc]
    for n in range(N):
        assert node[n, 0] == cell[n] // D and node[n, 1] == depth[n] and node[n, 3] == parent.get(n, -1)
        if not kids[n]:
            assert node[n, 2] == -1
    assert (pos == np.array([depth[n] for n in range(N)])).all()


def _op_off(tb):
    ti, pos = tb["tree_indices"], tb["tree_position_ids"]
    N = len(ti)
    mask = tb["tree_attn_mask"]
    R = int(((ti[1:] - 1) // 10).max()) + 1
    par_row = np.zeros(R, np.int64)
    for n in range(1, N):
        anc = [a for a in np.nonzero(mask[n] > 0)[0] if pos[a] == pos[n] - 1]
        par_row[(ti[n] - 1) // 10] = anc[0]
    d = pos[par_row]
    return np.array([np.nonzero(d == x)[0][0] for x in range(int(d.max()) + 1)], np.int32)


def test_node_tables_reject_a_non_tree():
    ret = np.array([[0, 1, 2], [0, 3, 2]], np.int64)           # node 2 under two parents
    with pytest.raises(Exception):
        ops.tree_node_tables(ret, 4)


# ----------------------------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _prob_rows(spec, nl, lo, W):
    """Window rows as probabilities, produced where the product produces them (the windowed O7: processors + softmax)."""
    m = CS.MODELS[spec["model"]]
    T, tk = spec.get("temperature", 1.0), spec.get("top_k", 0)
    if spec["model"] == "lumina":
        T, tk = 1.0, 0
    pr, _ = ops.cfg_mask_topk_window(dev(nl), None, 1.0, lo, W, model=ops.MODEL_ANOLE if spec["model"] != "llamagen" else ops.MODEL_PLAIN,
                                     img_lo=lo, img_hi=lo + W, top_k=min(tk, m["V"]) if tk else 0, temperature=T if T > 1e-5 else 1.0,
                                     probs=True)
    return pr


def _static_ok(s):
    tp = s.get("top_p", 1.0)
    return s["kind"] == "static" and not (0.0 < tp < 1.0) and s.get("temperature", 1.0) > 1e-5 and (not s["lantern"] or s["k"] + 1 <= 1024)


@gpu
@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if _static_ok(s)])
def test_nodes_static_golden(i):
    from test_gpu_parity import hip_cfg, table_dev
    from test_gpu_window import window_of
    spec, case = SPECS[i], H.ep_case(i)
    tb, g = H.static_inputs(spec, case)
    m = CS.MODELS[spec["model"]]
    lo, W = window_of(spec["model"])
    N = len(tb["tree_indices"])
    pr = _prob_rows(spec, g["node_logits"], lo, W)
    cfg = hip_cfg(spec)
    cfg.temperature, cfg.top_k, cfg.top_p = 1.0, 0, 1.0          # probability rows are final
    aux = ops.StaticAux(cart_prob=dev(case["cart_prob"])[None], orig_prob=dev(g["orig_prob"])[None], op_off=dev(g["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]),
                        b_idx=dev(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)), tree_cand=dev(case["tree_cand"])[None])
    nt = ops.tree_node_tables(tb["retrieve"], N, tb["p_indices"], tb["b_off"], g["op_off"], device="cuda")
    u = 0.1 + 0.8 * ((i * 37) % 100) / 100.0
    args = (cfg, m["V"], pr[None], lo, dev(H.row_index_from_retrieve(tb["retrieve"], N)), dev(case["cand"])[None], dev(case["uniforms"])[None])
    kw = dict(table=table_dev(m["K"]), aux=aux, u_bonus=dev(np.array([u])), want_dense=True, rows_probs=True)
    chain = ops.evaluate_posterior_window(*args, **kw)
    node = ops.evaluate_posterior_window(*args, nodes=nt, **kw)
    st = int(node["counters"][0, 5])
    if st == 8:
        # duplicate sibling tokens (a golden edge case): the node view does not hold, the kernel says so and the chain kernel is the path
        toks = case["tree_cand"]
        kids = _tables_py(tb["retrieve"], N)[0]
        assert any(len({int(toks[c]) for c in ks}) < len(ks) for ks in kids.values() if ks)
        return
    assert st == int(chain["counters"][0, 5])
    if st != 0:
        return
    for key in ("best", "accept_len", "counters", "token", "out_tok"):
        assert torch.equal(node[key], chain[key]), (key, node[key], chain[key])
    assert torch.equal(node["sample_p"], chain["sample_p"]) and torch.equal(node["sample_win"], chain["sample_win"])
    assert int(node["best"][0]) == int(case["best"]) and int(node["accept_len"][0]) == int(case["accept_len"])
    assert int(node["counters"][0, 3]) == int(case["n_draws"])
    np.testing.assert_allclose(node["sample_p"][0].cpu().numpy(), case["sample_p"], rtol=0, atol=PROB_TOL)
    assert int(node["token"][0]) == oracle.sample_inverse_cdf(node["sample_p"][0].cpu().numpy(), u)


@gpu
@pytest.mark.parametrize("model,tree,lantern,k,delta,sigma,seed,packed", [
    ("lumina", "mc_sim_7b_63", True, 100, 0.1, 1.0, 1, True), ("lumina", "mc_sim_7b_63", True, 300, 5.0, 2.0, 2, False),
    ("lumina", "naive_extend_57", True, 10, 0.3, 0.5, 3, True), ("lumina", "mc_sim_7b_63", False, 1, 0.1, 3.0, 4, False),
    ("llamagen", "naive_extend_57", True, 50, 0.1, 1.0, 5, True), ("llamagen", "mc_sim_7b_63", True, 200, 10.0, 2.0, 6, False),
    ("anole", "naive_extend_57", True, 10, 5.0, 1.0, 7, True), ("anole", "naive_extend_57", True, 5, 20.0, 3.0, 8, False),
    ("anole", "mc_sim_7b_63", False, 1, 0.1, 0.5, 9, False), ("lumina", "rand07", True, 60, 0.2, 2.5, 11, True),
    ("lumina", "rand21", True, 500, 0.1, 4.0, 12, True)])
def test_nodes_static_batches_vs_oracle(model, tree, lantern, k, delta, sigma, seed, packed):
    """32 sequences per launch, every sequence its own rows / candidates / drafter rows / uniform stream and its own cursor
    into it; the oracle is the judge, the chain kernel must agree bit for bit."""
    from test_gpu_fuzz import cfgs
    B = 32
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
    nl = np.stack([g["node_logits"] for g in gs])
    spec = dict(model=model)
    pr = _prob_rows(spec, nl.reshape(B * N, V), lo, W).reshape(B, N, W)
    aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack([g["orig_prob"] for g in gs])), op_off=dev(gs[0]["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)),
                        tree_cand=dev(np.stack(tcs)))
    nu = gs[0]["uniforms"].shape[0]
    start = np.arange(B, dtype=np.int32) % 5                   # every sequence reads its stream from its own cursor
    uni = np.stack([np.concatenate([np.full(start[b], 0.5), g["uniforms"]]) [:nu] for b, g in enumerate(gs)])
    tab = dev(table.view(np.int16))
    if packed and lantern:
        tab = ops.pack_vq_table(tab, -(-(k + 1) // 8) * 8)
    ub = np.random.RandomState(seed).random_sample(B)
    nt = ops.tree_node_tables(tb["retrieve_indices"], N, tb["p_indices"], tb["b_off"], gs[0]["op_off"], device="cuda")
    outs = {}
    for name, nodes in (("chain", None), ("nodes", nt)):
        cur = dev(start.copy())
        outs[name] = ops.evaluate_posterior_window(ch, V, pr, lo, dev(ri), dev(np.stack(cands)), dev(uni), table=tab if lantern else None, aux=aux,
                                                   cursor=cur, u_bonus=dev(ub), want_dense=True, rows_probs=True, nodes=nodes)
        outs[name]["cursor"] = cur
    n_rej = n_acc = 0
    needs_dense = []
    for b, g in enumerate(gs):
        a = oracle.StaticAux(cart_prob=cps[b], orig_prob=g["orig_prob"], op_off=g["op_off"], p_idx=tb["p_indices"], b_off=tb["b_off"],
                             b_idx=tb["b_idx"], tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(co, g["node_logits"], ri, cands[b], uni[b, start[b]:], table=table if lantern else None, aux=a)
        n_rej += int(ocnt[2]); n_acc += oa
        if all(int(o["counters"][b, 5]) == 6 for o in outs.values()):
            needs_dense.append(b)          # the residual became uniform over all V (gtp.sum() == 0 -> ones): every windowed kernel says so, the dense set handles it
            continue
        for name, o in outs.items():
            st = int(o["counters"][b, 5])
            assert st == 0, (name, b, st)
            assert (int(o["best"][b]), int(o["accept_len"][b])) == (ob, oa), (name, b, int(o["best"][b]), int(o["accept_len"][b]), ob, oa)
            assert np.array_equal(o["counters"][b, :5].cpu().numpy(), ocnt[:5]), (name, b, o["counters"][b].tolist(), ocnt.tolist())
            assert int(o["cursor"][b]) == start[b] + int(ocnt[3])
            np.testing.assert_allclose(o["sample_p"][b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
            assert int(o["token"][b]) == oracle.sample_inverse_cdf(o["sample_p"][b].cpu().numpy(), float(ub[b])), (name, b)
    assert n_rej > 0 and n_acc > 0
    ok = torch.ones(B, dtype=torch.bool, device="cuda")
    ok[needs_dense] = False
    assert len(needs_dense) <= B // 4
    for key in ("best", "accept_len", "counters", "token", "sample_p", "sample_win", "out_tok", "out_mass", "cursor"):
        assert torch.equal(outs["nodes"][key][ok], outs["chain"][key][ok]), key


@gpu
def test_nodes_one_hot_rows_and_no_outputs():
    """Lumina newline / end-of-image rows (one-hot OUTSIDE the window) through the node kernels, with the optional outputs
    off (no sample_p, no sample_win: the serving loop's configuration) -- results equal the chain kernel's."""
    m = CS.MODELS["lumina"]
    V, lo, W = m["V"], m["img_lo"], m["img_hi"] - m["img_lo"]
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["tree_position_ids"], tree_attn_mask=tb["tree_attn_mask"],
                retrieve_indices=tb["retrieve_indices"])
    B = 16
    gs = [CS.gen_static(777 + b, "lumina", bufs, sigma=2.0) for b in range(B)]
    cands, cps, tcs = [], [], []
    for g in gs:
        c, cp, tc = oracle.gather_candidates(g["ss_token"], CS.ss_prob_from(g["orig_prob"], g["ss_token"]), g["sample_token"],
                                             tb["tree_indices"], tb["retrieve_indices"])
        cands.append(c); cps.append(cp); tcs.append(tc)
    nl = np.stack([g["node_logits"] for g in gs])
    pr = _prob_rows(dict(model="lumina"), nl.reshape(B * N, V), lo, W).reshape(B, N, W).clone()
    hot = np.full((B, N), -1, np.int32)
    rs = np.random.RandomState(5)
    for b in range(B):                      # a few rows per sequence become forced syntax rows
        for n in rs.choice(N, 3, replace=False):
            hot[b, n] = m["syntax"][2] if rs.rand() < 0.7 else m["syntax"][0]
    ch = ops.EpConfig.lumina(True, lantern=True, k=64, delta=0.2)
    ch.img_lo, ch.img_hi, ch.tok_offset, ch.syntax = m["img_lo"], m["img_hi"], m["off"], tuple(m["syntax"])
    aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack([g["orig_prob"] for g in gs])), op_off=dev(gs[0]["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(np.stack(tcs)))
    nt = ops.tree_node_tables(tb["retrieve_indices"], N, tb["p_indices"], tb["b_off"], gs[0]["op_off"], device="cuda")
    tab = ops.pack_vq_table(dev(CS.build_table(m["K"]).view(np.int16)), 72)
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    uni = np.stack([g["uniforms"] for g in gs])
    ub = rs.random_sample(B)
    kw = dict(table=tab, aux=aux, u_bonus=dev(ub), row_hot=dev(hot), rows_probs=True)
    a = ops.evaluate_posterior_window(ch, V, pr, lo, dev(ri), dev(np.stack(cands)), dev(uni), want_dense=False, want_window=False, **kw)
    b_ = ops.evaluate_posterior_window(ch, V, pr, lo, dev(ri), dev(np.stack(cands)), dev(uni), want_dense=False, want_window=False, nodes=nt, **kw)
    for key in ("best", "accept_len", "counters", "token", "out_tok", "out_mass"):
        assert torch.equal(a[key], b_[key]), (key, a[key], b_[key])
    assert int((a["counters"][:, 5] != 0).sum()) == 0
    full = ops.evaluate_posterior_window(ch, V, pr, lo, dev(ri), dev(np.stack(cands)), dev(uni), want_dense=True, nodes=nt, **kw)
    full_c = ops.evaluate_posterior_window(ch, V, pr, lo, dev(ri), dev(np.stack(cands)), dev(uni), want_dense=True, **kw)
    assert torch.equal(full["sample_p"], full_c["sample_p"]) and torch.equal(full["token"], a["token"])

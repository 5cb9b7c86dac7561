"""GPU (-m gpu): the other BASELINE.json configurations at their full sizes, HIP vs oracle.
  C2  LlamaGen + EAGLE, standard (non-relaxed) verify: V=K=16384, dynamic tree N=59, HF processors T=1 / top_k=2000.
  C4  Anole + LANTERN++ static tree naive_extend_57 (N=58,P=33,D=6): V=65536, offset 4, lambda in {5,10,20}, k in {5,10}.
Both kernel sets (dense and windowed) must agree with the oracle and with each other."""
import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle
from lantern_amd import ops
from lantern_amd.drafters.choices import naive_extend_57
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-5


def perm_table(K, cols, seed):
    rs = np.random.RandomState(seed)
    base = np.stack([rs.permutation(K - 1)[:cols] for _ in range(64)]).astype(np.int64)
    tab = base[np.arange(K) % 64]
    return np.where(tab >= np.arange(K)[:, None], tab + 1, tab).astype(np.uint16)


def test_c2_llamagen_dynamic_standard_verify():
    V, B, k = 16384, 5, 10
    rs = np.random.RandomState(21)
    depth, T = 4, 58
    cfg_o = oracle.EpConfig.llamagen(False, lantern=False, temperature=1.0, top_p=1.0, top_k=2000)
    cfg_h = ops.EpConfig.llamagen(False, lantern=False, temperature=1.0, top_p=1.0, top_k=2000)
    for b in range(B):
        # drafter tree through the HIP O3/O4 kernels, checked against the oracle's
        script = [(4 * rs.standard_normal((1 if d == 0 else k, V))).astype(np.float32) for d in range(depth + 1)]
        ti, cu, ci, sc = ops.expand_dynamic(dev(CS.topk_filter(script[0], 2000))[None], None, k)
        oti, ocu, oci, osc = oracle.expand_dynamic(CS.topk_filter(script[0], 2000), None, k)
        assert np.array_equal(ti[0].cpu().numpy(), oti) and np.allclose(cu[0].cpu().numpy(), ocu, atol=1e-5)
        sl, tl, pl = [cu.reshape(-1)], [ti.reshape(-1)], [torch.zeros(1, dtype=torch.int64, device="cuda")]
        cs = torch.arange(k, device="cuda")
        for d in range(depth):
            pl.append(cs + 1 + k * k * max(0, d - 1) + (k if d > 0 else 0))
            ti, cu, ci, sc = ops.expand_dynamic(dev(CS.topk_filter(script[d + 1], 2000))[None], sc, k)
            cs = ci[0]
            sl.append(cu.reshape(-1)); tl.append(ti.reshape(-1))
        sample = int(rs.randint(0, V))
        draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(torch.cat(sl)[None], torch.cat(tl)[None], torch.cat(pl)[None],
                                                                   torch.tensor([sample], device="cuda"), k, T)
        od, oret, omask, opos = oracle.tree_dynamic_finalize(torch.cat(sl).cpu().numpy(), torch.cat(tl).cpu().numpy(),
                                                             torch.cat(pl).cpu().numpy(), k, T, sample)
        nl, md = int(nl[0]), int(md[0])
        assert np.array_equal(draft[0].cpu().numpy(), od) and np.array_equal(ret[0, :nl, :md].cpu().numpy(), oret)
        assert np.array_equal(mask[0].cpu().numpy(), omask) and np.array_equal(pos[0].cpu().numpy(), opos)
        N = T + 1
        node_logits = (4 * rs.standard_normal((N, V))).astype(np.float32)
        for p in range(oret.shape[0]):            # drafted tokens plausible under the target
            for d in range(1, oret.shape[1]):
                if oret[p, d] >= 0:
                    node_logits[oret[p, d - 1], od[oret[p, d]]] = node_logits[oret[p, d - 1]].max() - rs.uniform(0, 3)
        cand = np.where(oret >= 0, od[np.maximum(oret, 0)], -1)
        ri = H.row_index_from_retrieve(oret, N)
        uni = rs.random_sample(64)
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, node_logits, ri, cand, uni)
        d1 = ops.evaluate_posterior(cfg_h, dev(node_logits)[None], dev(ri), dev(cand)[None], dev(uni)[None])
        w = ops.evaluate_posterior_window(cfg_h, V, dev(node_logits)[None], 0, dev(ri), dev(cand)[None], dev(uni)[None], want_dense=True)
        for best, alen, sp, cnt in ((d1[0], d1[1], d1[2], d1[3]), (w["best"], w["accept_len"], w["sample_p"], w["counters"])):
            assert (int(best[0]), int(alen[0])) == (ob, oa)
            assert np.array_equal(cnt[0, :5].cpu().numpy(), ocnt[:5])
            np.testing.assert_allclose(sp[0].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)


@pytest.mark.parametrize("lam,k", [(5.0, 10), (10.0, 5), (20.0, 5)])
def test_c4_anole_static_lantern_pp(lam, k):
    V, K, off, lo, W = 65536, 8192, 4, 4, 8192
    rs = np.random.RandomState(int(lam) * 10 + k)
    tb = oracle.tree_static_build(naive_extend_57)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    assert (N, P, D) == (58, 33, 6)
    ti, pos = tb["tree_indices"], tb["tree_position_ids"]
    par = CS.node_parents(tb["tree_attn_mask"], pos)
    R = int(((ti[1:] - 1) // 10).max()) + 1
    par_row = np.zeros(R, np.int64)
    for n in range(1, N):
        par_row[(ti[n] - 1) // 10] = par[n]
    depth_of_row = pos[par_row]
    op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
    tab = perm_table(K, 64, 3)
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    B = 4
    fmin = np.finfo(np.float32).min
    cfg_kw = dict(lantern=True, k=k, delta=lam, temperature=1.0, top_p=1.0, top_k=2000)
    cfg_o, cfg_h = oracle.EpConfig.anole(True, **cfg_kw), ops.EpConfig.anole(True, **cfg_kw)
    logits, ops_l, cands, cps, tcs = [], [], [], [], []
    for b in range(B):
        nl = np.full((N, V), fmin, np.float32)             # ea_model_anole.py:931: non-image -> finfo.min
        nl[:, lo:lo + W] = (4 * rs.standard_normal((N, W))).astype(np.float32)
        dr = np.full((R, V), -np.inf, np.float32)
        dr[:, lo:lo + W] = nl[par_row][:, lo:lo + W] + (1.0 + b) * rs.standard_normal((R, W)).astype(np.float32)
        op = CS.softmax64(CS.topk_filter(dr, 2000)).astype(np.float32)
        sst = np.stack([rs.choice(V, 10, replace=False, p=op[r].astype(np.float64) / op[r].astype(np.float64).sum()) for r in range(R)])
        c, cp, tc = oracle.gather_candidates(sst, CS.ss_prob_from(op, sst), 100 + b, ti, tb["retrieve_indices"])
        logits.append(nl); ops_l.append(op); cands.append(c); cps.append(cp); tcs.append(tc)
    uni = rs.random_sample((B, 64))
    aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack(ops_l)), op_off=dev(op_off), p_idx=dev(tb["p_indices"]),
                        b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(np.stack(tcs)))
    dense = ops.evaluate_posterior(cfg_h, dev(np.stack(logits)), dev(ri), dev(np.stack(cands)), dev(uni), table=dev(tab.view(np.int16)), aux=aux)
    win = ops.evaluate_posterior_window(cfg_h, V, dev(np.ascontiguousarray(np.stack(logits)[:, :, lo:lo + W])), lo, dev(ri), dev(np.stack(cands)),
                                        dev(uni), table=dev(tab.view(np.int16)), aux=aux, want_dense=True)
    n_rej = 0
    for b in range(B):
        a = oracle.StaticAux(cart_prob=cps[b], orig_prob=ops_l[b], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"],
                             tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, logits[b], ri, cands[b], uni[b], table=tab, aux=a)
        n_rej += int(ocnt[2])
        for best, alen, sp, cnt in ((dense[0], dense[1], dense[2], dense[3]), (win["best"], win["accept_len"], win["sample_p"], win["counters"])):
            assert int(cnt[b, 5]) == 0
            assert (int(best[b]), int(alen[b])) == (ob, oa), (b, int(best[b]), int(alen[b]), ob, oa)
            assert np.array_equal(cnt[b, :5].cpu().numpy(), ocnt[:5])
            np.testing.assert_allclose(sp[b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
    assert n_rej > 0


@pytest.mark.parametrize("k,delta", [(3000, 0.3), (2047, 5.0), (1024, 0.1)])
def test_window_large_k_reads_ids_from_hbm(k, delta):
    """k + 1 > 1024 staged ids: the windowed kernel's HBM-id scan (several 1024-neighbour rounds with an early exit) -- Lumina
    static tree at full size, both kernel sets against the oracle."""
    V, K, lo, W = 65536, 8192, 4, 8192
    rs = np.random.RandomState(k)
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    ti, pos = tb["tree_indices"], tb["tree_position_ids"]
    par = CS.node_parents(tb["tree_attn_mask"], pos)
    R = int(((ti[1:] - 1) // 10).max()) + 1
    par_row = np.zeros(R, np.int64)
    for n in range(1, N):
        par_row[(ti[n] - 1) // 10] = par[n]
    depth_of_row = pos[par_row]
    op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
    tab = perm_table(K, 3072, 9)
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    B = 3
    cfg_o, cfg_h = oracle.EpConfig.lumina(True, lantern=True, k=k, delta=delta), ops.EpConfig.lumina(True, lantern=True, k=k, delta=delta)
    logits, ops_l, cands, cps, tcs = [], [], [], [], []
    for b in range(B):
        nl = np.full((N, V), -np.inf, np.float32)
        nl[:, lo:lo + W] = CS.topk_filter((4 * rs.standard_normal((N, W))).astype(np.float32), 2000)
        dr = np.full((R, V), -np.inf, np.float32)
        base = np.where(np.isfinite(nl[par_row][:, lo:lo + W]), nl[par_row][:, lo:lo + W], -30.0)
        dr[:, lo:lo + W] = base + (1.0 + b) * rs.standard_normal((R, W)).astype(np.float32)
        op = CS.softmax64(CS.topk_filter(dr, 2000)).astype(np.float32)
        sst = np.stack([rs.choice(V, 10, replace=False, p=op[r].astype(np.float64) / op[r].astype(np.float64).sum()) for r in range(R)])
        c, cp, tc = oracle.gather_candidates(sst, CS.ss_prob_from(op, sst), 100 + b, ti, tb["retrieve_indices"])
        logits.append(nl); ops_l.append(op); cands.append(c); cps.append(cp); tcs.append(tc)
    uni = rs.random_sample((B, 64))
    aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack(ops_l)), op_off=dev(op_off), p_idx=dev(tb["p_indices"]),
                        b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(np.stack(tcs)))
    dense = ops.evaluate_posterior(cfg_h, dev(np.stack(logits)), dev(ri), dev(np.stack(cands)), dev(uni), table=dev(tab.view(np.int16)), aux=aux)
    win = ops.evaluate_posterior_window(cfg_h, V, dev(np.ascontiguousarray(np.stack(logits)[:, :, lo:lo + W])), lo, dev(ri), dev(np.stack(cands)),
                                        dev(uni), table=dev(tab.view(np.int16)), aux=aux, want_dense=True)
    for b in range(B):
        a = oracle.StaticAux(cart_prob=cps[b], orig_prob=ops_l[b], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"],
                             tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, logits[b], ri, cands[b], uni[b], table=tab, aux=a)
        for best, alen, sp, cnt in ((dense[0], dense[1], dense[2], dense[3]), (win["best"], win["accept_len"], win["sample_p"], win["counters"])):
            assert int(cnt[b, 5]) == 0
            assert (int(best[b]), int(alen[b])) == (ob, oa), (b, int(best[b]), int(alen[b]), ob, oa)
            assert np.array_equal(cnt[b, :5].cpu().numpy(), ocnt[:5])
            np.testing.assert_allclose(sp[b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)


# a tree with the default tree's sizes (26 nodes, 15 paths, depth 6 -- what the host dispatches the fixed-tree instance on) but SIX children under
# the root: more unique candidates at a level than that instance stages ahead (4), so its restage path runs
SIX_WIDE_26 = [[0], [1], [2], [3], [4], [5], [0, 0], [0, 1], [0, 2], [0, 3], [1, 0], [2, 0], [2, 1], [3, 0], [3, 1], [0, 0, 0], [0, 0, 1], [0, 0, 2],
               [1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1], [1, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 1], [1, 0, 0, 0, 0]]


@pytest.mark.parametrize("tp4", [0, 2, 3, 4])
def test_throughput_instance_variants_of_the_default_tree(tp4):
    """The compact instance's measurement variants (lantern_tuning_set("epw_tp4", ..): 0 = round 4's three-per-CU form, 2 = round 5's compact form, 3 = the
    neighbour scan on all waves, 4 = the default without the raised priority of the serial section) hold the same oracle comparison as the default (1: the serial
    wave rotates with the sequence and runs at raised priority, the residual is normalised by a second pass over LDS)."""
    from lantern_amd import _lib
    _lib.set_tuning("epw_tp4", tp4)
    try:
        test_window_two_workgroups_per_cu_build("mc_sim_7b_63", 33)
    finally:
        _lib.set_tuning("epw_tp4", 1)


@pytest.mark.parametrize("tree_name,REP", [("mc_sim_7b_63", 33), ("six_wide", 33), ("six_wide", 2)])
def test_window_two_workgroups_per_cu_build(tree_name, REP):
    """Above 256 sequences per launch the throughput build of the windowed kernel runs (two workgroups per CU): 264 sequences
    (8 distinct full-size Lumina steps, tiled) must reproduce the oracle exactly like the one-per-CU build does.  `six_wide`: a tree of the
    default tree's sizes with six children under the root, in the throughput build and (REP 2: 16 sequences) in the latency build."""
    V, K, lo, W, k = 65536, 8192, 4, 8192, 1000
    rs = np.random.RandomState(77)
    tree = CS.mc_sim_7b_63 if tree_name == "mc_sim_7b_63" else SIX_WIDE_26
    tb = oracle.tree_static_build(tree)
    assert (len(tb["tree_indices"]),) + tuple(tb["retrieve_indices"].shape) == (26, 15, 6)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    ti, pos = tb["tree_indices"], tb["tree_position_ids"]
    par = CS.node_parents(tb["tree_attn_mask"], pos)
    R = int(((ti[1:] - 1) // 10).max()) + 1
    par_row = np.zeros(R, np.int64)
    for n in range(1, N):
        par_row[(ti[n] - 1) // 10] = par[n]
    depth_of_row = pos[par_row]
    op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
    tab = perm_table(K, 1008, 5)
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    U = 8
    cfg_o, cfg_h = oracle.EpConfig.lumina(True, lantern=True, k=k, delta=0.1), ops.EpConfig.lumina(True, lantern=True, k=k, delta=0.1)
    wins, ops_w, cands, cps, tcs, full = [], [], [], [], [], []
    for b in range(U):
        nlw = CS.topk_filter((4 * rs.standard_normal((N, W))).astype(np.float32), 2000)
        base = np.where(np.isfinite(nlw[par_row]), nlw[par_row], -30.0)
        drw = CS.topk_filter((base + (1.0 + 0.5 * b) * rs.standard_normal((R, W))).astype(np.float32), 2000)
        opw = CS.softmax64(drw).astype(np.float32)
        sst = np.stack([rs.choice(W, 10, replace=False, p=opw[r].astype(np.float64) / opw[r].astype(np.float64).sum()) for r in range(R)]) + lo
        opd = np.zeros((R, V), np.float32); opd[:, lo:lo + W] = opw
        c, cp, tc = oracle.gather_candidates(sst, CS.ss_prob_from(opd, sst), 100 + b, ti, tb["retrieve_indices"])
        nld = np.full((N, V), -np.inf, np.float32); nld[:, lo:lo + W] = nlw
        wins.append(nlw); ops_w.append(opw); cands.append(c); cps.append(cp); tcs.append(tc); full.append((nld, opd))
    uni = rs.random_sample((U, 64))
    tile = lambda a: np.concatenate([np.stack(a)] * REP)
    aux = ops.StaticAux(cart_prob=dev(tile(cps)), orig_prob=dev(tile(ops_w)), op_off=dev(op_off), p_idx=dev(tb["p_indices"]),
                        b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(tile(tcs)))
    out = ops.evaluate_posterior_window(cfg_h, V, dev(tile(wins)), lo, dev(ri), dev(tile(cands)), dev(np.concatenate([uni] * REP)),
                                        table=ops.pack_vq_table(dev(tab.view(np.int16)), 1008), aux=aux, orig_windowed=True, want_dense=False)
    assert out["best"].shape[0] == U * REP and (REP < 33 or U * REP > 256)
    for b in range(U):
        a = oracle.StaticAux(cart_prob=cps[b], orig_prob=full[b][1], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"],
                             tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, full[b][0], ri, cands[b], uni[b], table=tab, aux=a)
        for rep in range(REP):
            j = rep * U + b
            assert int(out["counters"][j, 5]) == 0
            assert (int(out["best"][j]), int(out["accept_len"][j])) == (ob, oa), (j, b)
            assert np.array_equal(out["counters"][j, :5].cpu().numpy(), ocnt[:5])
        np.testing.assert_allclose(out["sample_win"][b].cpu().numpy(), osp[lo:lo + W], rtol=0, atol=PROB_TOL)
        assert torch.equal(out["sample_win"][b], out["sample_win"][(REP - 1) * U + b])


@pytest.mark.parametrize("Vp", [2048, 4096])
@pytest.mark.parametrize("static", [False, True])
def test_window_mid_size_vocabularies(Vp, static):
    """Window widths 2048 and 4096 select their own workgroup shapes (256x2 / 512x2 float4 per thread) in both windowed kernels:
    a V == K vocabulary of that size (LlamaGen-style: no mask, processors inside evaluate_posterior), oracle vs both kernel sets."""
    rs = np.random.RandomState(Vp + int(static))
    tb = oracle.tree_static_build(CS.mc_sim_7b_63 if static else naive_extend_57)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    ti, pos = tb["tree_indices"], tb["tree_position_ids"]
    par = CS.node_parents(tb["tree_attn_mask"], pos)
    R = int(((ti[1:] - 1) // 10).max()) + 1
    par_row = np.zeros(R, np.int64)
    for n in range(1, N):
        par_row[(ti[n] - 1) // 10] = par[n]
    depth_of_row = pos[par_row]
    op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
    tab = perm_table(Vp, 400, 3)
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    B = 4
    kw = dict(lantern=True, k=300, delta=0.2, temperature=0.9, top_p=1.0, top_k=500)
    cfg_o, cfg_h = oracle.EpConfig.llamagen(static, **kw), ops.EpConfig.llamagen(static, **kw)
    nls, ops_l, cands, cps, tcs = [], [], [], [], []
    for b in range(B):
        nl = (4 * rs.standard_normal((N, Vp))).astype(np.float32)
        dr = nl[par_row] + (1.0 + b) * rs.standard_normal((R, Vp)).astype(np.float32)
        op = CS.softmax64(CS.topk_filter(dr, 500)).astype(np.float32)
        sst = np.stack([rs.choice(Vp, 10, replace=False, p=op[r].astype(np.float64) / op[r].astype(np.float64).sum()) for r in range(R)])
        c, cp, tc = oracle.gather_candidates(sst, CS.ss_prob_from(op, sst), int(rs.randint(0, Vp)), ti, tb["retrieve_indices"])
        nls.append(nl); ops_l.append(op); cands.append(c); cps.append(cp); tcs.append(tc)
    uni = rs.random_sample((B, 64))
    aux = None
    if static:
        aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack(ops_l)), op_off=dev(op_off), p_idx=dev(tb["p_indices"]),
                            b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(np.stack(tcs)))
    dense = ops.evaluate_posterior(cfg_h, dev(np.stack(nls)), dev(ri), dev(np.stack(cands)), dev(uni), table=dev(tab.view(np.int16)), aux=aux)
    win = ops.evaluate_posterior_window(cfg_h, Vp, dev(np.stack(nls)), 0, dev(ri), dev(np.stack(cands)), dev(uni), table=dev(tab.view(np.int16)),
                                        aux=aux, want_dense=True)
    # O7w at this width: probability rows (processors applied there) must give the same decisions as logits rows
    pw, hot = ops.cfg_mask_topk_window(dev(np.stack(nls)).reshape(B * N, Vp), None, 1.0, 0, Vp, model=ops.MODEL_PLAIN, top_k=500, temperature=0.9,
                                       probs=True)
    cfg_p = ops.EpConfig.llamagen(static, lantern=True, k=300, delta=0.2, temperature=1.0, top_p=1.0, top_k=0)
    winp = ops.evaluate_posterior_window(cfg_p, Vp, pw.reshape(B, N, Vp), 0, dev(ri), dev(np.stack(cands)), dev(uni),
                                         table=dev(tab.view(np.int16)), aux=aux, want_dense=True, rows_probs=True, row_hot=hot.reshape(B, N))
    for b in range(B):
        a = None
        if static:
            a = oracle.StaticAux(cart_prob=cps[b], orig_prob=ops_l[b], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"],
                                 tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, nls[b], ri, cands[b], uni[b], table=tab, aux=a)
        for best, alen, sp, cnt in ((dense[0], dense[1], dense[2], dense[3]), (win["best"], win["accept_len"], win["sample_p"], win["counters"]),
                                    (winp["best"], winp["accept_len"], winp["sample_p"], winp["counters"])):
            assert int(cnt[b, 5]) == 0
            assert (int(best[b]), int(alen[b])) == (ob, oa), (b, int(best[b]), int(alen[b]), ob, oa)
            assert np.array_equal(cnt[b, :5].cpu().numpy(), ocnt[:5])
            np.testing.assert_allclose(sp[b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)


@pytest.mark.parametrize("delta", [0.1, 5.0])
def test_c3_lumina_dynamic_tree_full_size(delta):
    """C3 with the EAGLE-2 dynamic tree (eagle_version 2): V=65536, window [4,8196), N=59, k=1000, LANTERN delta- and lambda-mode.
    Drafter tree from the HIP O3/O4 kernels (checked against the oracle), O7w over a position range that crosses a newline row,
    then O8w on probability rows with the packed neighbour table, batched over sequences -- against oracle O7 -> O8."""
    V, K, lo, W, B, k10 = 65536, 8192, 4, 8192, 3, 10
    rs = np.random.RandomState(77 + int(delta))
    depth, T = 4, 58
    N = T + 1
    tab = perm_table(K, K - 1, 3)
    packed = ops.pack_vq_table(dev(tab.view(np.int16)), 1008)
    cfg_o = oracle.EpConfig.lumina(False, lantern=True, k=1000, delta=delta)
    cfg_h = ops.EpConfig.lumina(False, lantern=True, k=1000, delta=delta)
    prompt = 20
    seq_len = np.array([prompt + 3 + 5, prompt + 3 + 46, prompt + 3 + 300], np.int64)      # the 2nd sequence's tree straddles a newline
    drafts, rets, poss, cands, ris = [], [], [], [], []
    P_max = D_max = 0
    for b in range(B):
        def img_logits(rows):
            x = np.full((rows, V), -np.inf, np.float32)
            x[:, lo:lo + W] = 4 * rs.standard_normal((rows, W))
            return CS.topk_filter(x, 2000)
        script = [img_logits(1 if d == 0 else k10) for d in range(depth + 1)]
        ti, cu, ci, sc = ops.expand_dynamic(dev(script[0])[None], None, k10)
        sl, tl, pl = [cu.reshape(-1)], [ti.reshape(-1)], [torch.zeros(1, dtype=torch.int64, device="cuda")]
        cs = torch.arange(k10, device="cuda")
        for d in range(depth):
            pl.append(cs + 1 + k10 * k10 * max(0, d - 1) + (k10 if d > 0 else 0))
            ti, cu, ci, sc = ops.expand_dynamic(dev(script[d + 1])[None], sc, k10)
            cs = ci[0]
            sl.append(cu.reshape(-1)); tl.append(ti.reshape(-1))
        sample = int(rs.randint(lo, lo + W))
        draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(torch.cat(sl)[None], torch.cat(tl)[None], torch.cat(pl)[None],
                                                                   torch.tensor([sample], device="cuda"), k10, T, sort_rows=True)
        od, oret, omask, opos = oracle.tree_dynamic_finalize(torch.cat(sl).cpu().numpy(), torch.cat(tl).cpu().numpy(),
                                                             torch.cat(pl).cpu().numpy(), k10, T, sample, sort_rows=True)
        nl, md = int(nl[0]), int(md[0])
        assert np.array_equal(draft[0].cpu().numpy(), od) and np.array_equal(ret[0, :nl, :md].cpu().numpy(), oret)
        assert np.array_equal(pos[0].cpu().numpy(), opos)
        drafts.append(od); rets.append(oret); poss.append(opos)
        P_max, D_max = max(P_max, oret.shape[0]), max(D_max, oret.shape[1])
    # target rows: bf16 cond / uncond pairs whose CFG mix makes the drafted tokens plausible
    cond = (2 * rs.standard_normal((B, N, V))).astype(np.float32)
    unc = rs.standard_normal((B, N, V)).astype(np.float32)
    for b in range(B):
        oret, od = rets[b], drafts[b]
        for p in range(oret.shape[0]):
            for d in range(1, oret.shape[1]):
                if oret[p, d] >= 0:
                    cond[b, oret[p, d - 1], od[oret[p, d]]] = cond[b, oret[p, d - 1], lo:lo + W].max() - rs.uniform(0, 1.0)
    cond_t, unc_t = torch.from_numpy(cond).to(torch.bfloat16), torch.from_numpy(unc).to(torch.bfloat16)
    cb, ub = cond_t.view(torch.int16).numpy().view(np.uint16), unc_t.view(torch.int16).numpy().view(np.uint16)
    pos1 = np.stack(poss) + 1                                                      # [B,N]: every sequence has its own tree
    procs = [oracle.cfg_mask_topk(cb[b], ub[b], 3.0, model=oracle.MODEL_LUMINA, pos_ids=pos1[b] + seq_len[b], pos_base=prompt + 3,
                                  top_k=2000, bf16=True) for b in range(B)]
    cand = np.full((B, P_max, D_max), -1, np.int64)
    ri = np.zeros((B, P_max, D_max), np.int32)
    for b in range(B):
        oret, od = rets[b], drafts[b]
        c = np.where(oret >= 0, od[np.maximum(oret, 0)], -1)
        cand[b, :c.shape[0], :c.shape[1]] = c
        r = H.row_index_from_retrieve(oret, N)
        ri[b, :r.shape[0], :r.shape[1]] = r
    assert any(int(np.isfinite(procs[1][n]).sum()) == 1 for n in range(N))         # a forced newline row is among the nodes
    pw_rows = []
    for b in range(B):                 # per-sequence position ids: one O7w call per sequence (pos_ids is shared by a batch)
        pw, hot = ops.cfg_mask_topk_window(cond_t[b].cuda(), unc_t[b].cuda(), 3.0, lo, W, model=ops.MODEL_LUMINA, pos_ids=dev(pos1[b] + seq_len[b]),
                                           pos_base=prompt + 3, top_k=2000, probs=True)
        pw_rows.append((pw, hot))
    win = torch.stack([p for p, _ in pw_rows])
    hot = torch.stack([h for _, h in pw_rows])
    uni = rs.random_sample((B, 64))
    ubon = rs.random_sample(B)
    out = ops.evaluate_posterior_window(cfg_h, V, win, lo, dev(ri), dev(cand), dev(uni), row_hot=hot, table=packed, u_bonus=dev(ubon),
                                        want_dense=True, rows_probs=True)
    ops.raise_on_status(out["counters"])
    tried = 0
    for b in range(B):
        Pb, Db = rets[b].shape
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, procs[b], ri[b, :Pb, :Db], cand[b, :Pb, :Db], uni[b], table=tab)
        assert (int(out["best"][b]), int(out["accept_len"][b])) == (ob, oa), b
        assert np.array_equal(out["counters"][b, :5].cpu().numpy(), ocnt[:5])
        np.testing.assert_allclose(out["sample_p"][b].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
        assert int(out["token"][b]) == oracle.sample_inverse_cdf(osp, ubon[b])
        tried += int(ocnt[1])
    assert tried >= B


def test_llamagen_throughput_instance_two_per_cu():
    """C2 at more sequences than CUs: LlamaGen's standard verify on probability rows takes `epw_kernel<512, 8, 1, 4, true, false, 5, ..>` (two workgroups per CU on the
    16384-id window, EwSharedLite).  264 sequences = 6 distinct steps with their own EAGLE-2 trees, tiled: the oracle's results, and bit for bit those of the
    generic one-per-CU instance (lantern_tuning_set("epw_tp_lg", 0)) and of the measurement variants (2: the residual normalised by a second LDS pass, 3: rows
    through registers instead of LDS-DMA, 4: raised priority of the serial section)."""
    from lantern_amd import _lib
    V, k, depth, T, U, REP = 16384, 10, 4, 58, 6, 44
    N, P, D = T + 1, T + 1, depth + 2
    rs = np.random.RandomState(2121)
    cfg_o = oracle.EpConfig.llamagen(False, lantern=False, temperature=1.0, top_p=1.0, top_k=2000)
    cfg_h = ops.EpConfig.llamagen(False, lantern=False, temperature=1.0, top_p=1.0, top_k=0)          # (probability rows are final)
    rows, cands, ris, nps, nds, ref = [], [], [], [], [], []
    for b in range(U):
        script = [(4 * rs.standard_normal((1 if d == 0 else k, V))).astype(np.float32) for d in range(depth + 1)]
        ti, cu, ci, sc = oracle.expand_dynamic(CS.topk_filter(script[0], 2000), None, k)
        sl, tl, pl, cs = [cu.reshape(-1)], [ti.reshape(-1)], [np.zeros(1, np.int64)], np.arange(k)
        for d in range(depth):
            pl.append(cs + 1 + k * k * max(0, d - 1) + (k if d > 0 else 0))
            ti, cu, ci, sc = oracle.expand_dynamic(CS.topk_filter(script[d + 1], 2000), sc, k)
            cs = ci
            sl.append(cu.reshape(-1)); tl.append(ti.reshape(-1))
        od, oret, omask, opos = oracle.tree_dynamic_finalize(np.concatenate(sl), np.concatenate(tl), np.concatenate(pl), k, T, int(rs.randint(0, V)))
        node_logits = (4 * rs.standard_normal((N, V))).astype(np.float32)
        for p in range(oret.shape[0]):
            for d in range(1, oret.shape[1]):
                if oret[p, d] >= 0:
                    node_logits[oret[p, d - 1], od[oret[p, d]]] = node_logits[oret[p, d - 1]].max() - rs.uniform(0, 2 + b)
        cand = np.where(oret >= 0, od[np.maximum(oret, 0)], -1)
        uni = rs.random_sample(64)
        ref.append(oracle.evaluate_posterior(cfg_o, node_logits, H.row_index_from_retrieve(oret, N), cand, uni) + (uni,))
        pr, _ = ops.cfg_mask_topk_window(dev(node_logits), None, 1.0, 0, V, model=ops.MODEL_PLAIN, img_lo=0, img_hi=V, top_k=2000, temperature=1.0, probs=True)
        padded = np.full((P, D), -1, np.int64)
        padded[:oret.shape[0], :oret.shape[1]] = oret
        cpad = np.full((P, D), -1, np.int64)
        cpad[:oret.shape[0], :oret.shape[1]] = cand
        rows.append(pr); cands.append(cpad); ris.append(H.row_index_from_retrieve(padded, N)); nps.append(oret.shape[0]); nds.append(oret.shape[1])
    tile = lambda a, dt: dev(np.concatenate([np.stack(a)] * REP).astype(dt))
    win = torch.cat([torch.stack(rows)] * REP)
    args = (cfg_h, V, win, 0, tile(ris, np.int32), tile(cands, np.int64), tile([r[4] for r in ref], np.float64))
    kw = dict(n_paths=tile(nps, np.int32), n_depth=tile(nds, np.int32), rows_probs=True, want_dense=False,
              u_bonus=dev(np.tile(rs.random_sample(U), REP)))
    outs = {}
    try:
        for knob in (1, 2, 3, 4, 0):
            _lib.set_tuning("epw_tp_lg", knob)
            outs[knob] = ops.evaluate_posterior_window(*args, **kw)
            torch.cuda.synchronize()
    finally:
        _lib.set_tuning("epw_tp_lg", 1)
    out = outs[1]
    assert out["best"].shape[0] == U * REP > 256
    for key in ("best", "accept_len", "counters", "sample_win", "out_tok", "out_mass", "token"):
        assert all(torch.equal(outs[v][key], out[key]) for v in (0, 2, 3, 4)), key
    for b in range(U):
        ob, oa, osp, ocnt, _ = ref[b]
        for rep in (0, REP // 2, REP - 1):
            j = rep * U + b
            assert int(out["counters"][j, 5]) == 0
            assert (int(out["best"][j]), int(out["accept_len"][j])) == (ob, oa), (j, b)
            assert np.array_equal(out["counters"][j, :5].cpu().numpy(), ocnt[:5])
            np.testing.assert_allclose(out["sample_win"][j].cpu().numpy(), osp, rtol=0, atol=PROB_TOL)
            assert int(out["token"][j]) == oracle.sample_inverse_cdf(out["sample_win"][j].cpu().numpy(), float(kw["u_bonus"][j]))

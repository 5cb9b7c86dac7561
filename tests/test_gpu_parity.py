"""GPU (-m gpu): the HIP path, called through the C-ABI, against (a) the golden vectors captured
from the reference and (b) the oracle on the same seeded inputs.  Integer outputs (best path,
accept length, uniforms consumed, tokens, tree buffers, gathered bytes) must be bit-exact;
probabilities within 1e-5 (north_star), in practice ~1e-7."""
import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle
from lantern_amd import ops

pytestmark = pytest.mark.gpu
SPECS = H.ep_specs()
PROB_TOL = 1e-5


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def table_dev(K):
    return dev(H.table(K).view(np.int16))


def hip_cfg(spec):
    o = H.ep_config(spec)
    return ops.EpConfig(mode=o.mode, syntax_shortcut=o.syntax_shortcut, tok_offset=o.tok_offset, img_lo=o.img_lo,
                        img_hi=o.img_hi, syntax=tuple(o.syntax), lantern=o.lantern, k=o.k, delta=o.delta,
                        temperature=o.temperature, top_p=o.top_p, top_k=o.top_k)


def _supported(spec):
    return True          # (top_p < 1 included: TopPLogitsWarper runs inside the dense kernel per visited row)


def run_static(spec, case):
    tb, g = H.static_inputs(spec, case)
    m = CS.MODELS[spec["model"]]
    N = len(tb["tree_indices"])
    aux = ops.StaticAux(cart_prob=dev(case["cart_prob"])[None], orig_prob=dev(g["orig_prob"])[None], op_off=dev(g["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]),
                        b_idx=dev(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)), tree_cand=dev(case["tree_cand"])[None])
    return ops.evaluate_posterior(hip_cfg(spec), dev(g["node_logits"])[None], dev(H.row_index_from_retrieve(tb["retrieve"], N)),
                                  dev(case["cand"])[None], dev(case["uniforms"])[None], table=table_dev(m["K"]), aux=aux)


def run_dynamic(spec, case):
    nl, uniforms = H.dynamic_node_logits(spec, case)
    m = CS.MODELS[spec["model"]]
    N = len(case["draft_tokens"])
    return ops.evaluate_posterior(hip_cfg(spec), dev(nl)[None], dev(H.row_index_from_retrieve(case["retrieve"], N)),
                                  dev(case["cand"])[None], dev(uniforms)[None], table=table_dev(m["K"]))


def check_ep(out, case):
    best, alen, sp, cnt = [x.cpu().numpy() for x in out]
    assert cnt[0, 5] == 0, f"status {cnt[0, 5]}"
    assert int(best[0]) == int(case["best"])
    assert int(alen[0]) == int(case["accept_len"])
    assert int(cnt[0, 3]) == int(case["n_draws"])
    np.testing.assert_allclose(sp[0], case["sample_p"], rtol=0, atol=PROB_TOL)


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "static" and _supported(s)])
def test_evaluate_posterior_static_golden(i):
    check_ep(run_static(SPECS[i], H.ep_case(i)), H.ep_case(i))


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "dynamic" and _supported(s)])
def test_evaluate_posterior_dynamic_golden(i):
    check_ep(run_dynamic(SPECS[i], H.ep_case(i)), H.ep_case(i))


def test_reference_cases_with_top_p_exist():
    """The reference-generated cases include nucleus filtering (prepare_logits_processor with top_p < 1): the golden tests above run them."""
    assert any(0.0 < s.get("top_p", 1.0) < 1.0 for s in SPECS if s["kind"] in ("static", "dynamic"))


def test_evaluate_posterior_batched_with_cursor():
    """Many sequences per launch, shared static tree, per-sequence uniform cursor: every sequence
    must equal its own B=1 oracle run, and the cursor must advance by the draws consumed."""
    name = "mc_sim_7b_63"
    tb = H.tree_buffers(name)
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["pos"], tree_attn_mask=tb["mask"],
                retrieve_indices=tb["retrieve"])
    m = CS.MODELS["lumina"]
    N = len(tb["tree_indices"])
    B = 24
    gs, cands, cps, tcs = [], [], [], []
    for s in range(B):
        g = CS.gen_static(5000 + s, "lumina", bufs, sigma=[0.5, 1.0, 1.5][s % 3])
        ssp = CS.ss_prob_from(g["orig_prob"], g["ss_token"])
        cand, cp, tc = oracle.gather_candidates(g["ss_token"], ssp, g["sample_token"], tb["tree_indices"], tb["retrieve"])
        gs.append(g); cands.append(cand); cps.append(cp); tcs.append(tc)
    cfg_o = oracle.EpConfig.lumina(True, lantern=True, k=300, delta=0.1)
    cfg_o.img_hi, cfg_o.syntax = m["img_hi"], m["syntax"]
    cfg_h = ops.EpConfig(mode=cfg_o.mode, syntax_shortcut=True, tok_offset=4, img_lo=4, img_hi=m["img_hi"], syntax=m["syntax"],
                         lantern=True, k=300, delta=0.1)
    start = np.arange(B) % 5
    uni = np.stack([np.concatenate([np.full(start[s], 0.123), g["uniforms"]])[:64] for s, g in enumerate(gs)])
    cursor = dev(start.astype(np.int32))
    aux = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(np.stack([g["orig_prob"] for g in gs])), op_off=dev(gs[0]["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(np.stack(tcs)))
    out = ops.evaluate_posterior(cfg_h, dev(np.stack([g["node_logits"] for g in gs])), dev(H.row_index_from_retrieve(tb["retrieve"], N)),
                                 dev(np.stack(cands)), dev(uni), table=table_dev(m["K"]), aux=aux, cursor=cursor)
    best, alen, sp, cnt = [x.cpu().numpy() for x in out]
    for s in range(B):
        a = oracle.StaticAux(cart_prob=cps[s], orig_prob=gs[s]["orig_prob"], op_off=gs[s]["op_off"], p_idx=tb["p_indices"],
                             b_off=tb["b_off"], b_idx=tb["b_idx"], tree_cand=tcs[s])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, gs[s]["node_logits"], H.row_index_from_retrieve(tb["retrieve"], N),
                                                      cands[s], uni[s, start[s]:], table=H.table(m["K"]), aux=a)
        assert (best[s], alen[s]) == (ob, oa), s
        assert np.array_equal(cnt[s, :5], ocnt[:5]), s
        np.testing.assert_allclose(sp[s], osp, rtol=0, atol=PROB_TOL)
    assert np.array_equal(cursor.cpu().numpy(), start + cnt[:, 3])


@pytest.mark.parametrize("mode,delta", [("static", 0.1), ("static", 5.0), ("dynamic", 0.1), ("dynamic", 5.0)])
def test_evaluate_posterior_full_size_lumina(mode, delta):
    """BASELINE config C3 shapes: V=65536, K=8192, k=1000, real token ids; HIP vs oracle."""
    V, K, off = 65536, 8192, 4
    rs = np.random.RandomState(77)
    table = np.stack([rs.permutation(K - 1) for _ in range(64)]).astype(np.uint16)   # rows for 64 codes only
    # map every code to one of the 64 stored rows to keep the fixture small; entries skip `self`
    full = np.zeros((K, K - 1), np.uint16)
    for c in range(K):
        row = table[c % 64].astype(np.int64)
        full[c] = np.where(row >= c, row + 1, row).astype(np.uint16)
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    B = 6
    cfg_o = oracle.EpConfig.lumina(mode == "static", lantern=True, k=1000, delta=delta)
    cfg_h = ops.EpConfig.lumina(mode == "static", lantern=True, k=1000, delta=delta)
    logits = np.full((B, N, V), -np.inf, np.float32)
    logits[:, :, 4:8196] = (4 * rs.standard_normal((B, N, K))).astype(np.float32)
    logits = np.stack([CS.topk_filter(l, 2000) for l in logits])
    R = 11
    dr = np.full((B, R, V), -np.inf, np.float32)
    dr[:, :, 4:8196] = (4 * rs.standard_normal((B, R, K))).astype(np.float32)
    par_row = np.zeros(R, np.int64)
    pos, ti = tb["tree_position_ids"], tb["tree_indices"]
    par = CS.node_parents(tb["tree_attn_mask"], pos)
    for n in range(1, N):
        par_row[(ti[n] - 1) // 10] = par[n]
    dr = 0.5 * dr + 0.5 * logits[:, par_row]          # drafter correlated with the target
    dr = np.where(np.isfinite(dr), dr, -np.inf).astype(np.float32)
    op = np.stack([CS.softmax64(CS.topk_filter(x, 2000)).astype(np.float32) for x in dr])
    depth_of_row = pos[par_row]
    op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
    outs = []
    cands, cps, tcs = [], [], []
    for b in range(B):
        sst = np.stack([rs.choice(V, 10, replace=False, p=op[b, r].astype(np.float64) / op[b, r].astype(np.float64).sum()) for r in range(R)])
        ssp = CS.ss_prob_from(op[b], sst)
        c, cp, tc = oracle.gather_candidates(sst, ssp, 100 + b, ti, tb["retrieve_indices"])
        cands.append(c); cps.append(cp); tcs.append(tc)
    uni = rs.random_sample((B, 64))
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    aux_h = None
    if mode == "static":
        aux_h = ops.StaticAux(cart_prob=dev(np.stack(cps)), orig_prob=dev(op), op_off=dev(op_off), p_idx=dev(tb["p_indices"]),
                              b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(np.stack(tcs)))
    out = ops.evaluate_posterior(cfg_h, dev(logits), dev(ri), dev(np.stack(cands)), dev(uni), table=dev(full.view(np.int16)), aux=aux_h)
    best, alen, sp, cnt = [x.cpu().numpy() for x in out]
    for b in range(B):
        a = None
        if mode == "static":
            a = oracle.StaticAux(cart_prob=cps[b], orig_prob=op[b], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"],
                                 b_idx=tb["b_idx"], tree_cand=tcs[b])
        ob, oa, osp, ocnt = oracle.evaluate_posterior(cfg_o, logits[b], ri, cands[b], uni[b], table=full, aux=a)
        assert cnt[b, 5] == 0
        assert (best[b], alen[b]) == (ob, oa), b
        assert np.array_equal(cnt[b, :5], ocnt[:5]), (b, cnt[b], ocnt)
        np.testing.assert_allclose(sp[b], osp, rtol=0, atol=PROB_TOL)


def _bf16(x):
    return torch.from_numpy(x).to(torch.bfloat16).cuda()


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_cfg_mask_topk_golden(tag):
    g = H.load("o7.npz")
    m = CS.MODELS["lumina"]
    cond, unc = (dev(g["cond"]), dev(g["uncond"])) if tag == "f32" else (_bf16(g["cond"]), _bf16(g["uncond"]))
    kw = dict(w=int(g["w"]), h=int(g["h"]), img_lo=m["img_lo"], img_hi=m["img_hi"], newline_id=m["syntax"][2], eos_id=m["syntax"][0])
    out = ops.cfg_mask_topk(cond, unc, 3.0, model=ops.MODEL_LUMINA, pos_ids=dev(g["pos"]), pos_base=int(g["img_start"]) + 3,
                            top_k=100, **kw)
    assert np.array_equal(out.cpu().numpy(), g[f"lumina_{tag}"])
    out = ops.cfg_mask_topk(cond, unc, 3.0, model=ops.MODEL_ANOLE, **kw)
    assert np.array_equal(out.cpu().numpy(), g[f"anole_{tag}"])
    out = ops.cfg_mask_topk(cond, unc, 3.0, model=ops.MODEL_PLAIN, **kw)
    assert np.array_equal(out.cpu().numpy(), g[f"plain_{tag}"])


def test_cfg_mask_topk_full_size_vs_oracle():
    V = 65536
    rs = np.random.RandomState(9)
    N = 8
    cond = torch.from_numpy((4 * rs.standard_normal((N, V))).astype(np.float32)).to(torch.bfloat16)
    unc = torch.from_numpy((4 * rs.standard_normal((N, V))).astype(np.float32)).to(torch.bfloat16)
    pos = np.array([103, 104, 150, 151, 152, 2453, 2454, 200], np.int64)   # includes newline rows and the eos row
    out = ops.cfg_mask_topk(cond.cuda(), unc.cuda(), 3.0, model=ops.MODEL_LUMINA, pos_ids=dev(pos), pos_base=100 + 3, top_k=2000)
    exp = oracle.cfg_mask_topk(cond.view(torch.int16).numpy().view(np.uint16), unc.view(torch.int16).numpy().view(np.uint16), 3.0,
                               model=oracle.MODEL_LUMINA, pos_ids=pos, pos_base=103, top_k=2000, bf16=True)
    assert np.array_equal(out.cpu().numpy(), exp)
    # unmasked full-width selection (the slow generic path) with f32 inputs
    out = ops.cfg_mask_topk(cond.float().cuda(), unc.float().cuda(), 2.5, model=ops.MODEL_PLAIN, top_k=3000)
    exp = oracle.cfg_mask_topk(cond.float().numpy(), unc.float().numpy(), 2.5, model=oracle.MODEL_PLAIN, top_k=0)
    exp = CS.topk_filter(exp, 3000)
    assert np.array_equal(out.cpu().numpy(), exp)


def test_kv_and_accept_gather_golden():
    g = H.load("kv.npz")
    slab = dev(g["before"])
    best = dev(np.array([int(g["best"])], np.int32))
    alen = dev(np.array([int(g["accept_len"])], np.int32))
    new_len = ops.kv_gather([slab], dev(np.zeros(1, np.int32)), dev(np.array([int(g["prev"])], np.int64)), dev(g["retrieve"]),
                            best, alen)
    assert np.array_equal(slab.cpu().numpy(), g["after"])
    assert int(new_len[0]) == int(g["current_length"][0])
    hid = dev(g["hidden"])[None]          # [B=1,G=1,N,H]
    sp = np.zeros(20, np.float32)
    sp[7] = 1.0
    cand = np.arange(15).reshape(1, 3, 5)
    # H=6 floats = 24 bytes is not 16-byte aligned: pad hidden to 8 columns for the kernel
    hid8 = torch.zeros(1, 1, hid.shape[2], 8, device="cuda")
    hid8[..., :6] = hid
    out_h, acc, tok = ops.accept_gather(hid8, dev(g["retrieve"]), dev(cand), best, alen, sample_p=dev(sp)[None],
                                        u=dev(np.array([0.3])))
    n = int(g["accept_len"]) + 1
    assert np.array_equal(out_h.cpu().numpy()[0, :, :n, :6], g["accept_hidden"])
    assert np.all(out_h.cpu().numpy()[0, :, n:] == 0)
    assert int(tok[0]) == 7 == int(g["token"].reshape(-1)[0])
    exp_ids = g["new_ids"][0, int(g["prev"]):]
    assert np.array_equal(acc.cpu().numpy()[0, :n], exp_ids)
    assert np.all(acc.cpu().numpy()[0, n:] == -1)


def test_kv_gather_lumina_geometry_vs_oracle():
    """bf16 slab with the 7B head geometry (reduced layers/S_max), two sequences x (cond,uncond) slabs with
    different prev_len, per-sequence retrieve rows."""
    rs = np.random.RandomState(1)
    slabs_np = [rs.randint(0, 65535, size=(4, 1, 32, 96, 128)).astype(np.uint16) for _ in range(4)]
    slabs = [dev(s.view(np.int16)) for s in slabs_np]
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    ret = tb["retrieve_indices"]
    best = np.array([3, 0], np.int32)
    alen = np.array([2, 5], np.int32)
    seq = np.array([0, 0, 1, 1], np.int32)
    prev = np.array([40, 17, 60, 33], np.int64)
    new_len = ops.kv_gather(slabs, dev(seq), dev(prev), dev(ret), dev(best), dev(alen))
    for s in range(4):
        b = seq[s]
        exp = oracle.kv_gather(slabs_np[s].copy(), ret[best[b]], int(alen[b]) + 1, int(prev[s]))
        assert np.array_equal(slabs[s].cpu().numpy().view(np.uint16), exp), s
        assert int(new_len[s]) == prev[s] + alen[b] + 1


def test_update_inference_inputs_fused_equals_separate_ops():
    """O9 + O10 in one launch == lantern_kv_gather followed by lantern_accept_gather (and the oracle)."""
    rs = np.random.RandomState(2)
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    ret = tb["retrieve_indices"]
    P, D = ret.shape
    N = len(tb["tree_indices"])
    B, G, Hd = 3, 2, 256
    slabs_np = [rs.randint(0, 65535, size=(4, 1, 8, 80, 128)).astype(np.uint16) for _ in range(2 * B)]
    a_slabs = [dev(s.view(np.int16)) for s in slabs_np]
    b_slabs = [dev(s.view(np.int16)) for s in slabs_np]
    best = np.array([3, 0, 14], np.int32)
    alen = np.array([2, 5, 0], np.int32)
    seq = np.array([0, 1, 2, 0, 1, 2], np.int32)
    prev = np.array([40, 17, 33, 9, 3, 20], np.int64)
    hidden = torch.randn(B, G, N, Hd, device="cuda").to(torch.bfloat16)
    cand = rs.randint(0, 8192, size=(B, P, D)).astype(np.int64)
    nl_a = ops.kv_gather(a_slabs, dev(seq), dev(prev), dev(ret), dev(best), dev(alen))
    h_a, t_a, _ = ops.accept_gather(hidden, dev(ret), dev(cand), dev(best), dev(alen))
    nl_b, h_b, t_b = ops.update_inference_inputs(b_slabs, dev(seq), dev(prev), dev(ret), dev(best), dev(alen), hidden, dev(cand))
    assert torch.equal(nl_a, nl_b) and torch.equal(h_a, h_b) and torch.equal(t_a, t_b)
    for s in range(2 * B):
        assert torch.equal(a_slabs[s], b_slabs[s]), s
        exp = oracle.kv_gather(slabs_np[s].copy(), ret[best[seq[s]]], int(alen[seq[s]]) + 1, int(prev[s]))
        assert np.array_equal(b_slabs[s].cpu().numpy().view(np.uint16), exp), s
    for b in range(B):
        n = int(alen[b]) + 1
        assert torch.equal(h_b[b, :, :n], hidden[b][:, torch.as_tensor(ret[best[b], :n]).cuda()])
        assert t_b[b, :n].tolist() == cand[b, best[b], :n].tolist() and (t_b[b, n:] == -1).all()


@pytest.mark.parametrize("shape,n_slabs", [((4, 1, 8, 80, 128), 1), ((4, 1, 8, 80, 128), 5), ((4, 1, 8, 80, 128), 11), ((2, 1, 3, 70, 64), 7),
                                           ((24, 1, 16, 40, 128), 3)])
def test_update_inference_inputs_slab_blocks_vs_oracle(shape, n_slabs):
    """The commit with several small slabs per workgroup (slab counts that do not fill the last block, slabs of odd sizes) and, last case, slabs
    too big for it (one workgroup tile per slab piece): every path of the tree somewhere, accept lengths 0 .. depth-1, against the oracle's move."""
    rs = np.random.RandomState(n_slabs + shape[0])
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    ret = tb["retrieve_indices"]
    P, D = ret.shape
    N = len(tb["tree_indices"])
    B = n_slabs
    depth = (ret >= 0).sum(1)
    slabs_np = [rs.randint(0, 65535, size=shape).astype(np.uint16) for _ in range(n_slabs)]
    slabs = [dev(s.view(np.int16)) for s in slabs_np]
    best = rs.randint(0, P, size=B).astype(np.int32)
    alen = np.array([rs.randint(0, depth[b]) for b in best], np.int32)
    seq = rs.permutation(n_slabs).astype(np.int32)
    prev = rs.randint(0, shape[-2] - N - 1, size=n_slabs).astype(np.int64)
    hidden = torch.randn(B, 2, N, 64, device="cuda").to(torch.bfloat16)
    cand = rs.randint(0, 8192, size=(B, P, D)).astype(np.int64)
    nl, h, t = ops.update_inference_inputs(slabs, dev(seq), dev(prev), dev(ret), dev(best), dev(alen), hidden, dev(cand))
    for s in range(n_slabs):
        b = seq[s]
        exp = oracle.kv_gather(slabs_np[s].copy(), ret[best[b]], int(alen[b]) + 1, int(prev[s]))
        assert np.array_equal(slabs[s].cpu().numpy().view(np.uint16), exp), s
        assert int(nl[s]) == prev[s] + alen[b] + 1
    for b in range(B):
        n = int(alen[b]) + 1
        assert torch.equal(h[b, :, :n], hidden[b][:, torch.as_tensor(ret[best[b], :n]).cuda()]) and not h[b, :, n:].any()
        assert t[b, :n].tolist() == cand[b, best[b], :n].tolist() and (t[b, n:] == -1).all()


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] in ("dynamic", "greedy")][::3])
def test_dynamic_tree_golden(i):
    spec, case = SPECS[i], H.ep_case(i)
    depth = int(case["depth"])
    script = H.dynamic_script(spec["seed"], spec["model"], depth)
    k = CS.TOPK
    ti, cu, ci, scores = ops.expand_dynamic(dev(H.hf_process_rows(script[0][None], H.DYN_TOP_K))[None], None, k)
    scores_list, tokens_list = [cu.reshape(-1)], [ti.reshape(-1)]
    parents_list = [torch.zeros(1, dtype=torch.int64, device="cuda")]
    topk_cs_index = torch.arange(k, device="cuda")
    for d in range(depth):
        bias = 1 + k * k * max(0, d - 1) + (k if d > 0 else 0)
        parents_list.append(topk_cs_index + bias)
        ti, cu, ci, scores = ops.expand_dynamic(dev(H.hf_process_rows(script[d + 1], H.DYN_TOP_K))[None], scores, k)
        topk_cs_index = ci[0]
        scores_list.append(cu.reshape(-1))
        tokens_list.append(ti.reshape(-1))
    T = int(case["total_tokens"])
    draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(
        torch.cat(scores_list)[None], torch.cat(tokens_list)[None], torch.cat(parents_list)[None],
        torch.tensor([int(case["sample_token"])], device="cuda"), k, T, sort_rows=True)
    nl, md = int(nl[0]), int(md[0])
    assert np.array_equal(draft[0].cpu().numpy(), case["draft_tokens"])
    assert np.array_equal(ret[0, :nl, :md].cpu().numpy(), case["retrieve"])
    assert torch.all(ret[0, nl:] == -1) and torch.all(ret[0, :, md:] == -1)
    assert np.array_equal(mask[0].cpu().numpy(), case["mask"])
    assert np.array_equal(pos[0].cpu().numpy(), case["pos"])


def test_gather_candidates_and_sample_static_golden():
    i = next(i for i, s in enumerate(SPECS) if s["kind"] == "static")
    spec, case = SPECS[i], H.ep_case(i)
    tb = H.tree_buffers(spec["tree"])
    cand, cp, tc = ops.gather_candidates(dev(case["ss_token"])[None], dev(case["ss_prob"])[None],
                                         dev(np.array([int(case["sample_token"])])), dev(tb["tree_indices"]), dev(tb["retrieve"]))
    assert np.array_equal(cand[0].cpu().numpy(), case["cand"])
    assert np.array_equal(cp[0].cpu().numpy(), case["cart_prob"])
    assert np.array_equal(tc[0].cpu().numpy(), case["tree_cand"])
    g = H.load("sample.npz")
    out = ops.sample_static(dev(g["full"]), dev(g["idx"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["prob"], rtol=0, atol=1e-7)


def test_bonus_token_inverse_cdf_vs_oracle():
    rs = np.random.RandomState(4)
    V = 65536
    p = np.zeros((16, V), np.float32)
    for b in range(16):
        idx = rs.choice(V, 2000, replace=False)
        p[b, idx] = rs.random_sample(2000).astype(np.float32)
        p[b] /= p[b].sum()
    u = rs.random_sample(16)
    u[0], u[1] = 0.0, 1.0 - 1e-12
    best = dev(np.zeros(16, np.int32)); alen = dev(np.zeros(16, np.int32))
    _, _, tok = ops.accept_gather(None, dev(np.zeros((1, 1), np.int64)), None, best, alen, sample_p=dev(p), u=dev(u))
    exp = [oracle.sample_inverse_cdf(p[b], u[b]) for b in range(16)]
    assert tok.cpu().tolist() == exp
    _, _, tok = ops.accept_gather(None, dev(np.zeros((1, 1), np.int64)), None, best, alen, sample_p=dev(p), u=None)
    assert tok.cpu().tolist() == p.argmax(1).tolist()


@pytest.mark.parametrize("D,d", [(9, 128), (12, 64), (16, 128)])
def test_kv_gather_deep_paths(D, d):
    """Paths deeper than 8 nodes take the second kernel variant (up to 16 moved rows per slab row group); LlamaGen's 64-wide heads
    and 128-wide ones; accept lengths from 0 to D-1, retrieve rows with gaps, -1 padded tails."""
    rs = np.random.RandomState(D)
    P, N, S = 5, 40, 96
    ret = np.full((P, D), -1, np.int64)
    for p in range(P):
        n = D - (p % 3)
        ret[p, :n] = np.concatenate([[0], np.sort(rs.choice(np.arange(1, N), size=n - 1, replace=False))])
    slabs_np = [rs.randint(0, 65535, size=(4, 1, 6, S, d)).astype(np.uint16) for _ in range(4)]
    slabs = [dev(s.view(np.int16)) for s in slabs_np]
    best = np.array([0, 4], np.int32)
    alen = np.array([D - 1, 0 if D % 2 else D - 3], np.int32)
    seq = np.array([0, 0, 1, 1], np.int32)
    prev = np.array([11, 30, 0, 47], np.int64)
    new_len = ops.kv_gather(slabs, dev(seq), dev(prev), dev(ret), dev(best), dev(alen))
    for s in range(4):
        b = seq[s]
        exp = oracle.kv_gather(slabs_np[s].copy(), ret[best[b]], int(alen[b]) + 1, int(prev[s]))
        assert np.array_equal(slabs[s].cpu().numpy().view(np.uint16), exp), s
        assert int(new_len[s]) == prev[s] + alen[b] + 1
    with pytest.raises(Exception, match="D="):
        ops.kv_gather(slabs, dev(seq), dev(prev), dev(np.zeros((P, 17), np.int64)), dev(best), dev(alen))

"""GPU (-m gpu): greedy/TVD accept (a9), MFMA drafter input contraction (O11), VQ-distance table builder (8f-1)."""
import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle
from lantern_amd import ops
from test_gpu_parity import dev, table_dev

pytestmark = pytest.mark.gpu
SPECS = H.ep_specs()


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "greedy"])
def test_greedy_golden(i):
    spec, case = SPECS[i], H.ep_case(i)
    nl, _ = H.dynamic_node_logits(spec, case, greedy=True)
    m = CS.MODELS[spec["model"]]
    N = len(case["draft_tokens"])
    best, alen, row = ops.evaluate_posterior_greedy(dev(nl)[None], dev(H.row_index_from_retrieve(case["retrieve"], N)), dev(case["cand"])[None],
                                                    lantern=spec["lantern"], k=spec["k"], delta=spec["delta"], tok_offset=m["off"],
                                                    table=table_dev(m["K"]))
    assert (int(best[0]), int(alen[0])) == (int(case["best"]), int(case["accept_len"]))
    assert np.array_equal(row[0].cpu().numpy(), case["out_row"])


@pytest.mark.parametrize("model,lantern,delta", [("llamagen", True, 0.2), ("llamagen", True, 4.0), ("anole", True, 0.2), ("anole", False, 0.1)])
def test_greedy_batched_vs_oracle(model, lantern, delta):
    """Full-width rows: LlamaGen V=K=16384 whole-vocabulary window; Anole V=65536 with the image window [4,8196)."""
    rs = np.random.RandomState(3)
    if model == "llamagen":
        V, K, off, lo, W = 16384, 16384, 0, 0, 16384
    else:
        V, K, off, lo, W = 65536, 8192, 4, 4, 8192
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    N, (P, D) = len(tb["tree_indices"]), tb["retrieve_indices"].shape
    ri = H.row_index_from_retrieve(tb["retrieve_indices"], N)
    tab = np.stack([rs.permutation(K - 1)[:600] for _ in range(K)]).astype(np.uint16)   # 600 columns are enough for k=500
    tab = np.where(tab >= np.arange(K)[:, None], tab + 1, tab).astype(np.uint16)
    B = 3
    logits = (3 * rs.standard_normal((B, N, V))).astype(np.float32)
    if model == "anole":
        logits[..., :lo] = np.finfo(np.float32).min
        logits[..., lo + W:] = np.finfo(np.float32).min
    cands = []
    for b in range(B):
        tok = rs.randint(lo, lo + W, size=N)
        for n in range(N):                           # make most drafted tokens the (near-)argmax of their parent's row
            par = np.nonzero(tb["tree_attn_mask"][n] > 0)[0]
            par = [a for a in par if tb["tree_position_ids"][a] == tb["tree_position_ids"][n] - 1]
            if par and rs.random_sample() < 0.8:
                logits[b, par[0], tok[n]] = logits[b, par[0]].max() + rs.uniform(-0.3, 1.0)
        c = np.where(tb["retrieve_indices"] >= 0, tok[np.maximum(tb["retrieve_indices"], 0)], -1)
        cands.append(c)
    cand = np.stack(cands).astype(np.int64)
    best, alen, row = ops.evaluate_posterior_greedy(dev(logits), dev(ri), dev(cand), lantern=lantern, k=500, delta=delta, tok_offset=off,
                                                    table=dev(tab.view(np.int16)), win_lo=lo, win_len=W)
    for b in range(B):
        ob, oa, orow = oracle.evaluate_posterior_greedy(logits[b], ri, cand[b], lantern=lantern, k=500, delta=delta, tok_offset=off, table=tab)
        assert (int(best[b]), int(alen[b])) == (ob, oa), (b, int(best[b]), int(alen[b]), ob, oa)
        assert np.array_equal(row[b].cpu().numpy(), orow)
    assert int(alen.max()) > 0


@pytest.mark.parametrize("M,H,scale,bias,packed", [(2, 1280, 1.0, True, False), (20, 4096, 1.0, True, False), (59, 1280, 1.0, False, False),
                                                   (120, 4096, 2.0, True, False), (20, 4096, 2.0, True, True), (32, 1280, 1.0, False, True), (5, 64, 3.0, True, True)])
def test_drafter_fc_mfma_vs_oracle(M, H, scale, bias, packed):
    """O11 against the oracle: up to 32 rows run the stream-K kernel (row-major weight, or packed with `packed`), more rows the per-tile kernel."""
    rs = np.random.RandomState(M + H)
    vocab = 1000
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)
    hidden, embed = bf(rs.standard_normal((M, H))), bf(rs.standard_normal((vocab, H)))
    W = bf(rs.standard_normal((H, 2 * H)) / np.sqrt(2 * H))
    b = bf(rs.standard_normal(H)) if bias else None
    ids = rs.randint(0, vocab, size=M)
    pk = ops.pack_linear_weight(W.cuda()) if packed else None
    out = ops.drafter_fc(dev(ids), hidden.cuda(), embed.cuda(), W.cuda(), None if b is None else b.cuda(), embed_scale=scale, packed=pk)
    bits = lambda t: t.view(torch.int16).numpy().view(np.uint16)
    exp = oracle.drafter_fc(ids, bits(hidden), bits(embed), bits(W), None if b is None else bits(b), embed_scale=scale)
    got = out.float().cpu().numpy()
    exp_bf = torch.from_numpy(exp).to(torch.bfloat16).float().numpy()
    # f32 accumulation over K = 2H bf16 products, then one bf16 rounding: within one bf16 ulp of the f64 reference
    ulp = np.maximum(np.abs(exp_bf), 1e-3) * 2.0 ** -7
    assert np.all(np.abs(got - exp_bf) <= ulp), float(np.max(np.abs(got - exp_bf) / ulp))
    assert np.mean(got == exp_bf) > 0.97
    # A = I check with an asymmetric B: rows of W must land in the right output columns
    Hs = 64
    eye_h = torch.zeros(Hs, Hs); eye_h[torch.arange(Hs), torch.arange(Hs)] = 1
    Wa = torch.arange(Hs * 2 * Hs, dtype=torch.float32).reshape(Hs, 2 * Hs) % 251 - 125
    o2 = ops.drafter_fc(torch.zeros(Hs, dtype=torch.int64).cuda(), eye_h.to(torch.bfloat16).cuda(), torch.zeros(4, Hs, dtype=torch.bfloat16).cuda(),
                        Wa.to(torch.bfloat16).cuda())
    assert torch.equal(o2.float().cpu(), Wa.to(torch.bfloat16).float()[:, Hs:].T.contiguous())


def test_vq_table_builder():
    g = H.load("codebook.npz")
    t = ops.build_vq_table(dev(g["codebook"])).cpu().numpy().view(np.uint16)
    assert np.array_equal(t, oracle.build_vq_table(g["codebook"]))          # same tie rule as the oracle: exact
    r = g["table"]
    cb = g["codebook"].astype(np.float64)
    d = np.sqrt(((cb[:, None] - cb[None]) ** 2).sum(-1))
    for a, c in np.argwhere(t != r):                                           # vs the reference: only float32 ties may differ
        assert abs(d[a, t[a, c]] - d[a, r[a, c]]) <= 1e-6 * d[a, t[a, c]]
    rs = np.random.RandomState(0)
    cb2 = rs.standard_normal((1000, 16)).astype(np.float32)                    # non power-of-two K
    t2 = ops.build_vq_table(dev(cb2)).cpu().numpy().view(np.uint16)
    assert np.array_equal(t2, oracle.build_vq_table(cb2))


def test_vq_table_builder_llamagen_size():
    """K = 16384, C = 8 (LlamaGen's codebook) and K = 9000 (non power of two): the packed-key kernel (distance and index in one 64-bit sort
    key) against the oracle's table (`lo_build_vq_table`, the recipe of generate_codebook.py:53-65 in f32): the whole K = 9000 table and 1024
    sampled rows of the K = 16384 one.  Rows may differ from the oracle only where two codes are equally far within the key's precision
    (the upper 48 bits of the f64 distance); every row is a permutation of the other codes in non-decreasing f64 distance."""
    rs = np.random.RandomState(4)
    for K, Cc, rows in ((9000, 8, None), (16384, 8, 1024)):
        cb = rs.standard_normal((K, Cc)).astype(np.float32)
        t = ops.build_vq_table(dev(cb)).cpu().numpy().view(np.uint16)
        assert t.shape == (K, K - 1)
        cb64 = cb.astype(np.float64)
        ref = oracle.build_vq_table(cb) if rows is None else None
        pick = np.arange(K) if rows is None else np.unique(np.concatenate([[0, 1, 777, 8191, 8192, K - 1], rs.randint(0, K, rows)]))
        n_diff = 0
        for a in pick:
            row = t[a].astype(np.int64)
            d = ((cb64 - cb64[a]) ** 2).sum(-1)
            exp = ref[a].astype(np.int64) if ref is not None else np.argsort(np.where(np.arange(K) == a, np.inf, d), kind="stable")[:K - 1]
            diff = np.nonzero(row != exp)[0]
            n_diff += len(diff)
            if len(diff) or a in (0, K - 1):
                assert a not in row and len(np.unique(row)) == K - 1
                dr = d[row]
                assert (np.diff(dr) >= -1e-9 * dr[1:]).all()
                # a differing position holds a code at the same distance (f32 arithmetic of the reference / 48-bit keys here)
                assert (np.abs(d[row[diff]] - d[exp[diff]]) <= 2e-6 * np.maximum(d[exp[diff]], 1e-30)).all(), a
        assert n_diff <= 2e-4 * len(pick) * K, (K, n_diff)
    packed = ops.pack_vq_table(torch.from_numpy(t.view(np.int16)).cuda(), 1008).cpu().numpy().view(np.uint16)
    assert np.array_equal(packed[:, :1008], t[:, :1008])


@pytest.mark.parametrize("n,K,special", [(10, 512, ""), (1, 256, ""), (10, 256, "newline"), (7, 128, "eos"), (16, 64, "few")])
def test_head_expand_vs_the_oracle_behind_the_gemm(n, K, special):
    """lantern_head_expand_streamk against the ORACLE's restatement of what follows the head's GEMM (cnets_lumina_mgpt.py:1271-1320: CFG in
    bf16, MultiModalLogitsProcessor + InterleavedTopK, log-softmax, top-k, cumulative scores, best k of n * k): the oracle gets the head's
    bf16 logits from the same GEMM kernel (held to f64 in test_drafter_layer) and must give the same token ids, parents and -- to f32
    rounding -- scores."""
    torch.manual_seed(100 * n + K + 7)
    V, lo, W = 65536, 4, 8192
    A = (0.5 * torch.randn(2 * n, K, device="cuda")).to(torch.bfloat16)
    Wt = (0.3 * torch.randn(V, K, device="cuda")).to(torch.bfloat16)
    bias = (0.1 * torch.randn(V, device="cuda")).to(torch.bfloat16)
    if special == "few":
        Wt[lo + 5:lo + W] = 0
        bias[lo + 5:lo + W] = -30000.0
    pos = torch.full((n,), 2 + 3, device="cuda", dtype=torch.int64) + torch.arange(n, device="cuda")
    if special == "newline":
        pos[2] = 2 + 48
    if special == "eos":
        pos[1] = 2 + 49 * 48
    scores_in = torch.randn(n, device="cuda") if n > 1 else None
    tk = 2000 if special != "few" else 3
    pk = ops.pack_linear_weight(Wt[lo:lo + W].contiguous())
    fused = ops.head_expand(A, Wt, lo, W, 3.0, bias=bias, model=ops.MODEL_LUMINA, pos_ids=pos, pos_base=2, top_k_filter=tk, scores_in=scores_in, top_k=10,
                            packed=pk)
    win = ops.linear_rows_streamk(A, pk, bias=bias[lo:lo + W].contiguous())                  # [2n, W] bf16: the head's logits on the image ids
    bits = np.zeros((2 * n, V), np.uint16)
    bits[:, lo:lo + W] = win.cpu().view(torch.int16).numpy().view(np.uint16)
    rows = oracle.cfg_mask_topk(bits[:n], bits[n:], 3.0, model=oracle.MODEL_LUMINA, pos_ids=pos.cpu().numpy(), pos_base=2, w=48, h=48, img_lo=lo,
                                img_hi=lo + W, newline_id=8803, eos_id=8196, top_k=tk, bf16=True)
    ti, cu, ci, so = oracle.expand_dynamic(rows, None if scores_in is None else scores_in.cpu().numpy(), 10)
    assert np.array_equal(fused[0][0].cpu().numpy(), ti) and np.array_equal(fused[2][0].cpu().numpy(), ci)
    fin = np.isfinite(cu)
    got = fused[1][0].cpu().numpy()
    assert np.array_equal(np.isfinite(got), fin) and np.allclose(got[fin], cu[fin], rtol=0, atol=2e-6)
    fs = np.isfinite(so)
    assert np.allclose(fused[3][0].cpu().numpy()[fs], so[fs], rtol=0, atol=2e-6)


@pytest.mark.parametrize("model,n,K,tk", [("anole", 10, 512, 2000), ("anole", 1, 4096, 2000), ("anole", 10, 128, 0), ("anole", 7, 256, 10000),
                                          ("llamagen", 10, 1280, 2000), ("llamagen", 1, 1280, 2000), ("llamagen", 10, 128, 0), ("llamagen", 16, 256, 50), ("anole", 16, 256, 12)])
def test_head_expand_anole_and_llamagen_vs_the_oracle_behind_the_gemm(model, n, K, tk):
    """lantern_head_expand_streamk for the two models beside Lumina (cnets_anole.py:835-903: non-image ids to finfo.min after the CFG mix, then the
    HF processors; cnets_llamagen.py:783-821: the whole 16384-id vocabulary, no mask).  What follows the head's GEMM is restated in torch f32 the way
    the reference computes it -- CFG in bf16 steps, HF TopKLogitsWarper over the FULL row (`scores < kth largest -> -inf`: ties at the threshold stay; a
    top_k wider than Anole's window removes nothing inside it), log-softmax -- and handed to the ORACLE's expand (top-k, cumulative scores, best k of
    n * k): same token ids and parents, scores to f32 rounding.  (oracle.cfg_mask_topk restates tree_decoding, where these two models apply no top-k.)"""
    torch.manual_seed(100 * n + K + tk)
    anole = model == "anole"
    V, lo, W = (65536, 4, 8192) if anole else (16384, 0, 16384)
    A = (0.5 * torch.randn(2 * n, K, device="cuda")).to(torch.bfloat16)
    Wt = (0.3 * torch.randn(V, K, device="cuda")).to(torch.bfloat16)
    scores_in = torch.randn(n, device="cuda") if n > 1 else None
    pk = ops.pack_linear_weight(Wt[lo:lo + W].contiguous()) if K % 64 == 0 else None
    fused = ops.head_expand(A, Wt, lo, W, 3.0, model=ops.MODEL_ANOLE if anole else ops.MODEL_PLAIN, pos_ids=None, top_k_filter=min(tk, V),
                            scores_in=scores_in, top_k=10, packed=pk)
    win = ops.linear_rows_streamk(A, pk if pk is not None else Wt[lo:lo + W].contiguous())          # [2n, W] bf16: the head's logits on the window
    bf = lambda t: t.to(torch.bfloat16).float()
    c, u = win[:n].float(), win[n:].float()
    mix = torch.full((n, V), torch.finfo(torch.bfloat16).min, dtype=torch.float32, device="cuda")      # (Anole: non-image ids at finfo.min)
    mix[:, lo:lo + W] = bf(u + bf(3.0 * bf(c - u)))
    if tk > 0:
        kth = torch.topk(mix, min(tk, V), dim=-1).values[:, -1:]
        mix = mix.masked_fill(mix < kth, float("-inf"))
    ti, cu, ci, so = oracle.expand_dynamic(mix.cpu().numpy(), None if scores_in is None else scores_in.cpu().numpy(), 10)
    assert np.array_equal(fused[0][0].cpu().numpy(), ti) and np.array_equal(fused[2][0].cpu().numpy(), ci)
    got = fused[1][0].cpu().numpy()
    assert np.isfinite(cu).all() and np.allclose(got, cu, rtol=0, atol=4e-6)
    assert np.allclose(fused[3][0].cpu().numpy(), so, rtol=0, atol=4e-6)


@pytest.mark.parametrize("form", ["streamk_packed", "streamk", "per_tile"])
@pytest.mark.parametrize("n,K,special", [(10, 512, ""), (1, 256, ""), (10, 4096, ""), (10, 256, "newline"), (7, 128, "eos"), (16, 64, "few")])
def test_head_expand_fused_equals_the_three_step_composition(n, K, special, form):
    """8f-2: lantern_head_expand[_streamk] (head GEMM with the CFG combination as its epilogue -> per-row processors + log-softmax + top-k ->
    best k of n*k) against head GEMM -> lantern_cfg_mask_topk -> lantern_expand_dynamic on the same inputs: token ids and parent indices
    exact, cumulative scores to the last bit or two (the f64 sum of exponentials is taken in another order).  The composition's GEMM is the
    one with the fused form's accumulation order: lantern_linear_rows for the per-tile kernel, lantern_linear_rows_streamk (row-major or
    packed weight) for the stream-K forms."""
    torch.manual_seed(100 * n + K)
    V, lo, W = 65536, 4, 8192
    A = (0.5 * torch.randn(2 * n, K, device="cuda")).to(torch.bfloat16)
    Wt = (0.3 * torch.randn(V, K, device="cuda")).to(torch.bfloat16)
    bias = (0.1 * torch.randn(V, device="cuda")).to(torch.bfloat16)
    if special == "few":          # a head that leaves only 5 distinct large logits: ties and the -inf tail of the top-k
        Wt[lo + 5:lo + W] = 0
        bias[lo + 5:lo + W] = -30000.0
    pos = torch.full((n,), 2 + 3, device="cuda", dtype=torch.int64)                         # image tokens 4, 5, ... of row 0
    pos = pos + torch.arange(n, device="cuda")
    if special == "newline":
        pos[2] = 2 + 48                                    # token 49 of a 48-wide row: forced newline
    if special == "eos":
        pos[1] = 2 + 49 * 48                               # forced end of image
    scores_in = torch.randn(n, device="cuda") if n > 1 else None
    tk = 2000 if special != "few" else 3
    pk = ops.pack_linear_weight(Wt[lo:lo + W].contiguous()) if form == "streamk_packed" else None
    fused = ops.head_expand(A, Wt, lo, W, 3.0, bias=bias, model=ops.MODEL_LUMINA, pos_ids=pos, pos_base=2, top_k_filter=tk,
                            scores_in=scores_in, top_k=10, packed=pk, streamk=form != "per_tile")
    buf = torch.zeros((2 * n, V), dtype=torch.bfloat16, device="cuda")
    if form == "per_tile":
        logits = ops.linear_rows(A, Wt, lo, W, bias=bias, out=buf)
    else:
        buf[:, lo:lo + W] = ops.linear_rows_streamk(A, pk if pk is not None else Wt[lo:lo + W].contiguous(), bias=bias[lo:lo + W].contiguous())
        logits = buf
    rows = ops.cfg_mask_topk(logits[:n], logits[n:], 3.0, model=ops.MODEL_LUMINA, pos_ids=pos, pos_base=2, img_lo=lo, img_hi=lo + W, top_k=tk)
    ref = ops.expand_dynamic(rows[None], None if scores_in is None else scores_in[None], 10)
    assert torch.equal(fused[0], ref[0]), (fused[0], ref[0])
    assert torch.equal(fused[2], ref[2])
    fin = torch.isfinite(ref[1])
    assert torch.equal(torch.isfinite(fused[1]), fin)
    assert torch.allclose(fused[1][fin], ref[1][fin], rtol=0, atol=2e-6)
    assert torch.allclose(fused[3][torch.isfinite(ref[3])], ref[3][torch.isfinite(ref[3])], rtol=0, atol=2e-6)


# ----------------------------------------------------------------------------- static drafter: head + sample (round 5)
@pytest.mark.parametrize("model,n,K,tk,special", [("lumina", 10, 512, 2000, ""), ("lumina", 1, 4096, 2000, ""), ("lumina", 8, 256, 2000, "newline"),
                                                  ("lumina", 4, 128, 2000, "eos"), ("anole", 16, 256, 2000, ""), ("anole", 13, 128, 0, ""),
                                                  ("llamagen", 16, 256, 300, ""), ("llamagen", 1, 1280, 2000, ""), ("lumina", 5, 64, 12, "few")])
def test_head_sample_vs_the_oracle_behind_the_gemm(model, n, K, tk, special):
    """lantern_head_sample (the static drafter's head stage: cnets_lumina_mgpt.py:1234-1243 + Model.sample :936-955) against the oracle behind the head's
    GEMM: the rows' distributions = softmax of the oracle's processed logits (<= 1e-6), forced rows one-hot; the draws = the oracle's successive
    inverse-CDF draws on the SAME injected uniforms (exact, taken on the kernel's own distribution so that a last-ulp difference of a probability
    cannot move a crossing), conditional probabilities = lo_sample_static's arithmetic; and with injected indices the indices come back as given."""
    torch.manual_seed(100 * n + K + tk)
    V, lo, W = (16384, 0, 16384) if model == "llamagen" else (65536, 4, 8192)
    mid = {"lumina": ops.MODEL_LUMINA, "anole": ops.MODEL_ANOLE, "llamagen": ops.MODEL_PLAIN}[model]
    A = (0.5 * torch.randn(2 * n, K, device="cuda")).to(torch.bfloat16)
    Wt = (0.3 * torch.randn(V, K, device="cuda")).to(torch.bfloat16)
    bias = (0.1 * torch.randn(V, device="cuda")).to(torch.bfloat16)
    if special == "few":          # 15 live logits, top-k 12: hardly more positive entries than draws
        Wt[lo + 15:lo + W] = 0
        bias[lo + 15:lo + W] = -30000.0
    pos = None
    if model == "lumina":
        pos = torch.full((n,), 2 + 3, device="cuda", dtype=torch.int64) + torch.arange(n, device="cuda")
        if special == "newline":
            pos[2] = 2 + 48
        if special == "eos":
            pos[1] = 2 + 49 * 48
    k = 10
    u = torch.rand((n, k), dtype=torch.float64, device="cuda")
    pk = ops.pack_linear_weight(Wt[lo:lo + W].contiguous()) if K % 64 == 0 else None
    probs, tok, prob = ops.head_sample(A, Wt, lo, W, 3.0, bias=bias, model=mid, pos_ids=pos, pos_base=2, top_k_filter=min(tk, V), n_draw=k, draw_u=u, packed=pk)
    win = ops.linear_rows_streamk(A, pk if pk is not None else Wt[lo:lo + W].contiguous(), bias=bias[lo:lo + W].contiguous())
    if model == "lumina":
        bits = np.zeros((2 * n, V), np.uint16)
        bits[:, lo:lo + W] = win.cpu().view(torch.int16).numpy().view(np.uint16)
        rows = oracle.cfg_mask_topk(bits[:n], bits[n:], 3.0, model=oracle.MODEL_LUMINA, pos_ids=pos.cpu().numpy(), pos_base=2, w=48, h=48, img_lo=lo,
                                    img_hi=lo + W, newline_id=8803, eos_id=8196, top_k=tk, bf16=True)
    else:
        bf = lambda t: t.to(torch.bfloat16).float()
        c, uu = win[:n].float(), win[n:].float()
        mix = torch.full((n, V), float("-inf"), dtype=torch.float32, device="cuda")          # (masked ids carry no probability either way)
        mix[:, lo:lo + W] = bf(uu + bf(3.0 * bf(c - uu)))
        if tk > 0:
            kth = torch.topk(mix, min(tk, V), dim=-1).values[:, -1:]
            mix = mix.masked_fill(mix < kth, float("-inf"))
        rows = mix.cpu().numpy()
    r64 = rows.astype(np.float64)
    e = np.exp(r64 - r64.max(-1, keepdims=True))
    want = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    got = probs.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-6
    assert np.abs(got.sum(-1) - 1).max() <= 1e-5
    tok_h, prob_h, u_h = tok.cpu().numpy(), prob.cpu().numpy(), u.cpu().numpy()
    for r in range(n):
        hot = int(np.argmax(got[r])) if got[r].max() == 1.0 else -1
        if hot >= 0 and model == "lumina" and special in ("newline", "eos") and hot in (8803, 8196):
            assert tok_h[r, 0] == hot and prob_h[r, 0] == 1.0 and (prob_h[r, 1:] == 0).all()
            assert len(set(tok_h[r].tolist())) == k and ((tok_h[r, 1:] >= lo) & (tok_h[r, 1:] < lo + W)).all()
            continue
        idx, cp = oracle.sample_draws(got[r], u_h[r])
        assert np.array_equal(tok_h[r], idx), (r, tok_h[r], idx)
        assert np.array_equal(prob_h[r], cp)
        assert len(set(idx.tolist())) == k and (got[r][idx] > 0).all()
    # injected indices come back as given, with their conditional probabilities
    inj = torch.stack([torch.randperm(W, device="cuda")[:k] + lo for _ in range(n)])
    probs2, tok2, prob2 = ops.head_sample(A, Wt, lo, W, 3.0, bias=bias, model=mid, pos_ids=pos, pos_base=2, top_k_filter=min(tk, V), n_draw=k, draw_idx=inj, packed=pk)
    assert torch.equal(tok2, inj) and torch.equal(probs2, probs)
    assert np.array_equal(prob2.cpu().numpy(), oracle.sample_static(got, inj.cpu().numpy()))


@pytest.mark.gpu
def test_mask_left_padding_equals_the_torch_reductions():
    """lantern_mask_left_padding: per row torch.argmax of the mask, its count of ones (`(mask.cumsum(-1) - 1)[:, -1] + 1`, cnets_lumina_mgpt.py:1180-1186) and
    the left-padding test `(mask.cummax(1).values != mask).any(1)` -- bool, uint8 and int64 masks, rows with holes, all ones, all zeros, strided rows."""
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    for S in (1, 7, 64, 257, 1500, 4099):
        rows = []
        for pad in (0, 1, S // 3, S - 1, S):
            r = torch.ones(S, dtype=torch.int64)
            r[:pad] = 0
            rows.append(r)
        holes = torch.ones(S, dtype=torch.int64)
        if S > 2:
            holes[torch.randint(1, S, (max(1, S // 50),), generator=g)] = 0
        rows.append(holes)
        rows.append((torch.rand(S, generator=g) < 0.5).to(torch.int64))
        m = torch.stack(rows)
        for dt in (torch.bool, torch.uint8, torch.int64):
            md = m.to(dt).to(dev)
            out = ops.mask_left_padding(md).cpu()
            m64 = m
            assert out[0].tolist() == m64.argmax(dim=1).tolist()
            assert out[1].tolist() == m64.sum(dim=1).tolist()
            assert out[2].tolist() == (m64.cummax(dim=1).values != m64).any(dim=1).to(torch.int64).tolist()
        wide = torch.zeros((m.shape[0], S + 5), dtype=torch.int64)
        wide[:, :S] = m
        out = ops.mask_left_padding(wide.to(dev)[:, :S]).cpu()          # rows S + 5 apart
        assert out[0].tolist() == m.argmax(dim=1).tolist() and out[1].tolist() == m.sum(dim=1).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["lumina", "llamagen"])
def test_head_sample_edge_draws(model):
    """The draw walk of sample_window_kernel at its edges: uniforms 0 and 1 - 2^-30 (the first positive entry; a crossing deep in the tail), and a row with FEWER positive entries than draws (top-k 4, ten draws): the first four are
    the oracle's draws, the rest the lowest window ids not drawn yet at conditional probability 0 (the reference's multinomial returns arbitrary
    zero-probability ids there; the verify side never accepts them)."""
    torch.manual_seed(7)
    V, lo, W = (16384, 0, 16384) if model == "llamagen" else (65536, 4, 8192)
    mid = ops.MODEL_PLAIN if model == "llamagen" else ops.MODEL_ANOLE
    n, K, k = 3, 64, 10
    A = (0.5 * torch.randn(2 * n, K, device="cuda")).to(torch.bfloat16)
    Wt = (0.3 * torch.randn(V, K, device="cuda")).to(torch.bfloat16)
    pk = ops.pack_linear_weight(Wt[lo:lo + W].contiguous())
    u = torch.rand((n, k), dtype=torch.float64, device="cuda")
    # (1 - 2^-30: the tail of the distribution without reaching the last ulps of the f64 running sum, where the order of the additions decides --
    # tests/fuzz_soak.py `draws` covers 1 - 2^-53 with a bracket check)
    u[0, 0], u[0, 1], u[1, 0], u[1, 5] = 0.0, 1.0 - 2.0 ** -30, 1.0 - 2.0 ** -30, 0.0
    for tk in (2000, 4):
        probs, tok, prob = ops.head_sample(A, Wt, lo, W, 3.0, model=mid, top_k_filter=tk, n_draw=k, draw_u=u, packed=pk)
        got, tok_h, prob_h, u_h = probs.cpu().numpy(), tok.cpu().numpy(), prob.cpu().numpy(), u.cpu().numpy()
        for r in range(n):
            npos = int((got[r] > 0).sum())
            live = min(k, npos)
            idx, cp = oracle.sample_draws(got[r], u_h[r][:live])
            assert np.array_equal(tok_h[r, :live], idx), (tk, r, tok_h[r], idx)
            assert np.array_equal(prob_h[r, :live], cp[:live])
            if live < k:
                assert npos == 4 or tk != 4
                rest, nxt = [], lo
                while len(rest) < k - live:
                    if nxt not in idx.tolist() and nxt not in rest:
                        rest.append(nxt)
                    nxt += 1
                assert tok_h[r, live:].tolist() == rest, (tok_h[r], rest)
                assert (prob_h[r, live:] == 0).all()

"""GPU (-m gpu): the drafter `Model` mirror (lantern_amd/drafters/cnets.py).
  * forward(): input stage + attention mask + position ids against vectors recorded from the REFERENCE's own Model.forward
    (tests/golden/make_golden_drafter.py: the decoder layer there only records what it is handed);
  * topK_genrate(): the whole dynamic-tree loop through the class, on a scripted head, against the goldens the reference's
    topK_genrate produced for the same script (tests/golden/make_golden.py: run_dynamic_tree);
  * topK_genrate_v1() / topK_generate(): shapes, ranges, the conditional-probability identity and the attention-mask / KV
    bookkeeping with the default plain-torch decoder layer."""
import types

import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
from lantern_amd import ops
from lantern_amd.drafters import cnets
from lantern_amd.drafters.choices import mc_sim_7b_63, naive_extend_57

pytestmark = pytest.mark.gpu
SPECS = H.ep_specs()


class Recorder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.seen = []

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False, use_cache=False):
        self.seen.append((hidden_states.clone(), attention_mask.clone(), position_ids.clone()))
        B, T, _ = hidden_states.shape
        k = torch.zeros(B, 1, T, 2, device=hidden_states.device)
        if past_key_value is not None:
            k = torch.cat([past_key_value[0], k], dim=2)
        return hidden_states, (k, k)


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_drafter_forward_input_stage_and_mask_vs_reference(ci):
    g = H.load("drafter.npz")
    pre = f"c{ci}."
    bf16 = bool(g[pre + "bf16"])
    dt = torch.bfloat16 if bf16 else torch.float32
    V, Hd = g[pre + "embed"].shape
    cfg = types.SimpleNamespace(vocab_size=V, hidden_size=Hd, pad_token_id=None, num_hidden_layers=1)
    rec = Recorder()
    has_bias = g[pre + "fc_b"].size > 0
    m = cnets.Model(cfg, layers=[rec], bias=has_bias, embed_upscale=float(g[pre + "upscale"])).cuda()
    m.embed_tokens.weight.data = torch.from_numpy(g[pre + "embed"]).cuda().to(dt)
    m.fc.weight.data = torch.from_numpy(g[pre + "fc_w"]).cuda().to(dt)
    if has_bias:
        m.fc.bias.data = torch.from_numpy(g[pre + "fc_b"]).cuda().to(dt)
    am = torch.from_numpy(g[pre + "attn"]).cuda()
    t = lambda k: torch.from_numpy(g[pre + k]).cuda()
    _, kv = m(t("a.hidden").to(dt), t("a.ids"), attention_mask=am, position_ids=t("a.pos"), use_cache=True)
    m.tree_mask = t("tree_mask")
    m(t("b.hidden").to(dt), t("b.ids"), attention_mask=am, position_ids=t("b.pos"), past_key_values=kv, use_cache=True)
    for tag, seen in (("a", rec.seen[0]), ("b", rec.seen[1])):
        assert np.array_equal(seen[1].cpu().numpy(), g[pre + tag + ".mask"]), (ci, tag)        # additive mask: exact (incl. -inf where both mask)
        assert np.array_equal(seen[2].cpu().numpy(), g[pre + tag + ".layer_pos"])
        got, ref = seen[0].detach().float().cpu().numpy(), g[pre + tag + ".post_fc"]
        if bf16:   # MFMA accumulates the 2H products in another order than the CPU GEMM: at most one bf16 ulp
            assert np.abs(got - ref).max() <= np.abs(ref).max() * 2 ** -7
            assert (got != ref).mean() < 0.2
        else:
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)


class ScriptedHead:
    """head() returns the next block of the script (cond == uncond), like the golden generator's FakeDrafter."""

    def __init__(self, script):
        self.script, self.calls = script, 0

    def __call__(self, hidden):
        blk = torch.from_numpy(self.script[self.calls]).cuda()
        self.calls += 1
        return torch.stack([blk, blk]) if hidden.dim() == 2 and blk.dim() == 1 else blk[None].repeat(2, 1, 1) if blk.dim() == 2 else blk


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["kind"] == "dynamic" and s["model"] == "llamagen"][::3])
def test_drafter_topK_genrate_reproduces_reference_tree(i):
    from transformers.generation.logits_process import LogitsProcessorList, TopKLogitsWarper
    spec, case = SPECS[i], H.ep_case(i)
    depth = int(case["depth"])
    m_ = CS.MODELS[spec["model"]]
    script = H.dynamic_script(spec["seed"], spec["model"], depth)
    cfg = types.SimpleNamespace(vocab_size=m_["V"], hidden_size=32, pad_token_id=None, num_hidden_layers=1, num_attention_heads=4,
                                intermediate_size=64)
    mdl = cnets.Model(cfg, total_tokens=int(case["total_tokens"]) + 1, depth=depth, top_k=CS.TOPK, model_type="llamagen", allow_torch_layers=True).cuda()
    mdl.init_tree()
    head = ScriptedHead(script)
    hidden = torch.randn(2, 5, 32, device="cuda")
    input_ids = torch.randint(0, m_["V"], (2, 6), device="cuda")
    input_ids[:, -1] = int(case["sample_token"])
    proc = LogitsProcessorList([TopKLogitsWarper(H.DYN_TOP_K)])
    draft, ret, mask, pos = mdl.topK_genrate(hidden, input_ids, head, proc, 3.0)
    assert head.calls == depth + 1
    assert np.array_equal(draft[0].cpu().numpy(), case["draft_tokens"])
    assert np.array_equal(ret.cpu().numpy(), case["retrieve"])
    assert np.array_equal(mask[0, 0].cpu().numpy(), case["mask"])
    assert np.array_equal(pos.cpu().numpy(), case["pos"])
    # KV bookkeeping of the loop: prefill 5 positions, then `depth` tree steps of top_k tokens each
    assert mdl.stable_kv[0][0].shape[2] == 5


def _tiny(model_type, V, dtype=torch.bfloat16, **kw):
    cfg = types.SimpleNamespace(vocab_size=V, hidden_size=64, pad_token_id=None, num_hidden_layers=1, num_attention_heads=4,
                                intermediate_size=128)
    torch.manual_seed(0)
    return cnets.Model(cfg, top_k=CS.TOPK, model_type=model_type, allow_torch_layers=True, **kw).cuda().to(dtype)


@pytest.mark.parametrize("model_type,V", [("llamagen", 16384), ("anole", 65536)])
def test_drafter_static_tree_v1(model_type, V):
    from transformers.generation.logits_process import LogitsProcessorList, TemperatureLogitsWarper, TopKLogitsWarper
    mdl = _tiny(model_type, V)
    mdl.init_tree_v1(naive_extend_57)
    W = torch.randn(V, 64, device="cuda", dtype=torch.bfloat16) * 0.5
    head = lambda h: (h @ W.T).float()
    hidden = torch.randn(2, 9, 64, device="cuda", dtype=torch.bfloat16)
    ids = torch.randint(4, 8000, (2, 10), device="cuda")
    proc = LogitsProcessorList([TemperatureLogitsWarper(0.9), TopKLogitsWarper(300)])
    tok, prob, ops_l = mdl.topK_genrate_v1(hidden, ids, head, proc, 4.0)
    counts = [1] + [len(t) for t in mdl.tree_buffer["tree_indices"]]
    assert tok.shape == (sum(counts), 10) and prob.shape == tok.shape and [o.shape[0] for o in ops_l] == counts
    full = torch.cat(ops_l)
    assert torch.allclose(full.sum(-1), torch.ones_like(full.sum(-1)), atol=1e-5)
    nz = (full > 0).sum(-1)
    assert (nz >= 300).all() and (nz <= 330).all()      # `scores < kth` keeps every tie of the k-th value (bf16-valued logits tie often)
    if model_type == "anole":
        assert (tok >= 4).all() and (tok < 8196).all() and float(full[:, :4].sum() + full[:, 8196:].sum()) == 0.0
    # conditional probabilities of draws without replacement: p_i / (1 - sum_{j<i} p_j), first one is the plain probability
    p = full.gather(1, tok)
    assert torch.allclose(prob[:, 0], p[:, 0], atol=1e-6)
    cum = torch.cumsum(p, 1) - p
    ref = (p / (1 - cum)).clamp(0, 1)
    assert torch.allclose(prob, torch.where(torch.isfinite(ref), ref, torch.zeros_like(ref)), atol=1e-4)
    # second call continues on the cache
    n0 = mdl.stable_kv[0][0].shape[2]
    ids2 = torch.cat([ids, torch.randint(4, 8000, (2, 3), device="cuda")], 1)
    mdl.topK_genrate_v1(torch.randn(2, 3, 64, device="cuda", dtype=torch.bfloat16), ids2, head, proc, 4.0)
    assert mdl.stable_kv[0][0].shape[2] == n0 + 3


@pytest.mark.parametrize("tree_type", ["static", "dynamic"])
def test_drafter_lumina_topK_generate(tree_type):
    from lantern_amd.ea_model_lumina_mgpt import InterleavedTopKLogitsWarper, MultiModalLogitsProcessor
    V = 65536
    mdl = _tiny("lumina_mgpt", V, total_tokens=59, depth=4)
    mdl.cfg_scale = 3.0
    mdl.init_tree(mc_sim_7b_63 if tree_type == "static" else None)
    W = torch.randn(V, 64, device="cuda", dtype=torch.bfloat16) * 0.5
    head = lambda h: (h @ W.T)
    procs = [MultiModalLogitsProcessor(), InterleavedTopKLogitsWarper(image_top_k=500)]
    prompt, n_img = 6, 49 - 3            # uncond stream = [8197, 8828, 8828] + 45 image tokens + ... so the next row ends soon
    S = prompt + 3 + n_img
    hidden = torch.randn(1, S, 64, device="cuda", dtype=torch.bfloat16)
    uncond = torch.randn(1, 3 + n_img, 64, device="cuda", dtype=torch.bfloat16)
    ids = torch.randint(4, 8000, (1, S + 1), device="cuda")
    cond_mask = torch.ones(S, dtype=torch.bool, device="cuda")
    unc_mask = torch.cat([torch.zeros(prompt, dtype=torch.bool, device="cuda"), torch.ones(3 + n_img, dtype=torch.bool, device="cuda")])
    out = mdl.topK_generate(hidden, uncond, ids, head, procs, attention_mask=torch.stack([cond_mask, unc_mask]), tree_type=tree_type)
    if tree_type == "static":
        tok, prob, ops_l = out
        full = torch.cat(ops_l)
        assert tok.shape == (11, 10) and prob.shape == (11, 10)
        # every drafter row is either an image-token row (mass only on ids 4..8195) or a forced newline row
        img_rows = full[:, 8803] < 0.5
        assert float(full[img_rows][:, :4].sum() + full[img_rows][:, 8196:].sum()) == 0.0
        assert ((full[img_rows] > 0).sum(-1) <= 540).all()
    else:
        draft, ret, mask, pos = out
        assert draft.shape == (1, 59) and mask.shape == (1, 1, 59, 59) and pos.shape == (59,)
        assert int(draft[0, 0]) == int(ids[0, -1]) and ret.shape[1] == int(pos.max()) + 1
        tok = draft[0, 1:]
        assert (((tok >= 4) & (tok < 8196)) | (tok == 8803)).all()
        # the uncond stream holds 49 tokens (3 header + 46 image): the root's children are image tokens, the nodes of tree depth 2
        # sit at the end of the 48-token image row and must be the forced newline, depth 3 opens the next row
        p1 = pos[1:]
        assert ((tok[p1 == 1] >= 4) & (tok[p1 == 1] < 8196)).all()
        assert (tok[p1 == 2] == 8803).all() and int((p1 == 2).sum()) > 0
        assert ((tok[p1 == 3] >= 4) & (tok[p1 == 3] < 8196)).all()
    assert mdl.stable_kv[0][0].shape[2] == S


@pytest.mark.parametrize("M,H,lo,n", [(20, 4096, 4, 8192), (2, 4096, 4, 8192), (33, 256, 0, 100), (120, 512, 7, 1000)])
def test_linear_rows_matches_torch(M, H, lo, n):
    """lm_head restricted to rows [lo, lo+n): bf16 MFMA with f32 accumulation vs torch's bf16 linear on the same slice."""
    V = max(lo + n + 5, 9000 if H == 4096 else lo + n + 5)
    g = torch.Generator().manual_seed(M + H)
    A = (torch.randn(M, H, generator=g) * 0.5).to(torch.bfloat16).cuda()
    W = (torch.randn(V, H, generator=g) / H ** 0.5).to(torch.bfloat16).cuda()
    bias = (torch.randn(V, generator=g) * 0.1).to(torch.bfloat16).cuda()
    out = ops.linear_rows(A, W, lo, n, bias=bias)
    ref = (A.float() @ W[lo:lo + n].float().T + bias[lo:lo + n].float())
    assert out.shape == (M, n)
    err = (out.float() - ref).abs().max().item()
    assert err <= 2 ** -7 * ref.abs().max().item() + 1e-3, err
    # into a wider row buffer at its own column offset, other columns untouched
    buf = torch.full((M, lo + n + 3), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.linear_rows(A, W, lo, n, bias=bias, out=buf)
    assert torch.equal(buf[:, lo:lo + n], out) and (buf[:, :lo] == 7).all() and (buf[:, lo + n:] == 7).all()


def test_drafter_head_window_equals_full_head():
    """Lumina drafter with a real nn.Linear head: the image-row GEMM feeds the same tree as the full 65536-row head."""
    from lantern_amd.ea_model_lumina_mgpt import InterleavedTopKLogitsWarper, MultiModalLogitsProcessor
    V = 65536
    outs = []
    for use_window in (True, False):
        mdl = _tiny("lumina_mgpt", V, total_tokens=59, depth=4)
        mdl.cfg_scale = 3.0
        mdl.init_tree()
        torch.manual_seed(3)
        head = torch.nn.Linear(64, V, bias=False).cuda().to(torch.bfloat16)
        hd = head if use_window else (lambda h, _hd=head: _hd(h))       # a plain callable takes the full-head route
        procs = [MultiModalLogitsProcessor(), InterleavedTopKLogitsWarper(image_top_k=300)]
        S = 6 + 3 + 20
        torch.manual_seed(4)
        hidden = torch.randn(1, S, 64, device="cuda", dtype=torch.bfloat16)
        uncond = torch.randn(1, 23, 64, device="cuda", dtype=torch.bfloat16)
        ids = torch.randint(4, 8000, (1, S + 1), device="cuda")
        am = torch.stack([torch.ones(S, dtype=torch.bool, device="cuda"),
                          torch.cat([torch.zeros(6, dtype=torch.bool, device="cuda"), torch.ones(23, dtype=torch.bool, device="cuda")])])
        outs.append(mdl.topK_generate(hidden, uncond, ids, hd, procs, attention_mask=am, tree_type="dynamic"))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


class RecordingAnoleModel(cnets.Model):
    """The mirror with its network replaced by a recorder (as tests/golden/make_golden_anole_drafter.py does to the reference):
    what reaches forward() and what the tree logic makes of scripted head logits is compared, not the transformer."""

    def forward(self, hidden_states, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, use_cache=None):
        T = input_ids.shape[1]
        past = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        self.seen.append(dict(ids=input_ids.clone(), pos=position_ids.clone(), past=past, attn=attention_mask))
        dev = input_ids.device
        return torch.zeros(2, T, 4, device=dev), ((torch.zeros(2, 1, past + T, 1, device=dev),),)


@pytest.mark.parametrize("ci", [0, 1])
def test_drafter_anole_calling_convention_vs_reference(ci):
    """`topK_genrate(..., cfg_scale, input_position_diff, attention_mask)` (cnets_anole.py:795-993): cond / uncond position ids
    of the prefill (clamped at 0), of the second call on top of the drafter cache, and of every tree depth (not clamped), the
    attention mask handed through unchanged, and the resulting tree -- against vectors recorded from the reference's method."""
    g = H.load("anole_drafter.npz")
    V, lo, hi, topk, depth, total = (int(x) for x in g["dims"])
    pre = f"c{ci}."
    rs = np.random.RandomState(int(g[pre + "seed"]))
    script = []
    for _ in range(2):
        script.append((4.0 * rs.standard_normal(V)).astype(np.float32))
        for _ in range(depth):
            script.append((4.0 * rs.standard_normal((topk, V))).astype(np.float32))
    dcfg = types.SimpleNamespace(num_hidden_layers=1, hidden_size=16, num_attention_heads=2, intermediate_size=32, vocab_size=V, pad_token_id=None)
    m = RecordingAnoleModel(dcfg, total_tokens=total, depth=depth, top_k=topk, model_type="anole", image_lo=lo, image_hi=hi, allow_torch_layers=True).cuda()
    m.seen = []
    m.init_tree()
    calls = {"n": 0}

    def head(hidden):
        blk = torch.from_numpy(script[calls["n"]]).cuda()
        calls["n"] += 1
        return torch.stack([blk, blk])

    from lantern_amd.verify import prepare_logits_processor
    proc = prepare_logits_processor(temperature=1.0, top_p=1.0, top_k=300)
    attn = torch.from_numpy(g[pre + "attn"]).cuda()
    diff, L0 = int(g[pre + "diff"]), int(g[pre + "L0"])
    ids1, ids2 = torch.from_numpy(g[pre + "ids1"]).cuda(), torch.from_numpy(g[pre + "ids2"]).cuda()
    outs = [m.topK_genrate(torch.zeros(2, L0, 4, device="cuda"), ids1, head, proc, 3.0, diff, attn)]
    assert len(m.seen) == int(g[pre + "n_first"])
    outs.append(m.topK_genrate(torch.zeros(2, 3, 4, device="cuda"), ids2, head, proc, 3.0, diff, attn))
    assert len(m.seen) == int(g[pre + "n_calls"])
    for j, s in enumerate(m.seen):
        assert s["past"] == int(g[pre + f"call{j}.past"]), j
        assert np.array_equal(s["ids"].cpu().numpy(), g[pre + f"call{j}.ids"]), j
        assert np.array_equal(s["pos"].cpu().numpy().reshape(g[pre + f"call{j}.pos"].shape), g[pre + f"call{j}.pos"]), j
        assert int(g[pre + f"call{j}.attn_same"]) == 1 and s["attn"] is not None and torch.equal(s["attn"], attn), j
    for tag, d in zip(("out1", "out2"), outs):
        assert np.array_equal(d[0].cpu().numpy(), g[pre + tag + ".draft"])
        assert np.array_equal(d[1].cpu().numpy(), g[pre + tag + ".retrieve"])
        assert np.array_equal(d[2].cpu().numpy().reshape(g[pre + tag + ".mask"].shape), g[pre + tag + ".mask"])
        assert np.array_equal(d[3].cpu().numpy(), g[pre + tag + ".pos"])


class RecordingLuminaModel(cnets.Model):
    def forward(self, hidden_states, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, use_cache=None):
        T = input_ids.shape[1]
        past = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        self.seen.append(dict(ids=input_ids.clone(), pos=position_ids.clone(), past=past, attn=attention_mask.clone(),
                              hid_shape=tuple(hidden_states.shape), tree=None if self.tree_mask is None else self.tree_mask.clone()))
        dev = input_ids.device
        return torch.zeros(2, T, 4, device=dev), ((torch.zeros(2, 1, past + T, 1, device=dev),),)


def _check_recorded_calls(seen, g, pre, check_ids=True):
    assert len(seen) == int(g[pre + "n_calls"])
    for j, s in enumerate(seen):
        assert s["past"] == int(g[pre + f"call{j}.past"]), j
        assert list(s["hid_shape"][:2]) == g[pre + f"call{j}.hid_shape"][:2].tolist(), j
        assert np.array_equal(s["pos"].cpu().numpy().reshape(g[pre + f"call{j}.pos"].shape), g[pre + f"call{j}.pos"]), j
        assert np.array_equal(s["attn"].cpu().numpy().astype(bool), g[pre + f"call{j}.attn"].astype(bool)), j
        want_tree = g[pre + f"call{j}.tree"]
        if want_tree.size == 0:
            assert s["tree"] is None, j
        else:
            assert np.array_equal(s["tree"].cpu().numpy().reshape(want_tree.shape), want_tree), j
        if check_ids:
            assert np.array_equal(s["ids"].cpu().numpy(), g[pre + f"call{j}.ids"]), j
        else:
            assert tuple(s["ids"].shape) == g[pre + f"call{j}.ids"].shape, j


def _lumina_recording_model(g):
    from lantern_amd.ea_model_lumina_mgpt import InterleavedTopKLogitsWarper, MultiModalLogitsProcessor
    V, lo, hi, nl, eos, topk, depth, total = (int(x) for x in g["dims"])
    dcfg = types.SimpleNamespace(num_hidden_layers=1, hidden_size=16, num_attention_heads=2, intermediate_size=32, vocab_size=V, pad_token_id=None)
    m = RecordingLuminaModel(dcfg, total_tokens=total, depth=depth, top_k=topk, model_type="lumina_mgpt", image_lo=lo, image_hi=hi, allow_torch_layers=True).cuda()
    m.seen, m.cfg_scale = [], 3.0
    procs = [MultiModalLogitsProcessor(image_next_line_token_id=nl, image_end_token_id=eos, voc_size=V), InterleavedTopKLogitsWarper(image_top_k=300)]
    return m, procs, V, topk, depth


def _scripted_head(seed, blocks):
    rs = np.random.RandomState(seed)
    script = [(3.0 * rs.standard_normal(shape)).astype(np.float32) for shape in blocks]
    calls = {"n": 0}

    def head(hidden):
        blk = torch.from_numpy(script[calls["n"]]).cuda()
        calls["n"] += 1
        return torch.stack([blk, 0.5 * blk])
    return head


@pytest.mark.parametrize("ci", [0, 1])
def test_drafter_lumina_dynamic_loop_vs_reference(ci):
    """`topK_generate(tree_type="dynamic")` (cnets_lumina_mgpt.py:1148-1393) at the real vocabulary: what reaches forward()
    (ids, cond / uncond position ids from the zero-padded mask, padded attention mask, tree masks, hidden shapes) and the drafted
    tree, first call and call on top of the drafter cache; the uncond stream is placed so that the depths cross a newline row."""
    g = H.load("lumina_drafter.npz")
    m, procs, V, topk, depth = _lumina_recording_model(g)
    m.init_tree()
    pre = f"dyn{ci}."
    head = _scripted_head(int(g[pre + "seed"]), ([(V,)] + [(topk, V)] * depth) * 2)
    L, Lu, extra = int(g[pre + "L"]), int(g[pre + "Lu"]), int(g[pre + "extra"])
    attn = torch.from_numpy(g[pre + "attn"]).cuda()
    ids1, ids2 = torch.from_numpy(g[pre + "ids1"]).cuda(), torch.from_numpy(g[pre + "ids2"]).cuda()
    z = lambda *s: torch.zeros(*s, device="cuda")      # noqa: E731
    outs = [m.topK_generate(z(1, L, 4), z(1, Lu, 4), ids1, head, procs, attention_mask=attn, tree_type="dynamic")]
    assert len(m.seen) == int(g[pre + "n_first"])
    outs.append(m.topK_generate(z(1, extra, 4), z(1, extra, 4), ids2, head, procs, attention_mask=attn, tree_type="dynamic"))
    _check_recorded_calls(m.seen, g, pre)
    for tag, d in zip(("out1", "out2"), outs):
        assert np.array_equal(d[0].cpu().numpy(), g[pre + tag + ".draft"])
        assert np.array_equal(d[1].cpu().numpy(), g[pre + tag + ".retrieve"])
        assert np.array_equal(d[2].cpu().numpy().reshape(g[pre + tag + ".mask"].shape), g[pre + tag + ".mask"])
        assert np.array_equal(d[3].cpu().numpy(), g[pre + tag + ".pos"])


def test_drafter_lumina_static_loop_calls_vs_reference():
    """`topK_generate(tree_type="static")`: the per-level forward calls (position ids, stacked tree-mask slices, repeated hidden
    rows, padded attention mask) against the reference's; the sampled tokens themselves come from each side's own generator."""
    g = H.load("lumina_drafter.npz")
    m, procs, V, topk, depth = _lumina_recording_model(g)
    m.init_tree(mc_sim_7b_63)
    pre = "sta0."
    tb = ops.tree_drafter_build(mc_sim_7b_63)
    head = _scripted_head(int(g[pre + "seed"]), [(V,)] + [(len(t), V) for t in tb["tree_indices"]])
    L, Lu = int(g[pre + "L"]), int(g[pre + "Lu"])
    attn = torch.from_numpy(g[pre + "attn"]).cuda()
    ids1 = torch.from_numpy(g[pre + "ids1"]).cuda()
    out = m.topK_generate(torch.zeros(1, L, 4, device="cuda"), torch.zeros(1, Lu, 4, device="cuda"), ids1, head, procs, attention_mask=attn,
                          tree_type="static")
    _check_recorded_calls(m.seen[:1], {**{k: g[k] for k in g.files}, pre + "n_calls": np.int64(1)}, pre)      # the prefill: ids too
    _check_recorded_calls(m.seen, g, pre, check_ids=False)
    assert tuple(out[0].shape) == tuple(g[pre + "ss_token_shape"]) and len(out[2]) == int(g[pre + "n_op"])
    assert ((out[0] >= 4) & (out[0] < 8196) | (out[0] == 8803)).all()


@pytest.mark.parametrize("model_type,V,H,heads", [("llamagen", 16384, 128, 2), ("anole", 65536, 256, 4), ("lumina_mgpt", 65536, 256, 2)])
def test_default_layers_draft_on_the_hip_path(model_type, V, H, heads, monkeypatch):
    """cnets.Model without injected layers builds the HIP decoder layer of the model family (LlamaDecoderLayer with the 2-D freqs_cis table it then
    owns / DecoderLayer, Anole's with one head-norm row per head) and a whole dynamic drafting call -- prefill, `depth` tree steps, head expansion per
    depth -- runs on the library's kernels: stream-K GEMMs, the head stage, tree attention and the fused head_expand; no torch attention call at the
    drafting shape, no full-vocabulary head GEMM.  The drafting forward agrees with the same layer on torch's ops (`fused = False`: the composition the
    golden vectors pin to the reference layer) within the bf16 tolerance, and a config outside the kernels raises instead of falling back."""
    import torch.nn.functional as F
    from lantern_amd._lib import LanternError
    from lantern_amd.drafters.decoder_layer import DecoderLayer, LlamaDecoderLayer
    from transformers.generation.logits_process import LogitsProcessorList, TopKLogitsWarper
    dev, bf = torch.device("cuda"), torch.bfloat16
    cfg = types.SimpleNamespace(vocab_size=V, hidden_size=H, pad_token_id=None, num_hidden_layers=1, num_attention_heads=heads, num_key_value_heads=heads,
                                intermediate_size=2 * H, max_position_embeddings=512, rms_norm_eps=1e-5, model_parallel_size=1, input_type="t2i")
    torch.manual_seed(5)
    mdl = cnets.Model(cfg, total_tokens=30, depth=3, top_k=CS.TOPK, model_type=model_type).to(dev).to(bf)
    assert isinstance(mdl.layers[0], LlamaDecoderLayer if model_type == "llamagen" else DecoderLayer)
    if model_type == "llamagen":
        assert mdl.freqs_cis.shape == (119 + 256 + 20, H // heads // 2, 2) and "input_layernorm.weight" not in mdl.layers[0].state_dict()
    if model_type == "anole":
        assert tuple(mdl.layers[0].self_attn.q_norm.weight.shape) == (heads, H // heads)
    with pytest.raises(LanternError, match="outside the HIP decoder layer"):
        cnets.Model(types.SimpleNamespace(vocab_size=V, hidden_size=64, pad_token_id=None, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128),
                    model_type=model_type)
    head = torch.nn.Linear(H, V, bias=False).to(dev).to(bf)
    calls = []
    for fn in ("linear_rows_streamk", "qk_norm_rope", "qk_rope_pairs", "tree_attention", "head_expand", "linear_rows", "drafter_fc"):
        real = getattr(ops, fn)
        monkeypatch.setattr(ops, fn, (lambda real, fn: (lambda *a, **k_: (calls.append(fn), real(*a, **k_))[1]))(real, fn))
    real_sdpa = F.scaled_dot_product_attention
    monkeypatch.setattr(F, "scaled_dot_product_attention", lambda *a, **k_: (calls.append(("sdpa", a[0].shape[2])), real_sdpa(*a, **k_))[1])
    mdl.init_tree()
    mdl.use_depth_plan = False          # the Python depth loop, kernel by kernel (lantern_draft_depth: test_depth_plan_equals_the_python_depth_loop)
    hidden = torch.randn(2, 5, H, device=dev, dtype=bf)
    ids = torch.randint(4, 8000, (2, 6), device=dev)
    proc = LogitsProcessorList([TopKLogitsWarper(300)])
    if model_type == "lumina_mgpt":
        class P2(list):
            pass
        pr = P2([None, types.SimpleNamespace(image_top_k=300)])
        out = mdl.topK_generate(hidden[:1], hidden[1:], ids[:1], head, pr, attention_mask=torch.ones(2, 5, dtype=torch.bool, device=dev), tree_type="dynamic")
    else:
        mdl.cfg_scale = 3.0
        kw = dict(input_position_diff=torch.zeros((), dtype=torch.long, device=dev), attention_mask=torch.ones(2, 5, dtype=torch.bool, device=dev)) if model_type == "anole" else {}
        out = mdl.topK_genrate(hidden, ids, head, proc, 3.0, **kw)
    draft, ret, mask, pos = out
    assert draft.shape == (1, 30) and mask.shape[-1] == 30 and int(ret.max()) < 30
    # the drafting calls: 3 depths x (4 stream-K GEMMs + head stage + tree attention) + the prefill's GEMMs; 4 fused head expansions
    assert calls.count("head_expand") == 4, calls
    assert calls.count("tree_attention") == 4 and calls.count("qk_rope_pairs" if model_type == "llamagen" else "qk_norm_rope") == 4, calls
    assert calls.count("linear_rows_streamk") == 16 and calls.count("linear_rows") == 0, calls
    assert not [c for c in calls if isinstance(c, tuple) and c[0] == "sdpa"], calls          # (the 5-token prefill too: block-causal tree attention)
    # the drafting forward against the same layer on torch's ops
    mdl.reset_kv()
    mdl.tree_mask = None
    x = torch.randn(2, 9, H, device=dev, dtype=bf)
    iid = torch.randint(4, 8000, (2, 9), device=dev)
    with torch.no_grad():
        y_hip, kv = mdl(x, iid, use_cache=True)
    mdl.tree_mask = mdl.tree_mask_init
    pos_t = (9 + mdl.position_ids)[None].expand(2, -1) if model_type != "llamagen" else 9 + mdl.position_ids
    xt, it = torch.randn(2, CS.TOPK, H, device=dev, dtype=bf), torch.randint(4, 8000, (2, CS.TOPK), device=dev)
    with torch.no_grad():
        t_hip, _ = mdl(xt, it, past_key_values=tuple((k.clone(), v.clone()) for k, v in kv), position_ids=pos_t, use_cache=True)
    for l in mdl.layers:
        l.fused = False
        l.inplace_cache = False
    monkeypatch.setattr("lantern_amd.drafters.decoder_layer._hip_ok", lambda a, b: False)
    with torch.no_grad():
        t_ref, _ = mdl(xt, it, past_key_values=tuple((k.clone(), v.clone()) for k, v in kv), position_ids=pos_t, use_cache=True)
    np.testing.assert_allclose(t_hip.float().cpu().numpy(), t_ref.float().cpu().numpy(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("model_type,V,H,heads", [("llamagen", 16384, 128, 2), ("anole", 65536, 256, 4), ("lumina_mgpt", 65536, 256, 2)])
def test_prompt_prefill_stays_on_the_hip_kernels(model_type, V, H, heads, monkeypatch):
    """The drafter's first call of a prompt (cnets_lumina_mgpt.py:1066-1098: the whole prompt through the layer, here 2 x 90 rows with the second row
    left-padded by 7) runs on the library's kernels -- drafter_fc in slices, lantern_linear_rows_packed for the layer's four GEMMs, the head stage,
    block-causal lantern_tree_attention -- with no F.linear, no scaled_dot_product_attention and no eager softmax; it agrees with the same model
    on torch's ops on every row that is not padding, and a drafting step behind the cache it leaves agrees too."""
    import torch.nn.functional as F
    dev, bf = torch.device("cuda"), torch.bfloat16
    cfg = types.SimpleNamespace(vocab_size=V, hidden_size=H, pad_token_id=None, num_hidden_layers=1, num_attention_heads=heads, num_key_value_heads=heads,
                                intermediate_size=2 * H, max_position_embeddings=512, rms_norm_eps=1e-5, model_parallel_size=1, input_type="t2i")
    torch.manual_seed(11)
    mdl = cnets.Model(cfg, total_tokens=30, depth=3, top_k=CS.TOPK, model_type=model_type).to(dev).to(bf)
    mdl.init_tree()
    T, pad = 90, 7
    x = torch.randn(2, T, H, device=dev, dtype=bf)
    iid = torch.randint(4, 8000, (2, T), device=dev)
    am = torch.ones(2, T, dtype=torch.bool, device=dev)
    if model_type != "llamagen":
        am[1, :pad] = False
    pos = (am.long().cumsum(1) - 1).clamp(min=0) if model_type != "llamagen" else None
    calls = []
    for fn in ("linear_rows_packed", "linear_rows_streamk", "tree_attention", "drafter_fc"):
        real = getattr(ops, fn)
        monkeypatch.setattr(ops, fn, (lambda real, fn: (lambda *a, **k_: (calls.append(fn), real(*a, **k_))[1]))(real, fn))
    for mod, fn in ((F, "scaled_dot_product_attention"), (F, "linear"), (F, "softmax"), (torch, "softmax")):
        real = getattr(mod, fn)
        monkeypatch.setattr(mod, fn, (lambda real, fn: (lambda *a, **k_: (calls.append("torch." + fn), real(*a, **k_))[1]))(real, fn))
    mdl.tree_mask = None
    with torch.no_grad():
        y_hip, kv = mdl(x, iid, attention_mask=am, position_ids=pos, use_cache=True)
    assert calls.count("linear_rows_packed") == 4 and calls.count("tree_attention") == 2 and calls.count("drafter_fc") == 2, calls
    assert not [c for c in calls if c.startswith("torch.") or c == "linear_rows_streamk"], calls
    # a drafting step behind that cache
    mdl.tree_mask = mdl.tree_mask_init
    base = (am.long().sum(1) if model_type != "llamagen" else torch.full((2,), T, device=dev))
    pos_t = (base[:, None] + mdl.position_ids[None]) if model_type != "llamagen" else T + mdl.position_ids
    am_t = torch.cat([am, torch.ones(2, CS.TOPK, dtype=torch.bool, device=dev)], dim=1)
    xt, it = torch.randn(2, CS.TOPK, H, device=dev, dtype=bf), torch.randint(4, 8000, (2, CS.TOPK), device=dev)
    with torch.no_grad():
        t_hip, _ = mdl(xt, it, attention_mask=am_t, past_key_values=tuple((k.clone(), v.clone()) for k, v in kv), position_ids=pos_t, use_cache=True)
    # the same two calls on torch's ops
    for l in mdl.layers:
        l.fused = False
        l.inplace_cache = False
    monkeypatch.setattr("lantern_amd.drafters.decoder_layer._hip_ok", lambda a, b: False)
    mdl.tree_mask = None
    with torch.no_grad():
        y_ref, kv_ref = mdl(x, iid, attention_mask=am, position_ids=pos, use_cache=True)
        mdl.tree_mask = mdl.tree_mask_init
        t_ref, _ = mdl(xt, it, attention_mask=am_t, past_key_values=tuple((k.clone(), v.clone()) for k, v in kv_ref), position_ids=pos_t, use_cache=True)
    for b in range(2):
        st = pad if (b == 1 and model_type != "llamagen") else 0
        np.testing.assert_allclose(y_hip[b, st:].float().cpu().numpy(), y_ref[b, st:].float().cpu().numpy(), rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(t_hip.float().cpu().numpy(), t_ref.float().cpu().numpy(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("model_type,V,H,heads,depth", [("llamagen", 16384, 128, 2, 4), ("anole", 65536, 256, 4, 4), ("lumina_mgpt", 65536, 256, 2, 5),
                                                        ("lumina_mgpt", 65536, 256, 4, 3)])
def test_depth_plan_equals_the_python_depth_loop(model_type, V, H, heads, depth, monkeypatch):
    """lantern_draft_depth (one C call per drafting depth: input stage, decoder layer, fused head expansion, the next depth's inputs) against the Python
    depth loop over the same kernels: identical drafted tree -- tokens, retrieve rows, mask, positions -- over three consecutive drafting calls (the
    second and third start from the cached prefix), and exactly `depth` lantern_draft_depth calls per drafting call."""
    from transformers.generation.logits_process import LogitsProcessorList, TopKLogitsWarper
    from lantern_amd import _lib
    dev, bf = torch.device("cuda"), torch.bfloat16
    cfg = types.SimpleNamespace(vocab_size=V, hidden_size=H, pad_token_id=None, num_hidden_layers=1, num_attention_heads=heads, num_key_value_heads=heads,
                                intermediate_size=2 * H, max_position_embeddings=512, rms_norm_eps=1e-5, model_parallel_size=1, input_type="t2i")
    torch.manual_seed(11)
    mdl = cnets.Model(cfg, total_tokens=40, depth=depth, top_k=CS.TOPK, model_type=model_type).to(dev).to(bf)
    mdl.init_tree()
    head = torch.nn.Linear(H, V, bias=False).to(dev).to(bf)
    proc = LogitsProcessorList([TopKLogitsWarper(300)])
    lum = [None, types.SimpleNamespace(image_top_k=300)]
    gen = torch.Generator(device="cuda").manual_seed(3)
    calls = []
    real = _lib.lib().lantern_draft_depth

    def run(plan_on):
        mdl.use_depth_plan = plan_on
        mdl.reset_kv()
        outs, total = [], 7
        g = torch.Generator(device="cuda").manual_seed(5)
        for c in range(3):
            n_new = total if c == 0 else 2
            hid = torch.randn(2, n_new, H, device=dev, dtype=bf, generator=g)
            ids = torch.randint(4, 8000, (2, total + 1), device=dev, generator=g)
            if model_type == "lumina_mgpt":
                am = torch.ones(2, total, dtype=torch.bool, device=dev)
                am[1, :2] = False                                      # the uncond stream is left-padded (sequential CFG)
                out = mdl.topK_generate(hid[:1], hid[1:], ids[:1], head, lum, attention_mask=am, tree_type="dynamic")
            elif model_type == "anole":
                am = torch.ones(2, total, dtype=torch.bool, device=dev)
                am[1, :1] = False
                out = mdl.topK_genrate(hid, ids, head, proc, 3.0, input_position_diff=torch.ones((), dtype=torch.long, device=dev), attention_mask=am)
            else:
                out = mdl.topK_genrate(hid, ids, head, proc, 3.0)
            outs.append([t.clone() for t in out])
            total += 2
        return outs
    py = run(False)
    n_calls = [0]
    import ctypes as C

    class Spy:
        def __call__(self, *a):
            n_calls[0] += 1
            return real(*a)
    plan_mod = cnets.DraftPlan
    orig_run = plan_mod.run
    monkeypatch.setattr(plan_mod, "run", lambda self, i: (n_calls.__setitem__(0, n_calls[0] + 1), orig_run(self, i))[1])
    pl = run(True)
    assert n_calls[0] == 3 * depth
    for c, (a, b) in enumerate(zip(py, pl)):
        for x, y in zip(a, b):
            assert x.shape == y.shape and torch.equal(x, y), (c, x, y)


@pytest.mark.parametrize("model_type,V,H,heads,tree", [("lumina_mgpt", 65536, 256, 2, "mc_sim_7b_63"), ("anole", 65536, 256, 4, "naive_extend_57"),
                                                       ("llamagen", 16384, 128, 2, "naive_extend_57"), ("lumina_mgpt", 65536, 256, 4, "naive_extend_57"),
                                                       ("llamagen", 16384, 128, 2, "mc_sim_7b_63")])
def test_static_plan_equals_the_python_static_loop(model_type, V, H, heads, tree, monkeypatch):
    """Round 5: the static-tree loops (Lumina's default eagle_version 1: topK_generate(tree_type="static"), cnets_lumina_mgpt.py:1245-1328; LlamaGen / Anole
    topK_genrate_v1, cnets_llamagen.py:944-1023 -- BASELINE config 4's LANTERN++ drafting) through StaticDraftPlan -- ONE lantern_head_sample for the root
    row, then ONE lantern_draft_depth per tree level -- against the Python level loop (sample -> cat -> repeat_hidden -> forward -> head -> processors)
    fed the SAME draws: the same conditional probabilities and drafter distributions at every level (5e-5), hence the same forward inputs through the
    tree's tables, over three consecutive drafting calls; exactly one C call per level."""
    from transformers.generation.logits_process import LogitsProcessorList, TopKLogitsWarper
    from lantern_amd.drafters import choices
    dev, bf = torch.device("cuda"), torch.bfloat16
    cfg = types.SimpleNamespace(vocab_size=V, hidden_size=H, pad_token_id=None, num_hidden_layers=1, num_attention_heads=heads, num_key_value_heads=heads,
                                intermediate_size=2 * H, max_position_embeddings=512, rms_norm_eps=1e-5, model_parallel_size=1, input_type="t2i")
    torch.manual_seed(13)
    mdl = cnets.Model(cfg, total_tokens=40, depth=4, top_k=CS.TOPK, model_type=model_type).to(dev).to(bf)
    mdl.init_tree(getattr(choices, tree))
    levels = [len(t) for t in mdl.tree_buffer["tree_indices"]]
    head = torch.nn.Linear(H, V, bias=False).to(dev).to(bf)
    proc = LogitsProcessorList([TopKLogitsWarper(300)])
    lum = [None, types.SimpleNamespace(image_top_k=300)]
    k = CS.TOPK

    def run(plan_on):
        mdl.use_depth_plan = plan_on
        mdl.reset_kv()
        outs, total = [], 7
        g = torch.Generator(device="cuda").manual_seed(5)
        for c in range(3):
            n_new = total if c == 0 else 2
            hid = torch.randn(2, n_new, H, device=dev, dtype=bf, generator=g)
            ids = torch.randint(4, 8000, (2, total + 1), device=dev, generator=g)
            state["call"] = c
            if model_type == "lumina_mgpt":
                am = torch.ones(2, total, dtype=torch.bool, device=dev)
                am[1, :2] = False
                out = mdl.topK_generate(hid[:1], hid[1:], ids[:1], head, lum, attention_mask=am, tree_type="static")
            elif model_type == "anole":
                am = torch.ones(2, total, dtype=torch.bool, device=dev)
                am[1, :1] = False
                out = mdl.topK_genrate_v1(hid, ids, head, proc, 3.0, input_position_diff=torch.ones((), dtype=torch.long, device=dev), attention_mask=am)
            else:
                out = mdl.topK_genrate_v1(hid, ids, head, proc, 3.0)
            outs.append((out[0].clone(), out[1].clone(), [o.clone() for o in out[2]]))
            total += 2
        return outs
    state = {"call": 0}
    n_calls = [0]
    orig = cnets.StaticDraftPlan.run_level
    monkeypatch.setattr(cnets.StaticDraftPlan, "run_level", lambda self, *a: (n_calls.__setitem__(0, n_calls[0] + 1), orig(self, *a))[1])
    ug = torch.Generator(device="cuda").manual_seed(77)
    mdl.static_draw_uniforms = lambda R, kk: torch.rand((R, kk), dtype=torch.float64, device=dev, generator=ug)
    pl = run(True)
    assert n_calls[0] == 3 * len(levels), (n_calls, levels)
    R = 1 + sum(levels)
    for tok, prob, ol in pl:
        assert tok.shape == (R, k) and prob.shape == (R, k) and [o.shape[0] for o in ol] == [1] + levels
        full = torch.cat(ol)
        assert torch.allclose(full.sum(-1), torch.ones(R, device=dev), atol=1e-5)
        p = full.gather(1, tok)
        assert (p > 0).all() and all(len(set(r)) == k for r in tok.tolist())            # k distinct draws with mass, in every row
    # ---- the Python loop on the same draws
    off = [0, 1] + [1 + sum(levels[:i + 1]) for i in range(len(levels))]
    seen = {"n": 0}

    def fake_sample(logits, logits_processor=None, k=1):
        if logits_processor is not None and not isinstance(logits_processor, (list, tuple)):
            logits = logits_processor(None, logits)
        probs = torch.softmax(logits.float().view(-1, logits.shape[-1]), dim=-1)
        lvl = seen["n"] % (len(levels) + 1)
        seen["n"] += 1
        idx = pl[state["call"]][0][off[lvl]:off[lvl + 1]]
        assert idx.shape[0] == probs.shape[0]
        from lantern_amd import ops as O
        return idx, O.sample_static(probs, idx), probs
    monkeypatch.setattr(mdl, "sample", fake_sample)
    py = run(False)
    assert seen["n"] == 3 * (len(levels) + 1)
    for c, ((t0, p0, o0), (t1, p1, o1)) in enumerate(zip(py, pl)):
        # (tolerance: the two forms split the tree attention's keys differently -- the plan sizes its workspace for 64 query rows -- so a hidden value
        # can differ by one bf16 ulp and with it a logit: ~1e-5 on a probability of a few 1e-3; a wrong table entry or position moves them by 1e-2)
        assert torch.equal(t0, t1)
        assert torch.allclose(p0, p1, rtol=2e-2, atol=5e-5), (c, (p0 - p1).abs().max())
        for lvl, (a, b) in enumerate(zip(o0, o1)):
            assert a.shape == b.shape and torch.allclose(a, b, rtol=0, atol=5e-5), (c, lvl, (a - b).abs().max())

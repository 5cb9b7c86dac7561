"""GPU (-m gpu): the host mirror classes called exactly like the reference's methods were called when
the golden vectors were captured (tests/golden/make_golden.py): same argument lists, `random.random`
patched with the recorded stream, HF logits processors for LlamaGen/Anole."""
import random
import types

import numpy as np
import pytest
from lantern_amd._lib import LanternError
import torch

import cases as CS
import helpers as H
from lantern_amd import ea_model_anole, ea_model_llamagen, ea_model_lumina_mgpt, ops, verify

pytestmark = pytest.mark.gpu
SPECS = H.ep_specs()


class Stream:
    def __init__(self, vals):
        self.vals, self.n = list(map(float, vals)), 0

    def __call__(self):
        v = self.vals[self.n] if self.n < len(self.vals) else 0.5
        self.n += 1
        return v


def fake_base(V, H=8):
    lm = types.SimpleNamespace(weight=torch.zeros(V, H, device="cuda"))
    return types.SimpleNamespace(lm_head=lm, config=None, model=types.SimpleNamespace())


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_lumina(spec, static):
    m = CS.MODELS["lumina"]
    mdl = ea_model_lumina_mgpt.EaLumina_mGPT(fake_base(m["V"]), None, H.table(m["K"]), eagle_version=1 if static else 2)
    mdl.uniform_window = 64
    mdl.image_tokens = torch.arange(m["img_lo"], m["img_hi"], device="cuda")          # reduced-vocabulary constants
    mdl.image_syntax_tokens = torch.tensor(m["syntax"], device="cuda")
    return mdl


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["model"] == "lumina" and s["kind"] in ("static", "dynamic")][::2])
def test_lumina_mirror_evaluate_posterior(i, monkeypatch):
    spec, case = SPECS[i], H.ep_case(i)
    static = spec["kind"] == "static"
    mdl = make_lumina(spec, static)
    if static:
        tb, g = H.static_inputs(spec, case)
        mdl.tree_buffers = verify.generate_tree_buffers(H.tree_choices(spec["tree"]), device="cuda")
        mdl.tree_choices = H.tree_choices(spec["tree"])
        tbuf = mdl.tree_buffers
        # O6 through the mirror, reference call shape: (tree_logits, tree_indices, retrieve_indices, sample_token)
        offs = list(g["op_off"]) + [g["R"]]
        op_list = [cuda(g["orig_prob"][offs[d]:offs[d + 1]]) for d in range(len(offs) - 1)]
        tree_logits = (cuda(case["ss_token"]), cuda(case["ss_prob"]), op_list)
        cand, cprob, tcand = mdl.generate_candidates(tree_logits, tbuf["tree_indices"], tbuf["retrieve_indices"],
                                                     torch.tensor([[int(case["sample_token"])]], device="cuda"))
        assert np.array_equal(cand.cpu().numpy(), case["cand"])
        assert np.array_equal(cprob.cpu().numpy(), case["cart_prob"])
        assert tcand.shape == (1, len(tb["tree_indices"]))
        logits = cuda(g["node_logits"])[tbuf["retrieve_indices"]]        # the reference's materialised [P,D,V]
        uniforms = case["uniforms"]
        call = lambda lg: mdl.evaluate_posterior(lg, cand, cart_candidates_prob=cprob, original_prob=op_list,
                                                 p_indices=tbuf["p_indices"], tree_candidates=tcand, b_indices=tbuf["b_indices"],
                                                 do_sample=True, lantern=spec["lantern"], lantern_k=spec["k"], lantern_delta=spec["delta"])
        node_view = verify.NodeLogits(cuda(g["node_logits"]), tbuf["retrieve_indices"])
    else:
        nl, uniforms = H.dynamic_node_logits(spec, case)
        ret = cuda(case["retrieve"])
        cand = cuda(case["cand"])
        logits = cuda(nl)[ret]
        call = lambda lg: mdl.evaluate_posterior(lg, cand, do_sample=True, lantern=spec["lantern"], lantern_k=spec["k"],
                                                 lantern_delta=spec["delta"])
        node_view = verify.NodeLogits(cuda(nl), ret)
    for lg in (logits, node_view):              # reference tensor form and the zero-copy node view
        mdl._fifo = None
        st = Stream(uniforms)
        monkeypatch.setattr(random, "random", st)
        best, alen, sp = call(lg)
        assert int(best) == int(case["best"]) and alen == int(case["accept_len"])
        assert isinstance(alen, int) and best.dtype == torch.int64
        np.testing.assert_allclose(sp.cpu().numpy(), case["sample_p"], rtol=0, atol=1e-5)
        assert int(mdl._fifo.cursor.item()) == int(case["n_draws"])


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["model"] in ("llamagen", "anole") and s["kind"] in ("static", "dynamic")
                               and not s.get("plain_eagle")][::2])
def test_llamagen_anole_mirror_evaluate_posterior(i, monkeypatch):
    from transformers.generation.logits_process import LogitsProcessorList, TemperatureLogitsWarper, TopKLogitsWarper
    spec, case = SPECS[i], H.ep_case(i)
    m = CS.MODELS[spec["model"]]
    cls = ea_model_llamagen.EaModel if spec["model"] == "llamagen" else ea_model_anole.EaModel
    mdl = cls(fake_base(m["V"]), None, H.table(m["K"]))
    mdl.uniform_window = 64
    if spec["model"] == "anole":
        mdl.image_lo, mdl.image_hi = m["img_lo"], m["img_hi"]
    proc = LogitsProcessorList()                       # what prepare_logits_processor builds in the reference
    T, tk, tp = spec.get("temperature", 1.0), spec.get("top_k", 0), spec.get("top_p", 1.0)
    if T != 1.0:
        proc.append(TemperatureLogitsWarper(T))
    if 1e-8 <= tp < 1.0:                               # (drafters/utils.py:36-52: Temperature, TopP, TopK)
        from transformers.generation.logits_process import TopPLogitsWarper
        proc.append(TopPLogitsWarper(tp))
    if tk > 0:
        proc.append(TopKLogitsWarper(tk))
    if spec["kind"] == "static":
        tb, g = H.static_inputs(spec, case)
        mdl.tree_buffers = mdl.generate_tree_buffers(H.tree_choices(spec["tree"]), device="cuda")
        tbuf = mdl.tree_buffers
        offs = list(g["op_off"]) + [g["R"]]
        op_list = [cuda(g["orig_prob"][offs[d]:offs[d + 1]]) for d in range(len(offs) - 1)]
        tree_logits = (cuda(case["ss_token"]), cuda(case["ss_prob"]), op_list)
        cand, cprob, tcand = mdl.generate_candidates(tree_logits, tbuf["tree_indices"], tbuf["retrieve_indices"],
                                                     torch.tensor([[int(case["sample_token"])]], device="cuda"), proc)
        assert np.array_equal(cand.cpu().numpy(), case["cand"])
        logits = verify.NodeLogits(cuda(g["node_logits"]), tbuf["retrieve_indices"])
        monkeypatch.setattr(random, "random", Stream(case["uniforms"]))
        best, alen, sp = mdl.evaluate_posterior_v1(logits, cand, proc, cprob, op_list, tbuf["p_indices"], torch.cat([tcand, tcand]),
                                                   tbuf["b_indices"], spec["lantern"], spec["k"], spec["delta"])
    else:
        nl, uniforms = H.dynamic_node_logits(spec, case)
        logits = verify.NodeLogits(cuda(nl), cuda(case["retrieve"]))
        monkeypatch.setattr(random, "random", Stream(uniforms))
        best, alen, sp = mdl.evaluate_posterior(logits, cuda(case["cand"]), proc, lantern=spec["lantern"], lantern_k=spec["k"],
                                                lantern_delta=spec["delta"])
    assert int(best) == int(case["best"]) and alen == int(case["accept_len"])
    np.testing.assert_allclose(sp.cpu().numpy(), case["sample_p"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("i", [i for i, s in enumerate(SPECS) if s["model"] in ("llamagen", "anole") and s["kind"] in ("static", "dynamic")
                               and not s.get("plain_eagle")])
def test_llamagen_anole_mirror_window_kernel_set(i, monkeypatch):
    """The windowed kernel set behind the same mirror methods: tree rows leave O7w as probabilities with the HF processors
    (Temperature -> TopP -> TopK, top_p < 1 included) applied to every row, evaluate_posterior[_v1] receives WindowRows."""
    spec, case = SPECS[i], H.ep_case(i)
    m = CS.MODELS[spec["model"]]
    anole = spec["model"] == "anole"
    cls = ea_model_anole.EaModel if anole else ea_model_llamagen.EaModel
    mdl = cls(fake_base(m["V"]), None, H.table(m["K"]))
    mdl.uniform_window = 64
    if anole:
        mdl.image_lo, mdl.image_hi = m["img_lo"], m["img_hi"]
    T, tk, tp = spec.get("temperature", 1.0), spec.get("top_k", 0), spec.get("top_p", 1.0)
    proc = verify.ProcessorSpec(temperature=T, top_p=tp if 0 < tp < 1 else 1.0, top_k=tk)
    lo, W = (m["img_lo"], m["img_hi"] - m["img_lo"]) if anole else (0, m["V"])

    def rows_of(nl, retrieve):
        win, hot = ops.cfg_mask_topk_window(cuda(nl), None, 1.0, lo, W, model=ops.MODEL_ANOLE if anole else ops.MODEL_PLAIN,
                                            img_lo=lo, img_hi=lo + W, top_k=min(tk, m["V"]), temperature=T, top_p=proc.top_p, probs=True)
        return verify.WindowRows(win, hot, retrieve, m["V"], lo)
    try:
        if spec["kind"] == "static":
            tb, g = H.static_inputs(spec, case)
            mdl.tree_buffers = mdl.generate_tree_buffers(H.tree_choices(spec["tree"]), device="cuda")
            tbuf = mdl.tree_buffers
            offs = list(g["op_off"]) + [g["R"]]
            op_list = [cuda(g["orig_prob"][offs[d]:offs[d + 1]]) for d in range(len(offs) - 1)]
            monkeypatch.setattr(random, "random", Stream(case["uniforms"]))
            tcand = cuda(case["tree_cand"])[None]
            best, alen, sp = mdl.evaluate_posterior_v1(rows_of(g["node_logits"], tbuf["retrieve_indices"]), cuda(case["cand"]), proc,
                                                       cuda(case["cart_prob"]), op_list, tbuf["p_indices"], torch.cat([tcand, tcand]),
                                                       tbuf["b_indices"], spec["lantern"], spec["k"], spec["delta"])
        else:
            nl, uniforms = H.dynamic_node_logits(spec, case)
            monkeypatch.setattr(random, "random", Stream(uniforms))
            best, alen, sp = mdl.evaluate_posterior(rows_of(nl, cuda(case["retrieve"])), cuda(case["cand"]), proc, lantern=spec["lantern"],
                                                    lantern_k=spec["k"], lantern_delta=spec["delta"])
    except LanternError as e:
        assert "dense" in str(e).lower() and spec["k"] >= m["K"] - 2      # `gtp.sum()==0 -> ones`: only the dense kernel holds it
        return
    assert int(best) == int(case["best"]) and alen == int(case["accept_len"])
    np.testing.assert_allclose(sp.cpu().numpy(), case["sample_p"], rtol=0, atol=1e-5)


def test_lumina_processors_match_golden():
    """MultiModalLogitsProcessor / InterleavedTopKLogitsWarper mirrors on the O7 golden (separate calls, as the drafter uses them)."""
    g = H.load("o7.npz")
    m = CS.MODELS["lumina"]
    c, u = cuda(g["cond"]), cuda(g["uncond"])
    cfg = u + 3.0 * (c - u)
    p = ea_model_lumina_mgpt.MultiModalLogitsProcessor(image_next_line_token_id=m["syntax"][2], image_end_token_id=m["syntax"][0],
                                                       voc_size=m["V"])
    import lantern_amd.ea_model_lumina_mgpt as LM
    old = (LM.IMAGE_LO, LM.IMAGE_HI)
    LM.IMAGE_LO, LM.IMAGE_HI = m["img_lo"], m["img_hi"]
    try:
        x = p(cfg, h_latent_dim=int(g["h"]), w_latent_dim=int(g["w"]), image_start_token_id_index=int(g["img_start"]),
              position_ids=cuda(g["pos"]))
        x = ea_model_lumina_mgpt.InterleavedTopKLogitsWarper(image_top_k=100)(x)
    finally:
        LM.IMAGE_LO, LM.IMAGE_HI = old
    assert np.array_equal(x.cpu().numpy(), g["lumina_f32"])


def test_kv_cache_mirror_copy():
    from lantern_amd.drafters.kv_cache import initialize_past_key_values
    g = H.load("kv.npz")
    cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=2, max_position_embeddings=32, hidden_size=16,
                                num_attention_heads=2)
    lin = types.SimpleNamespace(weight=torch.zeros(1, device="cuda"))
    layer = types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin))
    fake = types.SimpleNamespace(config=cfg, dtype=torch.float32, model=types.SimpleNamespace(layers=[layer, layer]))
    pkv, data_list, cur = initialize_past_key_values(fake, batch_size=2)
    assert tuple(data_list[0].shape) == (4, 2, 2, 32, 8) and cur.device.type == "cpu" and cur.dtype == torch.int64
    data_list[0].copy_(cuda(g["before"]))
    best, alen, prev = int(g["best"]), int(g["accept_len"]), int(g["prev"])
    sel = cuda(g["retrieve"][best, :alen + 1] + prev)
    for layer_kv in pkv:
        for kv in layer_kv:
            kv.copy(sel, prev)                      # KVCache.copy(indices, prev_length) as in kv_cache.py:38-50
    assert np.array_equal(data_list[0].cpu().numpy(), g["after"])
    assert torch.all(cur == prev + alen + 1)


def test_lumina_mirror_falls_back_to_the_dense_kernel(monkeypatch):
    """k + 1 neighbours cover the whole codebook and the root row's mass sits on the first candidate's neighbours: its rejection
    zeroes the residual completely (`gtp.sum()==0 -> ones`), a state the windowed kernel reports (LANTERN_ST_NEEDS_DENSE) instead
    of representing.  The mirror then runs the same step on the dense HIP kernel from the same position of the uniform stream:
    the caller sees the reference's result (here: the oracle's) either way."""
    import oracle
    i = next(i for i, s in enumerate(SPECS) if s["model"] == "lumina" and s["kind"] == "static" and s["lantern"] and not s.get("special"))
    spec, case = dict(SPECS[i]), H.ep_case(i)
    m = CS.MODELS["lumina"]
    spec["k"], spec["delta"] = m["K"] - 2, 0.1
    mdl = make_lumina(spec, True)
    tb, g = H.static_inputs(SPECS[i], case)
    mdl.tree_buffers = verify.generate_tree_buffers(H.tree_choices(spec["tree"]), device="cuda")
    mdl.tree_choices = H.tree_choices(spec["tree"])
    tbuf = mdl.tree_buffers
    offs = list(g["op_off"]) + [g["R"]]
    op_list = [cuda(g["orig_prob"][offs[d]:offs[d + 1]]) for d in range(len(offs) - 1)]
    tree_logits = (cuda(case["ss_token"]), cuda(case["ss_prob"]), op_list)
    cand, cprob, tcand = mdl.generate_candidates(tree_logits, tbuf["tree_indices"], tbuf["retrieve_indices"],
                                                 torch.tensor([[int(case["sample_token"])]], device="cuda"))
    lo, W = m["img_lo"], m["img_hi"] - m["img_lo"]
    nl_np = g["node_logits"].copy()
    x0 = int(case["cand"][0, 1])                        # the first candidate of level 1
    nl_np[0, lo:lo + W] = 0.0                           # root row: uniform over the image codes ...
    nl_np[0, x0] = -np.inf                              # ... except the candidate itself: p(x) = 0, all mass on its neighbours
    uniforms = np.full(64, 0.999)                       # every candidate is rejected
    aux = H.static_aux(tb, g, case)
    N = len(tb["tree_indices"])
    ob, oa, osp, ocnt = oracle.evaluate_posterior(H.ep_config(spec), nl_np, H.row_index_from_retrieve(tb["retrieve"], N), case["cand"], uniforms,
                                                  table=H.table(m["K"]), aux=aux)
    nl = cuda(nl_np)
    win, hot = ops.cfg_mask_topk_window(nl, None, 1.0, lo, W, model=ops.MODEL_ANOLE, img_lo=lo, img_hi=lo + W, top_k=0, probs=True)
    wr = verify.WindowRows(win, hot, tbuf["retrieve_indices"], m["V"], lo)
    built = []
    wr.dense_source = lambda: (built.append(1), verify.NodeLogits(nl, tbuf["retrieve_indices"]))[1]
    monkeypatch.setattr(random, "random", Stream(uniforms))
    best, alen, sp = mdl.evaluate_posterior(wr, cand, cart_candidates_prob=cprob, original_prob=op_list, p_indices=tbuf["p_indices"],
                                            tree_candidates=tcand, b_indices=tbuf["b_indices"], do_sample=True, lantern=True,
                                            lantern_k=spec["k"], lantern_delta=spec["delta"])
    assert built, "the windowed kernel did not report NEEDS_DENSE: the case no longer exercises the fallback"
    assert (int(best), alen) == (ob, oa)
    np.testing.assert_allclose(sp.cpu().numpy(), osp, rtol=0, atol=1e-5)
    assert int(mdl._fifo.cursor.item()) == int(ocnt[3])          # the retry re-read the same uniforms, not the next ones

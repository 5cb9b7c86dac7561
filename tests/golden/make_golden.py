#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
functions (jadohu/LANTERN at /root/reference) on CPU tensors.

Runs ONLY in the build container (the GPU box has no /root/reference); the resulting
``*.npz`` files are committed.  Nothing of the reference is copied: it is imported,
called, and its outputs are stored next to the (small) inputs it consumed.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

Import shims: LogitsWarper alias for transformers>=5, ftfy/bs4 stubs,
methods called unbound on a SimpleNamespace `self`.
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as CS  # noqa: E402

DYN_TOP_K = 200   # HF TopK processor used inside the scripted topK_genrate
REF = os.environ.get("LANTERN_REFERENCE", "/root/reference")


def import_reference():
    import transformers.generation.logits_process as lp
    if not hasattr(lp, "LogitsWarper"):
        lp.LogitsWarper = lp.LogitsProcessor
    for m in ("ftfy", "bs4"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["bs4"].BeautifulSoup = object
    sys.path.insert(0, REF)
    import models.ea_model_lumina_mgpt as lum
    import models.ea_model_llamagen as lg
    import models.ea_model_anole as an
    import models.drafters.utils_c as uc
    import models.drafters.utils as ut
    import models.drafters.choices as ch
    import models.drafters.cnets_llamagen as cl
    import models.drafters.cnets_lumina_mgpt as clu
    import models.drafters.kv_cache as kvc
    return types.SimpleNamespace(lum=lum, lg=lg, an=an, uc=uc, ut=ut, ch=ch, cl=cl, clu=clu, kvc=kvc)


class UniformStream:
    def __init__(self, vals):
        self.vals = list(map(float, vals))
        self.n = 0

    def __call__(self):
        v = self.vals[self.n]
        self.n += 1
        return v


def with_uniforms(vals, fn):
    st = UniformStream(vals)
    old = random.random
    random.random = st
    try:
        out = fn()
    finally:
        random.random = old
    return out, st.n


def nested_b(ref_b):
    """reference b_indices (nested lists of tensors / []) -> CSR."""
    off, idx = [0], []
    for row in ref_b:
        for cell in row:
            vals = cell.tolist() if isinstance(cell, torch.Tensor) else list(cell)
            idx.extend(int(v) for v in vals)
            off.append(len(idx))
    return np.asarray(off, np.int32), np.asarray(idx, np.int32)


# ----------------------------------------------------------------------------- trees

def golden_trees(R, out):
    names = ["mc_sim_7b_63", "mc_sim_7b_63_balanced", "naive_extend_57", "medusa_2_7b_63",
             "reverse_balanced_25", "chain"]
    data = {}
    for nm in names:
        choices = getattr(R.ch, nm)
        tb = R.lum.generate_tree_buffers(choices, device="cpu")
        # the LlamaGen/Anole method copies must agree with the module-level function
        for mod in (R.lg, R.an):
            tb2 = mod.EaModel.generate_tree_buffers(types.SimpleNamespace(pad_path=lambda path, length, pad_value=-2: path + [pad_value] * (length - len(path))), choices, device="cpu")
            for key in ("tree_attn_mask", "tree_indices", "tree_position_ids", "retrieve_indices"):
                assert torch.equal(tb[key], tb2[key]), (nm, key)
            assert tb["p_indices"] == tb2["p_indices"]
        boff, bidx = nested_b(tb["b_indices"])
        flat = [x for c in choices for x in c]
        coff = np.cumsum([0] + [len(c) for c in choices])
        data[f"{nm}.choices"] = np.asarray(flat, np.int32)
        data[f"{nm}.choice_off"] = np.asarray(coff, np.int32)
        data[f"{nm}.mask"] = tb["tree_attn_mask"][0, 0].numpy()
        data[f"{nm}.tree_indices"] = tb["tree_indices"].numpy()
        data[f"{nm}.pos"] = tb["tree_position_ids"].numpy()
        data[f"{nm}.retrieve"] = tb["retrieve_indices"].numpy()
        data[f"{nm}.p_indices"] = np.asarray(tb["p_indices"], np.int32)
        data[f"{nm}.b_off"] = boff
        data[f"{nm}.b_idx"] = bidx
        # drafter side (utils_c) -- chain has no level with >1 node but still valid
        db = R.uc.generate_tree_buffers(choices, device="cpu")
        data[f"{nm}.d_levels"] = np.asarray([len(db["tree_indices"])], np.int32)
        for l in range(len(db["tree_indices"])):
            data[f"{nm}.d_mask{l}"] = db["attn_mask"][l][0, 0].numpy()
            data[f"{nm}.d_ti{l}"] = db["tree_indices"][l].numpy()
            data[f"{nm}.d_rep{l}"] = np.asarray(db["repeat_nums"][l], np.int32)
    data["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(out, "trees.npz"), **data)
    print("trees.npz:", len(names), "trees")
    return {nm: getattr(R.ch, nm) for nm in names}


# --------------------------------------------------------------- evaluate_posterior

def ref_buffers(R, choices):
    tb = R.lum.generate_tree_buffers(choices, device="cpu")
    return tb


def run_static_case(R, spec, choices, table_cache):
    model, seed = spec["model"], spec["seed"]
    m = CS.model_dims(spec)
    tb = ref_buffers(R, choices)
    bufs = dict(tree_indices=tb["tree_indices"].numpy(), tree_position_ids=tb["tree_position_ids"].numpy(),
                tree_attn_mask=tb["tree_attn_mask"][0, 0].numpy(), retrieve_indices=tb["retrieve_indices"].numpy())
    g = CS.gen_static(seed, model, bufs, sigma=spec.get("sigma", 1.0), top_k=spec.get("gen_top_k", 200),
                      special=spec.get("special", ""), m=m)
    ss_prob = CS.ss_prob_from(g["orig_prob"], g["ss_token"])
    ss_token_t = torch.from_numpy(g["ss_token"])
    ss_prob_t = torch.from_numpy(ss_prob)
    op_list = []
    offs = list(g["op_off"]) + [g["R"]]
    for d in range(len(offs) - 1):
        op_list.append(torch.from_numpy(g["orig_prob"][offs[d]:offs[d + 1]]).clone())
    sample_token = torch.tensor([[g["sample_token"]]], dtype=torch.long)
    ns = types.SimpleNamespace()
    if model == "lumina":
        cand, cprob, tcand = R.lum.EaLumina_mGPT.generate_candidates(
            ns, (ss_token_t, ss_prob_t, op_list), tb["tree_indices"], tb["retrieve_indices"], sample_token)
    else:
        mod = R.lg if model == "llamagen" else R.an
        cand, cprob, tcand = mod.EaModel.generate_candidates(
            ns, (ss_token_t, ss_prob_t, op_list), tb["tree_indices"], tb["retrieve_indices"], sample_token, object())
    node_logits = torch.from_numpy(g["node_logits"])
    logits = node_logits[tb["retrieve_indices"]]
    table = table_cache(spec)
    lantern, k, delta = spec["lantern"], spec["k"], spec["delta"]
    if model == "lumina":
        ns = types.SimpleNamespace(eagle_version=1, image_syntax_tokens=torch.tensor(m["syntax"]),
                                   image_tokens=torch.arange(m["img_lo"], m["img_hi"]),
                                   nearest_latents=table, image_token_offset=m["off"])
        fn = lambda: R.lum.EaLumina_mGPT.evaluate_posterior(
            ns, logits, cand, cart_candidates_prob=cprob, original_prob=op_list, p_indices=tb["p_indices"],
            tree_candidates=tcand, b_indices=tb["b_indices"], do_sample=True, lantern=lantern,
            lantern_k=k, lantern_delta=delta)
    else:
        mod = R.lg if model == "llamagen" else R.an
        tab = table if model == "llamagen" else table.astype(np.int64)
        ns = types.SimpleNamespace(nearest_latents=tab, image_token_offset=m["off"])
        proc = R.ut.prepare_logits_processor(temperature=spec.get("temperature", 1.0),
                                             top_p=spec.get("top_p", 1.0), top_k=spec.get("top_k", 0))
        fn = lambda: mod.EaModel.evaluate_posterior_v1(
            ns, logits, cand, proc, cprob, op_list, tb["p_indices"], tcand, tb["b_indices"],
            lantern=lantern, lantern_k=k, lantern_delta=delta)
    (best, alen, sample_p), ndraw = with_uniforms(g["uniforms"], fn)
    return dict(cand=cand.numpy(), cart_prob=cprob.numpy(), tree_cand=tcand[0].numpy(),
                ss_token=g["ss_token"], ss_prob=ss_prob, sample_token=np.int64(g["sample_token"]),
                uniforms=g["uniforms"], best=np.int32(int(best)), accept_len=np.int32(int(alen)),
                sample_p=sample_p.numpy().astype(np.float32), n_draws=np.int32(ndraw),
                chk_logits=np.float64(CS.checksum(g["node_logits"])), chk_op=np.float64(CS.checksum(g["orig_prob"])))


class FakeDrafter:
    """Scripted stand-in for the drafter network so that the reference's own
    Model.topK_genrate (cnets_llamagen.py:732-912) runs its tree logic (O3+O4) on
    logits we control.  forward() returns hidden states that are ignored; head()
    returns the next scripted logits block with cond == uncond."""

    def __init__(self, script, total_tokens, depth, top_k, H=4):
        self.script = script          # list: [V] then depth x [top_k, V]
        self.calls = 0
        self.total_tokens = total_tokens
        self.depth = depth
        self.top_k = top_k
        self.H = H
        self.logsoftmax = torch.nn.LogSoftmax(dim=-1)
        self.embed_tokens = types.SimpleNamespace(weight=torch.zeros(1))
        self.tree_mask_init = torch.eye(top_k)[None, None]
        self.position_ids = torch.zeros(top_k, dtype=torch.long)
        self.stable_kv = None
        self.tree_mask = None

    def reset(self):
        self.tree_mask = None

    def __call__(self, hidden_states, input_ids=None, past_key_values=None, position_ids=None, use_cache=True):
        T = input_ids.shape[1]
        return torch.zeros(2, T, self.H), ((torch.zeros(1, 1, 1, 1),),)

    def head(self, hidden):
        blk = self.script[self.calls]
        self.calls += 1
        t = torch.from_numpy(blk)
        if hidden.dim() == 2:      # [2,H] -> [2,V]
            return torch.stack([t, t])
        return torch.stack([t, t])  # [2,top_k,V]


def dynamic_script(seed, model, depth, scale=4.0, m=None):
    """Finite raw drafter logits (cond == uncond); the top-k filtering is done by the
    reference's own HF processor inside topK_genrate (-inf inputs would turn into NaN in
    its CFG combine u + (c-u)*s)."""
    m = m or CS.MODELS[model]
    rs = np.random.RandomState(seed)
    V = m["V"]
    script = [(scale * rs.standard_normal(V)).astype(np.float32)]
    for _ in range(depth):
        script.append((scale * rs.standard_normal((CS.TOPK, V))).astype(np.float32))
    if model in ("lumina", "anole"):      # keep drafted tokens inside the image range (finite floor)
        for blk in script:
            blk[..., :m["img_lo"]] = -30000.0
            blk[..., m["img_hi"]:] = -30000.0
    return script


def run_dynamic_tree(R, seed, model, depth, total_token=59, m=None):
    script = dynamic_script(seed, model, depth, m=m)
    fake = FakeDrafter(script, total_tokens=total_token - 1, depth=depth, top_k=CS.TOPK)
    sample_token = 5 + seed % 100
    input_ids = torch.tensor([[0, 0, sample_token], [0, 0, sample_token]], dtype=torch.long)
    hidden = torch.zeros(2, 2, fake.H)
    proc = R.ut.prepare_logits_processor(temperature=1.0, top_p=1.0, top_k=DYN_TOP_K)
    draft, retrieve, tmask, tpos = R.cl.Model.topK_genrate(fake, hidden, input_ids, fake.head, proc, 3.0)
    return dict(draft_tokens=draft[0].numpy(), retrieve=retrieve.numpy(), mask=tmask[0, 0].numpy(),
                pos=tpos.numpy(), sample_token=np.int64(sample_token), depth=np.int32(depth),
                total_tokens=np.int32(total_token - 1), chk_script=np.float64(sum(CS.checksum(s) for s in script)))


def run_dynamic_case(R, spec, table_cache):
    model, seed = spec["model"], spec["seed"]
    m = CS.model_dims(spec)
    tree = run_dynamic_tree(R, seed, model, spec.get("depth", 4), m=m)
    N = len(tree["draft_tokens"])
    rs = np.random.RandomState(seed + 7919)
    node_logits = (4.0 * rs.standard_normal((N, m["V"]))).astype(np.float32)
    if model in ("lumina", "anole"):
        node_logits[:, :m["img_lo"]] = -np.inf
        node_logits[:, m["img_hi"]:] = -np.inf
    if model == "lumina":
        node_logits = CS.topk_filter(node_logits, spec.get("gen_top_k", 200))
    # make the drafted tokens plausible under the target: boost each child's logit at its parent row
    retrieve = tree["retrieve"]
    draft = tree["draft_tokens"]
    for p in range(retrieve.shape[0]):
        for d in range(1, retrieve.shape[1]):
            if retrieve[p, d] >= 0:
                par, tok = retrieve[p, d - 1], draft[retrieve[p, d]]
                mx = np.max(node_logits[par][np.isfinite(node_logits[par])])
                node_logits[par, tok] = mx - rs.uniform(0.0, 3.0)
    uniforms = rs.random_sample(64)
    if spec.get("special") == "accept_all":
        uniforms[:] = 0.0
    if spec.get("special") == "reject_all":
        uniforms[:] = 0.999999
    draft_ext = torch.cat([torch.from_numpy(draft), torch.tensor([-1])])
    cand = draft_ext[torch.from_numpy(retrieve)]
    logits = torch.from_numpy(node_logits)[torch.from_numpy(retrieve)]
    table = table_cache(spec)
    lantern, k, delta = spec["lantern"], spec["k"], spec["delta"]
    if model == "lumina":
        ns = types.SimpleNamespace(eagle_version=2, image_syntax_tokens=torch.tensor(m["syntax"]),
                                   image_tokens=torch.arange(m["img_lo"], m["img_hi"]),
                                   nearest_latents=table, image_token_offset=m["off"])
        fn = lambda: R.lum.EaLumina_mGPT.evaluate_posterior(ns, logits, cand, do_sample=True, lantern=lantern,
                                                            lantern_k=k, lantern_delta=delta)
    elif model == "eagle":
        raise NotImplementedError
    else:
        mod = R.lg if model == "llamagen" else R.an
        tab = table if model == "llamagen" else table.astype(np.int64)
        ns = types.SimpleNamespace(nearest_latents=tab, image_token_offset=m["off"])
        proc = R.ut.prepare_logits_processor(temperature=spec.get("temperature", 1.0),
                                             top_p=spec.get("top_p", 1.0), top_k=spec.get("top_k", 0))
        if spec.get("plain_eagle"):
            fn = lambda: R.ut.evaluate_posterior(logits, cand, proc)
        else:
            fn = lambda: mod.EaModel.evaluate_posterior(ns, logits, cand, proc, lantern=lantern, lantern_k=k,
                                                        lantern_delta=delta)
    (best, alen, sample_p), ndraw = with_uniforms(uniforms, fn)
    out = dict(tree)
    out.update(cand=cand.numpy(), uniforms=uniforms, best=np.int32(int(best)), accept_len=np.int32(int(alen)),
               sample_p=sample_p.numpy().astype(np.float32), n_draws=np.int32(ndraw),
               chk_logits=np.float64(CS.checksum(node_logits)))
    return out


def run_greedy_case(R, spec, table_cache):
    """a9: greedy/TVD branch (temperature<=1e-5 -> logits_processor None)."""
    model, seed = spec["model"], spec["seed"]
    m = CS.model_dims(spec)
    tree = run_dynamic_tree(R, seed, model, 4, m=m)
    N = len(tree["draft_tokens"])
    rs = np.random.RandomState(seed + 104729)
    node_logits = (4.0 * rs.standard_normal((N, m["V"]))).astype(np.float32)
    retrieve, draft = tree["retrieve"], tree["draft_tokens"]
    for p in range(retrieve.shape[0]):
        for d in range(1, retrieve.shape[1]):
            if retrieve[p, d] >= 0:
                par, tok = retrieve[p, d - 1], draft[retrieve[p, d]]
                node_logits[par, tok] = node_logits[par].max() - rs.uniform(-0.5, 1.5)
    draft_ext = torch.cat([torch.from_numpy(draft), torch.tensor([-1])])
    cand = draft_ext[torch.from_numpy(retrieve)]
    logits = torch.from_numpy(node_logits)[torch.from_numpy(retrieve)]
    table = table_cache(spec).astype(np.int64)  # uint16 torch tensors break masked assignment
    mod = R.lg if model == "llamagen" else R.an
    ns = types.SimpleNamespace(nearest_latents=table, image_token_offset=m["off"])
    best, alen, row = mod.EaModel.evaluate_posterior(ns, logits, cand, None, lantern=spec["lantern"],
                                                     lantern_k=spec["k"], lantern_delta=spec["delta"])
    out = dict(tree)
    out.update(cand=cand.numpy(), best=np.int32(int(best)), accept_len=np.int32(int(alen)),
               out_row=row.numpy().astype(np.float32), chk_logits=np.float64(CS.checksum(node_logits)))
    return out


# -------------------------------------------------------------------- O7 processors

def golden_o7(R, out):
    m = CS.MODELS["lumina"]
    V = m["V"]
    rs = np.random.RandomState(42)
    w = h = 6   # reduced latent grid: rows of 6 image tokens + newline
    N = 12
    cond = (4 * rs.standard_normal((N, V))).astype(np.float32)
    unc = (4 * rs.standard_normal((N, V))).astype(np.float32)
    proc = R.lum.MultiModalLogitsProcessor.__new__(R.lum.MultiModalLogitsProcessor)
    proc.image_next_line_token_id = m["syntax"][2]
    proc.image_end_token_id = m["syntax"][0]
    supp = torch.ones(V, dtype=torch.bool)
    supp[m["img_lo"]:m["img_hi"]] = False
    proc.suppress_token_mask = supp
    warp = R.lum.InterleavedTopKLogitsWarper(image_top_k=100)
    data = {}
    img_start = 9
    # positions chosen to hit: grid rows, a newline row, the final (eos) row, negative/zero n1
    pos = np.array([img_start + 3 + t for t in [0, 1, 5, 6, 7, 12, 13, 20, 41, 42, 27, 34]], np.int64)
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        c, u = torch.from_numpy(cond).to(dt), torch.from_numpy(unc).to(dt)
        cfg = u + 3.0 * (c - u)
        x = proc(cfg, h_latent_dim=h, w_latent_dim=w, image_start_token_id_index=img_start,
                 position_ids=torch.from_numpy(pos))
        x = warp(x)
        data[f"lumina_{tag}"] = x.float().numpy()
        # Anole: cfg then non-image -> finfo.min ; LlamaGen: cfg only
        cat = torch.cat([c[None], u[None]])           # [2,N,V]
        y = R.an.cfg_logit_process(cat, 3.0).clone()
        non_img = torch.tensor([i for i in range(0, m["img_lo"])] + [i for i in range(m["img_hi"], V)])
        y2 = y.clone()
        y2[:, :, non_img] = torch.finfo(y2.dtype).min
        data[f"plain_{tag}"] = y[0].float().numpy()
        data[f"anole_{tag}"] = y2[0].float().numpy()
    data["pos"] = pos
    data["img_start"] = np.int64(img_start)
    data["w"] = np.int32(w)
    data["h"] = np.int32(h)
    data["cond"] = cond
    data["uncond"] = unc
    np.savez_compressed(os.path.join(out, "o7.npz"), **data)
    print("o7.npz ok")


# --------------------------------------------------------------------------- O9/O10

def golden_kv(R, out):
    rs = np.random.RandomState(3)
    cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=2, max_position_embeddings=32,
                                hidden_size=16, num_attention_heads=2)
    lin = types.SimpleNamespace(weight=torch.zeros(1))
    layer = types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin))
    fake = types.SimpleNamespace(config=cfg, dtype=torch.float32,
                                 model=types.SimpleNamespace(layers=[layer, layer]))
    pkv, data_list, cur = R.kvc.initialize_past_key_values(fake, batch_size=2)
    slab = data_list[0]
    assert tuple(slab.shape) == (4, 2, 2, 32, 8)
    slab.copy_(torch.from_numpy(rs.standard_normal(tuple(slab.shape)).astype(np.float32)))
    before = slab.numpy().copy()
    retrieve = torch.tensor([[0, 1, 4, 9, -1], [0, 2, 5, -1, -1], [0, 3, 6, 7, 8]])
    best, alen, prev = 2, 3, 11
    ns = types.SimpleNamespace(cfg_mode="parallel", ea_layer=types.SimpleNamespace(
        topK_generate=lambda **kw: None), base_model=types.SimpleNamespace(lm_head=None),
        drafter_logits_processors=None, eagle_version=2)
    input_ids = torch.zeros(2, prev, dtype=torch.long)
    cand = torch.arange(15).view(3, 5)
    hid = torch.from_numpy(rs.standard_normal((1, 10, 6)).astype(np.float32))
    uhid = torch.from_numpy(rs.standard_normal((1, 10, 6)).astype(np.float32))
    sample_p = torch.zeros(20)
    sample_p[7] = 1.0
    captured = {}

    def fake_topk(**kw):
        captured.update(kw)
        return None
    ns.ea_layer.topK_generate = fake_topk
    new_ids, _, new_token, token = R.lum.EaLumina_mGPT.update_inference_inputs(
        ns, input_ids, None, cand, torch.tensor(best), alen, retrieve, True, 0, data_list, cur, hid, uhid, sample_p)
    np.savez_compressed(os.path.join(out, "kv.npz"), before=before, after=slab.numpy(), retrieve=retrieve.numpy(),
                        best=np.int32(best), accept_len=np.int32(alen), prev=np.int64(prev),
                        current_length=cur.numpy(), hidden=hid.numpy(),
                        accept_hidden=captured["hidden_states"].numpy(),
                        new_ids=new_ids.numpy(), token=token.numpy(), new_token=np.int64(new_token))
    print("kv.npz ok")


def golden_sample(R, out):
    """O5: sample() with multinomial indices captured from the reference call."""
    rs = np.random.RandomState(11)
    logits = torch.from_numpy(CS.topk_filter((4 * rs.standard_normal((5, 512))).astype(np.float32), 50))
    # a degenerate row: only 3 finite entries -> 1-cumsum hits 0 -> inf/nan -> clamp path
    logits[4, :] = -float("inf")
    logits[4, [3, 9, 27]] = torch.tensor([1.0, 0.5, 0.0])
    torch.manual_seed(0)
    try:
        idx, prob, full = R.clu.sample(logits, k=10)
    except RuntimeError:
        logits[4, :13] = torch.linspace(0, 1, 13)
        idx, prob, full = R.clu.sample(logits, k=10)
    np.savez_compressed(os.path.join(out, "sample.npz"), logits=logits.numpy(), idx=idx.numpy(), prob=prob.numpy(),
                        full=full.numpy())
    print("sample.npz ok")


def golden_codebook(R, out):
    """8f-1: the table recipe of generate_codebook.py:53-65 run verbatim on a random codebook."""
    rs = np.random.RandomState(5)
    cb = torch.from_numpy(rs.standard_normal((256, 8)).astype(np.float32))
    distances = torch.cdist(cb, cb)
    distances.fill_diagonal_(float("inf"))
    _, top_k_indices = torch.topk(distances, 255, dim=-1, largest=False)
    np.savez_compressed(os.path.join(out, "codebook.npz"), codebook=cb.numpy(),
                        table=top_k_indices.numpy().astype(np.uint16))
    print("codebook.npz ok")


def main():
    out = HERE
    torch.set_num_threads(4)
    R = import_reference()
    trees = golden_trees(R, out)
    tables = {}

    def table_cache(spec):
        K = CS.model_dims(spec)["K"]
        if K not in tables:
            tables[K] = CS.build_table(K)
        return tables[K]

    specs = []
    sid = 0
    for model in ("lumina", "llamagen", "anole"):
        for tree in ("mc_sim_7b_63", "naive_extend_57"):
            for (lantern, k, delta) in [(False, 10, 0.1), (True, 1, 0.1), (True, 10, 0.1), (True, 300, 0.1),
                                        (True, 10, 5.0), (True, 300, 5.0), (True, 300, 20.0), (True, 1022, 0.4)]:
                for rep in range(2):
                    sid += 1
                    specs.append(dict(kind="static", model=model, tree=tree, seed=1000 + sid, lantern=lantern, k=k,
                                      delta=delta, sigma=[0.5, 1.5][rep]))
    for model in ("lumina", "llamagen", "anole"):
        for special in ("syntax", "nonimage", "accept_all", "reject_all"):
            if special in ("syntax",) and model != "lumina":
                continue
            if special == "nonimage" and model != "lumina":   # Anole has no guard: the reference raises IndexError
                continue
            sid += 1
            specs.append(dict(kind="static", model=model, tree="mc_sim_7b_63", seed=1000 + sid, lantern=True, k=50,
                              delta=0.2, special=special))
    # processors inside evaluate_posterior (LlamaGen/Anole)
    for model in ("llamagen", "anole"):
        for (T, tp, tk) in [(1.0, 1.0, 100), (0.8, 1.0, 50), (1.0, 0.9, 0), (0.7, 0.8, 100)]:
            sid += 1
            specs.append(dict(kind="static", model=model, tree="naive_extend_57", seed=1000 + sid, lantern=True, k=20,
                              delta=0.1, temperature=T, top_p=tp, top_k=tk))
            sid += 1
            specs.append(dict(kind="dynamic", model=model, seed=1000 + sid, lantern=True, k=20, delta=3.0,
                              temperature=T, top_p=tp, top_k=tk))
    for model in ("lumina", "llamagen", "anole"):
        for (lantern, k, delta) in [(False, 10, 0.1), (True, 1, 0.1), (True, 10, 0.1), (True, 300, 0.1),
                                    (True, 10, 5.0), (True, 300, 5.0), (True, 1022, 0.4)]:
            for rep in range(2):
                sid += 1
                specs.append(dict(kind="dynamic", model=model, seed=1000 + sid, lantern=lantern, k=k, delta=delta,
                                  depth=[4, 5][rep]))
        for special in ("accept_all", "reject_all"):
            sid += 1
            specs.append(dict(kind="dynamic", model=model, seed=1000 + sid, lantern=True, k=300, delta=5.0,
                              special=special))
    sid += 1
    specs.append(dict(kind="dynamic", model="llamagen", seed=1000 + sid, lantern=False, k=1, delta=0.1,
                      plain_eagle=True, top_k=0))
    for model in ("llamagen", "anole"):
        for (lantern, k, delta) in [(False, 10, 0.1), (True, 10, 0.1), (True, 300, 0.3), (True, 100, 5.0)]:
            sid += 1
            specs.append(dict(kind="greedy", model=model, seed=1000 + sid, lantern=lantern, k=k, delta=delta))

    data = {}
    for i, spec in enumerate(specs):
        if spec["kind"] == "static":
            res = run_static_case(R, spec, trees[spec["tree"]], table_cache)
        elif spec["kind"] == "dynamic":
            res = run_dynamic_case(R, spec, table_cache)
        else:
            res = run_greedy_case(R, spec, table_cache)
        for key, val in res.items():
            data[f"c{i}.{key}"] = val
        print(i, json.dumps(spec), "-> best", int(res["best"]), "alen", int(res["accept_len"]),
              "draws", int(res.get("n_draws", -1)))
    data["specs"] = np.asarray(json.dumps(specs))
    np.savez_compressed(os.path.join(out, "evaluate_posterior.npz"), **data)
    print("evaluate_posterior.npz:", len(specs), "cases")

    golden_o7(R, out)
    golden_kv(R, out)
    golden_sample(R, out)
    golden_codebook(R, out)


if __name__ == "__main__":
    main()

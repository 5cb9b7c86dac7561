"""Shared helpers: load golden cases and rebuild the full inputs the reference consumed."""
import json
import os

import numpy as np

import cases as CS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DYN_TOP_K = 200

_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return _cache[name]


def ep_specs():
    d = load("evaluate_posterior.npz")
    return json.loads(str(d["specs"]))


def ep_case(i):
    d = load("evaluate_posterior.npz")
    pre = f"c{i}."
    return {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}


def full_specs():
    return json.loads(str(load("evaluate_posterior_full.npz")["specs"]))


def full_case(i):
    """A case of the REAL-size fixture (make_golden_fullsize.py); `sample_p` is rebuilt from its sparse form."""
    d = load("evaluate_posterior_full.npz")
    pre = f"c{i}."
    c = {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}
    if "sample_p_ids" in c:
        p = np.zeros(int(c["sample_p_len"]), np.float32)
        p[c["sample_p_ids"]] = c["sample_p_vals"]
        c["sample_p"] = p
    return c


_tables = {}


def table(K):
    if K not in _tables:
        _tables[K] = CS.build_table(K)
    return _tables[K]


def table_for(spec):
    """Neighbour table a golden spec was run with: the reduced one, or the real-size one (rebuilt from its seed and held
    to the SHA-256 the fixture recorded when the reference consumed it)."""
    if spec.get("size") != "full":
        return table(CS.MODELS[spec["model"]]["K"])
    m = CS.FULL[spec["model"]]
    key = ("full", m["K"], m["C"])
    if key not in _tables:
        import hashlib
        t = CS.build_table_full(m["K"], m["C"])
        want = str(load("evaluate_posterior_full.npz")[f"table.{m['K']}x{m['C']}.sha256"])
        assert hashlib.sha256(t.tobytes()).hexdigest() == want, "the rebuilt neighbour table is not the one the reference consumed"
        _tables[key] = t
    return _tables[key]


def tree_buffers(name):
    d = load("trees_random.npz" if name.startswith("rand") else "trees.npz")      # rand00..rand39: make_golden_trees_random.py
    pre = name + "."
    return {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}


def tree_choices(name):
    b = tree_buffers(name)
    off = b["choice_off"]
    return [b["choices"][off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


def static_inputs(spec, case):
    """Rebuild (node_logits, orig_prob, op_off) from the seed; check the checksums stored
    when the reference consumed them."""
    tb = tree_buffers(spec["tree"])
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["pos"], tree_attn_mask=tb["mask"],
                retrieve_indices=tb["retrieve"])
    g = CS.gen_static(spec["seed"], spec["model"], bufs, sigma=spec.get("sigma", 1.0),
                      top_k=spec.get("gen_top_k", 200), special=spec.get("special", ""), m=CS.model_dims(spec))
    assert abs(CS.checksum(g["node_logits"]) - float(case["chk_logits"])) < 1e-6
    assert abs(CS.checksum(g["orig_prob"]) - float(case["chk_op"])) < 1e-9
    assert np.array_equal(g["ss_token"], case["ss_token"])
    return tb, g


def dynamic_script(seed, model, depth, scale=4.0, m=None):
    m = m or CS.MODELS[model]
    rs = np.random.RandomState(seed)
    V = m["V"]
    script = [(scale * rs.standard_normal(V)).astype(np.float32)]
    for _ in range(depth):
        script.append((scale * rs.standard_normal((CS.TOPK, V))).astype(np.float32))
    if model in ("lumina", "anole"):
        for blk in script:
            blk[..., :m["img_lo"]] = -30000.0
            blk[..., m["img_hi"]:] = -30000.0
    return script


def dynamic_node_logits(spec, case, greedy=False):
    """Target rows of the dynamic-tree cases (same construction as make_golden.py)."""
    m = CS.model_dims(spec)
    N = len(case["draft_tokens"])
    retrieve, draft = case["retrieve"], case["draft_tokens"]
    if greedy:
        rs = np.random.RandomState(spec["seed"] + 104729)
        nl = (4.0 * rs.standard_normal((N, m["V"]))).astype(np.float32)
        for p in range(retrieve.shape[0]):
            for d in range(1, retrieve.shape[1]):
                if retrieve[p, d] >= 0:
                    par, tok = retrieve[p, d - 1], draft[retrieve[p, d]]
                    nl[par, tok] = nl[par].max() - rs.uniform(-0.5, 1.5)
        assert abs(CS.checksum(nl) - float(case["chk_logits"])) < 1e-6
        return nl, None
    rs = np.random.RandomState(spec["seed"] + 7919)
    nl = (4.0 * rs.standard_normal((N, m["V"]))).astype(np.float32)
    if spec["model"] in ("lumina", "anole"):
        nl[:, :m["img_lo"]] = -np.inf
        nl[:, m["img_hi"]:] = -np.inf
    if spec["model"] == "lumina":
        nl = CS.topk_filter(nl, spec.get("gen_top_k", 200))
    for p in range(retrieve.shape[0]):
        for d in range(1, retrieve.shape[1]):
            if retrieve[p, d] >= 0:
                par, tok = retrieve[p, d - 1], draft[retrieve[p, d]]
                mx = np.max(nl[par][np.isfinite(nl[par])])
                nl[par, tok] = mx - rs.uniform(0.0, 3.0)
    uniforms = rs.random_sample(64)
    if spec.get("special") == "accept_all":
        uniforms[:] = 0.0
    if spec.get("special") == "reject_all":
        uniforms[:] = 0.999999
    assert abs(CS.checksum(nl) - float(case["chk_logits"])) < 1e-6
    assert np.array_equal(uniforms, case["uniforms"])
    return nl, uniforms


def row_index_from_retrieve(retrieve, N):
    """tree_logits[retrieve_indices]: a -1 wraps to the last node row."""
    r = np.asarray(retrieve, np.int64).copy()
    r[r < 0] += N
    return r.astype(np.int32)


def hf_process_rows(rows, top_k):
    """HF TopKLogitsWarper on rows (used to rebuild what topK_genrate saw)."""
    return CS.topk_filter(rows, top_k)


def ep_config(spec):
    """oracle.EpConfig for a golden spec (reduced-vocabulary model constants of cases.MODELS)."""
    import oracle
    m = CS.model_dims(spec)
    static = spec["kind"] == "static"
    common = dict(lantern=bool(spec["lantern"]), k=int(spec["k"]), delta=float(spec["delta"]))
    if spec["model"] == "lumina":
        return oracle.EpConfig(mode=oracle.MODE_STATIC_LUMINA if static else oracle.MODE_DYNAMIC,
                               syntax_shortcut=True, tok_offset=m["off"], img_lo=m["img_lo"], img_hi=m["img_hi"],
                               syntax=m["syntax"], **common)
    proc = dict(temperature=spec.get("temperature", 1.0), top_p=spec.get("top_p", 1.0), top_k=spec.get("top_k", 0))
    return oracle.EpConfig(mode=oracle.MODE_STATIC_LG if static else oracle.MODE_DYNAMIC, tok_offset=m["off"],
                           img_lo=m["img_lo"], img_hi=m["img_hi"], **common, **proc)


def static_aux(tb, g, case):
    import oracle
    return oracle.StaticAux(cart_prob=case["cart_prob"], orig_prob=g["orig_prob"], op_off=g["op_off"],
                            p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"], tree_cand=case["tree_cand"])

"""Deterministic synthetic inputs for the golden vectors (numpy RandomState only).

Shared by ``make_golden.py`` (which feeds them to the REFERENCE, in the build container)
and by the tests (which feed the same inputs to the oracle and to the HIP path).  Small
tensors the reference consumed (candidates, probabilities, uniforms) are stored in the
``.npz`` fixtures; the big ones (logits, drafter distributions, neighbour table) are
regenerated from the seed here and pinned by a float64 checksum stored in the fixture.

No file of the reference is read or copied here: this is our own input generator.
"""
from __future__ import annotations

import numpy as np

TOPK = 10

# tree shapes of the reference (models/drafters/choices.py:1-32) restated as data
mc_sim_7b_63 = [[0], [1], [2], [3], [0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [2, 0], [2, 1], [3, 0],
                [0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 1, 0], [0, 1, 1], [0, 2, 0], [0, 2, 1], [1, 0, 0],
                [0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 2], [0, 0, 0, 0, 0], [0, 0, 0, 0, 1]]

MODELS = {
    # name: (V, K, tok_offset, img_lo, img_hi, syntax tokens)
    "lumina": dict(V=2048, K=1024, off=4, img_lo=4, img_hi=1028, syntax=(1028, 1029, 1035, 1060)),
    "anole": dict(V=2048, K=1024, off=4, img_lo=4, img_hi=1028, syntax=()),
    "llamagen": dict(V=1024, K=1024, off=0, img_lo=0, img_hi=1024, syntax=()),
}


# BASELINE's real sizes (data/configs/lumina_mgpt_config.json, llamagen_t2i_config.json): the reference-run
# fixtures of make_golden_fullsize.py.  C = codebook width of the synthetic VQ codebook the neighbour table is built from.
FULL = {
    "lumina": dict(V=65536, K=8192, off=4, img_lo=4, img_hi=8196, syntax=(8196, 8197, 8803, 8828), C=256),
    "anole": dict(V=65536, K=8192, off=4, img_lo=4, img_hi=8196, syntax=(), C=256),
    "llamagen": dict(V=16384, K=16384, off=0, img_lo=0, img_hi=16384, syntax=(), C=8),
}


def model_dims(spec: dict) -> dict:
    """Vocabulary / codebook constants of a golden spec: reduced (MODELS) unless the spec says size = "full"."""
    return FULL[spec["model"]] if spec.get("size") == "full" else MODELS[spec["model"]]


def full_codebook(K: int, C: int, seed: int = 0) -> np.ndarray:
    """N(0,1) codebook rounded to multiples of 2^-16: every product and every partial sum of a squared distance is then
    exact in float64 whatever the summation order (19-bit values, 38-bit squares, <= 256 terms), so the neighbour table
    below comes out bit-identical on any machine / BLAS."""
    rs = np.random.RandomState(seed)
    return (np.round(rs.standard_normal((K, C)) * 65536.0) / 65536.0).astype(np.float32)


def build_table_full(K: int, C: int, seed: int = 0, chunk: int = 1024) -> np.ndarray:
    """The recipe of generate_codebook.py:53-65 (pairwise L2, diagonal = inf, the K-1 nearest in ascending order, uint16)
    at the real codebook sizes: exact float64 squared distances (sqrt is monotone), stable order on the (rare) exact ties."""
    cb = full_codebook(K, C, seed).astype(np.float64)
    n2 = (cb * cb).sum(1)
    out = np.empty((K, K - 1), np.uint16)
    for r0 in range(0, K, chunk):
        r1 = min(K, r0 + chunk)
        d2 = n2[r0:r1, None] + n2[None, :] - 2.0 * (cb[r0:r1] @ cb.T)
        d2[np.arange(r1 - r0), np.arange(r0, r1)] = np.inf
        out[r0:r1] = np.argsort(d2, axis=1, kind="stable")[:, :K - 1].astype(np.uint16)
    return out


def build_table(K: int, C: int = 8, seed: int = 0) -> np.ndarray:
    """Neighbour table with the recipe of generate_codebook.py:53-65 on a random codebook."""
    rs = np.random.RandomState(seed)
    cb = rs.standard_normal((K, C))
    d = ((cb[:, None, :] - cb[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    order = np.argsort(d, axis=1, kind="stable")[:, :K - 1]
    return order.astype(np.uint16)


def softmax64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float64)
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


def topk_filter(rows: np.ndarray, k: int) -> np.ndarray:
    """InterleavedTopKLogitsWarper semantics: < kth-largest -> -inf."""
    if k <= 0:
        return rows
    out = rows.copy()
    kk = min(k, rows.shape[-1])
    thr = np.sort(rows, axis=-1)[..., -kk][..., None]
    out[rows < thr] = -np.inf
    return out


def node_parents(mask: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """parent node id of every node from the ancestor mask + depth."""
    N = len(pos)
    par = np.full(N, -1, np.int64)
    for n in range(1, N):
        anc = np.nonzero(mask[n] > 0)[0]
        cand = [a for a in anc if pos[a] == pos[n] - 1]
        par[n] = cand[0]
    return par


def target_rows(rs, n_rows: int, m: dict, scale: float, top_k: int, mask_non_image: bool) -> np.ndarray:
    raw = (scale * rs.standard_normal((n_rows, m["V"]))).astype(np.float32)
    if mask_non_image:
        raw[:, :m["img_lo"]] = -np.inf
        raw[:, m["img_hi"]:] = -np.inf
    return topk_filter(raw, top_k)


def gen_static(seed: int, model: str, buffers: dict, sigma: float = 1.0, scale: float = 4.0,
               top_k: int = 200, special: str = "", m: dict = None) -> dict:
    """Inputs of one static-tree (EAGLE-1 / LANTERN++) verify step.

    buffers: tree_indices [N], retrieve_indices [P,D], tree_attn_mask [N,N], tree_position_ids [N].
    Returns raw drafter outputs (ss_token, ss_prob, orig_prob, op_off), processed target rows
    `node_logits` [N,V], sample_token, uniforms.
    """
    m = m or MODELS[model]
    rs = np.random.RandomState(seed)
    ti = np.asarray(buffers["tree_indices"])
    pos = np.asarray(buffers["tree_position_ids"])
    mask = np.asarray(buffers["tree_attn_mask"]).reshape(len(ti), len(ti))
    N = len(ti)
    par = node_parents(mask, pos)
    rows = (ti[1:] - 1) // TOPK
    R = int(rows.max()) + 1
    parent_of_row = np.zeros(R, np.int64)
    for n in range(1, N):
        parent_of_row[(ti[n] - 1) // TOPK] = par[n]
    mask_img = model in ("lumina", "anole")
    raw = (scale * rs.standard_normal((N, m["V"]))).astype(np.float32)
    tgt = raw.copy()
    if mask_img:
        tgt[:, :m["img_lo"]] = -np.inf
        tgt[:, m["img_hi"]:] = -np.inf
    if model == "lumina":
        tgt = topk_filter(tgt, top_k)   # Lumina filters before the gather (ea_model_lumina_mgpt.py:605)
    dr = raw[parent_of_row] + (sigma * rs.standard_normal((R, m["V"]))).astype(np.float32)
    if mask_img:
        dr[:, :m["img_lo"]] = -np.inf
        dr[:, m["img_hi"]:] = -np.inf
    dr = topk_filter(dr, top_k)
    probs64 = softmax64(dr)
    orig_prob = probs64.astype(np.float32)
    ss_token = np.zeros((R, TOPK), np.int64)
    for r in range(R):
        ss_token[r] = rs.choice(m["V"], size=TOPK, replace=False, p=probs64[r])
    sample_token = int(rs.randint(m["img_lo"], m["img_hi"]))
    uniforms = rs.random_sample(64)
    if special == "syntax" and m["syntax"]:
        ss_token[0, 0] = m["syntax"][2]          # a syntax token as first root child
    elif special == "nonimage" and mask_img:
        ss_token[0, 0] = m["img_hi"] + 7          # a non-image, non-syntax token
        uniforms[0] = 0.5
    elif special == "accept_all":
        uniforms[:] = 0.0
    elif special == "reject_all":
        uniforms[:] = 0.999999
    # level structure of the drafter rows: level d = rows whose parent has depth d
    depth_of_row = pos[parent_of_row]
    n_levels = int(depth_of_row.max()) + 1
    op_off = np.zeros(n_levels, np.int32)
    for d in range(n_levels):
        op_off[d] = int(np.nonzero(depth_of_row == d)[0][0])
    return dict(node_logits=tgt, ss_token=ss_token, orig_prob=orig_prob, op_off=op_off,
                sample_token=sample_token, uniforms=uniforms, R=R)


def ss_prob_from(orig_prob: np.ndarray, ss_token: np.ndarray) -> np.ndarray:
    """Conditional probabilities of sample() (cnets_lumina_mgpt.py:944-953), float32 op order."""
    R, k = ss_token.shape
    out = np.zeros((R, k), np.float32)
    for r in range(R):
        p = orig_prob[r, ss_token[r]].astype(np.float32)
        c = np.cumsum(p.astype(np.float64)).astype(np.float32)
        c = np.concatenate([np.zeros(1, np.float32), c[:-1]])
        with np.errstate(divide="ignore", invalid="ignore"):
            v = p / (np.float32(1.0) - c)
        v[np.isinf(v)] = -1
        v[np.isnan(v)] = -1
        out[r] = np.clip(v, 0.0, 1.0)
    return out


def gen_dynamic(seed: int, model: str, depth: int = 4, total_tokens: int = 58, sigma: float = 1.0,
                scale: float = 4.0, top_k: int = 200) -> dict:
    """Inputs of the EAGLE-2 dynamic tree: per-depth drafter log-prob top-10s, cumulative
    scores, parents -- exactly what the tail of topK_genrate consumes
    (cnets_llamagen.py:765-833) -- plus a pool of target rows indexed by flat score index."""
    m = MODELS[model]
    rs = np.random.RandomState(seed)
    V = m["V"]
    mask_img = model in ("lumina", "anole")

    def proc(x):
        x = x.copy()
        if mask_img:
            x[..., :m["img_lo"]] = -np.inf
            x[..., m["img_hi"]:] = -np.inf
        return topk_filter(x, top_k)

    def logsm(x):
        x = x.astype(np.float64)
        mx = x.max(-1, keepdims=True)
        return (x - mx - np.log(np.exp(x - mx).sum(-1, keepdims=True))).astype(np.float32)

    root_raw = (scale * rs.standard_normal(V)).astype(np.float32)
    # drafter at the root
    lp = logsm(proc(root_raw + (sigma * rs.standard_normal(V)).astype(np.float32)))
    order = np.argsort(-lp, kind="stable")[:TOPK]
    scores = lp[order]
    scores_list = [scores[None].copy()]
    parents_list = [np.zeros(1, np.int64)]
    ss_token = [order[None].copy()]
    # target rows per flat score index (node identity before selection)
    n_scores = TOPK + TOPK * TOPK * depth
    raw_rows = {"root": root_raw}
    flat_raw = np.zeros((n_scores, V), np.float32)
    cur_raw = (scale * rs.standard_normal((TOPK, V))).astype(np.float32)   # rows of the 10 depth-1 nodes
    flat_raw[:TOPK] = cur_raw
    topk_cs_index = np.arange(TOPK)
    for i in range(depth):
        bias1 = TOPK if i > 0 else 0
        bias2 = max(0, i - 1)
        bias = 1 + TOPK ** 2 * bias2 + bias1
        parents_list.append(topk_cs_index + bias)
        lp = logsm(proc(cur_raw + (sigma * rs.standard_normal((TOPK, V))).astype(np.float32)))
        idx = np.argsort(-lp, axis=-1, kind="stable")[:, :TOPK]
        tp = np.take_along_axis(lp, idx, axis=-1)
        cu = (tp + scores[:, None]).astype(np.float32)
        flat = cu.reshape(-1)
        topk_cs_index = np.argsort(-flat, kind="stable")[:TOPK]
        scores = flat[topk_cs_index]
        ss_token.append(idx.copy())
        scores_list.append(cu.copy())
        child_raw = (scale * rs.standard_normal((TOPK * TOPK, V))).astype(np.float32)
        base = TOPK + TOPK * TOPK * i
        flat_raw[base:base + TOPK * TOPK] = child_raw
        cur_raw = child_raw[topk_cs_index]
    scores_flat = np.concatenate([s.reshape(-1) for s in scores_list]).astype(np.float32)
    tokens_flat = np.concatenate([t.reshape(-1) for t in ss_token]).astype(np.int64)
    parents_flat = np.concatenate(parents_list).astype(np.int64)
    sample_token = int(rs.randint(m["img_lo"], m["img_hi"]))
    uniforms = rs.random_sample(64)
    tgt_flat = flat_raw
    if mask_img:
        tgt_flat[:, :m["img_lo"]] = -np.inf
        tgt_flat[:, m["img_hi"]:] = -np.inf
    if model == "lumina":
        tgt_flat = topk_filter(tgt_flat, top_k)
        root_t = topk_filter(proc(root_raw)[None], top_k)[0]
    else:
        root_t = root_raw.copy()
        if mask_img:
            root_t[:m["img_lo"]] = -np.inf
            root_t[m["img_hi"]:] = -np.inf
    return dict(scores=scores_flat, tokens=tokens_flat, parents=parents_flat, sample_token=sample_token,
                uniforms=uniforms, root_logits=root_t.astype(np.float32), flat_logits=tgt_flat,
                total_tokens=total_tokens, depth=depth)


def csr_to_nested(b_off, b_idx, P, D):
    out = []
    for p in range(P):
        row = []
        for d in range(D):
            a, b = b_off[p * D + d], b_off[p * D + d + 1]
            row.append([int(x) for x in b_idx[a:b]])
        out.append(row)
    return out


def checksum(a: np.ndarray) -> float:
    a = np.asarray(a, np.float64)
    a = np.where(np.isfinite(a), a, 0.0)
    w = np.cos(np.arange(a.size, dtype=np.float64) * 0.37)
    return float((a.reshape(-1) * w).sum())


def kv_full_inputs(seed=77, S=96):
    """The 7B slab geometry (kv_cache.py:101-122: [2 L, B, Hkv, S_max, d] = [64, 2, 32, S, 128]) at a short S_max, f32 values that are small integers (exact in any
    dtype the copy goes through), regenerated from the seed on both sides."""
    rs = np.random.RandomState(seed)
    return rs.randint(-1000, 1000, size=(64, 2, 32, S, 128)).astype(np.float32)

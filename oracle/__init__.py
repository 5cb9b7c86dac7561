"""CPU oracle for the LANTERN verify/accept hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package, and only as the checker / the timed CPU baseline.  The product
(``lantern_amd``) never imports it.

numpy-in / numpy-out wrappers (ctypes) around ``liblantern_oracle.so``, the plain-C
restatement in ``lantern_oracle.c``.  Each wrapper names the reference function it
restates; parity of the restatement is pinned by ``tests/golden`` (see
``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LANTERN_ORACLE_LIBRARY: the ASan + UBSan build of the same file, `make -C oracle asan`; tests/test_sanitizers_cpu.py)
_SO = os.environ.get("LANTERN_ORACLE_LIBRARY") or os.path.join(_HERE, "liblantern_oracle.so")

MODE_DYNAMIC, MODE_STATIC_LUMINA, MODE_STATIC_LG = 0, 1, 2
MODEL_PLAIN, MODEL_LUMINA, MODEL_ANOLE = 0, 1, 2
F32, BF16 = 0, 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lantern_oracle.c")
    hdr = os.path.join(_HERE, "lantern_oracle.h")
    if os.environ.get("LANTERN_ORACLE_LIBRARY"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liblantern_oracle.so"])
    return _SO


class _EpParams(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("V", C.c_int32),
        ("mode", C.c_int32), ("syntax_shortcut", C.c_int32), ("tok_offset", C.c_int32),
        ("img_lo", C.c_int32), ("img_hi", C.c_int32),
        ("n_syntax", C.c_int32), ("syntax", C.c_int32 * 8),
        ("lantern", C.c_int32), ("k", C.c_int32),
        ("table_rows", C.c_int32), ("table_cols", C.c_int32),
        ("top_k", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float),
        ("delta", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.lo_sample_inverse_cdf.restype = C.c_int64
    return _lib


def _p(a, ty=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


@dataclass
class EpConfig:
    """Per-model switches of evaluate_posterior."""
    mode: int = MODE_DYNAMIC
    syntax_shortcut: bool = False
    tok_offset: int = 0
    img_lo: int = 0
    img_hi: int = 1 << 30
    syntax: Sequence[int] = ()
    lantern: bool = False
    k: int = 1000
    delta: float = 0.1
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = 0

    @staticmethod
    def lumina(static: bool, **kw) -> "EpConfig":
        return EpConfig(mode=MODE_STATIC_LUMINA if static else MODE_DYNAMIC, syntax_shortcut=True,
                        tok_offset=4, img_lo=4, img_hi=8196, syntax=(8196, 8197, 8803, 8828), **kw)

    @staticmethod
    def llamagen(static: bool, **kw) -> "EpConfig":
        return EpConfig(mode=MODE_STATIC_LG if static else MODE_DYNAMIC, **kw)

    @staticmethod
    def anole(static: bool, **kw) -> "EpConfig":
        return EpConfig(mode=MODE_STATIC_LG if static else MODE_DYNAMIC, tok_offset=4, img_lo=4, img_hi=8196, **kw)


@dataclass
class StaticAux:
    """Static-tree (EAGLE-1 / LANTERN++) side inputs of evaluate_posterior."""
    cart_prob: np.ndarray          # [P,D] f32
    orig_prob: np.ndarray          # [R,V] f32 (drafter levels concatenated)
    op_off: np.ndarray             # [D-1] i32 row offset of level d
    p_idx: np.ndarray              # [P,D] i32
    b_off: np.ndarray              # [P*D+1] i32
    b_idx: np.ndarray              # [nb] i32
    tree_cand: np.ndarray          # [N] i64


def evaluate_posterior(cfg: EpConfig, logits: np.ndarray, row_index: np.ndarray, cand: np.ndarray,
                       uniforms: np.ndarray, table: Optional[np.ndarray] = None,
                       aux: Optional[StaticAux] = None):
    """Sampling branch of evaluate_posterior (ea_model_lumina_mgpt.py:610-726,
    ea_model_llamagen.py:709-787 and :597-669).  logits: [rows,V] f32; row_index [P,D]
    maps (path,depth) to a logits row.  Returns (best, accept_len, sample_p, counters)."""
    logits = _c(logits, np.float32)
    P, D = cand.shape
    V = logits.shape[-1]
    logits2 = logits.reshape(-1, V)
    prm = _EpParams()
    prm.P, prm.D, prm.V = P, D, V
    prm.mode = cfg.mode
    prm.syntax_shortcut = int(cfg.syntax_shortcut)
    prm.tok_offset = cfg.tok_offset
    prm.img_lo, prm.img_hi = cfg.img_lo, min(cfg.img_hi, 2**31 - 1)
    prm.n_syntax = len(cfg.syntax)
    for i, s in enumerate(cfg.syntax):
        prm.syntax[i] = s
    prm.lantern, prm.k, prm.delta = int(cfg.lantern), cfg.k, float(cfg.delta)
    if table is not None:
        table = _c(table, np.uint16)
        prm.table_rows, prm.table_cols = table.shape
    prm.top_k, prm.temperature, prm.top_p = cfg.top_k, cfg.temperature, cfg.top_p
    row_index = _c(row_index, np.int32)
    cand = _c(cand, np.int64)
    uniforms = _c(uniforms, np.float64)
    best = C.c_int32(0)
    alen = C.c_int32(0)
    sample_p = np.empty(V, np.float32)
    counters = np.zeros(6, np.int32)
    if aux is not None:
        a = [_c(aux.cart_prob, np.float32), _c(aux.orig_prob, np.float32), _c(aux.op_off, np.int32),
             _c(aux.p_idx, np.int32), _c(aux.b_off, np.int32), _c(aux.b_idx, np.int32), _c(aux.tree_cand, np.int64)]
    else:
        a = [None] * 7
    rc = lib().lo_evaluate_posterior(C.byref(prm), _p(logits2), _p(row_index), _p(cand), *[_p(x) for x in a],
                                     _p(table), _p(uniforms), C.c_int32(len(uniforms)), C.byref(best),
                                     C.byref(alen), _p(sample_p), _p(counters))
    if rc != 0:
        raise RuntimeError(f"lo_evaluate_posterior rc={rc}")
    return best.value, alen.value, sample_p, counters


def evaluate_posterior_greedy(logits, row_index, cand, lantern=False, k=1000, delta=0.1, tok_offset=0, table=None):
    """Greedy/TVD branch (ea_model_llamagen.py:789-905)."""
    logits = _c(logits, np.float32)
    P, D = cand.shape
    V = logits.shape[-1]
    row_index = _c(row_index, np.int32)
    cand = _c(cand, np.int64)
    tr = tc = 0
    if table is not None:
        table = _c(table, np.uint16)
        tr, tc = table.shape
    best, alen = C.c_int32(0), C.c_int32(0)
    out = np.empty(V, np.float32)
    rc = lib().lo_evaluate_posterior_greedy(P, D, V, _p(logits.reshape(-1, V)), _p(row_index), _p(cand), int(lantern), k,
                                            C.c_double(delta), tok_offset, _p(table), tr, tc, C.byref(best),
                                            C.byref(alen), _p(out))
    if rc != 0:
        raise RuntimeError(f"lo_evaluate_posterior_greedy rc={rc}")
    return best.value, alen.value, out


def cfg_mask_topk(cond, uncond, cfg, model=MODEL_PLAIN, pos_ids=None, pos_base=0, w=48, h=48,
                  img_lo=4, img_hi=8196, newline_id=8803, eos_id=8196, top_k=0, bf16=False):
    """Logit post-processing of tree_decoding (ea_model_lumina_mgpt.py:597-605,45-86,106-112;
    ea_model_anole.py:930-931; ea_model_llamagen.py:930).  bf16=True: cond/uncond are uint16
    bf16 bit patterns and torch's per-op bf16 rounding is reproduced."""
    dt = np.uint16 if bf16 else np.float32
    cond, uncond = _c(cond, dt), _c(uncond, dt)
    N, V = cond.shape
    out = np.empty((N, V), np.float32)
    pos = _c(pos_ids if pos_ids is not None else np.zeros(N), np.int64)
    rc = lib().lo_cfg_mask_topk(_p(cond), _p(uncond), BF16 if bf16 else F32, N, V, C.c_float(cfg), model, _p(pos),
                                C.c_int64(pos_base), w, h, img_lo, img_hi, newline_id, eos_id, top_k, _p(out))
    if rc != 0:
        raise RuntimeError(f"lo_cfg_mask_topk rc={rc}")
    return out


def _flatten_choices(tree_choices):
    flat, off = [], [0]
    for c in tree_choices:
        flat.extend(c)
        off.append(len(flat))
    return np.asarray(flat, np.int32), np.asarray(off, np.int32)


def tree_static_build(tree_choices, top_k=10):
    """generate_tree_buffers, target side (ea_model_lumina_mgpt.py:140-277)."""
    flat, off = _flatten_choices(tree_choices)
    n = len(tree_choices)
    N, P, D, bt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib().lo_tree_static_sizes(_p(flat), _p(off), n, C.byref(N), C.byref(P), C.byref(D), C.byref(bt))
    N, P, D, bt = N.value, P.value, D.value, bt.value
    mask = np.empty((N, N), np.float32)
    ti = np.empty(N, np.int64)
    pos = np.empty(N, np.int64)
    ret = np.empty((P, D), np.int64)
    pidx = np.empty((P, D), np.int32)
    boff = np.empty(P * D + 1, np.int32)
    bidx = np.empty(max(bt, 1), np.int32)
    rc = lib().lo_tree_static_build(_p(flat), _p(off), n, top_k, _p(mask), _p(ti), _p(pos), _p(ret), _p(pidx),
                                    _p(boff), _p(bidx))
    if rc != 0:
        raise RuntimeError(f"lo_tree_static_build rc={rc}")
    return dict(tree_attn_mask=mask, tree_indices=ti, tree_position_ids=pos, retrieve_indices=ret,
                p_indices=pidx, b_off=boff, b_idx=bidx[:bt])


def tree_drafter_build(tree_choices, top_k=10):
    """generate_tree_buffers, drafter side (drafters/utils_c.py:100-179)."""
    flat, off = _flatten_choices(tree_choices)
    n = len(tree_choices)
    nl = C.c_int()
    counts = np.zeros(64, np.int32)
    lib().lo_tree_drafter_sizes(_p(flat), _p(off), n, C.byref(nl), _p(counts))
    L = nl.value
    counts = counts[:L]
    cum = np.cumsum(counts)
    masks = np.empty(int((counts * cum).sum()), np.float32)
    ti = np.empty(int(counts.sum()), np.int64)
    rep = np.empty(int(counts.sum()) + L, np.int32)
    roff = np.empty(L + 1, np.int32)
    rc = lib().lo_tree_drafter_build(_p(flat), _p(off), n, top_k, _p(masks), _p(ti), _p(rep), _p(roff))
    if rc != 0:
        raise RuntimeError(f"lo_tree_drafter_build rc={rc}")
    out_m, out_t, out_r = [], [], []
    mo = to = 0
    for l in range(L):
        out_m.append(masks[mo:mo + counts[l] * cum[l]].reshape(counts[l], cum[l]))
        mo += counts[l] * cum[l]
        out_t.append(ti[to:to + counts[l]])
        to += counts[l]
        out_r.append(rep[roff[l]:roff[l + 1]].tolist())
    return dict(attn_mask=out_m, tree_indices=out_t, repeat_nums=out_r,
                position_ids=[np.zeros(c, np.int64) for c in counts])


def tree_dynamic_finalize(scores, tokens, parents, top_k, total_tokens, sample_token, sort_rows=True):
    """Tail of Model.topK_genrate (cnets_llamagen.py:831-912)."""
    scores = _c(scores, np.float32).reshape(-1)
    tokens = _c(tokens, np.int64).reshape(-1)
    parents = _c(parents, np.int64).reshape(-1)
    N = total_tokens + 1
    draft = np.empty(N, np.int64)
    mask = np.empty((N, N), np.float32)
    pos = np.empty(N, np.int64)
    ret = np.full((N, N), -1, np.int64)
    nl, md = C.c_int32(), C.c_int32()
    rc = lib().lo_tree_dynamic_finalize(_p(scores), _p(tokens), _p(parents), len(scores), top_k, total_tokens,
                                        C.c_int64(int(sample_token)), int(sort_rows), _p(draft), _p(mask), _p(pos),
                                        _p(ret), C.byref(nl), C.byref(md))
    if rc != 0:
        raise RuntimeError(f"lo_tree_dynamic_finalize rc={rc}")
    return draft, ret[:nl.value, :md.value].copy(), mask, pos


def gather_candidates(ss_token, ss_prob, sample_token, tree_indices, retrieve):
    """generate_candidates (ea_model_lumina_mgpt.py:525-554)."""
    ss_token = _c(ss_token, np.int64).reshape(-1)
    prob = None if ss_prob is None else _c(ss_prob, np.float32).reshape(-1)
    tree_indices = _c(tree_indices, np.int64)
    retrieve = _c(retrieve, np.int64)
    N = len(tree_indices)
    P, D = retrieve.shape
    tc = np.empty(N, np.int64)
    cand = np.empty((P, D), np.int64)
    cp = np.empty((P, D), np.float32) if prob is not None else None
    rc = lib().lo_gather_candidates(_p(ss_token), _p(prob), len(ss_token), C.c_int64(int(sample_token)),
                                    _p(tree_indices), N, _p(retrieve), P, D, _p(tc), _p(cand), _p(cp))
    if rc != 0:
        raise RuntimeError(f"lo_gather_candidates rc={rc}")
    return cand, cp, tc


def kv_gather(slab: np.ndarray, retrieve_row, n_sel, prev_len):
    """In-place KV slab update (ea_model_lumina_mgpt.py:741-746; kv_cache.py:38-50).
    slab: [..., S_max, d] contiguous."""
    assert slab.flags.c_contiguous
    S, d = slab.shape[-2], slab.shape[-1]
    outer = slab.size // (S * d)
    rr = _c(retrieve_row, np.int64)
    rc = lib().lo_kv_gather(_p(slab), slab.itemsize, C.c_int64(outer), C.c_int64(S), C.c_int64(d), _p(rr), n_sel,
                            C.c_int64(prev_len))
    if rc != 0:
        raise RuntimeError(f"lo_kv_gather rc={rc}")
    return slab


def hidden_gather(hidden: np.ndarray, retrieve_row, n_sel):
    """hidden[:, retrieve][:, best, :a+1] (ea_model_lumina_mgpt.py:773-777)."""
    hidden = np.ascontiguousarray(hidden)
    B, N, H = hidden.shape
    out = np.empty((B, n_sel, H), hidden.dtype)
    rr = _c(retrieve_row, np.int64)
    rc = lib().lo_hidden_gather(_p(hidden), hidden.itemsize, B, N, H, _p(rr), n_sel, _p(out))
    if rc != 0:
        raise RuntimeError(f"lo_hidden_gather rc={rc}")
    return out


def sample_inverse_cdf(p, u):
    p = _c(p, np.float32)
    return int(lib().lo_sample_inverse_cdf(_p(p), len(p), C.c_double(u)))


def sample_static(probs, idx):
    """sample() with injected multinomial indices (cnets_lumina_mgpt.py:936-955)."""
    probs = _c(probs, np.float32)
    idx = _c(idx, np.int64)
    R, V = probs.shape
    k = idx.shape[1]
    out = np.empty((R, k), np.float32)
    rc = lib().lo_sample_static(_p(probs), R, V, _p(idx), k, _p(out))
    if rc != 0:
        raise RuntimeError(f"lo_sample_static rc={rc}")
    return out


def sample_draws(p, us):
    """Model.sample's k draws WITHOUT replacement from one row's distribution p [V] (cnets_lumina_mgpt.py:936-955: torch.multinomial(probs, k,
    replacement=False)), restated on INJECTED uniforms us [k] (SURVEY 8a RNG contract: the device RNG behind torch.multinomial is not reproducible
    across devices): draw j is the inverse CDF (token-id order, f64 running sum -- lo_sample_inverse_cdf) of p with the j tokens already drawn
    removed.  Successive inverse-CDF draws without replacement are Plackett-Luce distributed, as torch.multinomial's are.  Out of mass (fewer
    positive entries than draws): the lowest ids >= `first positive or 0` not drawn yet is NOT restated here -- the callers' rows have more than k
    positive entries.  Returns (idx [k] int64, conditional probabilities [k] f32 = sample_static's arithmetic)."""
    q = _c(p, np.float32).copy()
    idx = []
    for u in us:
        t = sample_inverse_cdf(q, float(u))
        idx.append(t)
        q[t] = 0.0
    idx = np.asarray(idx, np.int64)
    return idx, sample_static(_c(p, np.float32)[None], idx[None])[0]


def expand_dynamic(logits, scores_in, top_k=10):
    """One EAGLE-2 expansion depth (cnets_llamagen.py:798-820)."""
    logits = _c(logits, np.float32)
    R, V = logits.shape
    si = None if scores_in is None else _c(scores_in, np.float32)
    ti = np.empty((R, top_k), np.int64)
    cu = np.empty((R, top_k), np.float32)
    ci = np.empty(top_k, np.int64)
    so = np.empty(top_k, np.float32)
    rc = lib().lo_expand_dynamic(_p(logits), R, V, top_k, _p(si), _p(ti), _p(cu), _p(ci), _p(so))
    if rc != 0:
        raise RuntimeError(f"lo_expand_dynamic rc={rc}")
    return ti, cu, ci, so


def drafter_fc(ids, hidden_bf16, embed_bf16, w_bf16, bias_bf16=None, embed_scale=1.0):
    """fc(cat(embed(ids), hidden)) (cnets_lumina_mgpt.py:1071,1095-1098); uint16 bf16 bit patterns in, f32 out."""
    ids = _c(ids, np.int64).reshape(-1)
    hidden_bf16 = _c(hidden_bf16, np.uint16)
    M, H = hidden_bf16.shape
    out = np.empty((M, H), np.float32)
    rc = lib().lo_drafter_fc(_p(ids), _p(hidden_bf16), _p(_c(embed_bf16, np.uint16)), _p(_c(w_bf16, np.uint16)),
                             _p(None if bias_bf16 is None else _c(bias_bf16, np.uint16)), M, H,
                             C.c_float(embed_scale), _p(out))
    if rc != 0:
        raise RuntimeError(f"lo_drafter_fc rc={rc}")
    return out


def build_vq_table(codebook):
    """generate_codebook.py:53-65."""
    cb = _c(codebook, np.float32)
    K, Cc = cb.shape
    out = np.empty((K, K - 1), np.uint16)
    rc = lib().lo_build_vq_table(_p(cb), K, Cc, _p(out))
    if rc != 0:
        raise RuntimeError(f"lo_build_vq_table rc={rc}")
    return out


def num_threads() -> int:
    return int(lib().lo_num_threads())


class _LoopArgs(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("n_seq", "n_steps", "pool_steps", "n_threads", "N", "P", "D", "R", "V", "W", "win_lo", "H")]
                + [("prm", _EpParams)]
                + [(n, C.c_void_p) for n in ("tree_indices", "retrieve", "pos1", "row_index", "p_idx", "b_off", "b_idx", "op_off")]
                + [("seq_stride", C.c_int64)]
                + [(n, C.c_void_p) for n in ("ss_token", "ss_prob", "cond", "uncond", "orig_win", "hidden", "nn_table", "uniforms")]
                + [("n_uniforms", C.c_int64), ("u_bonus", C.c_void_p), ("first_token", C.c_void_p), ("cfg_scale", C.c_float)]
                + [(n, C.c_int32) for n in ("w_latent", "h_latent", "newline_id", "eos_id", "top_k")]
                + [("prompt_len", C.c_int64), ("tokens_per_image", C.c_int64), ("slabs", C.c_void_p), ("kv_outer", C.c_int64),
                   ("kv_smax", C.c_int64), ("kv_dim", C.c_int64), ("best", C.c_void_p), ("alen", C.c_void_p), ("token", C.c_void_p),
                   ("rc", C.c_void_p)])


def set_row_threads(n: int) -> int:
    """Rows of one cfg_mask_topk call shared over n OpenMP threads of the calling thread (the CPU baseline's all-core leg)."""
    L = lib()
    L.lo_set_row_threads.restype = C.c_int
    return int(L.lo_set_row_threads(int(n)))


def verify_loop_mt(cfg: EpConfig, tb: dict, op_off, pools: dict, uniforms, u_bonus, first_token, table, n_steps: int, n_threads: int,
                   cfg_scale: float, prompt_len: int, tokens_per_image: int, top_k: int, w_latent=48, h_latent=48, newline_id=8803,
                   eos_id=8196, win_lo=4, slabs=None):
    """lo_verify_loop_mt: the synthetic static-tree verify loop over pools [S, n_seq, ...] in ONE C call on n_threads pthreads
    (bench.py's cpu_baseline, C leg).  pools: ss_token [S,B,R,10] i64, ss_prob f32, cond / uncond [S,B,N,V] uint16 (bf16 bits),
    orig_win [S,B,R,W] f32, hidden [S,B,2,N,H] uint16.  slabs: list of 2*B uint16 arrays [outer, 1?, smax, dim] or None.
    Returns (best, accept_len, token) arrays [n_steps, B]."""
    ss_token = _c(pools["ss_token"], np.int64)
    S, B, R = ss_token.shape[:3]
    cond, uncond = _c(pools["cond"], np.uint16), _c(pools["uncond"], np.uint16)
    N, V = cond.shape[2], cond.shape[3]
    orig = _c(pools["orig_win"], np.float32)
    W = orig.shape[-1]
    hidden = _c(pools["hidden"], np.uint16)
    H = hidden.shape[-1]
    ret = _c(tb["retrieve_indices"], np.int64)
    P, D = ret.shape
    ri = ret.copy()
    ri[ri < 0] += N
    ri = _c(ri, np.int32)
    a = _LoopArgs()
    a.n_seq, a.n_steps, a.pool_steps, a.n_threads = B, n_steps, S, n_threads
    a.N, a.P, a.D, a.R, a.V, a.W, a.win_lo, a.H = N, P, D, R, V, W, win_lo, H
    prm = a.prm
    prm.P, prm.D, prm.V, prm.mode = P, D, V, cfg.mode
    prm.syntax_shortcut, prm.tok_offset = int(cfg.syntax_shortcut), cfg.tok_offset
    prm.img_lo, prm.img_hi, prm.n_syntax = cfg.img_lo, min(cfg.img_hi, 2 ** 31 - 1), len(cfg.syntax)
    for i, sx in enumerate(cfg.syntax):
        prm.syntax[i] = sx
    prm.lantern, prm.k, prm.delta = int(cfg.lantern), cfg.k, float(cfg.delta)
    table = _c(table, np.uint16)
    prm.table_rows, prm.table_cols = table.shape
    prm.top_k, prm.temperature, prm.top_p = cfg.top_k, cfg.temperature, cfg.top_p
    keep = dict(ti=_c(tb["tree_indices"], np.int64), ret=ret, pos1=_c(np.asarray(tb["tree_position_ids"]) + 1, np.int64), ri=ri,
                pi=_c(tb["p_indices"], np.int32), bo=_c(tb["b_off"], np.int32), bi=_c(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1), np.int32),
                oo=_c(op_off, np.int32), sst=ss_token, ssp=_c(pools["ss_prob"], np.float32), cond=cond, unc=uncond, orig=orig, hid=hidden,
                tab=table, uni=_c(uniforms, np.float64), ub=_c(u_bonus, np.float64), ft=_c(first_token, np.int64))
    ptr = lambda x: x.ctypes.data
    a.tree_indices, a.retrieve, a.pos1, a.row_index = ptr(keep["ti"]), ptr(keep["ret"]), ptr(keep["pos1"]), ptr(keep["ri"])
    a.p_idx, a.b_off, a.b_idx, a.op_off = ptr(keep["pi"]), ptr(keep["bo"]), ptr(keep["bi"]), ptr(keep["oo"])
    a.seq_stride = B
    a.ss_token, a.ss_prob, a.cond, a.uncond = ptr(keep["sst"]), ptr(keep["ssp"]), ptr(keep["cond"]), ptr(keep["unc"])
    a.orig_win, a.hidden, a.nn_table, a.uniforms = ptr(keep["orig"]), ptr(keep["hid"]), ptr(keep["tab"]), ptr(keep["uni"])
    assert keep["uni"].shape[0] >= B and keep["ub"].shape[0] >= n_steps and keep["ub"].shape[1] == B and keep["ft"].shape[0] >= B
    a.n_uniforms, a.u_bonus, a.first_token, a.cfg_scale = keep["uni"].shape[1], ptr(keep["ub"]), ptr(keep["ft"]), cfg_scale
    a.w_latent, a.h_latent, a.newline_id, a.eos_id, a.top_k = w_latent, h_latent, newline_id, eos_id, top_k
    a.prompt_len, a.tokens_per_image = prompt_len, tokens_per_image
    slab_ptrs = None
    if slabs is not None:
        assert len(slabs) == 2 * B
        sh = slabs[0].shape
        a.kv_smax, a.kv_dim = sh[-2], sh[-1]
        a.kv_outer = int(np.prod(sh[:-2]))
        slab_ptrs = (C.c_void_p * (2 * B))(*[x.ctypes.data for x in slabs])
        a.slabs = C.cast(slab_ptrs, C.c_void_p).value
    best = np.zeros((n_steps, B), np.int32)
    alen = np.zeros((n_steps, B), np.int32)
    token = np.zeros((n_steps, B), np.int64)
    rcs = np.zeros(n_threads, np.int32)
    a.best, a.alen, a.token, a.rc = ptr(best), ptr(alen), ptr(token), ptr(rcs)
    rc = lib().lo_verify_loop_mt(C.byref(a))
    if rc != 0:
        raise RuntimeError(f"lo_verify_loop_mt rc={rc} (per thread: {rcs.tolist()})")
    return best, alen, token

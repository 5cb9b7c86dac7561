"""Thin torch-tensor wrappers over the C-ABI of liblantern_hip.so.

PyTorch is plumbing here: device memory, streams.  Every function launches HIP kernels on
torch's current stream and returns device tensors; nothing synchronises, nothing falls
back to torch ops for the computation itself.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Tuple, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import EpBuffers, EpNodes, EpParams, EpWindow, check

MODE_DYNAMIC, MODE_STATIC_LUMINA, MODE_STATIC_LG = 0, 1, 2
MODEL_PLAIN, MODEL_LUMINA, MODEL_ANOLE = 0, 1, 2
ROWS_LOGITS, ROWS_PROBS, ROWS_RAW_BF16 = 0, 1, 2          # what a window row holds (include/lantern_hip.h LANTERN_ROWS_*)


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.LanternError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


# ------------------------------------------------------------------------ static trees (host)

def _flatten_choices(tree_choices):
    flat, off = [], [0]
    for c in tree_choices:
        flat.extend(int(x) for x in c)
        off.append(len(flat))
    return np.asarray(flat, np.int32), np.asarray(off, np.int32)


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def tree_static_build(tree_choices, top_k: int = 10) -> dict:
    """O1 (host): numpy buffers of generate_tree_buffers (ea_model_lumina_mgpt.py:140-277)."""
    L = _lib.lib()
    flat, off = _flatten_choices(tree_choices)
    n = len(tree_choices)
    N, P, D, bt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(L.lantern_tree_static_sizes(_np_ptr(flat), _np_ptr(off), n, C.byref(N), C.byref(P), C.byref(D), C.byref(bt)),
          "tree_static_sizes")
    N, P, D, bt = N.value, P.value, D.value, bt.value
    mask = np.empty((N, N), np.float32)
    ti = np.empty(N, np.int64)
    pos = np.empty(N, np.int64)
    ret = np.empty((P, D), np.int64)
    pidx = np.empty((P, D), np.int32)
    boff = np.empty(P * D + 1, np.int32)
    bidx = np.empty(max(bt, 1), np.int32)
    check(L.lantern_tree_static_build(_np_ptr(flat), _np_ptr(off), n, top_k, _np_ptr(mask), _np_ptr(ti), _np_ptr(pos),
                                      _np_ptr(ret), _np_ptr(pidx), _np_ptr(boff), _np_ptr(bidx)), "tree_static_build")
    return dict(tree_attn_mask=mask, tree_indices=ti, tree_position_ids=pos, retrieve_indices=ret, p_indices=pidx,
                b_off=boff, b_idx=bidx[:bt])


def tree_drafter_build(tree_choices, top_k: int = 10) -> dict:
    """O2 (host): drafter-side buffers (drafters/utils_c.py:100-179)."""
    L = _lib.lib()
    flat, off = _flatten_choices(tree_choices)
    n = len(tree_choices)
    nl = C.c_int()
    counts = np.zeros(64, np.int32)
    check(L.lantern_tree_drafter_sizes(_np_ptr(flat), _np_ptr(off), n, C.byref(nl), _np_ptr(counts)), "tree_drafter_sizes")
    Lv = nl.value
    counts = counts[:Lv]
    cum = np.cumsum(counts)
    masks = np.empty(max(int((counts * cum).sum()), 1), np.float32)
    ti = np.empty(max(int(counts.sum()), 1), np.int64)
    rep = np.empty(int(counts.sum()) + Lv + 1, np.int32)
    roff = np.empty(Lv + 1, np.int32)
    check(L.lantern_tree_drafter_build(_np_ptr(flat), _np_ptr(off), n, top_k, _np_ptr(masks), _np_ptr(ti), _np_ptr(rep),
                                       _np_ptr(roff)), "tree_drafter_build")
    out_m, out_t, out_r = [], [], []
    mo = to = 0
    for l in range(Lv):
        out_m.append(masks[mo:mo + counts[l] * cum[l]].reshape(counts[l], cum[l]).copy())
        mo += counts[l] * cum[l]
        out_t.append(ti[to:to + counts[l]].copy())
        to += counts[l]
        out_r.append(rep[roff[l]:roff[l + 1]].tolist())
    return dict(attn_mask=out_m, tree_indices=out_t, repeat_nums=out_r, position_ids=[np.zeros(c, np.int64) for c in counts])


# ----------------------------------------------------------------------------- device ops

def tree_dynamic_finalize(scores, tokens, parents, sample_token, top_k: int, total_tokens: int, sort_rows: bool = True):
    """O4.  scores [B,n] f32, tokens [B,n] i64, parents [B,m] i64, sample_token [B] i64 ->
    (draft_tokens [B,N], mask [B,N,N], pos_ids [B,N], retrieve [B,N,N], n_leaf [B], max_depth [B])."""
    scores = _dev(scores, torch.float32, "scores")
    tokens = _dev(tokens, torch.int64, "tokens")
    parents = _dev(parents, torch.int64, "parents")
    sample_token = _dev(sample_token, torch.int64, "sample_token").reshape(-1)
    B, n = scores.shape
    N = total_tokens + 1
    dev = scores.device
    draft = torch.empty((B, N), dtype=torch.int64, device=dev)
    mask = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    pos = torch.empty((B, N), dtype=torch.int64, device=dev)
    ret = torch.empty((B, N, N), dtype=torch.int64, device=dev)
    nl = torch.empty(B, dtype=torch.int32, device=dev)
    md = torch.empty(B, dtype=torch.int32, device=dev)
    check(_lib.lib().lantern_tree_dynamic_finalize(
        C.c_void_p(scores.data_ptr()), C.c_void_p(tokens.data_ptr()), C.c_void_p(parents.data_ptr()),
        C.c_void_p(sample_token.data_ptr()), B, n, parents.shape[1], top_k, total_tokens, int(sort_rows),
        C.c_void_p(draft.data_ptr()), C.c_void_p(mask.data_ptr()), C.c_void_p(pos.data_ptr()), C.c_void_p(ret.data_ptr()),
        C.c_void_p(nl.data_ptr()), C.c_void_p(md.data_ptr()), _stream()), "tree_dynamic_finalize")
    return draft, mask, pos, ret, nl, md


def expand_dynamic(logits, scores_in, top_k: int = 10):
    """O3.  logits [B,R,V] f32, scores_in [B,R] or None -> (topk_index [B,R,k], cu_scores [B,R,k],
    topk_cs_index [B,k], scores_out [B,k])."""
    logits = _dev(logits, torch.float32, "logits")
    B, R, V = logits.shape
    dev = logits.device
    si = None if scores_in is None else _dev(scores_in, torch.float32, "scores_in")
    ti = torch.empty((B, R, top_k), dtype=torch.int64, device=dev)
    cu = torch.empty((B, R, top_k), dtype=torch.float32, device=dev)
    ci = torch.empty((B, top_k), dtype=torch.int64, device=dev)
    so = torch.empty((B, top_k), dtype=torch.float32, device=dev)
    check(_lib.lib().lantern_expand_dynamic(C.c_void_p(logits.data_ptr()), C.c_void_p(_ptr(si)), B, R, V, top_k,
                                            C.c_void_p(ti.data_ptr()), C.c_void_p(cu.data_ptr()), C.c_void_p(ci.data_ptr()),
                                            C.c_void_p(so.data_ptr()), _stream()), "expand_dynamic")
    return ti, cu, ci, so


def gather_candidates(ss_token, ss_prob, sample_token, tree_indices, retrieve):
    """O6.  ss_token [B,R,10] i64, ss_prob [B,R,10] f32|None, sample_token [B] i64, tree_indices [N],
    retrieve [P,D] -> (cand [B,P,D] i64, cart_prob [B,P,D] f32|None, tree_cand [B,N] i64)."""
    ss_token = _dev(ss_token, torch.int64, "ss_token")
    B = ss_token.shape[0]
    n_flat = ss_token[0].numel()
    prob = None if ss_prob is None else _dev(ss_prob, torch.float32, "ss_prob")
    sample_token = _dev(sample_token, torch.int64, "sample_token").reshape(-1)
    tree_indices = _dev(tree_indices, torch.int64, "tree_indices")
    retrieve = _dev(retrieve, torch.int64, "retrieve")
    N = tree_indices.numel()
    P, D = retrieve.shape
    dev = ss_token.device
    tc = torch.empty((B, N), dtype=torch.int64, device=dev)
    cand = torch.empty((B, P, D), dtype=torch.int64, device=dev)
    cp = torch.empty((B, P, D), dtype=torch.float32, device=dev) if prob is not None else None
    check(_lib.lib().lantern_gather_candidates(
        C.c_void_p(ss_token.data_ptr()), C.c_void_p(_ptr(prob)), C.c_void_p(sample_token.data_ptr()),
        C.c_void_p(tree_indices.data_ptr()), C.c_void_p(retrieve.data_ptr()), B, n_flat, N, P, D,
        C.c_void_p(tc.data_ptr()), C.c_void_p(cand.data_ptr()), C.c_void_p(_ptr(cp)), _stream()), "gather_candidates")
    return cand, cp, tc


def cfg_mask_topk(cond, uncond, cfg: float, model: int = MODEL_PLAIN, pos_ids=None, pos_base: int = 0, w: int = 48,
                  h: int = 48, img_lo: int = 4, img_hi: int = 8196, newline_id: int = 8803, eos_id: int = 8196,
                  top_k: int = 0, out: Optional[torch.Tensor] = None, seq_len: Optional[torch.Tensor] = None,
                  rows_per_seq: int = 0):
    """O7.  cond/uncond [rows,V] bf16 or f32 -> processed f32 [rows,V].  With seq_len ([B] i64 device,
    each sequence's len(input_ids)) pos_ids is the shared [rows_per_seq] tree_position_ids + 1."""
    if not cond.is_cuda:
        raise _lib.LanternError("cfg_mask_topk: expected device tensors")
    assert cond.dtype in (torch.float32, torch.bfloat16) and (uncond is None or uncond.dtype == cond.dtype)
    cond = cond.contiguous()
    uncond = None if uncond is None else uncond.contiguous()
    V = cond.shape[-1]
    rows = cond.numel() // V
    if out is None:
        out = torch.empty(cond.shape, dtype=torch.float32, device=cond.device)
    pos = None if pos_ids is None else _dev(pos_ids, torch.int64, "pos_ids").reshape(-1)
    check(_lib.lib().lantern_cfg_mask_topk(
        C.c_void_p(cond.data_ptr()), C.c_void_p(_ptr(uncond)), 1 if cond.dtype == torch.bfloat16 else 0, rows, V,
        C.c_float(cfg), model, C.c_void_p(_ptr(pos)), C.c_int64(pos_base), w, h, img_lo, img_hi, newline_id, eos_id, top_k,
        C.c_void_p(_ptr(seq_len)), rows_per_seq, C.c_void_p(out.data_ptr()), _stream()), "cfg_mask_topk")
    return out


@dataclass
class NodeTables:
    """Node view of a verify tree for the node-parallel evaluate_posterior (lantern_tree_node_tables)."""
    tables: torch.Tensor      # [ints] i32 on the device
    host: np.ndarray
    n_nodes: int
    n_internal: int
    n_children: int
    max_children: int
    prefix_siblings: int = 1

    def struct(self, workspace: int, workspace_bytes: int, leaf_workgroups: int = -1) -> EpNodes:
        en = EpNodes()
        en.tables = self.tables.data_ptr()
        en.tables_host = self.host.ctypes.data
        en.n_nodes, en.n_internal, en.n_children, en.max_children = self.n_nodes, self.n_internal, self.n_children, self.max_children
        en.prefix_siblings, en.leaf_workgroups = self.prefix_siblings, leaf_workgroups
        en.workspace, en.workspace_bytes = workspace, workspace_bytes
        return en


def tree_node_tables(retrieve, N: int, p_idx=None, b_off=None, op_off=None, device=None, b_idx=None) -> NodeTables:
    """Host: retrieve [P,D] (-1 pad) (+ p_idx / b_off / op_off of a static tree) -> the packed per-node tables, uploaded."""
    L = _lib.lib()
    ret = np.ascontiguousarray(np.asarray(retrieve.cpu() if torch.is_tensor(retrieve) else retrieve), np.int64)
    P, D = ret.shape
    n = L.lantern_tree_node_tables_size(N, P, D)
    out = np.zeros(n, np.int32)
    h = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a.cpu() if torch.is_tensor(a) else a), np.int32)
    pi, bo, oo, bi = h(p_idx), h(b_off), h(op_off), h(b_idx)
    if bi is not None and bi.size == 0:
        bi = None
    nul = lambda a: None if a is None else _np_ptr(a)
    check(L.lantern_tree_node_tables(_np_ptr(ret), nul(pi), nul(bo), nul(bi), nul(oo), N, P, D, _np_ptr(out), n), "tree_node_tables")
    dev = torch.from_numpy(out).to(device) if device is not None else None
    return NodeTables(dev, out, int(out[0]), int(out[1]), int(out[2]), int(out[3]), int(out[6]))


@dataclass
class EpConfig:
    """Per-model switches of evaluate_posterior."""
    mode: int = MODE_DYNAMIC
    syntax_shortcut: bool = False
    tok_offset: int = 0
    img_lo: int = 0
    img_hi: int = 2 ** 31 - 1
    syntax: Sequence[int] = ()
    lantern: bool = False
    k: int = 1000
    delta: float = 0.1
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = 0

    @staticmethod
    def lumina(static: bool, **kw) -> "EpConfig":
        # ea_model_lumina_mgpt.py:322-324
        return EpConfig(mode=MODE_STATIC_LUMINA if static else MODE_DYNAMIC, syntax_shortcut=True, tok_offset=4,
                        img_lo=4, img_hi=8196, syntax=(8196, 8197, 8803, 8828), **kw)

    @staticmethod
    def llamagen(static: bool, **kw) -> "EpConfig":
        return EpConfig(mode=MODE_STATIC_LG if static else MODE_DYNAMIC, **kw)

    @staticmethod
    def anole(static: bool, **kw) -> "EpConfig":
        # ea_model_anole.py:142-146
        return EpConfig(mode=MODE_STATIC_LG if static else MODE_DYNAMIC, tok_offset=4, img_lo=4, img_hi=8196, **kw)


@dataclass
class StaticAux:
    cart_prob: torch.Tensor   # [B,P,D] f32
    orig_prob: torch.Tensor   # [B,R,V] f32
    op_off: torch.Tensor      # [D-1] i32
    p_idx: torch.Tensor       # [P,D] i32
    b_off: torch.Tensor       # [P*D+1] i32
    b_idx: torch.Tensor       # [nb] i32
    tree_cand: torch.Tensor   # [B,N] i64


def evaluate_posterior(cfg: EpConfig, logits, row_index, cand, uniforms, table=None, aux: Optional[StaticAux] = None,
                       n_paths=None, n_depth=None, cursor=None, out=None):
    """O8.  logits [B,rows,V] f32; row_index [P,D] or [B,P,D] i32; cand [B,P,D] i64; uniforms [B,n] f64;
    table [K,K-1] u16 (torch.uint16 or int16 view).  Returns (best [B] i32, accept_len [B] i32,
    sample_p [B,V] f32, counters [B,6] i32) -- all device tensors, no sync."""
    logits = _dev(logits, torch.float32, "logits")
    cand = _dev(cand, torch.int64, "cand")
    uniforms = _dev(uniforms, torch.float64, "uniforms")
    row_index = _dev(row_index, torch.int32, "row_index")
    B, P, D = cand.shape
    V = logits.shape[-1]
    rows = logits.shape[-2]
    assert logits.shape[0] == B
    dev = logits.device
    prm = EpParams()
    prm.B, prm.P, prm.D, prm.V, prm.rows_per_seq = B, P, D, V, rows
    prm.mode, prm.syntax_shortcut, prm.tok_offset = cfg.mode, int(cfg.syntax_shortcut), cfg.tok_offset
    prm.img_lo, prm.img_hi = cfg.img_lo, min(cfg.img_hi, 2 ** 31 - 1)
    prm.n_syntax = len(cfg.syntax)
    for i, s in enumerate(cfg.syntax):
        prm.syntax[i] = int(s)
    prm.lantern, prm.k, prm.delta = int(cfg.lantern), int(cfg.k), float(cfg.delta)
    prm.top_k, prm.temperature, prm.top_p = int(cfg.top_k), float(cfg.temperature), float(cfg.top_p)
    prm.n_uniforms = uniforms.shape[1]
    prm.row_index_per_seq = int(row_index.dim() == 3)
    buf = EpBuffers()
    if table is not None:
        if not table.is_cuda:
            raise _lib.LanternError("evaluate_posterior: table must be a device tensor")
        assert table.element_size() == 2
        table = table.contiguous()
        prm.table_rows, prm.table_cols = table.shape
        buf.nn_table = table.data_ptr()
    keep = []
    if aux is not None:
        a = [_dev(aux.cart_prob, torch.float32, "cart_prob"), _dev(aux.orig_prob, torch.float32, "orig_prob"),
             _dev(aux.op_off, torch.int32, "op_off"), _dev(aux.p_idx, torch.int32, "p_idx"),
             _dev(aux.b_off, torch.int32, "b_off"), _dev(aux.b_idx, torch.int32, "b_idx"),
             _dev(aux.tree_cand, torch.int64, "tree_cand")]
        keep.extend(a)
        buf.cart_prob, buf.orig_prob, buf.op_off, buf.p_idx, buf.b_off, buf.b_idx, buf.tree_cand = [x.data_ptr() for x in a]
        prm.R = a[1].shape[1]
        prm.N = a[6].shape[1]
        ws = torch.empty((B, V), dtype=torch.float32, device=dev)
        keep.append(ws)
        buf.workspace = ws.data_ptr()
    if out is None:
        best = torch.empty(B, dtype=torch.int32, device=dev)
        alen = torch.empty(B, dtype=torch.int32, device=dev)
        sample_p = torch.empty((B, V), dtype=torch.float32, device=dev)
        counters = torch.empty((B, 6), dtype=torch.int32, device=dev)
    else:
        best, alen, sample_p, counters = out
    buf.logits, buf.row_index, buf.cand, buf.uniforms = logits.data_ptr(), row_index.data_ptr(), cand.data_ptr(), uniforms.data_ptr()
    if n_paths is not None:
        n_paths = _dev(n_paths, torch.int32, "n_paths")
        buf.n_paths = n_paths.data_ptr()
    if n_depth is not None:
        n_depth = _dev(n_depth, torch.int32, "n_depth")
        buf.n_depth = n_depth.data_ptr()
    if cursor is not None:
        assert cursor.dtype == torch.int32 and cursor.is_cuda
        buf.cursor = cursor.data_ptr()
    buf.best, buf.accept_len, buf.sample_p, buf.counters = best.data_ptr(), alen.data_ptr(), sample_p.data_ptr(), counters.data_ptr()
    check(_lib.lib().lantern_evaluate_posterior(C.byref(prm), C.byref(buf), _stream()), "evaluate_posterior")
    return best, alen, sample_p, counters


_STATUS = {1: "candidate token outside [0,V)", 2: "uniform stream exhausted", 3: "token outside the neighbour table",
           4: "image syntax token rejected (reference assert, ea_model_lumina_mgpt.py:694)", 5: "no path matches the accepted prefix",
           6: "residual distribution vanished (`gtp.sum()==0 -> ones` is uniform over all V): only the dense kernel set represents it",
           7: "static tree beyond the windowed kernel's staging limits (more than 16 earlier siblings of a node / 1024 b_idx entries): "
              "use the dense kernel set"}


def raise_on_status(counters: torch.Tensor):
    """Host-side check of the per-sequence status word (synchronises)."""
    st = counters[:, 5].cpu()
    bad = torch.nonzero(st).reshape(-1)
    if len(bad):
        i = int(bad[0])
        raise _lib.LanternError(f"evaluate_posterior: sequence {i}: {_STATUS.get(int(st[i]), st[i])}")


def kv_gather(slabs: Sequence[torch.Tensor], slab_seq, slab_prev, retrieve, best, accept_len, slab_ptrs=None):
    """O9.  slabs: tensors [..., S_max, d] of identical shape/dtype; slab_seq [n] i32 (sequence of each slab),
    slab_prev [n] i64 device.  retrieve [P,D] or [B,P,D] i64; best/accept_len [B] i32 device.
    Returns new_len [n] i64 device.  In place."""
    s0 = slabs[0]
    S, d = s0.shape[-2], s0.shape[-1]
    outer = s0.numel() // (S * d)
    dev = s0.device
    for s in slabs:
        assert s.is_cuda and s.is_contiguous() and s.shape == s0.shape and s.dtype == s0.dtype
    if slab_ptrs is None:
        slab_ptrs = torch.tensor([s.data_ptr() for s in slabs], dtype=torch.int64, device=dev)
    slab_seq = _dev(slab_seq, torch.int32, "slab_seq")
    slab_prev = _dev(slab_prev, torch.int64, "slab_prev")
    retrieve = _dev(retrieve, torch.int64, "retrieve")
    P, D = retrieve.shape[-2:]
    new_len = torch.empty(len(slabs), dtype=torch.int64, device=dev)
    check(_lib.lib().lantern_kv_gather(
        C.c_void_p(slab_ptrs.data_ptr()), C.c_void_p(slab_seq.data_ptr()), C.c_void_p(slab_prev.data_ptr()), len(slabs),
        s0.element_size(), C.c_int64(outer), C.c_int64(S), C.c_int64(d), C.c_void_p(retrieve.data_ptr()),
        int(retrieve.dim() == 3), P, D, C.c_void_p(best.data_ptr()), C.c_void_p(accept_len.data_ptr()),
        C.c_void_p(new_len.data_ptr()), _stream()), "kv_gather")
    return new_len


def update_inference_inputs(slabs: Sequence[torch.Tensor], slab_seq, slab_prev, retrieve, best, accept_len, hidden, cand, slab_ptrs=None):
    """O9 + O10 in one launch (the reference's update_inference_inputs): KV rows of every slab move in place, accepted hidden
    rows [B,G,D,H] and tokens [B,D] are gathered.  Returns (new_len [n] i64, out_hidden, accepted_tokens)."""
    s0 = slabs[0]
    S, d = s0.shape[-2], s0.shape[-1]
    outer = s0.numel() // (S * d)
    dev = s0.device
    for s in slabs:
        assert s.is_cuda and s.is_contiguous() and s.shape == s0.shape and s.dtype == s0.dtype
    if slab_ptrs is None:
        slab_ptrs = torch.tensor([s.data_ptr() for s in slabs], dtype=torch.int64, device=dev)
    slab_seq = _dev(slab_seq, torch.int32, "slab_seq")
    slab_prev = _dev(slab_prev, torch.int64, "slab_prev")
    retrieve = _dev(retrieve, torch.int64, "retrieve")
    cand = _dev(cand, torch.int64, "cand")
    hidden = hidden.contiguous()
    P, D = retrieve.shape[-2:]
    B, G, N, H = hidden.shape
    new_len = torch.empty(len(slabs), dtype=torch.int64, device=dev)
    out_h = torch.empty((B, G, D, H), dtype=hidden.dtype, device=dev)
    acc = torch.empty((B, D), dtype=torch.int64, device=dev)
    check(_lib.lib().lantern_update_inference_inputs(
        C.c_void_p(slab_ptrs.data_ptr()), C.c_void_p(slab_seq.data_ptr()), C.c_void_p(slab_prev.data_ptr()), len(slabs),
        s0.element_size(), C.c_int64(outer), C.c_int64(S), C.c_int64(d), C.c_void_p(retrieve.data_ptr()),
        int(retrieve.dim() == 3), P, D, C.c_void_p(best.data_ptr()), C.c_void_p(accept_len.data_ptr()),
        C.c_void_p(new_len.data_ptr()), C.c_void_p(hidden.data_ptr()), hidden.element_size(), B, G, N, H, C.c_void_p(cand.data_ptr()),
        C.c_void_p(out_h.data_ptr()), C.c_void_p(acc.data_ptr()), _stream()), "update_inference_inputs")
    return new_len, out_h, acc


def accept_gather(hidden, retrieve, cand, best, accept_len, sample_p=None, u=None):
    """O10.  hidden [B,G,N,H]; retrieve [P,D]|[B,P,D]; cand [B,P,D]; sample_p [B,V]; u [B] f64 or None (greedy).
    Returns (out_hidden [B,G,D,H], accepted_tokens [B,D], token [B])."""
    retrieve = _dev(retrieve, torch.int64, "retrieve")
    P, D = retrieve.shape[-2:]
    B = best.shape[0]
    dev = best.device
    out_h = None
    G = N = H = 0
    eb = 0
    if hidden is not None:
        hidden = hidden.contiguous()
        _, G, N, H = hidden.shape
        eb = hidden.element_size()
        out_h = torch.empty((B, G, D, H), dtype=hidden.dtype, device=dev)
    acc = None
    if cand is not None:
        cand = _dev(cand, torch.int64, "cand")
        acc = torch.empty((B, D), dtype=torch.int64, device=dev)
    token = None
    V = 0
    if sample_p is not None:
        sample_p = _dev(sample_p, torch.float32, "sample_p")
        V = sample_p.shape[-1]
        token = torch.empty(B, dtype=torch.int64, device=dev)
        if u is not None:
            u = _dev(u, torch.float64, "u")
    check(_lib.lib().lantern_accept_gather(
        C.c_void_p(_ptr(hidden)), eb, B, G, N, H, C.c_void_p(retrieve.data_ptr()), int(retrieve.dim() == 3), P, D,
        C.c_void_p(_ptr(cand)), C.c_void_p(best.data_ptr()), C.c_void_p(accept_len.data_ptr()), C.c_void_p(_ptr(sample_p)), V,
        C.c_void_p(_ptr(u)), C.c_void_p(_ptr(out_h)), C.c_void_p(_ptr(acc)), C.c_void_p(_ptr(token)), _stream()),
        "accept_gather")
    return out_h, acc, token


def sample_static(probs, idx):
    """O5.  probs [R,V] f32, idx [R,k] i64 -> conditional probabilities [R,k]."""
    probs = _dev(probs, torch.float32, "probs")
    idx = _dev(idx, torch.int64, "idx")
    R, V = probs.shape
    k = idx.shape[1]
    out = torch.empty((R, k), dtype=torch.float32, device=probs.device)
    check(_lib.lib().lantern_sample_static(C.c_void_p(probs.data_ptr()), C.c_void_p(idx.data_ptr()), R, V, k,
                                           C.c_void_p(out.data_ptr()), _stream()), "sample_static")
    return out


def evaluate_posterior_greedy(logits, row_index, cand, lantern=False, k=1000, delta=0.1, tok_offset=0, table=None, win_lo=0,
                              win_len=None):
    """a9 greedy/TVD branch.  logits [B,rows,V] f32; row_index [P,D]|[B,P,D] i32; cand [B,P,D] i64; ids outside
    [win_lo, win_lo+win_len) are not read (LlamaGen: whole vocabulary; Anole: image range).
    Returns (best [B] i32, accept_len [B] i32, out_row [B,V] f32 = logits[best, accept_len])."""
    logits = _dev(logits, torch.float32, "logits")
    cand = _dev(cand, torch.int64, "cand")
    row_index = _dev(row_index, torch.int32, "row_index")
    B, P, D = cand.shape
    V, rows = logits.shape[-1], logits.shape[-2]
    if win_len is None:
        win_len = V - win_lo
    dev = logits.device
    best = torch.empty(B, dtype=torch.int32, device=dev)
    alen = torch.empty(B, dtype=torch.int32, device=dev)
    out = torch.empty((B, V), dtype=torch.float32, device=dev)
    scratch = torch.empty(B * P * max(D - 1, 1), dtype=torch.int32, device=dev)
    tr = tc = 0
    if table is not None:
        table = table.contiguous()
        tr, tc = table.shape
    check(_lib.lib().lantern_evaluate_posterior_greedy(
        C.c_void_p(logits.data_ptr()), C.c_void_p(row_index.data_ptr()), C.c_void_p(cand.data_ptr()), B, P, D, V, rows,
        int(row_index.dim() == 3), int(bool(lantern)), int(k), C.c_double(float(delta)), int(tok_offset), C.c_void_p(_ptr(table)),
        tr, tc, int(win_lo), int(win_len), C.c_void_p(scratch.data_ptr()), C.c_void_p(best.data_ptr()), C.c_void_p(alen.data_ptr()),
        C.c_void_p(out.data_ptr()), _stream()), "evaluate_posterior_greedy")
    return best, alen, out


# ------------------------------------------------------------------------- windowed (v2) path

def cfg_mask_topk_window(cond, uncond, cfg: float, win_lo: int, win_len: int, model: int = MODEL_PLAIN, pos_ids=None,
                         pos_base: int = 0, w: int = 48, h: int = 48, img_lo: int = 4, img_hi: int = 8196, newline_id: int = 8803,
                         eos_id: int = 8196, top_k: int = 0, seq_len=None, rows_per_seq: int = 0, out=None, row_hot=None,
                         probs: bool = False, temperature: float = 1.0, top_p: float = 1.0):
    """O7 windowed: (out_win [rows,win_len] f32, row_hot [rows] i32).  probs=True: rows leave as softmax probabilities
    (temperature -> top-k -> softmax applied here, to every row); pass rows_probs=True to evaluate_posterior_window."""
    if not cond.is_cuda:
        raise _lib.LanternError("cfg_mask_topk_window: expected device tensors")
    assert cond.dtype in (torch.float32, torch.bfloat16) and (uncond is None or uncond.dtype == cond.dtype)
    cond = cond.contiguous()
    uncond = None if uncond is None else uncond.contiguous()
    V = cond.shape[-1]
    rows = cond.numel() // V
    if out is None:
        out = torch.empty((*cond.shape[:-1], win_len), dtype=torch.float32, device=cond.device)
    if row_hot is None:
        row_hot = torch.empty(cond.shape[:-1], dtype=torch.int32, device=cond.device)
    pos = None if pos_ids is None else _dev(pos_ids, torch.int64, "pos_ids").reshape(-1)
    check(_lib.lib().lantern_cfg_mask_topk_window(
        C.c_void_p(cond.data_ptr()), C.c_void_p(_ptr(uncond)), 1 if cond.dtype == torch.bfloat16 else 0, rows, V, C.c_float(cfg),
        model, C.c_void_p(_ptr(pos)), C.c_int64(pos_base), w, h, img_lo, img_hi, newline_id, eos_id, top_k,
        C.c_void_p(_ptr(seq_len)), rows_per_seq, win_lo, win_len, C.c_void_p(out.data_ptr()), C.c_void_p(row_hot.data_ptr()),
        ROWS_PROBS if probs else ROWS_LOGITS, C.c_float(temperature), C.c_float(top_p), _stream()), "cfg_mask_topk_window")
    return out, row_hot


def evaluate_posterior_window(cfg: EpConfig, V: int, win_logits, win_lo: int, row_index, cand, uniforms, row_hot=None, table=None,
                              aux: Optional[StaticAux] = None, orig_windowed: bool = False, n_paths=None, n_depth=None,
                              cursor=None, u_bonus=None, want_dense: bool = False, want_window: bool = True, rows_probs: bool = False,
                              nodes: Optional[NodeTables] = None, leaf_workgroups: int = -1):
    """O8 windowed.  win_logits [B,rows,W] f32 (probabilities when rows_probs: cfg must then carry top_k=0, temperature=1).  aux.orig_prob is the dense [B,R,V] pool (orig_windowed=False) or a
    windowed [B,R,W] pool.  Returns dict(best, accept_len, counters, sample_win, out_tok, out_mass, token, sample_p)."""
    win_logits = _dev(win_logits, torch.float32, "win_logits")
    cand = _dev(cand, torch.int64, "cand")
    uniforms = _dev(uniforms, torch.float64, "uniforms")
    row_index = _dev(row_index, torch.int32, "row_index")
    B, P, D = cand.shape
    rows, W = win_logits.shape[-2], win_logits.shape[-1]
    dev = win_logits.device
    prm = EpParams()
    prm.B, prm.P, prm.D, prm.V, prm.rows_per_seq = B, P, D, V, rows
    prm.mode, prm.syntax_shortcut, prm.tok_offset = cfg.mode, int(cfg.syntax_shortcut), cfg.tok_offset
    prm.img_lo, prm.img_hi = cfg.img_lo, min(cfg.img_hi, 2 ** 31 - 1)
    prm.n_syntax = len(cfg.syntax)
    for i, s in enumerate(cfg.syntax):
        prm.syntax[i] = int(s)
    prm.lantern, prm.k, prm.delta = int(cfg.lantern), int(cfg.k), float(cfg.delta)
    prm.top_k, prm.temperature, prm.top_p = int(cfg.top_k), float(cfg.temperature), float(cfg.top_p)
    prm.n_uniforms = uniforms.shape[1]
    prm.row_index_per_seq = int(row_index.dim() == 3)
    buf, win = EpBuffers(), EpWindow()
    keep = []
    if table is not None:
        table = table.contiguous()
        prm.table_rows, prm.table_cols = table.shape
        buf.nn_table = table.data_ptr()
    if aux is not None:
        a = [_dev(aux.cart_prob, torch.float32, "cart_prob"), _dev(aux.orig_prob, torch.float32, "orig_prob"),
             _dev(aux.op_off, torch.int32, "op_off"), _dev(aux.p_idx, torch.int32, "p_idx"), _dev(aux.b_off, torch.int32, "b_off"),
             _dev(aux.b_idx, torch.int32, "b_idx"), _dev(aux.tree_cand, torch.int64, "tree_cand")]
        keep.extend(a)
        buf.cart_prob, buf.orig_prob, buf.op_off, buf.p_idx, buf.b_off, buf.b_idx, buf.tree_cand = [x.data_ptr() for x in a]
        prm.R, prm.N = a[1].shape[1], a[6].shape[1]
        win.orig_prob_stride = a[1].shape[-1]
        win.orig_prob_offset = 0 if orig_windowed else win_lo
    out = dict(best=torch.empty(B, dtype=torch.int32, device=dev), accept_len=torch.empty(B, dtype=torch.int32, device=dev),
               counters=torch.empty((B, 6), dtype=torch.int32, device=dev), out_tok=torch.empty(B, dtype=torch.int32, device=dev),
               out_mass=torch.empty(B, dtype=torch.float32, device=dev), sample_win=None, token=None, sample_p=None)
    buf.logits, buf.row_index, buf.cand, buf.uniforms = win_logits.data_ptr(), row_index.data_ptr(), cand.data_ptr(), uniforms.data_ptr()
    if n_paths is not None:
        n_paths = _dev(n_paths, torch.int32, "n_paths"); buf.n_paths = n_paths.data_ptr()
    if n_depth is not None:
        n_depth = _dev(n_depth, torch.int32, "n_depth"); buf.n_depth = n_depth.data_ptr()
    if cursor is not None:
        buf.cursor = cursor.data_ptr()
    buf.best, buf.accept_len, buf.counters = out["best"].data_ptr(), out["accept_len"].data_ptr(), out["counters"].data_ptr()
    if want_dense:
        out["sample_p"] = torch.empty((B, V), dtype=torch.float32, device=dev)
        buf.sample_p = out["sample_p"].data_ptr()
    win.win_lo, win.win_len = win_lo, W
    win.rows_kind = ROWS_PROBS if rows_probs else ROWS_LOGITS
    if row_hot is not None:
        row_hot = _dev(row_hot, torch.int32, "row_hot"); win.row_hot = row_hot.data_ptr()
    if want_window:
        out["sample_win"] = torch.empty((B, W), dtype=torch.float32, device=dev)
        win.sample_win = out["sample_win"].data_ptr()
    win.out_tok, win.out_mass = out["out_tok"].data_ptr(), out["out_mass"].data_ptr()
    if u_bonus is not None:
        u_bonus = _dev(u_bonus, torch.float64, "u_bonus")
        out["token"] = torch.empty(B, dtype=torch.int64, device=dev)
        win.u_bonus, win.token = u_bonus.data_ptr(), out["token"].data_ptr()
    if nodes is not None:      # node-parallel form: one workgroup per internal tree node, then the walk
        nbytes = _lib.lib().lantern_evaluate_posterior_nodes_workspace(C.byref(prm), C.byref(win), nodes.n_internal, int(want_dense or want_window))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        en = nodes.struct(ws.data_ptr(), int(nbytes), leaf_workgroups)
        check(_lib.lib().lantern_evaluate_posterior_nodes(C.byref(prm), C.byref(buf), C.byref(win), C.byref(en), _stream()), "evaluate_posterior_nodes")
        out["_workspace"] = ws
        return out
    check(_lib.lib().lantern_evaluate_posterior_window(C.byref(prm), C.byref(buf), C.byref(win), _stream()), "evaluate_posterior_window")
    return out


def window_to_dense(sample_win, out_tok, out_mass, V: int, win_lo: int):
    sample_win = _dev(sample_win, torch.float32, "sample_win")
    B, W = sample_win.shape
    dense = torch.empty((B, V), dtype=torch.float32, device=sample_win.device)
    check(_lib.lib().lantern_window_to_dense(C.c_void_p(sample_win.data_ptr()), C.c_void_p(_ptr(out_tok)), C.c_void_p(_ptr(out_mass)),
                                             B, V, win_lo, W, C.c_void_p(dense.data_ptr()), _stream()), "window_to_dense")
    return dense


def linear_rows(A, weight, row_lo: int, n_rows: int, bias=None, out=None, out_col0: Optional[int] = None):
    """8f-2: A [M,K] bf16 @ weight[row_lo:row_lo+n_rows].T -> bf16 columns [out_col0, out_col0+n_rows) of `out` [M, stride]
    (a fresh [M, n_rows] tensor, out_col0 = 0, when `out` is None).  The drafter's lm_head on the image-token rows only."""
    for t, n in ((A, "A"), (weight, "weight")):
        if not t.is_cuda or t.dtype != torch.bfloat16:
            raise _lib.LanternError(f"linear_rows: {n} must be a bf16 device tensor")
    A = A.contiguous()
    weight = weight.contiguous()
    M, K = A.shape
    if out is None:
        out = torch.empty((M, n_rows), dtype=torch.bfloat16, device=A.device)
        out_col0 = 0
    elif out_col0 is None:
        out_col0 = row_lo
    assert out.is_contiguous() and out.dtype == torch.bfloat16 and out.shape[0] == M
    b = None if bias is None else bias.contiguous()
    for s0 in range(0, M, 128):
        m = min(128, M - s0)
        check(_lib.lib().lantern_linear_rows(C.c_void_p(A[s0:].data_ptr()), C.c_void_p(weight.data_ptr()), C.c_void_p(_ptr(b)), m, K, row_lo,
                                             n_rows, C.c_void_p(out[s0:].data_ptr()), out.shape[1], out_col0, _stream()), "linear_rows")
    return out


EPI_RESIDUAL, EPI_SILU_MUL = 1, 2          # include/lantern_hip.h LANTERN_EPI_*


def linear_rows_epilogue(A, weight, epilogue: int, n_rows: Optional[int] = None, bias=None, residual=None, pair_rows: int = 0):
    """8f-2, decoder layer: A [M <= 32, K] bf16 @ weight[:n_rows].T with the layer's tail in the epilogue.  EPI_RESIDUAL: + residual [M, n_rows]
    (torch's two bf16 roundings); EPI_SILU_MUL: weight = cat(gate, up) rows, pair_rows = distance gate -> up row: silu(gate) * up, [M, n_rows]."""
    for t, n in ((A, "A"), (weight, "weight")):
        if not t.is_cuda or t.dtype != torch.bfloat16:
            raise _lib.LanternError(f"linear_rows_epilogue: {n} must be a bf16 device tensor")
    A, weight = A.contiguous(), weight.contiguous()
    M, K = A.shape
    n_rows = (weight.shape[0] if epilogue == EPI_RESIDUAL else pair_rows) if n_rows is None else n_rows
    out = torch.empty((M, n_rows), dtype=torch.bfloat16, device=A.device)
    r = None if residual is None else residual.contiguous()
    b = None if bias is None else bias.contiguous()
    check(_lib.lib().lantern_linear_rows_epilogue(C.c_void_p(A.data_ptr()), C.c_void_p(weight.data_ptr()), C.c_void_p(_ptr(b)), M, K, 0, n_rows,
                                                  C.c_void_p(out.data_ptr()), n_rows, 0, epilogue, C.c_void_p(_ptr(r)), 0 if r is None else r.shape[1],
                                                  pair_rows, _stream()), "linear_rows_epilogue")
    return out


def linear_rows_splitk(A, weight, bias=None, residual=None, ksplit: int = 0):
    """A [M <= 32, K] bf16 @ weight.T (+ bias) (+ residual) with K split over `ksplit` workgroups per column tile (0: enough to fill 256 CUs)."""
    A, weight = A.contiguous(), weight.contiguous()
    M, K = A.shape
    N = weight.shape[0]
    if ksplit <= 0:
        ksplit = max(1, min(8, -(-256 // (-(-N // 32)))))
    ws = torch.empty((ksplit, M, N), dtype=torch.float32, device=A.device)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    r = None if residual is None else residual.contiguous()
    b = None if bias is None else bias.contiguous()
    check(_lib.lib().lantern_linear_rows_splitk(C.c_void_p(A.data_ptr()), C.c_void_p(weight.data_ptr()), C.c_void_p(_ptr(b)), M, K, N, C.c_void_p(out.data_ptr()), N,
                                                C.c_void_p(_ptr(r)), 0 if r is None else r.shape[1], ksplit, C.c_void_p(ws.data_ptr()), _stream()), "linear_rows_splitk")
    return out


_SK_WS = {}


def _sk_workspace(device):
    """The stream-K workspace of `device` and the current stream (partial tiles + tile counters; zero-filled once, the kernels leave the counters
    zeroed; one launch at a time per workspace -- launches on one stream are ordered)."""
    need = int(_lib.lib().lantern_linear_rows_streamk_workspace(65536))
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _SK_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _SK_WS[key] = torch.zeros(need, dtype=torch.uint8, device=device)
    return ws


class PackedLinearWeight:
    """A [N, K] bf16 nn.Linear weight re-laid out for lantern_linear_rows_streamk (lantern_pack_linear_weight): `data` is the brick stream,
    `n_rows` / `K` / `pair_rows` describe the original."""

    def __init__(self, data, n_rows, K, pair_rows):
        self.data, self.n_rows, self.K, self.pair_rows = data, int(n_rows), int(K), int(pair_rows)


def pack_linear_weight(weight, pair_rows: int = 0) -> PackedLinearWeight:
    """weight [N, K] bf16 (K % 64 == 0) -> PackedLinearWeight; pair_rows > 0: weight = cat(gate, up) rows, N = 2 * pair_rows."""
    if not weight.is_cuda or weight.dtype != torch.bfloat16 or weight.dim() != 2 or weight.shape[1] % 64:
        raise _lib.LanternError("pack_linear_weight: a [N, K] bf16 device tensor with K % 64 == 0")
    weight = weight.contiguous()
    N, K = weight.shape
    n_rows = pair_rows if pair_rows > 0 else N
    L = _lib.lib()
    nbytes = int(L.lantern_pack_linear_weight_bytes(n_rows, K, pair_rows))
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
    check(L.lantern_pack_linear_weight(C.c_void_p(weight.data_ptr()), n_rows, K, int(pair_rows), C.c_void_p(out.data_ptr()), _stream()), "pack_linear_weight")
    return PackedLinearWeight(out, n_rows, K, pair_rows)


def linear_rows_streamk(A, weight, epilogue: int = 0, n_rows: Optional[int] = None, bias=None, residual=None, pair_rows: int = 0, row_lo: int = 0):
    """A [M <= 32, K] bf16 @ weight[row_lo : row_lo + n_rows].T in stream-K form (one launch, every workgroup streams an equal share of the
    weight): epilogue 0 (+ bias), EPI_RESIDUAL (+ residual [M, n_rows], torch's two roundings), EPI_SILU_MUL (weight = cat(gate, up) rows,
    pair_rows = distance gate -> up row: silu(gate) * up).  `weight`: the [N, K] tensor, or its PackedLinearWeight (the fast form).  The
    workspace (partial tiles + tile counters, zeroed once) is kept per device and current stream."""
    packed = isinstance(weight, PackedLinearWeight)
    if not A.is_cuda or A.dtype != torch.bfloat16:
        raise _lib.LanternError("linear_rows_streamk: A must be a bf16 device tensor")
    A = A.contiguous()
    M, K = A.shape
    if packed:
        if K != weight.K or row_lo != 0 or (epilogue == EPI_SILU_MUL) != (weight.pair_rows > 0):
            raise _lib.LanternError("linear_rows_streamk: the packed weight does not match this call (K, row_lo = 0, gate / up pairing)")
        n_rows, pair_rows, wt = weight.n_rows, weight.pair_rows, weight.data
    else:
        if not weight.is_cuda or weight.dtype != torch.bfloat16:
            raise _lib.LanternError("linear_rows_streamk: weight must be a bf16 device tensor")
        wt = weight.contiguous()
        if n_rows is None:
            n_rows = pair_rows if epilogue == EPI_SILU_MUL else wt.shape[0] - row_lo
    L = _lib.lib()
    ws = _sk_workspace(A.device)
    out = torch.empty((M, n_rows), dtype=torch.bfloat16, device=A.device)
    r = None if residual is None else residual.contiguous()
    b = None if bias is None else bias.contiguous()
    check(L.lantern_linear_rows_streamk(C.c_void_p(A.data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(_ptr(b)), M, K, row_lo, n_rows,
                                        C.c_void_p(out.data_ptr()), n_rows, 0, int(epilogue), C.c_void_p(_ptr(r)), 0 if r is None else r.shape[1], int(pair_rows),
                                        int(packed), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), _stream()), "linear_rows_streamk")
    return out


def linear_rows_packed(A, weight: "PackedLinearWeight", epilogue: int = 0, bias=None, residual=None):
    """A [M, K] bf16 (ANY number of rows: the drafter's prompt prefill) @ the packed weight, epilogues as linear_rows_streamk
    (lantern_linear_rows_packed: row blocks of 128 -- 64 for a gate / up pair -- re-stream the weight tile)."""
    if not isinstance(weight, PackedLinearWeight):
        raise _lib.LanternError("linear_rows_packed: weight must be a PackedLinearWeight (pack_linear_weight)")
    if not A.is_cuda or A.dtype != torch.bfloat16:
        raise _lib.LanternError("linear_rows_packed: A must be a bf16 device tensor")
    A = A.contiguous()
    M, K = A.shape
    if K != weight.K or (epilogue == EPI_SILU_MUL) != (weight.pair_rows > 0):
        raise _lib.LanternError("linear_rows_packed: the packed weight does not match this call (K, gate / up pairing)")
    out = torch.empty((M, weight.n_rows), dtype=torch.bfloat16, device=A.device)
    r = None if residual is None else residual.contiguous()
    b = None if bias is None else bias.contiguous()
    check(_lib.lib().lantern_linear_rows_packed(C.c_void_p(A.data_ptr()), C.c_void_p(weight.data.data_ptr()), C.c_void_p(_ptr(b)), M, K, weight.n_rows,
                                                C.c_void_p(out.data_ptr()), weight.n_rows, int(epilogue), C.c_void_p(_ptr(r)), 0 if r is None else r.shape[1],
                                                int(weight.pair_rows), _stream()), "linear_rows_packed")
    return out


def rmsnorm_rows(x, weight, eps: float):
    """ChameleonRMSNorm of bf16 rows [M, H] (lantern_rmsnorm_rows)."""
    x, weight = x.contiguous(), weight.contiguous()
    out = torch.empty_like(x)
    check(_lib.lib().lantern_rmsnorm_rows(C.c_void_p(x.data_ptr()), C.c_void_p(weight.data_ptr()), x.shape[0], x.shape[1], C.c_float(eps),
                                          C.c_void_p(out.data_ptr()), _stream()), "rmsnorm_rows")
    return out


def qk_norm_rope(qkv, B: int, T: int, nq: int, nk: int, d: int, qw, qb, kw, kb, cos, sin, position_ids, k_slab=None, v_slab=None, row0: int = 0):
    """Head stage of the drafter's attention on the fused projection qkv [B*T, (nq + 2 nk) d]: -> q [B,nq,T,d], k / v [B,nk,T,d] (bf16), or --
    k_slab / v_slab [B, nk, rows, d] given -- k / v written in place at rows [row0, row0 + T) of the slabs (returned as they are)."""
    dev = qkv.device
    q = torch.empty((B, nq, T, d), dtype=torch.bfloat16, device=dev)
    if k_slab is None:
        k_slab = torch.empty((B, nk, T, d), dtype=torch.bfloat16, device=dev)
        v_slab = torch.empty((B, nk, T, d), dtype=torch.bfloat16, device=dev)
        row0 = 0
    assert k_slab.is_contiguous() and v_slab.is_contiguous() and k_slab.shape == v_slab.shape and k_slab.shape[0] == B and k_slab.shape[1] == nk
    pos = position_ids.to(torch.int64)
    if pos.numel() == T and B > 1:          # one row of positions shared by the batch rows
        pos = pos.reshape(1, T).expand(B, T)
    if pos.numel() != B * T:
        raise _lib.LanternError(f"qk_norm_rope: position_ids must hold T = {T} (shared by the batch rows) or B x T positions, got {tuple(position_ids.shape)}")
    pos = pos.contiguous()
    check(_lib.lib().lantern_qk_norm_rope(C.c_void_p(qkv.contiguous().data_ptr()), B, T, nq, nk, d, C.c_void_p(qw.contiguous().data_ptr()),
                                          C.c_void_p(qb.contiguous().data_ptr()), C.c_void_p(kw.contiguous().data_ptr()), C.c_void_p(kb.contiguous().data_ptr()),
                                          qw.shape[0], C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), cos.shape[0], C.c_void_p(pos.data_ptr()),
                                          C.c_void_p(q.data_ptr()), C.c_void_p(k_slab.data_ptr()), C.c_void_p(v_slab.data_ptr()), k_slab.shape[2], row0,
                                          _stream()), "qk_norm_rope")
    return q, k_slab, v_slab


def qk_rope_pairs(qkv, B: int, T: int, nq: int, nk: int, d: int, freqs, position_ids, k_slab=None, v_slab=None, row0: int = 0):
    """Head stage of the LlamaGen drafter's attention (lantern_qk_rope_pairs; cnets_llamagen.py:315-323) on the fused projection
    qkv [B*T, (nq + 2 nk) d] bf16: rotary on adjacent pairs with freqs [rows, d/2, 2] f32 indexed by position_ids ([T] shared by the batch rows, or
    [B, T]) -> q [B,nq,T,d], k / v [B,nk,T,d] (bf16), or -- k_slab / v_slab [B, nk, rows, d] given -- k / v written in place at rows
    [row0, row0 + T) of the slabs."""
    dev = qkv.device
    q = torch.empty((B, nq, T, d), dtype=torch.bfloat16, device=dev)
    if k_slab is None:
        k_slab = torch.empty((B, nk, T, d), dtype=torch.bfloat16, device=dev)
        v_slab = torch.empty((B, nk, T, d), dtype=torch.bfloat16, device=dev)
        row0 = 0
    assert k_slab.is_contiguous() and v_slab.is_contiguous() and k_slab.shape == v_slab.shape and k_slab.shape[0] == B and k_slab.shape[1] == nk
    freqs = _dev(freqs, torch.float32, "freqs")
    if freqs.dim() != 3 or freqs.shape[1] != d // 2 or freqs.shape[2] != 2:
        raise _lib.LanternError(f"qk_rope_pairs: freqs must be [rows, {d // 2}, 2] f32, got {tuple(freqs.shape)}")
    pos = position_ids.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
    if pos.numel() == T:
        per_row = 0
    elif pos.numel() == B * T:
        per_row = 1
    else:
        raise _lib.LanternError(f"qk_rope_pairs: position_ids must hold T = {T} (shared by the batch rows) or B x T positions, got {tuple(position_ids.shape)}")
    check(_lib.lib().lantern_qk_rope_pairs(C.c_void_p(qkv.contiguous().data_ptr()), B, T, nq, nk, d, C.c_void_p(freqs.data_ptr()), freqs.shape[0],
                                           C.c_void_p(pos.data_ptr()), per_row, C.c_void_p(q.data_ptr()), C.c_void_p(k_slab.data_ptr()),
                                           C.c_void_p(v_slab.data_ptr()), k_slab.shape[2], row0, _stream()), "qk_rope_pairs")
    return q, k_slab, v_slab


def head_sample(A, weight, row_lo: int, n_cols: int, cfg: float, bias=None, model: int = MODEL_LUMINA, pos_ids=None, pos_base: int = 2,
                w: int = 48, h: int = 48, newline_id: int = 8803, eos_id: int = 8196, top_k_filter: int = 0, n_draw: int = 10, draw_u=None, draw_idx=None,
                packed: Optional["PackedLinearWeight"] = None):
    """The static drafter's head stage (lantern_head_sample): A [2n, K] bf16 (n cond rows, then n uncond rows) -> head window GEMM + CFG -> processors
    -> softmax -> n_draw draws without replacement per row from injected uniforms draw_u [n, n_draw] f64 (or the injected indices draw_idx).
    Returns (probs [n, V] f32, ss_token [n, n_draw] i64, ss_prob [n, n_draw] f32)."""
    A = _dev(A, torch.bfloat16, "A")
    weight = _dev(weight, torch.bfloat16, "weight")
    n = A.shape[0] // 2
    K, V = A.shape[1], weight.shape[0]
    dev = A.device
    ws = torch.empty((n, n_cols), dtype=torch.bfloat16, device=dev)
    probs = torch.empty((n, V), dtype=torch.float32, device=dev)
    tok = torch.empty((n, n_draw), dtype=torch.int64, device=dev)
    prob = torch.empty((n, n_draw), dtype=torch.float32, device=dev)
    b = None if bias is None else _dev(bias, torch.bfloat16, "bias").contiguous()
    pos = None if pos_ids is None else _dev(pos_ids, torch.int64, "pos_ids").contiguous()
    du = None if draw_u is None else _dev(draw_u, torch.float64, "draw_u").contiguous()
    di = None if draw_idx is None else _dev(draw_idx, torch.int64, "draw_idx").contiguous()
    if packed is not None and (packed.K != K or packed.n_rows != n_cols or packed.pair_rows):
        raise _lib.LanternError("head_sample: the packed weight is not the window rows of this head")
    sk = _sk_workspace(dev)
    wt = weight if packed is None else packed.data
    check(_lib.lib().lantern_head_sample(C.c_void_p(A.contiguous().data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(_ptr(b)), n, K, int(row_lo), int(n_cols), V,
                                         C.c_float(cfg), int(model), C.c_void_p(_ptr(pos)), C.c_int64(pos_base), w, h, newline_id, eos_id, int(top_k_filter),
                                         int(n_draw), C.c_void_p(_ptr(du)), C.c_void_p(_ptr(di)), C.c_void_p(ws.data_ptr()), C.c_void_p(probs.data_ptr()),
                                         C.c_void_p(tok.data_ptr()), C.c_void_p(prob.data_ptr()), int(packed is not None), C.c_void_p(sk.data_ptr()),
                                         C.c_size_t(sk.numel()), _stream()), "head_sample")
    return probs, tok, prob


def head_expand(A, weight, row_lo: int, n_cols: int, cfg: float, bias=None, model: int = MODEL_LUMINA, pos_ids=None, pos_base: int = 2,
                w: int = 48, h: int = 48, newline_id: int = 8803, eos_id: int = 8196, top_k_filter: int = 0, scores_in=None, top_k: int = 10,
                packed: Optional["PackedLinearWeight"] = None, streamk: bool = True):
    """8f-2 fused: one drafter expansion depth from hidden states to top-k (lantern_head_expand[_streamk]).  A [2n, K] bf16 (n cond rows, then
    n uncond rows), weight [V, K] bf16; `packed`: pack_linear_weight(weight[row_lo : row_lo + n_cols]) (the fast form of the window GEMM).
    Returns (topk_index [1,n,k], cu_scores [1,n,k], topk_cs_index [1,k], scores_out [1,k]) like expand_dynamic on a batch of one sequence."""
    A = _dev(A, torch.bfloat16, "A")
    weight = _dev(weight, torch.bfloat16, "weight")
    n = A.shape[0] // 2
    K, V = A.shape[1], weight.shape[0]
    dev = A.device
    ws = torch.empty((n, n_cols), dtype=torch.bfloat16, device=dev)
    ti = torch.empty((1, n, top_k), dtype=torch.int64, device=dev)
    cu = torch.empty((1, n, top_k), dtype=torch.float32, device=dev)
    ci = torch.empty((1, top_k), dtype=torch.int64, device=dev)
    so = torch.empty((1, top_k), dtype=torch.float32, device=dev)
    pos = None if pos_ids is None else _dev(pos_ids.reshape(-1), torch.int64, "pos_ids")
    si = None if scores_in is None else _dev(scores_in.reshape(-1), torch.float32, "scores_in")
    b = None if bias is None else _dev(bias, torch.bfloat16, "bias")
    if streamk or packed is not None:
        if packed is not None and (packed.K != K or packed.n_rows != n_cols or packed.pair_rows):
            raise _lib.LanternError("head_expand: the packed weight is not the window rows of this head")
        sk = _sk_workspace(dev)
        wt = weight if packed is None else packed.data
        check(_lib.lib().lantern_head_expand_streamk(C.c_void_p(A.data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(_ptr(b)), n, K, int(row_lo), int(n_cols), V,
                                                     C.c_float(cfg), int(model), C.c_void_p(_ptr(pos)), C.c_int64(pos_base), w, h, newline_id, eos_id,
                                                     int(top_k_filter), C.c_void_p(_ptr(si)), int(top_k), C.c_void_p(ws.data_ptr()), C.c_void_p(ti.data_ptr()),
                                                     C.c_void_p(cu.data_ptr()), C.c_void_p(ci.data_ptr()), C.c_void_p(so.data_ptr()), int(packed is not None),
                                                     C.c_void_p(sk.data_ptr()), C.c_size_t(sk.numel()), _stream()), "head_expand_streamk")
        return ti, cu, ci, so
    check(_lib.lib().lantern_head_expand(C.c_void_p(A.data_ptr()), C.c_void_p(weight.data_ptr()), C.c_void_p(_ptr(b)), n, K, int(row_lo), int(n_cols), V,
                                         C.c_float(cfg), int(model), C.c_void_p(_ptr(pos)), C.c_int64(pos_base), w, h, newline_id, eos_id, int(top_k_filter),
                                         C.c_void_p(_ptr(si)), int(top_k), C.c_void_p(ws.data_ptr()), C.c_void_p(ti.data_ptr()), C.c_void_p(cu.data_ptr()),
                                         C.c_void_p(ci.data_ptr()), C.c_void_p(so.data_ptr()), _stream()), "head_expand")
    return ti, cu, ci, so


def drafter_attention_mask(attention_mask, tree_mask, B: int, T: int, past: int, device=None):
    """a5: additive [B,1,T,past+T] f32 mask = causal + padding (+ tree), Model._prepare_decoder_attention_mask in one launch.
    attention_mask [B,L] bool or None; tree_mask [1|B,1,t0,t1] f32 or None."""
    dev = device or (attention_mask.device if attention_mask is not None else tree_mask.device)
    out = torch.empty((B, 1, T, past + T), dtype=torch.float32, device=dev)
    am = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.uint8).contiguous()
    tm = None if tree_mask is None else tree_mask.to(device=dev, dtype=torch.float32).contiguous()
    tb, t0, t1 = (tm.shape[0], tm.shape[-2], tm.shape[-1]) if tm is not None else (1, 0, 0)
    check(_lib.lib().lantern_drafter_attention_mask(C.c_void_p(_ptr(am)), am.shape[1] if am is not None else 0, C.c_void_p(_ptr(tm)), tb, t0,
                                                    t1, B, T, past, C.c_void_p(out.data_ptr()), _stream()), "drafter_attention_mask")
    return out


def mask_left_padding(attention_mask, out=None):
    """The reductions a drafting call makes over its attention mask [B, S] (bool / uint8 / int64, device), one launch: returns an int64 [3, B] tensor --
    [0] the first non-zero index per row (torch.argmax of a left-padded mask), [1] the number of non-zero entries (position_ids[:, -1] + 1 of
    `mask.cumsum(-1) - 1`), [2] 1 where a zero follows a one (not left padding)."""
    if not attention_mask.is_cuda or attention_mask.dim() != 2:
        raise _lib.LanternError("mask_left_padding: expected a [B, S] device tensor")
    m = attention_mask
    if m.dtype not in (torch.bool, torch.uint8, torch.int64):
        m = m.to(torch.int64)
    if m.stride(1) != 1:
        m = m.contiguous()
    B, S = m.shape
    if out is None or out.shape != (3, B) or out.device != m.device:
        out = torch.empty((3, B), dtype=torch.int64, device=m.device)
    check(_lib.lib().lantern_mask_left_padding(C.c_void_p(m.data_ptr()), m.element_size(), B, C.c_int64(S), C.c_int64(m.stride(0) if B > 1 else max(S, 1)),
                                               C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()), _stream()),
          "mask_left_padding")
    return out


def drafter_fc(ids, hidden, embed, weight, bias=None, embed_scale: float = 1.0, packed: Optional["PackedLinearWeight"] = None):
    """O11 (MFMA): fc(cat(embed[ids] * scale, hidden)) -> bf16 [M,H].  ids [M] i64, hidden [M,H] bf16, embed [vocab,H] bf16,
    weight [H,2H] bf16 (nn.Linear layout), bias [H] bf16 or None.  Up to 32 rows with H % 64 == 0 (the drafting shape) take the stream-K
    kernel, on `packed` = pack_linear_weight(weight) when given."""
    for t, n in ((hidden, "hidden"), (embed, "embed"), (weight, "weight")):
        if not t.is_cuda or t.dtype != torch.bfloat16:
            raise _lib.LanternError(f"drafter_fc: {n} must be a bf16 device tensor")
    ids = _dev(ids, torch.int64, "ids").reshape(-1)
    hidden, embed, weight = hidden.contiguous(), embed.contiguous(), weight.contiguous()
    M, H = hidden.reshape(-1, hidden.shape[-1]).shape
    assert weight.shape == (H, 2 * H) and embed.shape[1] == H and ids.numel() == M
    out = torch.empty((M, H), dtype=torch.bfloat16, device=hidden.device)
    b = None if bias is None else bias.contiguous()
    if M <= 32 and H % 64 == 0:
        if packed is not None and (packed.K != 2 * H or packed.n_rows != H or packed.pair_rows):
            raise _lib.LanternError("drafter_fc: the packed weight is not this fc's [H, 2H] weight")
        sk = _sk_workspace(hidden.device)
        wt = weight if packed is None else packed.data
        check(_lib.lib().lantern_drafter_fc_streamk(C.c_void_p(ids.data_ptr()), C.c_void_p(hidden.data_ptr()), C.c_void_p(embed.data_ptr()),
                                                    C.c_void_p(wt.data_ptr()), C.c_void_p(_ptr(b)), M, H, embed.shape[0], C.c_float(embed_scale),
                                                    C.c_void_p(out.data_ptr()), int(packed is not None), C.c_void_p(sk.data_ptr()), C.c_size_t(sk.numel()),
                                                    _stream()), "drafter_fc_streamk")
        return out
    check(_lib.lib().lantern_drafter_fc(C.c_void_p(ids.data_ptr()), C.c_void_p(hidden.data_ptr()), C.c_void_p(embed.data_ptr()),
                                        C.c_void_p(weight.data_ptr()), C.c_void_p(_ptr(b)), M, H, embed.shape[0], C.c_float(embed_scale),
                                        C.c_void_p(out.data_ptr()), _stream()), "drafter_fc")
    return out


def pack_vq_table(table, cols: int):
    """[K, K-1] uint16 table (int16-viewed) -> [K, cols] with cols % 8 == 0: the first `cols` neighbours of every code in
    16-byte aligned rows (the layout evaluate_posterior_window stages fastest); needs k <= cols - 1."""
    table = table.contiguous()
    assert table.dtype == torch.int16 and table.is_cuda and cols % 8 == 0
    out = torch.empty((table.shape[0], cols), dtype=torch.int16, device=table.device)
    check(_lib.lib().lantern_pack_vq_table(C.c_void_p(table.data_ptr()), table.shape[0], table.shape[1], C.c_void_p(out.data_ptr()), cols,
                                           _stream()), "pack_vq_table")
    return out


def build_vq_table(codebook):
    """8f-1: codebook [K,C] f32 -> uint16 table [K,K-1] (as an int16-viewed tensor), generate_codebook.py:53-65."""
    cb = _dev(codebook, torch.float32, "codebook")
    K, Cc = cb.shape
    table = torch.empty((K, K - 1), dtype=torch.int16, device=cb.device)
    check(_lib.lib().lantern_build_vq_table(C.c_void_p(cb.data_ptr()), K, Cc, C.c_void_p(table.data_ptr()), None, _stream()),
          "build_vq_table")
    return table


def save_vq_table(table: torch.Tensor, save_path: str) -> str:
    """The reference's on-disk format (entrypoints/generate_codebook.py:60-65): `<save_path>/top_{K-1}_indices.npy`, uint16 [K, K-1]
    (the file the models load: ea_model_lumina_mgpt.py:321 `np.load("ckpts/lumina_mgpt/vq_distances/top_8191_indices.npy")`)."""
    import os
    t = table.detach().cpu().contiguous()
    if t.dtype == torch.int16:
        arr = t.numpy().view(np.uint16)
    else:
        arr = t.numpy().astype(np.uint16)
    K, cols = arr.shape
    if cols != K - 1:
        raise _lib.LanternError(f"save_vq_table: expected the full [K, K-1] table, got {arr.shape}")
    os.makedirs(save_path, exist_ok=True)
    path = os.path.join(save_path, f"top_{K - 1}_indices.npy")
    np.save(path, arr)
    return path


def load_vq_table(path: str, device=None) -> torch.Tensor:
    """`top_{K-1}_indices.npy` -> the int16-viewed uint16 device tensor the kernels take (np.load as the reference does)."""
    arr = np.load(path)
    if arr.dtype != np.uint16 or arr.ndim != 2:
        raise _lib.LanternError(f"load_vq_table: {path}: expected uint16 [K, K-1], got {arr.dtype} {arr.shape}")
    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int16))
    return t.to(device) if device is not None else t


def tree_mask_bits(tree_mask: torch.Tensor) -> torch.Tensor:
    """[..., N, N] tree attention mask (non-zero = visible; the reference's `tree_attn_mask` / dynamic `tree_mask`) -> one
    64-bit ancestor word per node, int64 [N] or [B,N] (bit t of word n = node n sees tree key t)."""
    m = tree_mask
    while m.dim() > 3:
        assert m.shape[1] == 1 or m.dim() == 4
        m = m[:, 0] if m.dim() == 4 else m
    if m.dim() == 3 and m.shape[0] == 1:
        m = m[0]
    N = m.shape[-1]
    if N > 64 or m.shape[-2] != N:
        raise _lib.LanternError(f"tree_mask_bits: {tuple(tree_mask.shape)}: need a square tree block of at most 64 nodes")
    w = torch.ones(N, dtype=torch.int64, device=m.device) << torch.arange(N, dtype=torch.int64, device=m.device)
    return ((m != 0).to(torch.int64) * w).sum(-1).contiguous()      # bit 63 wraps to the sign bit: same 64-bit pattern


def drafter_tree_bits(tree_mask: torch.Tensor, T: int) -> Tuple[torch.Tensor, int]:
    """The drafter's tree block as ancestor words for `tree_attention`.  tree_mask: [1|B, 1, t0, t1] (or [t0, t1]) with t0 >= T, t1 <= 64: the last
    T rows are this call's T new tokens, the t1 columns the last t1 keys of the cache (cnets_lumina_mgpt.py:1014-1050: finfo.min wherever the block
    is zero, on top of the causal mask).  Returns (int64 [t1] | [B, t1] words for t1 query rows -- rows in front of the last T are placeholders that
    see only themselves -- , t1).  Built on the device, no synchronisation."""
    m = tree_mask
    while m.dim() > 3:
        m = m[:, 0]
    if m.dim() == 2:
        m = m[None]
    t0, t1 = m.shape[-2], m.shape[-1]
    if t1 > 64 or t0 < T or t1 < T:
        raise _lib.LanternError(f"drafter_tree_bits: block {t0} x {t1} for {T} new tokens (at most 64 tree keys)")
    dev = m.device
    rows = (m[:, t0 - T:] != 0)                                                  # [Bm, T, t1]
    j = torch.arange(t1, device=dev)
    causal = j[None, :] <= (t1 - T + torch.arange(T, device=dev))[:, None]      # row i is key t1 - T + i: nothing behind it
    rows = rows & causal[None]
    w = torch.ones(t1, dtype=torch.int64, device=dev) << j.to(torch.int64)
    words = (rows.to(torch.int64) * w).sum(-1)                                   # [Bm, T]
    pad = (torch.ones(t1 - T, dtype=torch.int64, device=dev) << torch.arange(t1 - T, dtype=torch.int64, device=dev))[None].expand(words.shape[0], -1)
    bits = torch.cat((pad, words), dim=1).contiguous()
    return (bits[0] if bits.shape[0] == 1 else bits), t1


def tree_attention(q, k_cache, v_cache, tree_bits, kv_len=None, kv_start=None, max_kv_len: Optional[int] = None,
                   scale: Optional[float] = None, out=None):
    """8f-3: attention of the tree-verify forward over the KV cache in place (no mask tensor, no repeat_kv, no [N,S] scores).
    q bf16 [B,N,Hq,d] (or any view with a contiguous last dim, e.g. the transposed [B,Hq,N,d] -- pass it as `q.transpose(1,2)`);
    k_cache / v_cache bf16 [B,Hkv,S_max,d] views of the slab (the N tree keys already appended); tree_bits int64 [N] | [B,N]
    (`tree_mask_bits`); kv_len int64 [B] keys per row incl. the tree keys (None: max_kv_len everywhere); kv_start int64 [B]
    first visible key (left padding; None: 0).  Returns bf16 [B,N,Hq*d]."""
    for t, n in ((q, "q"), (k_cache, "k_cache"), (v_cache, "v_cache")):
        if not t.is_cuda or t.dtype != torch.bfloat16:
            raise _lib.LanternError(f"tree_attention: {n} must be a bf16 device tensor")
    B, N, Hq, d = q.shape
    Bk, Hkv, S_max, dk = k_cache.shape
    if (Bk, dk) != (B, d) or v_cache.shape != k_cache.shape or k_cache.stride() != v_cache.stride():
        raise _lib.LanternError(f"tree_attention: q {tuple(q.shape)} vs caches {tuple(k_cache.shape)} / {tuple(v_cache.shape)}")
    if q.stride(3) != 1 or k_cache.stride(3) != 1 or k_cache.stride(2) != d:
        raise _lib.LanternError("tree_attention: q needs a contiguous last dim, the caches contiguous [S_max, d] rows")
    if max_kv_len is None:
        max_kv_len = S_max
    if not N <= max_kv_len <= S_max:
        raise _lib.LanternError(f"tree_attention: max_kv_len={max_kv_len} outside [N={N}, S_max={S_max}]")
    bits = _dev(tree_bits, torch.int64, "tree_attention: tree_bits").contiguous()
    if bits.shape not in ((N,), (B, N)):
        raise _lib.LanternError(f"tree_attention: tree_bits {tuple(bits.shape)}, expected [{N}] or [{B},{N}]")
    kl = None if kv_len is None else _dev(kv_len, torch.int64, "tree_attention: kv_len").contiguous()
    ks = None if kv_start is None else _dev(kv_start, torch.int64, "tree_attention: kv_start").contiguous()
    for t, n in ((kl, "kv_len"), (ks, "kv_start")):
        if t is not None and t.numel() != B:
            raise _lib.LanternError(f"tree_attention: {n} must have {B} entries")
    if out is None:
        out = torch.empty((B, N, Hq * d), dtype=torch.bfloat16, device=q.device)
    assert out.dtype == torch.bfloat16 and out.shape == (B, N, Hq * d) and out.stride(2) == 1
    L = _lib.lib()
    need = int(L.lantern_tree_attention_workspace(B, Hq, N, d, C.c_int64(max_kv_len)))
    ws = torch.empty(need, dtype=torch.uint8, device=q.device) if need else None
    check(L.lantern_tree_attention(C.c_void_p(q.data_ptr()), C.c_void_p(k_cache.data_ptr()), C.c_void_p(v_cache.data_ptr()),
                                   C.c_void_p(out.data_ptr()), B, Hq, Hkv, N, d, C.c_int64(q.stride(0)), C.c_int64(q.stride(1)),
                                   C.c_int64(q.stride(2)), C.c_int64(k_cache.stride(0)), C.c_int64(k_cache.stride(1)),
                                   C.c_int64(out.stride(0)), C.c_int64(out.stride(1)), C.c_void_p(_ptr(kl)), C.c_void_p(_ptr(ks)),
                                   C.c_int64(max_kv_len), C.c_void_p(bits.data_ptr()), int(bits.dim() == 2),
                                   C.c_float(float(scale) if scale is not None else d ** -0.5), C.c_void_p(_ptr(ws)), C.c_size_t(need),
                                   _stream()), "tree_attention")
    return out

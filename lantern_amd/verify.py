"""Shared host machinery of the verify/accept mirror classes.

Nothing here computes: it prepares arguments (tree buffers, uniform stream, lengths) and calls the
HIP entry points through lantern_amd.ops.
"""
from __future__ import annotations

import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops

TOPK = 10  # drafter fan-out of the static trees (models/ea_model_lumina_mgpt.py:23)


class UniformFifo:
    """The reference draws its acceptance uniforms from Python's module-level `random.random()`
    (MT19937), one per tried candidate, in a data-dependent count (SURVEY 8a RNG contract).  The
    kernel cannot call back into Python, so a window of the SAME stream is staged on the device and
    consumed through a device cursor; values are drawn from `random` in order, so the sequence of
    uniforms the acceptance tests see is exactly the reference's.  Host syncs only when the window
    may run out (every ~window/max_per_step steps).

    begin() / end() bracket one generate(): begin() stages from the stream's CURRENT position (a
    `random.seed()` between two prompts is honoured), end() puts the module-level generator back
    exactly where the reference's own draws would have left it -- the staged but unconsumed values
    are returned to the stream."""

    def __init__(self, device, window: int = 4096, rng=None):
        self.device, self.window = device, window
        self.rng = rng if rng is not None else random
        self.buf = torch.zeros((1, window), dtype=torch.float64, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)
        self.upper = 0  # host-side upper bound of the cursor
        self.active = False
        self.values: List[float] = []
        self._state0, self._consumed = None, 0

    def begin(self):
        self._state0, self._consumed = self.rng.getstate(), 0
        self.values = [self.rng.random() for _ in range(self.window)]
        self.buf.copy_(torch.tensor(self.values, dtype=torch.float64).reshape(1, self.window))
        self.cursor.zero_()
        self.upper = 0
        self.active = True

    def end(self) -> int:
        """Returns the number of uniforms the kernels consumed since begin()."""
        if not self.active:
            return 0
        total = self._consumed + int(self.cursor.item())
        self.rng.setstate(self._state0)
        for _ in range(total):
            self.rng.random()
        self.active = False
        return total

    def reserve(self, max_draws: int):
        if not self.active:
            self.begin()
        if self.upper + max_draws > self.window:
            used = int(self.cursor.item())          # the only sync
            self._consumed += used
            self.values = self.values[used:] + [self.rng.random() for _ in range(used)]
            self.buf.copy_(torch.tensor(self.values, dtype=torch.float64).reshape(1, self.window))
            self.cursor.zero_()
            self.upper = 0
        self.upper += max_draws
        self._last = max_draws

    def consumed(self, n_used: int):
        """The step behind the last reserve() really drew `n_used` uniforms (its verdict says so): the host-side bound becomes exact again, so the
        window is refilled -- the one host sync of this class -- only when it is really used up (every ~window / 4.5 steps instead of window / max)."""
        last = getattr(self, "_last", 0)
        if 0 <= n_used <= last:
            self.upper -= last - n_used
            self._last = n_used


class NodeLogits:
    """Processed tree-node logits `[N,V]` + `retrieve_indices [P,D]`: what the reference's
    `tree_decoding` returns as the gathered `[P,D,V]` tensor, without materialising it.
    `materialize()` / indexing give the reference's tensor for API compatibility."""

    def __init__(self, node_logits: torch.Tensor, retrieve_indices: torch.Tensor):
        self.node_logits = node_logits
        self.retrieve_indices = retrieve_indices

    @property
    def shape(self):
        return (*self.retrieve_indices.shape, self.node_logits.shape[-1])

    @property
    def device(self):
        return self.node_logits.device

    def row_index(self) -> torch.Tensor:
        r = self.retrieve_indices.to(torch.int64)
        return torch.where(r < 0, r + self.node_logits.shape[0], r).to(torch.int32)

    def materialize(self) -> torch.Tensor:
        return self.node_logits[self.retrieve_indices]

    def __getitem__(self, idx):
        return self.materialize()[idx]


class WindowRows:
    """What the windowed kernel set hands from `tree_decoding` to `evaluate_posterior`: softmax probabilities of the
    processed node rows restricted to the image-token window, `[N, W] f32`, plus `row_hot [N]` (token id of a forced one-hot
    row -- newline / end of image -- or -1) and `retrieve_indices [P,D]`.  `materialize()` rebuilds the reference's
    `[P,D,V]` tensor of PROBABILITIES for callers that want to look at it."""

    def __init__(self, win: torch.Tensor, row_hot: torch.Tensor, retrieve_indices: torch.Tensor, V: int, win_lo: int):
        self.win, self.row_hot, self.retrieve_indices, self.V, self.win_lo = win, row_hot, retrieve_indices, V, win_lo
        self.dense_source = None      # callable -> NodeLogits of the same step (set by tree_decoding)

    def dense_rows(self):
        """The same step's rows for the dense kernel set (full-vocabulary processed logits), for the states the windowed
        evaluate_posterior reports instead of representing (LANTERN_ST_NEEDS_DENSE and friends)."""
        if self.dense_source is None:
            raise RuntimeError("WindowRows.dense_rows: no dense source attached (tree_decoding attaches one)")
        return self.dense_source()

    @property
    def shape(self):
        return (*self.retrieve_indices.shape, self.V)

    @property
    def device(self):
        return self.win.device

    def row_index(self) -> torch.Tensor:
        r = self.retrieve_indices.to(torch.int64)
        return torch.where(r < 0, r + self.win.shape[0], r).to(torch.int32)

    def materialize(self) -> torch.Tensor:
        N, W = self.win.shape
        dense = torch.zeros((N, self.V), dtype=torch.float32, device=self.win.device)
        dense[:, self.win_lo:self.win_lo + W] = self.win
        hot = self.row_hot.long()
        rows = torch.nonzero(hot >= 0).reshape(-1)
        dense[rows] = 0.0
        dense[rows, hot[rows]] = 1.0
        return dense[self.retrieve_indices]

    def __getitem__(self, idx):
        return self.materialize()[idx]


def as_rows(logits):
    """(rows [R,V] f32, row_index [P,D] i32) from either a NodeLogits or the reference's [P,D,V] tensor."""
    if isinstance(logits, NodeLogits):
        return logits.node_logits, logits.row_index()
    P, D, V = logits.shape
    rows = logits.reshape(P * D, V)
    idx = torch.arange(P * D, dtype=torch.int32, device=logits.device).reshape(P, D)
    return rows, idx


def nested_from_csr(b_off: np.ndarray, b_idx: np.ndarray, P: int, D: int, device):
    out = []
    for p in range(P):
        row = []
        for d in range(D):
            a, b = int(b_off[p * D + d]), int(b_off[p * D + d + 1])
            row.append(torch.tensor(b_idx[a:b], device=device) if b > a else [])
        out.append(row)
    return out


def generate_tree_buffers(tree_choices, device="cuda"):
    """Same dictionary as the reference's generate_tree_buffers (ea_model_lumina_mgpt.py:140-277):
    tree_attn_mask [1,1,N,N], tree_indices, tree_position_ids, retrieve_indices, p_indices (nested list),
    b_indices (nested lists of tensors) -- plus `_hip`: the flat device buffers the kernels consume."""
    tb = ops.tree_static_build(tree_choices, TOPK)
    P, D = tb["retrieve_indices"].shape
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    buffers = {
        "tree_attn_mask": t(tb["tree_attn_mask"])[None, None],
        "tree_indices": t(tb["tree_indices"]),
        "tree_position_ids": t(tb["tree_position_ids"]),
        "retrieve_indices": t(tb["retrieve_indices"]),
        "p_indices": tb["p_indices"].tolist(),
        "b_indices": nested_from_csr(tb["b_off"], tb["b_idx"], P, D, device),
    }
    # drafter rows per level: row r of the concatenated original_prob belongs to level depth(parent(r))
    ti, pos, mask = tb["tree_indices"], tb["tree_position_ids"], tb["tree_attn_mask"]
    N = len(ti)
    R = int(((ti[1:] - 1) // TOPK).max()) + 1
    par_row = np.zeros(R, np.int64)
    for n in range(1, N):
        anc = [a for a in np.nonzero(mask[n] > 0)[0] if pos[a] == pos[n] - 1]
        par_row[(ti[n] - 1) // TOPK] = anc[0]
    depth_of_row = pos[par_row]
    op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
    buffers["_hip"] = dict(p_idx=t(tb["p_indices"]), b_off=t(tb["b_off"]),
                           b_idx=t(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)), op_off=t(op_off),
                           N=N, P=P, D=D, R=R)
    # per-node tables of the node-parallel evaluate_posterior (one workgroup per internal tree node: the faster form at the
    # reference's batch of one sequence); None when the tree is outside what those kernels stage (the chain kernel then runs)
    try:
        nt = ops.tree_node_tables(tb["retrieve_indices"], N, tb["p_indices"], tb["b_off"], op_off, device=device, b_idx=tb["b_idx"])
        buffers["_hip"]["nodes"] = nt if (nt.prefix_siblings and nt.n_nodes <= 128) else None
    except Exception:
        buffers["_hip"]["nodes"] = None
    return buffers


def concat_original_prob(original_prob: Sequence[torch.Tensor]) -> torch.Tensor:
    """list of [n_i,V] (ss_original_prob of the static drafter) -> [1,R,V] f32."""
    return torch.cat([o.reshape(-1, o.shape[-1]) for o in original_prob], dim=0).to(torch.float32)[None].contiguous()


class ProcessorSpec:
    """Numbers behind the HF LogitsProcessorList of prepare_logits_processor (drafters/utils.py:36-52):
    Temperature -> TopP -> TopK.  Callable like the HF list (`spec(None, logits)`) for the parts of the
    pipeline that stay in torch (drafter side); the verify kernels take the numbers."""

    def __init__(self, temperature: float = 1.0, top_p: float = 1.0, top_k: int = 0):
        self.temperature, self.top_p, self.top_k = float(temperature), float(top_p), int(top_k)

    @staticmethod
    def from_hf(proc) -> Optional["ProcessorSpec"]:
        if proc is None:
            return None
        if isinstance(proc, ProcessorSpec):
            return proc
        s = ProcessorSpec()
        for p in proc:
            n = type(p).__name__
            if n == "TemperatureLogitsWarper":
                s.temperature = float(p.temperature)
            elif n == "TopPLogitsWarper":
                s.top_p = float(p.top_p)
            elif n == "TopKLogitsWarper":
                s.top_k = int(p.top_k)
            elif n == "RepetitionPenaltyLogitsProcessor":
                raise NotImplementedError("repetition penalty is never enabled by the reference's generate()")
        return s

    def __call__(self, input_ids, scores):
        if self.temperature > 1e-5 and self.temperature != 1.0:
            scores = scores / self.temperature
        if 1e-8 <= self.top_p < 1.0:
            sl, si = torch.sort(scores, descending=False)
            cp = sl.softmax(dim=-1).cumsum(dim=-1)
            rm = cp <= (1 - self.top_p)
            rm[..., -1:] = 0
            scores = scores.masked_fill(rm.scatter(-1, si, rm), float("-inf"))
        if self.top_k > 0:
            k = min(self.top_k, scores.size(-1))
            scores = scores.masked_fill(scores < torch.topk(scores, k)[0][..., -1, None], float("-inf"))
        return scores


def prepare_logits_processor(temperature: float = 0.0, repetition_penalty: float = 0.0, top_p: float = 0.0, top_k: int = 0):
    """drafters/utils.py:36-52 with the same activation rules."""
    s = ProcessorSpec()
    if temperature > 1e-5:
        if temperature != 1.0:
            s.temperature = temperature
        if repetition_penalty > 1.0:
            raise NotImplementedError("repetition penalty")
        if 1e-8 <= top_p < 1.0:
            s.top_p = top_p
        if top_k > 0:
            s.top_k = top_k
    return s


def reference_loader(module: str, attr: str):
    """The reference's own loader class/function (checkpoint loading, tokenizers and item processors stay in the reference: this
    package replaces the accept loop, not the loaders).  Needs the jadohu/LANTERN checkout importable (its root on sys.path, as
    entrypoints/generate_images.py has it); raises a clear error otherwise -- nothing of the loader is re-implemented here."""
    import importlib
    try:
        return getattr(importlib.import_module(module), attr)
    except Exception as e:          # ImportError, or whatever the reference's own imports raise
        from ._lib import LanternError
        raise LanternError(f"lantern_amd delegates model / checkpoint loading to the reference ({module}.{attr}), which is not importable here: "
                           f"{type(e).__name__}: {e}.  Put the LANTERN checkout on sys.path, or load the model yourself and wrap it with "
                           "from_reference(ref_model) (INTEGRATION.md 3).") from e

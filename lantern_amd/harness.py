"""Synthetic Lumina-mGPT-7B-768 verify/accept workload (BASELINE.json config C3/C5).

No checkpoints exist on either machine, so the target-model and drafter forwards are replaced by
pre-generated pools; everything between them -- the hot path -- runs for
real, every step, on device-resident state with no host round trip:

    O6 gather_candidates -> O7 cfg_mask_topk -> O8 evaluate_posterior -> O9 kv_gather -> O10 accept_gather

Geometry follows the reference defaults (entrypoints/generate_images.py:47-60): V=65536, image ids
4..8195, static tree mc_sim_7b_63 (eagle_version 1), cfg 3.0, top_k 2000, lantern_k 1000,
cfg_mode "sequential" = two B=1 KV slabs per sequence [2L=64, 1, 32, S_max, 128] bf16 whose
prev_len differ by the prompt length (models/ea_model_lumina_mgpt.py:753-767).

This module is product-side host code: it never imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import random
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

from . import _lib, ops
from ._lib import EpBuffers, EpNodes, EpParams, EpWindow, StepDynamic, StepGroup, check

from contextlib import nullcontext as _nullctx

MC_SIM_7B_63 = [[0], [1], [2], [3], [0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [2, 0], [2, 1], [3, 0],
                [0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 1, 0], [0, 1, 1], [0, 2, 0], [0, 2, 1], [1, 0, 0],
                [0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 2], [0, 0, 0, 0, 0], [0, 0, 0, 0, 1]]

V = 65536
K_CODES = 8192
IMG_LO, IMG_HI = 4, 8196
NEWLINE, EOS = 8803, 8196
W_LATENT = H_LATENT = 48          # 768 / 16
TOKENS_PER_IMAGE = (W_LATENT + 1) * H_LATENT + 3   # 2352 grid+newline tokens + 3 header tokens
HIDDEN = 4096



_GROUP_STREAMS = {}
_SKIPPED = []


def group_streams(device, n: int):
    """The stream-group streams of a device, created once per process and shared by every workload built after (workloads run one at a time): a
    stream is bound to a hardware queue when it is created, and streams created late in a process that has made many land two groups on one
    queue -- measured: a side configuration's step 46 -> 80 us in the third workload of a bench run."""
    key = str(torch.device(device))
    have = _GROUP_STREAMS.get(key)
    if have is None:
        have = _GROUP_STREAMS[key] = []
        for _ in range(int(os.environ.get("LANTERN_GROUP_STREAM_SKIP", "0"))):          # tuning knob (diagnostic): start further along torch's stream pool
            _SKIPPED.append(torch.cuda.Stream(device=device))
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


@dataclass
class WorkloadConfig:
    model: str = "lumina"           # "lumina" (BASELINE C3 / C5) or "anole" (C4: Anole-7B 512x512, LANTERN++ static tree -- the same 7B KV geometry and
                                    # image-token window; no grammar rows, the neighbours of a rejected token are zeroed in the DRAFTER's row:
                                    # ea_model_anole.py:597-669, :930-931; rows post-processed by cfg_mask_topk, the chain kernel on probability rows)
    tree: str = "mc_sim_7b_63"      # static tree by its name in drafters/choices.py (the reference's default for Lumina, generate_images.py:59)
    n_seq: int = 64                 # 4.3 GB of KV slabs each (BASELINE.md geometry: 4096 rows) -> 276e9 of the GPU's 309e9 bytes
    pool_steps: int = 16
    lantern_k: int = 1000
    lantern_delta: float = 0.1
    cfg_scale: float = 3.0
    top_k: int = 2000
    top_p: float = 1.0              # < 1: TopPLogitsWarper in front of the top-k, wherever the rows are post-processed (O7, prepare_step, raw rows inside the
                                    # chain kernel): LlamaGen / Anole take top_p from generate() (drafters/utils.py:36-52)
    sigma: float = 5.0              # drafter noise, tuned once so accepted tokens/step ~2.6, then frozen
    logit_scale: float = 4.0
    prompt_len: int = 64
    kv_layers: int = 32
    kv_heads: int = 32
    kv_smax: int = 4096             # rows per slab = max_position_embeddings, as the reference allocates (kv_cache.py:124) and
                                    # BASELINE.md fixes.  A 768x768 image only ever reaches row 64 + 3 + 2355 + 59 = 2481:
                                    # `--kv-smax 2560 --seqs-per-gpu 96` keeps 96 sequences resident instead (Measured)
    kv_dim: int = 128
    kv_pad_rows: int = 16           # row stride = kv_smax + pad: a 4096*256 B = 1 MiB stride between (layer, head) groups lands
                                    # every group on the same HBM channel set (measured 56 -> 42 us per gather); DESIGN.md 3
    with_kv: bool = True
    seed_base: int = 3000           # 1000 * config index (C3)
    max_steps: int = 4096           # uniform stream sizing
    table_seed: int = 0
    use_graph: bool = False         # True: replay one captured hipGraph per pool slot (6 kernels, fixed arguments); measured
                                    # 167 us/step vs 158 us eager on MI355X (the eager queue already runs ahead of the GPU)
    path: str = "window"            # "window": v2 kernels (32 KB rows, LDS-resident residual); "dense": v1 kernels
    rows_probs: bool = True         # windowed path: O7 emits softmax probabilities for every row (1248 workgroups in parallel) so
                                    # O8's serial per-sequence chain only copies its visited rows into LDS
    side_stream: bool = False       # O6 runs beside O7 and O10 beside O9 on a second HIP stream (they are independent: O6 needs only
                                    # the sample token, O10 only evaluate_posterior's result); fork/join with events, capturable.
                                    # Measured SLOWER on MI355X (116 vs 103 us/step): the cross-stream event waits cost more than the
                                    # two small kernels they hide; kept as an option
    direct_logs: bool = True        # eager windowed launches: O8 writes (best, accept_len, counters, bonus token) straight into the
                                    # step's log row and the following kernels read them there (per-step pointers from the host), so
                                    # the bookkeeping kernel is only launched when an image can actually end (host-side bound)
    fuse_update: bool = True        # windowed path: O9 + O10 in one launch (lantern_update_inference_inputs)
    pack_table: bool = True         # windowed path: neighbour table packed to [K, ceil8(k+1)] (lantern_pack_vq_table)
    ep_kernel: str = "nodes"        # windowed path: "nodes" = node-parallel evaluate_posterior (one workgroup per internal tree node + the
                                    # walk; B * n_internal workgroups fill the GPU), "chain" = one serial chain per sequence (epw_kernel)
    fuse_o7: bool = False           # windowed chain kernel: LANTERN_ROWS_RAW_BF16 -- no O7 launch, evaluate_posterior post-processes (CFG, top-k,
                                    # softmax) the rows its walk visits from the raw cond / uncond logits
    spec_rows: int = 0              # with fuse_o7: this many of the tree's most likely nodes (the root first) get their rows post-processed
                                    # up front, in the same launch as the candidate assembly (lantern_prepare_step); the rest on demand
    fused_prepare: bool = False     # chain kernel on raw rows with prepared rows (fuse_o7, spec_rows >= 2): the prepare stage inside the chain launch
                                    # (LANTERN_STEP_FUSED_PREPARE): two launches per group and step instead of three
    dense_one_call: bool = True     # dense path with KV slabs: the step through ONE C call (lantern_step_group.dense); False: one ctypes call per kernel
    native_step: bool = True        # eager windowed path: the whole step of all groups through ONE C call (lantern_verify_step) instead of
                                    # 4 x n_groups ctypes calls (the Python launch loop caps the stream groups at ~2 otherwise)
    leaf_workgroups: int = -1       # node kernel: -1 = by batch size (include/lantern_hip.h lantern_ep_nodes)
    merge_prepare: bool = False     # HARNESS-ONLY variant (bench extra `merged_prepare_harness_only`, never the headline): step s + 1's lantern_prepare_step
                                    # rides in step s's commit launch (lantern_step_group.prepare_next).  Its precondition -- step s + 1's cond / uncond /
                                    # ss_token are FINAL when step s's commit launches -- only holds where those rows come out of a pre-generated pool; in a
                                    # real decode loop they are the outputs of the drafter and target forwards that run AFTER commit(s)
    commit_window: int = 0          # > 0 (one-call steps, chain kernel, KV slabs, more than one group): commit turn-taking between the stream groups
                                    # (lantern_step_group.turn) with this many commits in flight at most; 0 = the groups run free (they fall into
                                    # lock-step: every group commits at the same time)
    n_groups: int = 1               # >1: the sequences are split into groups, each launched on its own HIP stream, so that one
                                    # group's latency-bound evaluate_posterior overlaps the others' bandwidth-bound kernels
                                    # (independent sequences: no ordering between groups exists)


def build_neighbour_table(device, seed: int = 0) -> torch.Tensor:
    """Setup only: table recipe of entrypoints/generate_codebook.py:53-65 on a N(0,1) [8192,256]
    codebook, via torch on the device (the HIP builder is a 'next' row)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    cb = torch.randn(K_CODES, 256, generator=g).to(device)
    d = torch.cdist(cb, cb)
    d.fill_diagonal_(float("inf"))
    idx = torch.argsort(d, dim=-1, stable=True)[:, :K_CODES - 1]
    return idx.to(torch.int16)      # uint16 bit patterns (values < 8192)


def static_rejection_levels(cand, cart_prob, best, alen, counters):
    """Per sequence: the number of tree LEVELS at which the static-tree walk rejected at least one candidate (= distinct drafter rows the
    walk reads), from the step's own inputs and verdict -- the walk is replayed on the host without its arithmetic: at level i the tried
    candidates are the distinct tokens (!= -1, cart_candidates_prob > 0: ea_model_lumina_mgpt.py:655-667) of the paths that share the accepted
    prefix, in path order, up to the accepted one (levels <= accept_len) or all of them (the level that ended the walk).  Checked against the
    kernel's own counters: the replay's rejections per sequence must equal counters[:, 2].
    cand [B,P,D] i64, cart_prob [B,P,D] f32, best / alen [B], counters [B,6]."""
    B, P, D = cand.shape
    dev = cand.device
    ar = torch.arange(B, device=dev)
    best, alen = best.long(), alen.long()
    acc = cand[ar, best]                                     # [B, D] the accepted path's tokens
    alive = torch.ones((B, P), dtype=torch.bool, device=dev)
    levels = counters[:, 0].long()
    rej_levels = torch.zeros(B, dtype=torch.int64, device=dev)
    rej_total = torch.zeros(B, dtype=torch.int64, device=dev)
    for i in range(1, D):
        tok = cand[:, :, i]
        tried = alive & (tok != -1) & (cart_prob[:, :, i] > 0)
        # distinct tokens in path order: path j counts when no earlier tried path carries the same token
        same_earlier = (tok[:, :, None] == tok[:, None, :]) & tried[:, None, :] & (torch.arange(P, device=dev)[None, :, None] > torch.arange(P, device=dev)[None, None, :])
        first = tried & ~same_earlier.any(-1)
        visited = levels >= i
        accepted_here = alen >= i
        is_acc = first & (tok == acc[:, i, None])
        # position of the accepted token among the distinct tried ones (its first carrier)
        acc_pos = torch.where(is_acc.any(-1), is_acc.int().argmax(-1), torch.full((B,), P, device=dev))
        before = first & (torch.arange(P, device=dev)[None, :] < acc_pos[:, None])
        n_rej = torch.where(accepted_here, before.sum(-1), first.sum(-1))
        n_rej = torch.where(visited, n_rej, torch.zeros_like(n_rej))
        rej_total += n_rej
        rej_levels += (n_rej > 0).long()
        alive = alive & (tok == acc[:, i, None])
    if not torch.equal(rej_total, counters[:, 2].long()):
        bad = int((rej_total != counters[:, 2].long()).sum())
        raise _lib.LanternError(f"static_rejection_levels: the replayed walk disagrees with the kernel's rejection counters in {bad} of {B} sequences")
    return rej_levels


class LuminaVerifyWorkload:
    @staticmethod
    def windowed_cfg(cfg) -> bool:
        return cfg.path == "window"

    def __init__(self, cfg: WorkloadConfig, device: torch.device, rank: int = 0):
        self.cfg, self.device, self.rank = cfg, device, rank
        B, S = cfg.n_seq, cfg.pool_steps
        if cfg.n_groups < 1 or B % cfg.n_groups:
            raise ValueError(f"n_seq={B} must be a multiple of n_groups={cfg.n_groups}")
        self.G, self.Bg = cfg.n_groups, B // cfg.n_groups
        from .drafters import choices as _choices
        tree = MC_SIM_7B_63 if cfg.tree == "mc_sim_7b_63" else getattr(_choices, cfg.tree)
        tb = ops.tree_static_build(tree)
        self.tb = tb
        self.N = N = len(tb["tree_indices"])
        self.P, self.D = P, D = tb["retrieve_indices"].shape
        ti, pos = tb["tree_indices"], tb["tree_position_ids"]
        self.R = R = int(((ti[1:] - 1) // 10).max()) + 1
        # parent node of every drafter row, level offsets of the drafter rows
        mask = tb["tree_attn_mask"]
        par = np.zeros(N, np.int64)
        for n in range(1, N):
            anc = [a for a in np.nonzero(mask[n] > 0)[0] if pos[a] == pos[n] - 1]
            par[n] = anc[0]
        par_row = np.zeros(R, np.int64)
        for n in range(1, N):
            par_row[(ti[n] - 1) // 10] = par[n]
        depth_of_row = pos[par_row]
        op_off = np.array([np.nonzero(depth_of_row == d)[0][0] for d in range(int(depth_of_row.max()) + 1)], np.int32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.d_tree_indices, self.d_retrieve = t(ti), t(tb["retrieve_indices"])
        ri = tb["retrieve_indices"].copy()
        ri[ri < 0] += N
        self.d_row_index = t(ri.astype(np.int32))
        self.d_p_idx, self.d_b_off = t(tb["p_indices"]), t(tb["b_off"])
        self.d_b_idx = t(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32))
        self.d_op_off = t(op_off)
        self.d_pos_ids = t(pos + 1)                 # tree_position_ids + 1 (ea_model_lumina_mgpt.py:559,601)
        self.table_full = build_neighbour_table(device, cfg.table_seed)        # the reference's [K, K-1] layout
        # hot-path layout: the first k+1 neighbours of every code in 16-byte aligned rows (16 MiB instead of 128 MiB)
        self.table_cols = K_CODES - 1
        self.table = self.table_full
        if cfg.pack_table and self.windowed_cfg(cfg):
            self.table_cols = min(K_CODES, -(-(cfg.lantern_k + 1) // 8) * 8)
            self.table = ops.pack_vq_table(self.table_full, self.table_cols)

        # ---------------- pools (setup, untimed; torch is fine here)
        self.cond = torch.empty((S, B, N, V), dtype=torch.bfloat16, device=device)
        self.uncond = torch.empty((S, B, N, V), dtype=torch.bfloat16, device=device)
        self.windowed = cfg.path == "window"
        if cfg.model not in ("lumina", "anole"):
            raise ValueError(f"model={cfg.model}")
        self.anole = cfg.model == "anole"
        self.o7_model = ops.MODEL_ANOLE if self.anole else ops.MODEL_LUMINA
        self.tokens_per_image = 32 * 32 if self.anole else TOKENS_PER_IMAGE          # Anole 512x512: 1024 image tokens, no newline / header rows
        self.fused_o7 = self.windowed and cfg.fuse_o7 and cfg.ep_kernel == "chain"
        self.n_spec = min(max(int(cfg.spec_rows), 0), N) if self.fused_o7 else 0
        if self.n_spec:
            # likelihood order of the nodes, the root always: the walk reaches a node when every node on its path was accepted; a level accepts
            # its r-th candidate with probability ~ a (1 - a)^r (a = 0.65: the measured first-try acceptance of this workload), so a node's
            # visit probability is the product over its path -- root, c0, c0.c0, c0.c0.c0, c1, ... for the reference's trees
            paths = {0: ()}
            for n in range(1, N):
                sib_rank = sum(1 for m_ in range(1, n) if par[m_] == par[n])
                paths[n] = paths[int(par[n])] + (sib_rank,)
            a_first = 0.65
            visit = lambda n: float(np.prod([a_first * (1.0 - a_first) ** r for r in paths[n]])) if paths[n] else 1.0
            order = sorted(range(N), key=lambda n: (-visit(n), len(paths[n]), paths[n]))
            self.spec_nodes = order[:self.n_spec]
            self.d_node_list = t(np.asarray(self.spec_nodes, np.int32))
            flags = np.zeros(N, np.uint8)
            flags[self.spec_nodes] = 1
            self.d_pre = t(flags)
        self.win_lo, self.W = IMG_LO, IMG_HI - IMG_LO
        # drafter distributions: dense [R,V] rows for the dense path; for the windowed path the pool holds what a
        # windowed drafter softmax emits, [R,W] (zero outside the image range by construction)
        self.orig_prob = torch.empty((S, B, R, self.W if self.windowed else V), dtype=torch.float32, device=device)
        self.ss_token = torch.empty((S, B, R, 10), dtype=torch.int64, device=device)
        self.ss_prob = torch.empty((S, B, R, 10), dtype=torch.float32, device=device)
        self.hidden = torch.empty((S, B, 2, N, HIDDEN), dtype=torch.bfloat16, device=device)
        gen = torch.Generator(device=device)
        d_par_row = t(par_row)
        for s in range(S):
            gen.manual_seed(cfg.seed_base + 7919 * s + 104729 * rank)
            tgt = cfg.logit_scale * torch.randn((B, N, V), generator=gen, device=device)
            unc = torch.randn((B, N, V), generator=gen, device=device)
            cnd = unc + (tgt - unc) / cfg.cfg_scale
            self.cond[s] = cnd.to(torch.bfloat16)
            self.uncond[s] = unc.to(torch.bfloat16)
            c32, u32 = self.cond[s].float(), self.uncond[s].float()
            cfgd = u32 + cfg.cfg_scale * (c32 - u32)
            dr = cfgd[:, d_par_row] + cfg.sigma * torch.randn((B, R, V), generator=gen, device=device)
            dr[..., :IMG_LO] = float("-inf")
            dr[..., IMG_HI:] = float("-inf")
            kth = torch.topk(dr, cfg.top_k, dim=-1).values[..., -1:]
            dr = dr.masked_fill(dr < kth, float("-inf"))
            op = torch.softmax(dr, dim=-1)
            self.orig_prob[s] = op[..., IMG_LO:IMG_HI] if self.windowed else op
            tok = torch.multinomial(op.view(-1, V), 10, replacement=False, generator=gen)
            self.ss_token[s] = tok.view(B, R, 10)
            self.ss_prob[s] = ops.sample_static(op.view(-1, V), tok).view(B, R, 10)
            self.hidden[s] = torch.randn((B, 2, N, HIDDEN), generator=gen, device=device).to(torch.bfloat16)
            del tgt, unc, cnd, c32, u32, cfgd, dr, op
        torch.cuda.synchronize(device)

        # ---------------- per-sequence state (device resident)
        nu = cfg.max_steps * (N - 1) + 64
        uni = np.empty((B, nu), np.float64)
        for b in range(B):
            r = random.Random(cfg.seed_base + 1000 * rank + b)      # MT19937, Python's random.random()
            uni[b] = [r.random() for _ in range(nu)]
        self.uniforms_host = uni
        self.uniforms = t(uni)
        self.n_uniforms = nu
        gen.manual_seed(cfg.seed_base + 17 + rank)
        self.u_bonus = torch.rand((cfg.max_steps, B), generator=gen, device=device, dtype=torch.float64)
        self.first_token = torch.randint(IMG_LO, IMG_HI, (B,), generator=gen, device=device)
        self.slabs: List[torch.Tensor] = []
        if cfg.with_kv:
            # prompt + header + the image tokens this run can reach (<= D per step) + the tree's rows behind them
            longest = cfg.prompt_len + 3 + min(self.tokens_per_image, cfg.max_steps * self.D) + self.N
            if cfg.kv_smax < longest:
                raise _lib.LanternError(f"kv_smax={cfg.kv_smax} rows cannot hold a sequence of this workload ({longest} rows)")
            shape = (2 * cfg.kv_layers, 1, cfg.kv_heads, cfg.kv_smax + cfg.kv_pad_rows, cfg.kv_dim)
            need = 2 * B * int(np.prod(shape)) * 2
            free, _total = torch.cuda.mem_get_info(device)
            if need + (8 << 30) > free:      # never drive the box out of memory
                raise _lib.LanternError(f"KV slabs need {need / 2**30:.0f} GiB (+8 GiB head-room) but only {free / 2**30:.0f} GiB are free: "
                                        f"lower --seqs-per-gpu ({B})")
            for _ in range(2 * B):                 # order: per group [cond slabs of its sequences..., uncond slabs...]
                self.slabs.append(torch.zeros(shape, dtype=torch.bfloat16, device=device))
            self.slab_ptrs = torch.tensor([s.data_ptr() for s in self.slabs], dtype=torch.int64, device=device)
            # slab -> sequence, local to the slab's group (the kernels get group-offset best/accept_len pointers)
            self.slab_seq = torch.arange(self.Bg, dtype=torch.int32, device=device).repeat(2 * cfg.n_groups)

        # ---------------- work buffers
        # candidates by lens parity: step s's commit reads cand[s & 1] (the accepted tokens) while -- merge_prepare -- the same launch writes step
        # s + 1's candidates into cand[(s + 1) & 1]
        self.cand2 = torch.empty((2, B, P, D), dtype=torch.int64, device=device)
        self.cart_prob = torch.empty((B, P, D), dtype=torch.float32, device=device)
        self.tree_cand = torch.empty((B, N), dtype=torch.int64, device=device)
        if self.windowed:
            self.proc = torch.empty((B, N, self.W), dtype=torch.float32, device=device)
            self.row_hot = torch.empty((B, N), dtype=torch.int32, device=device)
            self.out_tok = torch.empty(B, dtype=torch.int32, device=device)
            self.out_mass = torch.empty(B, dtype=torch.float32, device=device)
            self.sample_p = self.workspace = None
        else:
            self.proc = torch.empty((B, N, V), dtype=torch.float32, device=device)
            self.sample_p = torch.empty((B, V), dtype=torch.float32, device=device)
            self.workspace = torch.empty((B, V), dtype=torch.float32, device=device)
        self.out_hidden = torch.empty((B, 2, D, HIDDEN), dtype=torch.bfloat16, device=device)
        self.acc_tokens = torch.empty((B, D), dtype=torch.int64, device=device)
        self.log_best = torch.zeros((cfg.max_steps, B), dtype=torch.int32, device=device)
        self.log_alen = torch.zeros((cfg.max_steps, B), dtype=torch.int32, device=device)
        self.log_cnt = torch.zeros((cfg.max_steps, B, 6), dtype=torch.int32, device=device)
        self.log_token = torch.zeros((cfg.max_steps, B), dtype=torch.int64, device=device)
        # per-step staging (fixed addresses: a step is then a fixed kernel sequence, capturable into a hipGraph)
        self.st_best = torch.zeros(B, dtype=torch.int32, device=device)
        self.st_alen = torch.zeros(B, dtype=torch.int32, device=device)
        self.st_cnt = torch.zeros((B, 6), dtype=torch.int32, device=device)
        self.st_token = torch.zeros(B, dtype=torch.int64, device=device)
        self.u_cur = torch.zeros(B, dtype=torch.float64, device=device)
        self.step_dev = torch.zeros(cfg.n_groups, dtype=torch.int64, device=device)    # one device step counter per group
        self.streams = group_streams(device, cfg.n_groups) if cfg.n_groups > 1 else [None]
        self.side = [torch.cuda.Stream(device=device) for _ in range(cfg.n_groups)] if cfg.side_stream else None
        self._ev = [[torch.cuda.Event() for _ in range(4)] for _ in range(cfg.n_groups)]
        self._L = _lib.lib()
        self._ep_prm = self._make_ep_params()
        self.graphs = None
        self.ep_nodes = None
        if self.windowed and cfg.ep_kernel == "nodes":
            self.node_tables = ops.tree_node_tables(tb["retrieve_indices"], N, tb["p_indices"], tb["b_off"], op_off, device=device, b_idx=tb["b_idx"])
            nt = self.node_tables
            win0 = EpWindow()
            win0.win_len = self.W
            nbytes = int(self._L.lantern_evaluate_posterior_nodes_workspace(C.byref(self._ep_prm), C.byref(win0), nt.n_internal, 0))
            self.node_ws = torch.empty((cfg.n_groups, max(nbytes, 16)), dtype=torch.uint8, device=device)
            self.ep_nodes = []
            for g in range(cfg.n_groups):
                self.ep_nodes.append(nt.struct(self.node_ws[g].data_ptr(), nbytes, cfg.leaf_workgroups))
        self.reset_state()
        # every (pool slot, parity, group) argument block is built HERE (setup, untimed): the step loop only patches the
        # step-dependent log-row addresses, computed arithmetically from these bases
        self._bases = dict(best=self.log_best.data_ptr(), alen=self.log_alen.data_ptr(), cnt=self.log_cnt.data_ptr(),
                           tok=self.log_token.data_ptr(), ub=self.u_bonus.data_ptr())
        for slot in range(cfg.pool_steps):
            for parity in (0, 1):
                for g in range(self.G):
                    self._group_args(slot, parity, g)
        self._steps = {}
        if self.windowed and cfg.native_step and cfg.direct_logs and cfg.fuse_update and not cfg.side_stream and not cfg.use_graph:
            for slot in range(cfg.pool_steps):
                for parity in (0, 1):
                    self._steps[(slot, parity)] = self._make_step_groups(slot, parity)
        # the prepare stage inside the chain launch (LANTERN_STEP_FUSED_PREPARE): Lumina static trees on raw rows, the one-call step
        self.fused_prepare = bool(cfg.fused_prepare and self.fused_o7 and self.n_spec >= 1 and self.ep_nodes is None and self._steps
                                  and cfg.top_p >= 1.0 and cfg.n_seq // max(1, cfg.n_groups) <= 256)
        if self.fused_prepare:
            self._row_ready = torch.zeros((cfg.n_seq, self.N), dtype=torch.int32, device=device)          # zeroed once: the epochs only grow
            self._row_epoch = 0          # (its own counter, NOT the step index: reset_state() restarts the steps, and a word left by an earlier run must never read as "published")

    # -------------------------------------------------------------------------------------
    def reset_state(self):
        B, dev = self.cfg.n_seq, self.device
        cond0 = self.cfg.prompt_len + 3
        # lens[0] = current, lens[1] = next (double buffer); layout per group: [cond lens of its seqs, uncond lens]
        base = torch.cat([torch.full((self.Bg,), cond0, dtype=torch.int64), torch.full((self.Bg,), 3, dtype=torch.int64)]).repeat(self.G).to(dev)
        if hasattr(self, "len_base"):
            self.len_base.copy_(base)
        else:
            self.len_base = base
        if hasattr(self, "lens"):
            self.lens[0].copy_(base)
            self.lens[1].copy_(base)
        else:
            self.lens = [base.clone(), base.clone()]
        self.cursor = torch.zeros(B, dtype=torch.int32, device=dev) if not hasattr(self, "cursor") else self.cursor.zero_()
        if not hasattr(self, "sample_token"):
            self.sample_token = self.first_token.clone()
        else:
            self.sample_token.copy_(self.first_token)
        self.step_idx = 0
        self._prepared_for = -1           # the step whose preparation already ran inside the previous step's commit launch
        self._len_ub = 0                  # host-side upper bound of (length - base) over all sequences (direct_logs)
        self._forked = False
        if hasattr(self, "step_dev"):
            self.step_dev.zero_()
            self.u_cur.copy_(self.u_bonus[0])

    def prime(self, min_seconds: float = 0.0):
        """Setup, untimed: launch one step on every pool slot (code objects loaded, every pool page and slab touched by
        the kernels themselves), then put the per-sequence state back to step 0.  Input residency, not work.
        `min_seconds`: keep doing that for at least this long -- a GPU that sat idle while the process started (imports, pool
        generation) runs its first milliseconds at idle clocks (measured on a fresh box: 136 instead of 84 us per step over a whole
        100-step run), and a 20-step timed region is 1.7 ms long."""
        import time
        t0 = time.perf_counter()
        while True:
            for _ in range(min(self.cfg.pool_steps, self.cfg.max_steps)):
                self.step()
            self.join()
            torch.cuda.synchronize(self.device)
            self.reset_state()
            torch.cuda.synchronize(self.device)
            if time.perf_counter() - t0 >= min_seconds:
                break
        # the sequence-management ops of an image end (a step that can end an image wraps the lengths with torch ops and reads one scalar back) run once
        # here: their FIRST use loads torch's code objects for them, which cost a run that crosses the bound 15 - 18 ms inside its timed region
        # (measured: 400 timed steps 113 us per step against 68 for 300; tools/probe/turn_long.py).  No state changes: nothing is at the bound.
        for g in range(self.G):
            s0, B = g * self.Bg, self.Bg
            nxt, base = self.lens[1][2 * s0:2 * s0 + 2 * B], self.len_base[2 * s0:2 * s0 + 2 * B]
            with torch.cuda.stream(self.streams[g]) if self.streams[g] is not None else _nullctx():
                torch.where(nxt - base >= self.tokens_per_image, base, nxt, out=nxt)
        torch.cuda.synchronize(self.device)
        int((self.lens[1] - self.len_base).max().item())

    def set_lantern_delta(self, delta: float):
        """Switch between LANTERN's delta mode (<= 1) and LANTERN++'s lambda mode (> 1: tau = (delta - 1) * p(x)) on the same pools."""
        self.cfg.lantern_delta = float(delta)
        self._ep_prm.delta = float(delta)
        for arr in self._steps.values():
            for g in range(self.G):
                arr[g].ep.delta = float(delta)

    def release_kv(self):
        """Free the KV slabs (most of the footprint) once the timed loop is over."""
        self.slabs = []
        self.slab_ptrs = None
        torch.cuda.empty_cache()

    def _make_ep_params(self) -> EpParams:
        c = self.cfg
        p = EpParams()
        p.B, p.P, p.D, p.V, p.rows_per_seq = self.Bg, self.P, self.D, V, self.N
        if self.anole:              # ea_model_anole.py: no syntax shortcut; static trees zero the neighbours in q (MODE_STATIC_LG)
            p.mode, p.syntax_shortcut, p.tok_offset = ops.MODE_STATIC_LG, 0, 4
            p.img_lo, p.img_hi, p.n_syntax = IMG_LO, IMG_HI, 0
        else:
            p.mode, p.syntax_shortcut, p.tok_offset = ops.MODE_STATIC_LUMINA, 1, 4
            p.img_lo, p.img_hi, p.n_syntax = IMG_LO, IMG_HI, 4
            for i, s in enumerate((8196, 8197, 8803, 8828)):
                p.syntax[i] = s
        p.lantern, p.k, p.delta = 1, c.lantern_k, c.lantern_delta
        p.table_rows, p.table_cols = K_CODES, self.table_cols
        p.top_k, p.temperature, p.top_p = 0, 1.0, (c.top_p if self.fused_o7 else 1.0)       # Lumina filters in O7, not per level (raw rows: the chain kernel is O7)
        p.n_uniforms, p.R, p.N, p.row_index_per_seq = self.n_uniforms, self.R, self.N, 0
        return p

    def ep_buffers(self, slot: int, g: int = 0, parity: int = 0) -> EpBuffers:
        """evaluate_posterior buffers of group g (sequences [g*Bg, (g+1)*Bg)) for pool slot `slot`."""
        s0 = g * self.Bg
        at = lambda t: t[s0:].data_ptr()
        b = EpBuffers()
        b.logits, b.row_index, b.cand = at(self.proc), self.d_row_index.data_ptr(), at(self.cand2[parity])
        b.cart_prob, b.orig_prob = at(self.cart_prob), at(self.orig_prob[slot])
        b.op_off, b.p_idx, b.b_off, b.b_idx = (self.d_op_off.data_ptr(), self.d_p_idx.data_ptr(), self.d_b_off.data_ptr(),
                                               self.d_b_idx.data_ptr())
        b.tree_cand, b.nn_table = at(self.tree_cand), self.table.data_ptr()
        b.uniforms, b.cursor = at(self.uniforms), at(self.cursor)
        b.best, b.accept_len = at(self.st_best), at(self.st_alen)
        b.counters = at(self.st_cnt)
        if not self.windowed:
            b.workspace, b.sample_p = at(self.workspace), at(self.sample_p)
        return b

    def ep_window(self, g: int = 0) -> EpWindow:
        s0 = g * self.Bg
        at = lambda t: t[s0:].data_ptr()
        w = EpWindow()
        w.win_lo, w.win_len, w.row_hot = self.win_lo, self.W, at(self.row_hot)
        w.orig_prob_stride, w.orig_prob_offset = self.W, 0
        w.out_tok, w.out_mass = at(self.out_tok), at(self.out_mass)
        w.u_bonus, w.token = at(self.u_cur), at(self.st_token)
        w.rows_kind = ops.ROWS_PROBS if self.cfg.rows_probs else ops.ROWS_LOGITS
        if self.fused_o7:
            w.rows_kind = ops.ROWS_RAW_BF16                   # logits / raw_uncond / raw_seq_len are set per (slot, parity)
            w.raw_pos_ids, w.raw_pos_base = self.d_pos_ids.data_ptr(), self.cfg.prompt_len + 3
            w.raw_cfg, w.raw_top_k = self.cfg.cfg_scale, self.cfg.top_k
            if self.anole:          # no grammar rows (ea_model_anole.py:930-931 masks the non-image ids, nothing else): raw_w_latent = raw_h_latent = 0
                w.raw_w_latent, w.raw_h_latent, w.raw_newline_id, w.raw_eos_id = 0, 0, 0, 0
            else:
                w.raw_w_latent, w.raw_h_latent, w.raw_newline_id, w.raw_eos_id = W_LATENT, H_LATENT, NEWLINE, EOS
            if self.n_spec:
                w.raw_probs, w.raw_pre = at(self.proc), self.d_pre.data_ptr()
        return w

    @property
    def cand(self) -> torch.Tensor:
        """The candidates of the step that ran last ([B, P, D]; the buffers alternate with the step's parity)."""
        return self.cand2[(self.step_idx - 1) & 1] if self.step_idx > 0 else self.cand2[0]

    def cond_lens(self, parity: int) -> torch.Tensor:
        """KV lengths of the cond slabs in sequence order (the device layout is per group [cond..., uncond...])."""
        return self.lens[parity].view(self.G, 2, self.Bg)[:, 0].reshape(-1)

    def uncond_lens(self, parity: int) -> torch.Tensor:
        return self.lens[parity].view(self.G, 2, self.Bg)[:, 1].reshape(-1)

    def slab(self, b: int, j: int) -> torch.Tensor:
        """KV slab j (0 = cond, 1 = uncond) of sequence b."""
        g, l = divmod(b, self.Bg)
        return self.slabs[g * 2 * self.Bg + j * self.Bg + l]

    # -------------------------------------------------------------------------------------
    def step(self, events=None, serial=False):
        """Enqueue one verify step of every sequence.  G == 1: on the current stream (eager, or one hipGraph replay with
        cfg.use_graph).  G > 1: group g's kernels go to its own stream with NO ordering against the other groups (they
        are independent sequences), eagerly or as one graph replay per group; call join() (or synchronize the device)
        before reading results on another stream.  `events`: dict name -> (start, end) torch events around the
        HBM-heavy kernels of group 0.  `serial`: launch all groups eagerly on the current stream (clean per-kernel
        durations for the event-timed pass)."""
        i = self.step_idx
        slot = i % self.cfg.pool_steps
        use_graph = events is None and not serial and self.cfg.use_graph and self.cfg.pool_steps % 2 == 0
        if use_graph and self.graphs is None:
            self._capture_graphs()
        if events is None and not serial and self._steps:
            if self.G > 1 and not self._forked:
                self._fork()
            self._native_step(slot, i & 1)
            self.step_idx += 1
            return
        if self.G == 1:
            if use_graph:
                self.graphs[slot][0].replay()
            else:
                self._launch_step(slot, i & 1, events, 0)
        elif serial:
            if self._forked:          # (only when group streams may still hold work: a join / fork pair per step puts cross-queue waits next to the timed
                self.join()           # kernels -- with more than four hardware queues they doubled the event-timed durations)
            for g in range(self.G):
                self._launch_step(slot, i & 1, events if g == 0 else None, g)
        else:
            if not self._forked:
                self._fork()
            for g in range(self.G):
                with torch.cuda.stream(self.streams[g]):
                    if use_graph:
                        self.graphs[slot][g].replay()
                    else:
                        self._launch_step(slot, i & 1, events if g == 0 else None, g)
        self.step_idx += 1

    def _fork(self):
        """Group streams pick up after everything enqueued so far on the current stream."""
        if self.G > 1:
            cur = torch.cuda.current_stream(self.device)
            for st in self.streams:
                st.wait_stream(cur)
        self._forked = True

    def join(self):
        """The current stream waits for every group stream."""
        if self.G > 1:
            cur = torch.cuda.current_stream(self.device)
            for st in self.streams:
                cur.wait_stream(st)
        self._forked = False

    def _capture_graphs(self):
        """One graph per (pool slot, group); the lens double buffer alternates with the slot parity."""
        dev = self.device
        self.join()
        torch.cuda.synchronize(dev)
        state = (self.lens[0], self.lens[1], self.cursor, self.sample_token, self.step_dev, self.u_cur, self.log_best, self.log_alen,
                 self.log_cnt, self.log_token)
        snap = [t.clone() for t in state]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):            # warm-up launches outside capture (module load, lazy init)
            for g in range(self.G):
                self._launch_step(0, 0, None, g)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graphs = []
        for slot in range(self.cfg.pool_steps):
            per_group = []
            for g in range(self.G):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    self._launch_step(slot, slot & 1, None, g)
                per_group.append(gr)
            self.graphs.append(per_group)
        torch.cuda.synchronize(dev)
        for t, v in zip(state, snap):
            t.copy_(v)                            # capture does not execute, the warm-up did: restore the state
        torch.cuda.synchronize(dev)

    def _make_step_groups(self, slot: int, parity: int):
        """lantern_step_group array of one (pool slot, lens parity): everything but this step's output rows is fixed."""
        c = self.cfg
        arr = (StepGroup * self.G)()
        for g in range(self.G):
            A, s = self._group_args(slot, parity, g), arr[g]
            val = lambda x: None if x is None else (x.value if isinstance(x, C.c_void_p) else x)
            s.stream = 0 if self.streams[g] is None else self.streams[g].cuda_stream       # patched per step for the current stream when G == 1
            s.ss_token, s.ss_prob, s.sample_token = val(A["ss_token"]), val(A["ss_prob"]), val(A["sample_token"])
            s.tree_indices, s.retrieve = self.d_tree_indices.data_ptr(), self.d_retrieve.data_ptr()
            s.B, s.n_flat, s.N, s.P, s.D = self.Bg, self.R * 10, self.N, self.P, self.D
            s.tree_cand, s.cand, s.cart_prob = val(A["tree_cand"]), val(A["cand"]), val(A["cart_prob"])
            s.cond, s.uncond, s.dtype, s.V, s.cfg, s.model = val(A["cond"]), val(A["uncond"]), 1, V, c.cfg_scale, self.o7_model
            s.pos_ids, s.pos_base = self.d_pos_ids.data_ptr(), c.prompt_len + 3
            s.w_latent, s.h_latent = (0, 0) if self.anole else (W_LATENT, H_LATENT)          # (Anole: no grammar rows)
            s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k = IMG_LO, IMG_HI, NEWLINE, EOS, c.top_k
            s.win_lo, s.win_len, s.out_kind = self.win_lo, self.W, ops.ROWS_PROBS if c.rows_probs else ops.ROWS_LOGITS
            s.seq_len, s.out_win, s.row_hot, s.temperature, s.top_p = val(A["cur"]), (None if (self.fused_o7 and not self.n_spec) else val(A["proc"])), val(A["row_hot"]), 1.0, c.top_p
            if self.n_spec:
                s.node_list, s.n_list = self.d_node_list.data_ptr(), self.n_spec
            C.memmove(C.byref(s.ep), C.byref(self._ep_prm), C.sizeof(EpParams))
            C.memmove(C.byref(s.ep_buf), C.byref(A["ep_buf"]), C.sizeof(EpBuffers))
            C.memmove(C.byref(s.ep_win), C.byref(A["ep_win"]), C.sizeof(EpWindow))
            if self.ep_nodes is not None:
                s.nodes = C.pointer(self.ep_nodes[g])
            if c.with_kv:
                s.slab_ptrs, s.slab_seq, s.slab_prev, s.new_len = val(A["slab_ptrs"]), val(A["slab_seq"]), val(A["cur"]), val(A["nxt"])
                s.n_slabs, s.elem_bytes, s.outer, s.S_max, s.d = 2 * self.Bg, 2, 2 * c.kv_layers * c.kv_heads, c.kv_smax + c.kv_pad_rows, c.kv_dim
                s.hidden, s.out_hidden, s.accepted_tokens = val(A["hidden"]), val(A["out_hidden"]), val(A["acc_tokens"])
                s.hid_elem_bytes, s.hid_groups, s.H = 2, 2, HIDDEN
        return arr

    def _dense_group(self, slot: int, parity: int, g: int):
        """lantern_step_group of the DENSE kernel set for group g (lantern_step_group.dense): cached, only the stream and the sample token change."""
        key = (slot, parity, g)
        cache = self.__dict__.setdefault("_densecache", {})
        if key in cache:
            return cache[key][0]
        c, A, s, q = self.cfg, self._group_args(slot, parity, g), StepGroup(), _lib.StepDense()
        val = lambda x: None if x is None else (x.value if isinstance(x, C.c_void_p) else x)
        s.ss_token, s.ss_prob = val(A["ss_token"]), val(A["ss_prob"])
        s.tree_indices, s.retrieve = self.d_tree_indices.data_ptr(), self.d_retrieve.data_ptr()
        s.B, s.n_flat, s.N, s.P, s.D = self.Bg, self.R * 10, self.N, self.P, self.D
        s.tree_cand, s.cand, s.cart_prob = val(A["tree_cand"]), val(A["cand"]), val(A["cart_prob"])
        s.cond, s.uncond, s.dtype, s.V, s.cfg, s.model = val(A["cond"]), val(A["uncond"]), 1, V, c.cfg_scale, self.o7_model
        s.pos_ids, s.pos_base = self.d_pos_ids.data_ptr(), c.prompt_len + 3
        s.w_latent, s.h_latent = (0, 0) if self.anole else (W_LATENT, H_LATENT)
        s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k = IMG_LO, IMG_HI, NEWLINE, EOS, c.top_k
        s.seq_len, s.temperature, s.top_p = val(A["cur"]), 1.0, c.top_p
        C.memmove(C.byref(s.ep), C.byref(self._ep_prm), C.sizeof(EpParams))
        C.memmove(C.byref(s.ep_buf), C.byref(A["ep_buf"]), C.sizeof(EpBuffers))
        s.ep_buf.best, s.ep_buf.accept_len, s.ep_buf.counters = val(A["st_best"]), val(A["st_alen"]), val(A["st_cnt"])
        s.slab_ptrs, s.slab_seq, s.slab_prev, s.new_len = val(A["slab_ptrs"]), val(A["slab_seq"]), val(A["cur"]), val(A["nxt"])
        s.n_slabs, s.elem_bytes, s.outer, s.S_max, s.d = 2 * self.Bg, 2, 2 * c.kv_layers * c.kv_heads, c.kv_smax + c.kv_pad_rows, c.kv_dim
        s.hidden, s.out_hidden, s.accepted_tokens = val(A["hidden"]), val(A["out_hidden"]), val(A["acc_tokens"])
        s.hid_elem_bytes, s.hid_groups, s.H = 2, 2, HIDDEN
        q.logits, q.sample_p, q.u_bonus, q.token = val(A["proc"]), val(A["sample_p"]), val(A["u_cur"]), val(A["st_token"])
        s.dense = C.pointer(q)
        cache[key] = (s, q)
        return s

    def _native_step(self, slot: int, parity: int):
        """One C call enqueues O6 -> O7 -> O8 -> O9 + O10 of every group (each on its stream); results land in the step's log row."""
        c = self.cfg
        step = self.step_idx
        if step >= c.max_steps:
            raise _lib.LanternError(f"step {step} >= max_steps {c.max_steps}: size the logs for the run")
        arr, bs = self._steps[(slot, parity)], self._bases
        cur_stream = torch.cuda.current_stream().cuda_stream if self.G == 1 else None
        for g in range(self.G):
            s = arr[g]
            e = step * c.n_seq + g * self.Bg
            s.stream = cur_stream if cur_stream is not None else self.streams[g].cuda_stream      # (the per-kernel path borrows these blocks)
            if step > 0:
                s.sample_token = bs["tok"] + 8 * (e - c.n_seq)
            else:
                s.sample_token = self.sample_token[g * self.Bg:].data_ptr()
            s.ep_buf.best, s.ep_buf.accept_len, s.ep_buf.counters = bs["best"] + 4 * e, bs["alen"] + 4 * e, bs["cnt"] + 24 * e
            s.ep_win.u_bonus, s.ep_win.token = bs["ub"] + 8 * e, bs["tok"] + 8 * e
        # the next step's preparation inside this step's commit launch: whenever the next step exists in the logs and no image can end in this step
        # (an image end rewrites the lengths on the host's side of the stream: those steps prepare themselves, as before)
        merge = (c.merge_prepare and c.with_kv and self.n_spec > 0 and step + 1 < c.max_steps and self._len_ub + 2 * self.D < self.tokens_per_image)
        nxt_arr = self._steps[((step + 1) % c.pool_steps, parity ^ 1)] if merge else None
        turns = c.commit_window > 0 and c.with_kv and self.G > 1 and self.ep_nodes is None
        fused = self.fused_prepare and not merge
        if fused:
            self._row_epoch += 1
        if turns and not hasattr(self, "_turn"):
            self._turn = torch.zeros(_lib.TURN_WORDS(self.G), dtype=torch.int64, device=self.device)          # never reset: tickets and epochs only grow
            self._turn_step = 0
        for g in range(self.G):
            s = arr[g]
            if turns:          # ticket = (commit launches so far) = turn_step * G + g; at most commit_window commits in flight
                s.turn, s.turn_group, s.turn_groups = self._turn.data_ptr(), g, self.G
                s.turn_wait = self._turn_step * self.G + g - (c.commit_window - 1)
            else:
                s.turn = None
            s.flags = _lib.STEP_PREPARED if self._prepared_for == step else 0
            if fused:
                s.flags = _lib.STEP_FUSED_PREPARE
                s.row_ready, s.row_epoch = self._row_ready[g * self.Bg:].data_ptr(), self._row_epoch
            if merge:
                n_ = nxt_arr[g]
                n_.sample_token = bs["tok"] + 8 * (step * c.n_seq + g * self.Bg)          # this step's bonus tokens = the next step's roots
                s.prepare_next = C.addressof(n_)
            else:
                s.prepare_next = None
        self._prepared_for = step + 1 if merge else -1
        check(self._L.lantern_verify_step(arr, self.G), "verify_step")
        if turns:
            self._turn_step += 1
        if not c.with_kv:
            for g in range(self.G):
                s0, B = g * self.Bg, self.Bg
                with torch.cuda.stream(self.streams[g]) if self.streams[g] is not None else _nullctx():
                    torch.add(self.lens[parity][2 * s0:2 * s0 + 2 * B], (self.log_alen[step, s0:s0 + B] + 1).repeat(2), out=self.lens[parity ^ 1][2 * s0:2 * s0 + 2 * B])
        # sequence management (not the hot path): an image can only end once the host-side bound says so
        self._len_ub += self.D
        if self._len_ub >= self.tokens_per_image:
            for g in range(self.G):
                s0, B = g * self.Bg, self.Bg
                with torch.cuda.stream(self.streams[g]) if self.streams[g] is not None else _nullctx():
                    nxt, base = self.lens[parity ^ 1][2 * s0:2 * s0 + 2 * B], self.len_base[2 * s0:2 * s0 + 2 * B]
                    torch.where(nxt - base >= self.tokens_per_image, base, nxt, out=nxt)
            # a device-wide synchronise, then the scalar: NOT join() + .item() on the current stream -- with hundreds of steps queued on the group streams the
            # cross-stream event waits of a join made the queued steps drain at ~110 us instead of ~70 us per step (tools/probe/turn_long.py: 22 ms for the
            # read-back behind 300 queued steps against 10 ms of queued work)
            torch.cuda.synchronize(self.device)
            self._len_ub = int((self.lens[parity ^ 1] - self.len_base).max().item())

    def _group_args(self, slot: int, parity: int, g: int):
        """Raw pointers of group g's slice of every buffer (cached: the addresses never change)."""
        key = (slot, parity, g)
        cache = self.__dict__.setdefault("_argcache", {})
        if key in cache:
            return cache[key]
        s0, Bg, N = g * self.Bg, self.Bg, self.N
        vp = C.c_void_p
        at = lambda t, o=s0: vp(t[o:].data_ptr())
        cur, nxt = self.lens[parity], self.lens[parity ^ 1]
        a = dict(
            ss_token=at(self.ss_token[slot]), ss_prob=at(self.ss_prob[slot]), sample_token=at(self.sample_token), tree_cand=at(self.tree_cand),
            cand=at(self.cand2[parity]), cart_prob=at(self.cart_prob), cond=at(self.cond[slot]), uncond=at(self.uncond[slot]), proc=at(self.proc),
            row_hot=at(self.row_hot) if self.windowed else None, cur=at(cur, 2 * s0), nxt=at(nxt, 2 * s0), len_base=at(self.len_base, 2 * s0),
            st_best=at(self.st_best), st_alen=at(self.st_alen), st_cnt=at(self.st_cnt), st_token=at(self.st_token),
            hidden=at(self.hidden[slot]), out_hidden=at(self.out_hidden), acc_tokens=at(self.acc_tokens),
            sample_p=None if self.windowed else at(self.sample_p), u_cur=at(self.u_cur),
            step_dev=at(self.step_dev, g), log_best=vp(self.log_best[:, s0:].data_ptr()), log_alen=vp(self.log_alen[:, s0:].data_ptr()),
            log_cnt=vp(self.log_cnt[:, s0:].data_ptr()), log_token=vp(self.log_token[:, s0:].data_ptr()),
            u_bonus_g=vp(self.u_bonus[:, s0:].data_ptr()),
            slab_ptrs=at(self.slab_ptrs, 2 * s0) if self.cfg.with_kv else None, slab_seq=at(self.slab_seq, 2 * s0) if self.cfg.with_kv else None,
            ep_buf=self.ep_buffers(slot, g, parity), ep_win=self.ep_window(g) if self.windowed else None)
        if self.fused_o7:
            a["ep_buf"].logits = a["cond"].value
            a["ep_win"].raw_uncond, a["ep_win"].raw_seq_len = a["uncond"].value, a["cur"].value
        cache[key] = a
        return a

    def _arm(self, events, name):
        """Kernel-only timing: the next launch records (start, stop) at kernel begin / end (lantern_profile_next_launch);
        kernel sets without the hook (dense path) fall back to a hipEventRecord bracket."""
        e0, e1 = events[name]
        if self.windowed:
            check(self._L.lantern_profile_next_launch(C.c_void_p(e0.cuda_event), C.c_void_p(e1.cuda_event)), "profile_next_launch")
        else:
            e0.record()

    def _disarm(self, events, name):
        if not self.windowed:
            events[name][1].record()

    def _launch_step(self, slot: int, parity: int, events, g: int = 0):
        c, L = self.cfg, self._L
        B, N, P, D = self.Bg, self.N, self.P, self.D
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        A = self._group_args(slot, parity, g)
        vp = C.c_void_p

        main = torch.cuda.current_stream()
        side = self.side[g] if (self.side is not None and not events) else None
        ev = self._ev[g]
        st_side = st
        if side is not None:                      # fork
            ev[0].record(main)
            side.wait_event(ev[0])
            st_side = C.c_void_p(side.cuda_stream)
        # where this step's results live: the staging buffers (+ the bookkeeping kernel copies them into the logs), or -- eager
        # windowed launches -- directly the step's log row
        direct = self.windowed and c.direct_logs and self.graphs is None and not torch.cuda.is_current_stream_capturing()
        if direct:
            step = self.step_idx
            if step >= c.max_steps:
                raise _lib.LanternError(f"step {step} >= max_steps {c.max_steps}: size the logs for the run")
            s0 = g * self.Bg
            e = step * c.n_seq + s0                      # element offset of this group's row in the [steps, n_seq(, 6)] logs
            bs = self._bases
            p_best, p_alen, p_cnt, p_tok = vp(bs["best"] + 4 * e), vp(bs["alen"] + 4 * e), vp(bs["cnt"] + 24 * e), vp(bs["tok"] + 8 * e)
            p_sample = A["sample_token"] if step == 0 else vp(bs["tok"] + 8 * (e - c.n_seq))
            eb, ew = A["ep_buf"], A["ep_win"]
            eb.best, eb.accept_len, eb.counters = p_best.value, p_alen.value, p_cnt.value
            ew.u_bonus, ew.token = bs["ub"] + 8 * e, p_tok.value
        else:
            p_best, p_alen, p_sample = A["st_best"], A["st_alen"], A["sample_token"]
            if self.windowed:                     # the cached structs may carry a previous direct step's pointers
                eb, ew = A["ep_buf"], A["ep_win"]
                eb.best, eb.accept_len, eb.counters = A["st_best"].value, A["st_alen"].value, A["st_cnt"].value
                ew.u_bonus, ew.token = A["u_cur"].value, A["st_token"].value
        # the dense kernel set as ONE call (lantern_step_group.dense): O6 -> O7 over all rows at the full vocabulary -> the dense evaluate_posterior -> the bonus
        # draw -> the commit launch; the per-kernel form below stays for the timing pass (events), the side-stream variant and workloads without KV slabs
        if (not self.windowed) and c.dense_one_call and c.with_kv and side is None and not events:
            sg = self._dense_group(slot, parity, g)
            sg.stream, sg.sample_token = st.value, p_sample.value
            check(L.lantern_verify_step(C.byref(sg), 1), "verify_step")
        else:
            one_launch = self.fused_prepare and side is None
            if one_launch:                # the chain launch with its prepare stage inside, alone (no commit behind it): a copy of the step's block without slabs / turn
                tmp = StepGroup()
                C.memmove(C.byref(tmp), C.byref(self._steps[(slot, parity)][g]), C.sizeof(StepGroup))
                C.memmove(C.byref(tmp.ep_buf), C.byref(A["ep_buf"]), C.sizeof(EpBuffers))
                C.memmove(C.byref(tmp.ep_win), C.byref(A["ep_win"]), C.sizeof(EpWindow))
                tmp.stream, tmp.sample_token, tmp.slab_ptrs, tmp.turn, tmp.prepare_next = st.value, p_sample.value, None, None, None
                self._row_epoch += 1
                tmp.flags, tmp.row_ready, tmp.row_epoch = _lib.STEP_FUSED_PREPARE, self._row_ready[g * self.Bg:].data_ptr(), self._row_epoch
                if events:
                    self._arm(events, "evaluate_posterior")
                check(L.lantern_verify_step(C.byref(tmp), 1), "verify_step (fused prepare)")
            elif self.n_spec:             # candidates + the likely rows in one launch (the step's own argument block, this step's sample token)
                sg = self._steps[(slot, parity)][g]
                sg.stream, sg.sample_token = st.value, p_sample.value
                if events:
                    self._arm(events, "cfg_mask_topk")
                check(L.lantern_prepare_step(C.byref(sg)), "prepare_step")
            # O6 candidate assembly (side stream: only needs the sample token)
            else:
                check(L.lantern_gather_candidates(A["ss_token"], A["ss_prob"], p_sample, vp(self.d_tree_indices.data_ptr()),
                                              vp(self.d_retrieve.data_ptr()), B, self.R * 10, N, P, D, A["tree_cand"], A["cand"], A["cart_prob"],
                                              st_side), "gather_candidates")
            if side is not None:
                ev[1].record(side)
            # O7 CFG + Lumina position mask + top-k, positions from the device-side lengths
            if events and not self.fused_o7:
                self._arm(events, "cfg_mask_topk")
            if one_launch:
                pass
            elif self.fused_o7:
                pass                                  # evaluate_posterior reads the raw logits itself
            elif self.windowed:
                check(L.lantern_cfg_mask_topk_window(A["cond"], A["uncond"], 1, B * N, V, C.c_float(c.cfg_scale), self.o7_model,
                                                     vp(self.d_pos_ids.data_ptr()), C.c_int64(c.prompt_len + 3), W_LATENT, H_LATENT, IMG_LO, IMG_HI,
                                                     NEWLINE, EOS, c.top_k, A["cur"], N, self.win_lo, self.W, A["proc"], A["row_hot"],
                                                     ops.ROWS_PROBS if c.rows_probs else ops.ROWS_LOGITS, C.c_float(1.0), C.c_float(c.top_p), st),
                      "cfg_mask_topk_window")
            else:
                check(L.lantern_cfg_mask_topk(A["cond"], A["uncond"], 1, B * N, V, C.c_float(c.cfg_scale), self.o7_model,
                                              vp(self.d_pos_ids.data_ptr()), C.c_int64(c.prompt_len + 3), W_LATENT, H_LATENT, IMG_LO, IMG_HI,
                                              NEWLINE, EOS, c.top_k, A["cur"], N, A["proc"], st), "cfg_mask_topk")
            if events and not one_launch:
                if not self.fused_o7:
                    self._disarm(events, "cfg_mask_topk")
                self._arm(events, "evaluate_posterior")
            if side is not None:
                main.wait_event(ev[1])                # O8 needs the candidates
            # O8 (windowed: the bonus token is drawn in the kernel epilogue)
            if one_launch:
                pass
            elif self.windowed and self.ep_nodes is not None:
                check(L.lantern_evaluate_posterior_nodes(C.byref(self._ep_prm), C.byref(A["ep_buf"]), C.byref(A["ep_win"]),
                                                         C.byref(self.ep_nodes[g]), st), "evaluate_posterior_nodes")
            elif self.windowed:
                check(L.lantern_evaluate_posterior_window(C.byref(self._ep_prm), C.byref(A["ep_buf"]), C.byref(A["ep_win"]), st),
                      "evaluate_posterior_window")
            else:
                check(L.lantern_evaluate_posterior(C.byref(self._ep_prm), C.byref(A["ep_buf"]), st), "evaluate_posterior")
            if events:
                self._disarm(events, "evaluate_posterior")
            if side is not None:
                ev[2].record(main)
                side.wait_event(ev[2])
            fused = c.with_kv and self.windowed and c.fuse_update and side is None
            if not fused:
                # O10 accepted hidden + token append (+ bonus token on the dense path); side stream: beside the KV gather
                check(L.lantern_accept_gather(A["hidden"], 2, B, 2, N, HIDDEN, vp(self.d_retrieve.data_ptr()), 0, P, D, A["cand"], p_best,
                                              p_alen, A["sample_p"], V, None if self.windowed else A["u_cur"], A["out_hidden"],
                                              A["acc_tokens"], None if self.windowed else A["st_token"], st_side), "accept_gather")
            if side is not None:
                ev[3].record(side)
            # O9 KV gather: both slabs of every sequence of the group in one launch (fused: + the O10 copy, as the reference's
            # update_inference_inputs does both)
            if c.with_kv:
                if events:
                    self._arm(events, "kv_gather")
                if fused:
                    check(L.lantern_update_inference_inputs(A["slab_ptrs"], A["slab_seq"], A["cur"], 2 * B, 2, C.c_int64(2 * c.kv_layers * c.kv_heads),
                                                            C.c_int64(c.kv_smax + c.kv_pad_rows), C.c_int64(c.kv_dim), vp(self.d_retrieve.data_ptr()),
                                                            0, P, D, p_best, p_alen, A["nxt"], A["hidden"], 2, B, 2, N, HIDDEN,
                                                            A["cand"], A["out_hidden"], A["acc_tokens"], st), "update_inference_inputs")
                else:
                    check(L.lantern_kv_gather(A["slab_ptrs"], A["slab_seq"], A["cur"], 2 * B, 2, C.c_int64(2 * c.kv_layers * c.kv_heads),
                                              C.c_int64(c.kv_smax + c.kv_pad_rows), C.c_int64(c.kv_dim), vp(self.d_retrieve.data_ptr()), 0, P, D,
                                              p_best, p_alen, A["nxt"], st), "kv_gather")
                if events:
                    self._disarm(events, "kv_gather")
            else:
                s0 = g * self.Bg
                torch.add(self.lens[parity][2 * s0:2 * s0 + 2 * B], ((self.log_alen[self.step_idx, s0:s0 + B] if direct else self.st_alen[s0:s0 + B]) + 1).repeat(2), out=self.lens[parity ^ 1][2 * s0:2 * s0 + 2 * B])
        if side is not None:
            main.wait_event(ev[3])                # join (the dense path's bonus token feeds the bookkeeping below)
        # harness bookkeeping (sequence management, not the hot path): logs, next sample token, image wrap-around, step counter
        if direct:
            # everything is already where it belongs; only the end of an image needs the kernel, and lengths grow by at most D
            # per step: skip it while no sequence can have reached the image length
            if g == 0:
                self._len_ub += D
            if self._len_ub >= self.tokens_per_image:
                s0 = g * self.Bg
                nxt, base = self.lens[parity ^ 1][2 * s0:2 * s0 + 2 * B], self.len_base[2 * s0:2 * s0 + 2 * B]
                torch.where(nxt - base >= self.tokens_per_image, base, nxt, out=nxt)
                if g == self.G - 1:             # refresh the bound: one host sync every few hundred steps
                    self.join()
                    self._len_ub = int((self.lens[parity ^ 1] - self.len_base).max().item())
            return
        check(L.lantern_harness_advance(B, 2 * B, c.n_seq, C.c_int64(self.tokens_per_image), C.c_int64(c.max_steps),
                                        C.c_int64(-1 if torch.cuda.is_current_stream_capturing() or self.graphs is not None else self.step_idx),
                                        A["step_dev"], A["st_best"],
                                        A["st_alen"], A["st_cnt"], A["st_token"], A["log_best"], A["log_alen"], A["log_cnt"], A["log_token"],
                                        A["sample_token"], A["nxt"], A["len_base"], A["u_bonus_g"], A["u_cur"], st), "harness_advance")

    # -------------------------------------------------------------------------------------
    def sync(self):
        """Everything submitted so far has run (worker-thread enqueues included)."""
        self.join()
        torch.cuda.synchronize(self.device)

    def accepted_tokens(self, i0: int, i1: int) -> int:
        self.sync()
        return int((self.log_alen[i0:i1].to(torch.int64) + 1).sum().item())

    # Byte accounting.  `group`: None = all sequences (one whole step), g = the sequences of group g (one LAUNCH when G > 1).
    def _seqs(self, group):
        return slice(None) if group is None else slice(group * self.Bg, (group + 1) * self.Bg)

    def ep_algorithmic_bytes(self, i0: int, i1: int, group=None) -> float:
        """SURVEY 8d contract figure summed over steps [i0,i1):
        L*V*4 + T*k*6 + R*(k+1)*4 + V*4 (+ V*4 when the final row is a fresh softmax)."""
        return self.ep_algorithmic_bytes_from(self.log_cnt[i0:i1, self._seqs(group)])

    def ep_algorithmic_bytes_from(self, cnt) -> float:
        c = cnt.to(torch.float64).reshape(-1, 6)
        k = self.cfg.lantern_k
        Lv, T, Rj, fresh = c[:, 0].sum(), c[:, 1].sum(), c[:, 2].sum(), (1 - c[:, 4]).sum()
        n = c.shape[0]
        return float(Lv * V * 4 + T * k * 6 + Rj * (k + 1) * 4 + n * V * 4 + fresh * V * 4)

    def ep_window_bytes(self, i0: int, i1: int, group=None) -> float:
        """HBM bytes the windowed kernel must move: per visited level and per fresh final row one window row
        (W*4); per tried candidate k table ids (k*2); per rejection one drafter window row (W*4, static trees).  The
        neighbour gathers, zeroing, scans and the bonus-token draw run in LDS."""
        return self.ep_window_bytes_from(self.log_cnt[i0:i1, self._seqs(group)])

    def ep_window_bytes_from(self, cnt, rejection_levels=None) -> float:
        """`rejection_levels` (static_rejection_levels): the drafter row is priced once per LEVEL with a rejection -- every candidate tried at a
        level hangs off the same parent, hence the same original_prob row (ea_model_lumina_mgpt.py:697), and the re-reads of the 2nd, 3rd...
        rejection are L2 hits, not HBM traffic (round 4's PMC pass: 0.79 x the per-rejection model).  None: once per rejection (upper bound)."""
        c = cnt.to(torch.float64).reshape(-1, 6)
        k, W = self.cfg.lantern_k, self.W
        Lv, T, Rj, fresh = c[:, 0].sum(), c[:, 1].sum(), c[:, 2].sum(), (1 - c[:, 4]).sum()
        rows = Rj if rejection_levels is None else rejection_levels.to(torch.float64).sum()
        return float((Lv + fresh) * W * 4 + T * k * 2 + rows * W * 4 + Rj * 2)

    def o7_algorithmic_bytes(self, n_launches: int, group=None) -> float:
        per_row = self.W * (2 * 2 + 4) if self.windowed else V * (2 * 2 + 4)
        return float(n_launches) * (self.cfg.n_seq if group is None else self.Bg) * self.N * per_row

    def kv_algorithmic_bytes(self, i0: int, i1: int, group=None) -> float:
        c = self.cfg
        per_pos = 2 * c.kv_layers * c.kv_heads * c.kv_dim * 2      # bytes per position per slab
        moved = (self.log_alen[i0:i1, self._seqs(group)].to(torch.float64) + 1).sum().item() * 2   # two slabs per sequence
        return float(2 * moved * per_pos)                            # read + write

    def kv_moved_bytes(self, i0: int, i1: int, group=None) -> float:
        """Bytes kv_gather really moves: accepted rows whose tree slot is not already their final position
        (retrieve[best, t] != t); the root and accepted first children stay where the target forward wrote them."""
        c = self.cfg
        per_pos = 2 * c.kv_layers * c.kv_heads * c.kv_dim * 2
        sq = self._seqs(group)
        ret = self.d_retrieve.reshape(self.P, self.D)[self.log_best[i0:i1, sq].long()]              # [steps, B, D]
        t = torch.arange(self.D, device=ret.device)
        live = t[None, None, :] <= self.log_alen[i0:i1, sq].long()[..., None]
        moved = ((ret != t) & live).sum().item() * 2
        return float(2 * moved * per_pos)

    def check_status(self, i0: int, i1: int):
        self.sync()
        st = self.log_cnt[i0:i1, :, 5]
        if int(st.abs().sum().item()) != 0:
            bad = torch.nonzero(st)[0].tolist()
            raise _lib.LanternError(f"evaluate_posterior status {int(st[bad[0], bad[1]])} at step {i0 + bad[0]} seq {bad[1]}")


# =====================================================================================================================
# The dynamic-tree (EAGLE-2, eagle_version 2) half of config C3: a different tree per sequence and step.
#
#   O4 lantern_tree_dynamic_finalize -> O6 lantern_gather_candidates_dynamic -> O7 lantern_cfg_mask_topk_window (N = 59 rows per
#   sequence, per-node positions) -> O8 lantern_evaluate_posterior_window (LANTERN_MODE_DYNAMIC, per-sequence row maps, ragged
#   paths) -> O9 + O10 lantern_update_inference_inputs (per-sequence retrieve rows)
#
# with every buffer resident and every per-step index on the device, like the static workload above.  Reference:
# models/drafters/cnets_lumina_mgpt.py:1229-1393 (topK_genrate), models/ea_model_lumina_mgpt.py:610-726 (eagle_version 2 branch).
@dataclass
class DynamicConfig:
    model: str = "lumina"           # "lumina": the dynamic half of C3; "llamagen": BASELINE C2 -- LlamaGen + EAGLE, standard (non-relaxed) verify:
                                    # V = 16384 (the whole vocabulary is the window), no CFG grammar, lantern off, HF processors T = 1 / top_k 2000
                                    # (ea_model_llamagen.py:709-787, :930), drafter depth 4, 256 tokens per image, LlamaGen-B KV geometry via kv_*
    n_seq: int = 64
    pool_steps: int = 4
    top_k: int = 10                 # drafter top_k (generate_images.py: drafter_top_k 10)
    depth: int = 5                  # Lumina drafter depth (ea_model_lumina_mgpt.py:355-357)
    total_tokens: int = 58          # total_token 59 - 1 (cnets_lumina_mgpt.py: self.total_tokens = total_tokens - 1) -> N = 59 nodes
    lantern_k: int = 1000
    lantern_delta: float = 0.1
    cfg_scale: float = 3.0
    logit_top_k: int = 2000
    top_p: float = 1.0              # < 1: TopPLogitsWarper in front of the top-k (see WorkloadConfig.top_p)
    prompt_len: int = 64
    kv_layers: int = 32
    kv_heads: int = 32
    kv_smax: int = 4096
    kv_dim: int = 128
    kv_pad_rows: int = 16
    with_kv: bool = True
    seed: int = 3700
    max_steps: int = 256
    plausible: float = 8.0          # drafted tokens get target logits in [plausible - 2, plausible]: the walk accepts a few levels
    fuse_o7: bool = False           # LANTERN_ROWS_RAW_BF16 with per-sequence positions: no O7 launch over all N rows, evaluate_posterior
                                    # post-processes the rows its walk visits (alen + 1 of the 59)
    commit_window: int = 0          # > 0: commit turn-taking between the stream groups (see WorkloadConfig.commit_window)
    spec_rows: int = 2              # with fuse_o7: rows post-processed up front beside the tree build (lantern_prepare_step): 1 = the root, 2 = + node 1
                                    # (the drafter's best first token, at depth 1 in every EAGLE-2 tree); 0 = every row on demand
    n_groups: int = 1               # >1: stream groups, as in WorkloadConfig (n_seq must divide)
    native_step: bool = True        # the whole step of all groups through ONE C call (lantern_verify_step with lantern_step_dynamic blocks);
                                    # False: one ctypes call per kernel and group


class DynamicVerifyWorkload:
    def __init__(self, cfg: DynamicConfig, device: torch.device):
        self.cfg, self.device = cfg, device
        B, S, TK, DP, TT = cfg.n_seq, cfg.pool_steps, cfg.top_k, cfg.depth, cfg.total_tokens
        self.N = N = TT + 1
        self.P, self.D = N, DP + 2
        self.lg = cfg.model == "llamagen"
        self.G = G = max(1, cfg.n_groups)
        if B % G:
            raise ValueError(f"n_seq={B} is not a multiple of n_groups={G}")
        self.Bg = Bg = B // G
        self.streams = group_streams(device, G) if G > 1 else [None]
        self._forked = False
        if cfg.model not in ("lumina", "llamagen"):
            raise ValueError(f"model={cfg.model}")
        # vocabulary, image-token window, hidden width, tokens per image of the model
        V, IMG_LO, IMG_HI = (16384, 0, 16384) if self.lg else (globals()["V"], globals()["IMG_LO"], globals()["IMG_HI"])
        HIDDEN = 768 if self.lg else globals()["HIDDEN"]
        self.V, self.lo, self.hi, self.hidden_w = V, IMG_LO, IMG_HI, HIDDEN
        self.tokens_per_image = 256 if self.lg else TOKENS_PER_IMAGE
        self.W, self.win_lo = IMG_HI - IMG_LO, IMG_LO
        g = torch.Generator(device=device).manual_seed(cfg.seed)
        W = self.W

        def img_logits(*shape):
            x = torch.full((*shape, V), float("-inf"), device=device)
            x[..., IMG_LO:IMG_HI] = 4.0 * torch.randn((*shape, W), generator=g, device=device)
            kth = torch.topk(x, cfg.logit_top_k, dim=-1).values[..., -1:]
            return x.masked_fill(x < kth, float("-inf"))

        self.lantern = not self.lg
        if self.lantern:
            table_full = build_neighbour_table(device, 0)
            self.table_full = table_full
            self.table_cols = min(K_CODES, -(-(cfg.lantern_k + 1) // 8) * 8)
            self.table = ops.pack_vq_table(table_full, self.table_cols)
        else:
            self.table_full = self.table = None
            self.table_cols = 0
        self.pools = []
        for s in range(S):          # the drafter side of a step (setup, untimed): O3 per depth on random rows, then one O4 to learn the shapes
            ti, cu, ci, sc = ops.expand_dynamic(img_logits(B, 1), None, TK)
            sl, tl, pl = [cu.reshape(B, -1)], [ti.reshape(B, -1)], [torch.zeros((B, 1), dtype=torch.int64, device=device)]
            cs = torch.arange(TK, device=device).expand(B, TK)
            for d in range(DP):
                pl.append(cs + 1 + TK * TK * max(0, d - 1) + (TK if d > 0 else 0))
                ti, cu, ci, sc = ops.expand_dynamic(img_logits(B, TK), sc, TK)
                cs = ci
                sl.append(cu.reshape(B, -1)); tl.append(ti.reshape(B, -1))
            scores, tokens, parents = torch.cat(sl, 1).contiguous(), torch.cat(tl, 1).contiguous(), torch.cat(pl, 1).contiguous()
            sample = torch.randint(IMG_LO, IMG_HI, (B,), generator=g, device=device)
            draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(scores, tokens, parents, sample, TK, TT)
            cond = (2.0 * torch.randn((B, N, V), generator=g, device=device)).to(torch.bfloat16)
            unc = torch.randn((B, N, V), generator=g, device=device).to(torch.bfloat16)
            r6 = ret[:, :, :self.D]
            par, ch = r6[:, :, :-1], r6[:, :, 1:]
            ok = ch >= 0
            bi = torch.arange(B, device=device)[:, None, None].expand_as(ch)[ok]
            tok = draft.gather(1, ch.clamp(min=0).reshape(B, -1)).reshape(ch.shape)[ok]
            # drafted tokens plausible under the target (cfg(cond, unc) ~ cond for large values)
            cond[bi, par[ok], tok] = (cfg.plausible - 2.0 * torch.rand(bi.shape, generator=g, device=device)).to(torch.bfloat16)
            hidden = torch.randn((B, 2, N, HIDDEN), generator=g, device=device).to(torch.bfloat16)
            self.pools.append(dict(scores=scores, tokens=tokens, parents=parents, cond=cond, unc=unc, hidden=hidden))
        self.n_scores, self.n_parents = self.pools[0]["scores"].shape[1], self.pools[0]["parents"].shape[1]
        # ---- per-sequence state and work buffers (device resident)
        self.uniforms = torch.rand((B, 64 * (cfg.max_steps + 8)), generator=g, device=device, dtype=torch.float64)
        self.cursor = torch.zeros(B, dtype=torch.int32, device=device)
        self.u_bonus = torch.rand((cfg.max_steps, B), generator=g, device=device, dtype=torch.float64)
        self.first_token = torch.randint(IMG_LO, IMG_HI, (B,), generator=g, device=device)
        i64 = lambda *sh: torch.empty(sh, dtype=torch.int64, device=device)
        self.draft, self.pos, self.ret = i64(B, N), i64(B, N), i64(B, N, N)
        self.mask = torch.empty((B, N, N), dtype=torch.float32, device=device)
        self.nleaf = torch.empty(B, dtype=torch.int32, device=device)
        self.mdepth = torch.empty(B, dtype=torch.int32, device=device)
        self.cand, self.ret_pd, self.pos_abs = i64(B, self.P, self.D), i64(B, self.P, self.D), i64(B, N)
        self.row_index = torch.empty((B, self.P, self.D), dtype=torch.int32, device=device)
        self.win = torch.empty((B, N, W), dtype=torch.float32, device=device)
        self.hot = torch.empty((B, N), dtype=torch.int32, device=device)
        self.out_hidden = torch.empty((B, 2, self.D, HIDDEN), dtype=torch.bfloat16, device=device)
        self.acc_tokens = i64(B, self.D)
        self.out_tok = torch.empty(B, dtype=torch.int32, device=device)
        self.out_mass = torch.empty(B, dtype=torch.float32, device=device)
        self.log_best = torch.zeros((cfg.max_steps, B), dtype=torch.int32, device=device)
        self.log_alen = torch.zeros((cfg.max_steps, B), dtype=torch.int32, device=device)
        self.log_cnt = torch.zeros((cfg.max_steps, B, 6), dtype=torch.int32, device=device)
        self.log_token = torch.zeros((cfg.max_steps, B), dtype=torch.int64, device=device)
        # slab / length order: G blocks of [conditional slabs of the group's Bg sequences | their unconditional slabs]
        base = torch.cat([torch.full((Bg,), cfg.prompt_len + 3, dtype=torch.int64), torch.full((Bg,), 3, dtype=torch.int64)]).repeat(G).to(device)
        self.len_base, self.lens = base, [base.clone(), base.clone()]
        self.slabs: List[torch.Tensor] = []
        if cfg.with_kv:
            shape = (2 * cfg.kv_layers, 1, cfg.kv_heads, cfg.kv_smax + cfg.kv_pad_rows, cfg.kv_dim)
            need = 2 * B * int(np.prod(shape)) * 2
            free, _t = torch.cuda.mem_get_info(device)
            if need + (8 << 30) > free:
                raise _lib.LanternError(f"KV slabs need {need / 2**30:.0f} GiB (+8 GiB head-room), {free / 2**30:.0f} GiB free")
            for _ in range(2 * B):
                self.slabs.append(torch.zeros(shape, dtype=torch.bfloat16, device=device))
            self.slab_ptrs = torch.tensor([s.data_ptr() for s in self.slabs], dtype=torch.int64, device=device)
            self.slab_seq = torch.arange(Bg, dtype=torch.int32, device=device).repeat(2 * G)      # sequence index INSIDE its group
        self._L = _lib.lib()
        p = EpParams()
        p.B, p.P, p.D, p.V, p.rows_per_seq = Bg, self.P, self.D, V, N
        if self.lg:
            p.mode, p.syntax_shortcut, p.tok_offset = ops.MODE_DYNAMIC, 0, 0
            p.img_lo, p.img_hi, p.n_syntax = 0, V, 0
            p.lantern, p.k, p.delta = 0, 1, 0.1
            p.table_rows, p.table_cols = 0, 0
        else:
            p.mode, p.syntax_shortcut, p.tok_offset = ops.MODE_DYNAMIC, 1, 4
            p.img_lo, p.img_hi, p.n_syntax = IMG_LO, IMG_HI, 4
            for i, sx in enumerate((8196, 8197, 8803, 8828)):
                p.syntax[i] = sx
            p.lantern, p.k, p.delta = 1, cfg.lantern_k, cfg.lantern_delta
            p.table_rows, p.table_cols = K_CODES, self.table_cols
        p.top_k, p.temperature, p.top_p = 0, 1.0, (cfg.top_p if cfg.fuse_o7 else 1.0)
        p.n_uniforms, p.row_index_per_seq = self.uniforms.shape[1], 1
        self._prm = p
        b = EpBuffers()
        b.logits, b.row_index, b.cand = self.win.data_ptr(), self.row_index.data_ptr(), self.cand.data_ptr()
        b.n_paths, b.n_depth = self.nleaf.data_ptr(), self.mdepth.data_ptr()
        b.nn_table, b.uniforms, b.cursor = (self.table.data_ptr() if self.table is not None else None), self.uniforms.data_ptr(), self.cursor.data_ptr()
        self._buf = b
        w = EpWindow()
        w.win_lo, w.win_len, w.row_hot = IMG_LO, W, self.hot.data_ptr()
        w.out_tok, w.out_mass, w.rows_kind = self.out_tok.data_ptr(), self.out_mass.data_ptr(), ops.ROWS_PROBS
        self.fused_o7 = bool(cfg.fuse_o7)
        if self.fused_o7:               # raw rows: positions are per sequence and absolute (O6 dynamic writes them), the processors' parameters ride along
            w.rows_kind, w.raw_pos_per_seq = ops.ROWS_RAW_BF16, 1
            w.raw_pos_ids, w.raw_seq_len, w.raw_pos_base = self.pos_abs.data_ptr(), None, cfg.prompt_len + 3
            w.raw_cfg, w.raw_top_k = cfg.cfg_scale, cfg.logit_top_k
            if self.lg:                 # LlamaGen: no grammar rows (raw_w_latent = raw_h_latent = 0), CFG mix + top-k + softmax only
                w.raw_w_latent, w.raw_h_latent, w.raw_newline_id, w.raw_eos_id = 0, 0, 0, 0
            else:
                w.raw_w_latent, w.raw_h_latent, w.raw_newline_id, w.raw_eos_id = W_LATENT, H_LATENT, NEWLINE, EOS
        self.n_spec = max(0, min(int(cfg.spec_rows), 2)) if self.fused_o7 else 0
        if self.n_spec:
            nodes = list(range(self.n_spec))
            self.d_node_list = torch.tensor(nodes + nodes, dtype=torch.int32, device=device)       # node ids, then the depth each is prepared for
            pre = torch.zeros(N, dtype=torch.uint8)
            for n in nodes:
                pre[n] = 1 + n
            self.d_pre = pre.to(device)
            w.raw_pre = self.d_pre.data_ptr()
        self._win = w
        self._bases = dict(best=self.log_best.data_ptr(), alen=self.log_alen.data_ptr(), cnt=self.log_cnt.data_ptr(),
                           tok=self.log_token.data_ptr(), ub=self.u_bonus.data_ptr())
        self.step_idx, self._len_ub = 0, 0
        # one argument block per (pool slot, parity, group), built here (setup, untimed); the step loop patches the log-row addresses
        self._dyn_keep, self._steps = [], {}
        for slot in range(S):
            for parity in (0, 1):
                self._steps[(slot, parity)] = self._make_groups(slot, parity)

    def _make_groups(self, slot: int, parity: int):
        c, vp = self.cfg, (lambda t, o=0: t[o:].data_ptr())
        pool, cur, nxt = self.pools[slot], self.lens[parity], self.lens[parity ^ 1]
        arr = (StepGroup * self.G)()
        for g in range(self.G):
            s0, Bg, N = g * self.Bg, self.Bg, self.N
            d = StepDynamic()
            d.scores, d.tokens, d.parents = vp(pool["scores"], s0), vp(pool["tokens"], s0), vp(pool["parents"], s0)
            d.n_scores, d.n_parents, d.top_k, d.total_tokens, d.sort_rows = self.n_scores, self.n_parents, c.top_k, c.total_tokens, 1
            d.draft_tokens, d.mask, d.pos_ids, d.retrieve = vp(self.draft, s0), vp(self.mask, s0), vp(self.pos, s0), vp(self.ret, s0)
            d.n_leaf, d.max_depth, d.seq_len = vp(self.nleaf, s0), vp(self.mdepth, s0), vp(cur, 2 * s0)
            d.retrieve_pd, d.row_index, d.pos_abs = vp(self.ret_pd, s0), vp(self.row_index, s0), vp(self.pos_abs, s0)
            self._dyn_keep.append(d)
            s = arr[g]
            s.dyn = C.pointer(d)
            s.B, s.N, s.P, s.D = Bg, N, self.P, self.D
            s.cand = vp(self.cand, s0)
            s.cond, s.uncond, s.dtype, s.V, s.cfg = vp(pool["cond"], s0), vp(pool["unc"], s0), 1, self.V, c.cfg_scale
            s.model = ops.MODEL_PLAIN if self.lg else ops.MODEL_LUMINA
            s.pos_ids, s.pos_base, s.seq_len = vp(self.pos_abs, s0), c.prompt_len + 3, None
            s.w_latent, s.h_latent, s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k = W_LATENT, H_LATENT, self.lo, self.hi, NEWLINE, EOS, c.logit_top_k
            if self.lg:                 # no grammar rows
                s.w_latent, s.h_latent, s.newline_id, s.eos_id = 0, 0, 0, 0
            s.win_lo, s.win_len, s.out_kind, s.temperature, s.top_p = self.lo, self.W, ops.ROWS_PROBS, 1.0, c.top_p
            s.out_win, s.row_hot = (None if self.fused_o7 else vp(self.win, s0)), vp(self.hot, s0)
            C.memmove(C.byref(s.ep), C.byref(self._prm), C.sizeof(EpParams))
            C.memmove(C.byref(s.ep_buf), C.byref(self._buf), C.sizeof(EpBuffers))
            C.memmove(C.byref(s.ep_win), C.byref(self._win), C.sizeof(EpWindow))
            b, w = s.ep_buf, s.ep_win
            b.logits = vp(pool["cond"], s0) if self.fused_o7 else vp(self.win, s0)
            b.row_index, b.cand, b.n_paths, b.n_depth = vp(self.row_index, s0), vp(self.cand, s0), vp(self.nleaf, s0), vp(self.mdepth, s0)
            b.uniforms, b.cursor = vp(self.uniforms, s0), vp(self.cursor, s0)
            w.row_hot, w.out_tok, w.out_mass = vp(self.hot, s0), vp(self.out_tok, s0), vp(self.out_mass, s0)
            if self.fused_o7:
                w.raw_uncond, w.raw_pos_ids = vp(pool["unc"], s0), vp(self.pos_abs, s0)
            if self.n_spec:
                s.node_list, s.n_list, s.out_win = self.d_node_list.data_ptr(), self.n_spec, vp(self.win, s0)
                w.raw_probs = vp(self.win, s0)
            if c.with_kv:
                s.slab_ptrs, s.slab_seq = vp(self.slab_ptrs, 2 * s0), vp(self.slab_seq, 2 * s0)
                s.slab_prev, s.new_len = vp(cur, 2 * s0), vp(nxt, 2 * s0)
                s.n_slabs, s.elem_bytes, s.outer = 2 * Bg, 2, 2 * c.kv_layers * c.kv_heads
                s.S_max, s.d = c.kv_smax + c.kv_pad_rows, c.kv_dim
                s.hidden, s.out_hidden, s.accepted_tokens = vp(pool["hidden"], s0), vp(self.out_hidden, s0), vp(self.acc_tokens, s0)
                s.hid_elem_bytes, s.hid_groups, s.H = 2, 2, self.hidden_w
        return arr

    def _fork(self):
        if self.G > 1:
            cur = torch.cuda.current_stream(self.device)
            for st in self.streams:
                st.wait_stream(cur)
        self._forked = True

    def join(self):
        """The current stream waits for every group stream."""
        if self.G > 1:
            cur = torch.cuda.current_stream(self.device)
            for st in self.streams:
                cur.wait_stream(st)
        self._forked = False

    def sync(self):
        self.join()
        torch.cuda.synchronize(self.device)

    def release_kv(self):
        self.sync()
        self.slabs, self.slab_ptrs, self._steps = [], None, {}
        torch.cuda.empty_cache()

    def step(self, events=None):
        """Enqueue one verify step of every sequence: O4 -> O6 -> O7 -> O8 -> O9 + O10 per group.  events None and cfg.native_step: ONE call
        of lantern_verify_step, group g on its own stream (G > 1: join() / sync() before reading results elsewhere).  Otherwise the same
        entry points one ctypes call each, group after group on the current stream; `events`: name -> (start, end) around group 0's launches."""
        c, L = self.cfg, self._L
        i = self.step_idx
        if i >= c.max_steps:
            raise _lib.LanternError(f"step {i} >= max_steps {c.max_steps}")
        arr = self._steps[(i % c.pool_steps, i & 1)]
        bs, native = self._bases, events is None and c.native_step
        if native and self.G > 1 and not self._forked:
            self._fork()
        if not native and self._forked:
            self.join()
        cur_stream = torch.cuda.current_stream().cuda_stream
        for g in range(self.G):
            s, e = arr[g], i * c.n_seq + g * self.Bg
            s.stream = self.streams[g].cuda_stream if (native and self.G > 1) else cur_stream
            s.sample_token = self.first_token[g * self.Bg:].data_ptr() if i == 0 else bs["tok"] + 8 * (e - c.n_seq)
            s.ep_buf.best, s.ep_buf.accept_len, s.ep_buf.counters = bs["best"] + 4 * e, bs["alen"] + 4 * e, bs["cnt"] + 24 * e
            s.ep_win.u_bonus, s.ep_win.token = bs["ub"] + 8 * e, bs["tok"] + 8 * e
        turns = native and c.commit_window > 0 and c.with_kv and self.G > 1          # commit turn-taking between the stream groups (lantern_step_group.turn)
        if turns and not hasattr(self, "_turn"):
            self._turn = torch.zeros(_lib.TURN_WORDS(self.G), dtype=torch.int64, device=self.device)
            self._turn_step = 0
        for g in range(self.G):
            s = arr[g]
            if turns:
                s.turn, s.turn_group, s.turn_groups = self._turn.data_ptr(), g, self.G
                s.turn_wait = self._turn_step * self.G + g - (c.commit_window - 1)
            else:
                s.turn = None
        if native:
            check(L.lantern_verify_step(arr, self.G), "verify_step")
            if turns:
                self._turn_step += 1
        else:
            for g in range(self.G):
                self._launch_group(arr[g], events if g == 0 else None)
        cur, nxt = self.lens[i & 1], self.lens[(i & 1) ^ 1]
        if not c.with_kv:
            for g in range(self.G):
                s0, Bg = g * self.Bg, self.Bg
                with torch.cuda.stream(self.streams[g]) if (native and self.G > 1) else _nullctx():
                    torch.add(cur[2 * s0:2 * s0 + 2 * Bg], (self.log_alen[i, s0:s0 + Bg] + 1).repeat(2), out=nxt[2 * s0:2 * s0 + 2 * Bg])
        self._len_ub += self.D
        if self._len_ub >= self.tokens_per_image:      # an image can end: wrap those sequences (host bound refreshed every few hundred steps)
            torch.cuda.synchronize(self.device)          # (device-wide, not join(): see LuminaVerifyWorkload._native_step)
            torch.where(nxt - self.len_base >= self.tokens_per_image, self.len_base, nxt, out=nxt)
            self._len_ub = int((nxt - self.len_base).max().item())
        self.step_idx += 1

    def _launch_group(self, s, events=None):
        """One group's step, one ctypes call per kernel on s.stream (the per-kernel timing pass; the parity tests' second path)."""
        L, vp, d = self._L, C.c_void_p, s.dyn.contents
        st = vp(s.stream)
        arm = lambda name: events and check(L.lantern_profile_next_launch(vp(events[name][0].cuda_event), vp(events[name][1].cuda_event)), "profile")
        if s.n_list > 0:          # tree + candidates + the likely rows in one launch
            check(L.lantern_prepare_step(C.byref(s)), "prepare_step")
        else:
            check(L.lantern_tree_dynamic_finalize(vp(d.scores), vp(d.tokens), vp(d.parents), vp(s.sample_token), s.B, d.n_scores, d.n_parents, d.top_k,
                                                  d.total_tokens, d.sort_rows, vp(d.draft_tokens), vp(d.mask), vp(d.pos_ids), vp(d.retrieve), vp(d.n_leaf),
                                                  vp(d.max_depth), st), "tree_dynamic_finalize")
            check(L.lantern_gather_candidates_dynamic(vp(d.draft_tokens), vp(d.retrieve), vp(d.pos_ids), vp(d.seq_len), s.B, s.N, s.P, s.D, vp(s.cand),
                                                      vp(d.retrieve_pd), vp(d.row_index), vp(d.pos_abs), st), "gather_candidates_dynamic")
            if s.out_win:
                arm("cfg_mask_topk")
                check(L.lantern_cfg_mask_topk_window(vp(s.cond), vp(s.uncond), s.dtype, s.B * s.N, s.V, C.c_float(s.cfg), s.model, vp(s.pos_ids), C.c_int64(s.pos_base),
                                                     s.w_latent, s.h_latent, s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k, None, 0, s.win_lo, s.win_len,
                                                     vp(s.out_win), vp(s.row_hot), s.out_kind, C.c_float(s.temperature), C.c_float(s.top_p), st), "cfg_mask_topk_window")
        arm("evaluate_posterior")
        check(L.lantern_evaluate_posterior_window(C.byref(s.ep), C.byref(s.ep_buf), C.byref(s.ep_win), st), "evaluate_posterior_window")
        if s.slab_ptrs:
            arm("kv_gather")
            check(L.lantern_update_inference_inputs(vp(s.slab_ptrs), vp(s.slab_seq), vp(s.slab_prev), s.n_slabs, s.elem_bytes, C.c_int64(s.outer), C.c_int64(s.S_max),
                                                    C.c_int64(s.d), vp(d.retrieve_pd), 1, s.P, s.D, vp(s.ep_buf.best), vp(s.ep_buf.accept_len), vp(s.new_len),
                                                    vp(s.hidden), s.hid_elem_bytes, s.B, s.hid_groups, s.N, s.H, vp(s.cand), vp(s.out_hidden),
                                                    vp(s.accepted_tokens), st), "update_inference_inputs")

    def lengths(self, parity: int):
        """(conditional, unconditional) cache lengths in sequence order (the length vector is stored group by group)."""
        v = self.lens[parity].view(self.G, 2, self.Bg)
        return v[:, 0].reshape(-1), v[:, 1].reshape(-1)

    def accepted_tokens(self, i0: int, i1: int) -> int:
        self.join()
        return int((self.log_alen[i0:i1].to(torch.int64) + 1).sum().item())

    def check_status(self, i0: int, i1: int):
        self.join()
        stt = self.log_cnt[i0:i1, :, 5]
        if int(stt.abs().sum().item()) != 0:
            bad = torch.nonzero(stt)[0].tolist()
            raise _lib.LanternError(f"evaluate_posterior status {int(stt[bad[0], bad[1]])} at step {i0 + bad[0]} seq {bad[1]}")

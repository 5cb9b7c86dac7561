"""Scripted stand-ins for the LlamaGen / Anole target models and their EAGLE drafters (dynamic EAGLE-2 trees and static trees),
shared by make_golden_generate_lg.py (which drives the REFERENCE's own models/ea_model_llamagen.EaModel.generate and
models/ea_model_anole.EaModel.generate with them on CPU, in the build container) and tests/test_gpu_generate_lg.py (which drives
the lantern_amd mirrors with the same objects on the GPU).  Same idea as gen_fakes.py: everything the two sides exchange is a
table lookup of numpy-seeded values, so both see bit-identical logits / drafts on any device.  Test infrastructure only."""
import types

import numpy as np
import torch

import gen_fakes as F

M, H, HKV, DH, SMAX = F.M, F.H, F.HKV, F.DH, F.SMAX
V = F.V                         # 16384: LlamaGen's vocabulary; for the Anole stand-in a small vocabulary with the image range 4..8195
PREFIX = 120                    # LlamaGen: T5 caption positions / zero ids in front of the image tokens
N_SHAPES, MAX_DEPTH = 6, 5

MODELS = {"llamagen": dict(img_lo=0, img_hi=V, offset=0, table_rows=V), "anole": dict(img_lo=F.IMG_LO, img_hi=F.IMG_HI, offset=4, table_rows=8192)}


def tables(model):
    """Target logits by (token, position) key, the drafter's distribution / 10 scripted draws per key, the neighbour table.  The target
    strongly prefers the drafter's first draws (a trained drafter's situation), so the dynamic-tree walk -- which accepts a drafted
    token with the target's own probability of it -- really accepts some."""
    lo, hi = MODELS[model]["img_lo"], MODELS[model]["img_hi"]
    rows = MODELS[model]["table_rows"]
    rs = np.random.RandomState(20240522 if model == "llamagen" else 20240523)
    tgt = (3.0 * rs.standard_normal((M, V))).astype(np.float32)
    dl = tgt[:, lo:hi].astype(np.float64) + 1.5 * rs.standard_normal((M, hi - lo))
    kth = np.sort(dl, axis=1)[:, -200][:, None]
    dl = np.where(dl < kth, -np.inf, dl)
    e = np.exp(dl - dl.max(1, keepdims=True))
    op = np.zeros((M, V), np.float32)
    op[:, lo:hi] = (e / e.sum(1, keepdims=True)).astype(np.float32)                # drafter distribution: top-200 of the image range
    gum = -np.log(-np.log(rs.uniform(1e-9, 1 - 1e-9, (M, V))))
    with np.errstate(divide="ignore"):
        key = np.where(op > 0, np.log(op.astype(np.float64)) + gum, -np.inf)
    tok = np.argsort(-key, axis=1)[:, :10].astype(np.int64)                        # 10 draws without replacement (Gumbel top-k), fixed
    prob = np.take_along_axis(op, tok, 1)
    for j, bonus in enumerate((9.0, 8.0, 7.0)):
        tgt[np.arange(M), tok[:, j]] += bonus
    nb = ((np.arange(rows)[:, None] + 1 + 37 * np.arange(F.TABLE_COLS)[None, :]) % rows).astype(np.int64)
    return dict(tgt=tgt, op=op, tok=tok, prob=prob, nb=nb)


def row_key(tok, pos):
    return (tok * 7 + pos * 13) % M


class Head:
    """lm_head: logits looked up from the (token, position, batch row) digits of the hidden state: the unconditional row of the
    [cond; uncond] batch is the conditional row plus half of another table row, so CFG really mixes two (related) distributions."""

    def __init__(self, T, dev):
        self.weight = torch.zeros(V, H, device=dev, dtype=torch.float32)
        self.tgt = torch.from_numpy(T["tgt"]).to(dev)

    def __call__(self, hidden):
        tok, pos = F.decode(hidden)
        row = hidden[..., 4:5]
        return self.tgt[row_key(tok, pos)] + 0.5 * row * self.tgt[(row_key(tok, pos) + 31) % M]


class Inner:
    def __init__(self, dev, n_layers=2):
        lin = types.SimpleNamespace(weight=torch.zeros(1, device=dev))
        self.layers = [types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin)) for _ in range(n_layers)]
        self.tree_mask, self.tree_mode, self.dev = None, None, dev
        self.cls_embedding = types.SimpleNamespace(uncond_embedding=torch.full((H,), 0.5, device=dev))
        self.calls = []

    def __call__(self, cond_idx=None, input_ids=None, attention_mask=None, past_key_values=None, position_ids=None, cache_position=None):
        cur = int(past_key_values[0][0].current_length)
        if cond_idx is not None:                  # LlamaGen prefill: caption embeddings, no token ids
            B, T = cond_idx.shape[:2]
            input_ids = torch.zeros((B, T), dtype=torch.long, device=self.dev)
        B, T = input_ids.shape
        if position_ids is None:
            position_ids = torch.arange(cur, cur + T, device=self.dev)[None].expand(B, T)
        position_ids = position_ids.reshape(-1, T).expand(B, T)
        self.calls.append((cur, T))
        hidden = torch.zeros(B, T, H, device=self.dev, dtype=torch.float32)
        hidden[..., 0] = (input_ids % 128).float()
        hidden[..., 2] = (input_ids // 128).float()
        hidden[..., 1] = (position_ids % 128).float()
        hidden[..., 3] = (position_ids // 128).float()
        hidden[..., 4] = torch.arange(B, device=self.dev)[:, None].float()
        kv = hidden[:, None, :, :DH].expand(B, HKV, T, DH).contiguous()
        for layer in past_key_values:
            for c in layer:
                c.cat(kv.to(c.data.dtype), dim=2)
        return (hidden,)


class T5:
    """`base_model.t5_model.get_text_embeddings(prompt)` -> (embeddings [B, 120, C], mask [B, 120], valid tokens first)."""

    def __init__(self, dev):
        self.dev = dev

    def get_text_embeddings(self, prompt):
        embs, masks = [], []
        for p in prompt:
            rs = np.random.RandomState(sum(map(ord, p)) % 100000)
            n = 5 + len(p) % 40
            embs.append(rs.standard_normal((PREFIX, H)).astype(np.float32))
            m = np.zeros(PREFIX, np.int64)
            m[:n] = 1
            masks.append(m)
        return torch.from_numpy(np.stack(embs)).to(self.dev), torch.from_numpy(np.stack(masks)).to(self.dev)


class Tokenizer:
    def tokenize_text(self, p):
        return [9000 + (ord(ch) % 500) for ch in p][:12]


def make_base(T, dev, model):
    cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=HKV, max_position_embeddings=SMAX, hidden_size=HKV * DH,
                                num_attention_heads=HKV)
    return types.SimpleNamespace(model=Inner(dev), lm_head=Head(T, dev), config=cfg, dtype=torch.float32, device=dev, t5_model=T5(dev))


# ----------------------------------------------------------------------------------------------------------------- dynamic trees
def make_shapes():
    """A few EAGLE-2 style tree shapes: node 0 = the root (the sampled token), every other node = (parent, slot of the parent's
    top-10), parents before children, nodes ordered by depth; leaves' root paths as -1 padded rows in the reference's row order
    (cnets_llamagen.py:895-906: sorted with the padding last)."""
    rs = np.random.RandomState(77)
    shapes = []
    for s in range(N_SHAPES):
        n_nodes = int(rs.randint(14, 27))
        par, slot, depth = [0], [0], [0]
        level = [0]
        d = 1
        while len(par) < n_nodes and d <= MAX_DEPTH:
            nxt = []
            for p in level:
                n_ch = int(rs.randint(1, 4)) if p == 0 or rs.rand() < 0.6 else 0
                if p == 0:
                    n_ch = max(n_ch, 2)
                for sl in range(n_ch):
                    if len(par) >= n_nodes:
                        break
                    par.append(p); slot.append(sl); depth.append(d)
                    nxt.append(len(par) - 1)
            if not nxt:
                break
            level, d = nxt, d + 1
        N = len(par)
        mask = np.zeros((N, N), np.float32)
        for i in range(N):
            j = i
            while True:
                mask[i, j] = 1
                if j == 0:
                    break
                j = par[j]
        leaves = [i for i in range(N) if i not in set(par[1:])]
        md = max(depth) + 1
        rows = []
        for lf in leaves:
            path, j = [], lf
            while True:
                path.append(j)
                if j == 0:
                    break
                j = par[j]
            path = path[::-1]
            rows.append(path + [-1] * (md - len(path)))
        rows.sort(key=lambda r: [x if x >= 0 else N + 5 for x in r])
        shapes.append(dict(par=par, slot=slot, depth=depth, mask=mask, retrieve=np.asarray(rows, np.int64)))
    return shapes


class Drafter:
    """The drafter interface of models/drafters/cnets_llamagen.py / cnets_anole.py: `topK_genrate` (dynamic tree: draft tokens, retrieve
    rows, tree mask, tree position ids) and `init_tree_v1` / `topK_genrate_v1` (static tree: ss_token [R,10], ss_prob [R,10], the
    per-level drafter distributions).  A node's children are the scripted top-10 draws of the row keyed by the node's (token, position):
    the same key the target's conditional row uses, so drafted tokens are plausible under the target."""

    def __init__(self, T, dev):
        self.dev = dev
        self.op, self.prob = torch.from_numpy(T["op"]).to(dev), torch.from_numpy(T["prob"]).to(dev)
        self.tok_np, self.tok = T["tok"], torch.from_numpy(T["tok"]).to(dev)
        self.shapes = make_shapes()
        self.calls = []

    def reset_kv(self):
        pass

    def init_tree(self):
        pass

    def init_tree_v1(self, tree=None):
        if tree is not None:
            self.levels = F.level_parents(tree)

    def _log(self, hidden_states, input_ids):
        last_tok, pos = int(input_ids[0, -1]), int(input_ids.shape[1])
        self.calls.append((last_tok, pos, tuple(F.decode(hidden_states)[0].reshape(-1).tolist())))
        return last_tok, pos

    def topK_genrate(self, hidden_states, input_ids, head, logits_processor, cfg_scale, *extra, **kw):
        last_tok, pos = self._log(hidden_states, input_ids)
        sh = self.shapes[(last_tok + 3 * pos) % len(self.shapes)]
        toks = [last_tok]
        for i in range(1, len(sh["par"])):
            p = sh["par"][i]
            toks.append(int(self.tok_np[row_key(toks[p], pos - 1 + sh["depth"][p]), sh["slot"][i]]))
        t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt, device=self.dev)
        return (t([toks], torch.long), t(sh["retrieve"], torch.long), t(sh["mask"], torch.float32)[None, None], t(sh["depth"], torch.long))

    def topK_genrate_v1(self, hidden_states, input_ids, head, logits_processor, cfg_scale, *extra, **kw):
        last_tok, pos = self._log(hidden_states, input_ids)
        tok_of = {(): last_tok}
        toks, probs, ops_ = [], [], []
        for lvl, parents in enumerate(self.levels):
            keys = [int(row_key(tok_of[p], pos - 1 + lvl)) for p in parents]
            for p, k in zip(parents, keys):
                for slot in range(10):
                    tok_of[p + (slot,)] = int(self.tok_np[k, slot])
            key = torch.tensor(keys, device=self.dev)
            toks.append(self.tok[key])
            probs.append(self.prob[key])
            ops_.append(self.op[key] if logits_processor is not None else None)
        return torch.cat(toks), torch.cat(probs), ops_


# name, model, tree ("dynamic" or a static choice name), lantern, k, delta, temperature, cfg, seed, max_length, prompt
CASES = [
    dict(name="lg_dyn_lantern", model="llamagen", tree="dynamic", lantern=True, k=32, delta=0.2, temperature=1.0, cfg=2.0, seed=21, max_length=48, prompt=["a red bird"]),
    dict(name="lg_dyn_plain", model="llamagen", tree="dynamic", lantern=False, k=32, delta=0.1, temperature=1.0, cfg=2.0, seed=22, max_length=40, prompt=["two cats on a sofa"]),
    dict(name="lg_static_lantern", model="llamagen", tree="naive_extend_57", lantern=True, k=16, delta=0.3, temperature=1.0, cfg=2.0, seed=23, max_length=48, prompt=["a boat"]),
    dict(name="lg_static_plain", model="llamagen", tree="mc_sim_7b_63", lantern=False, k=16, delta=0.1, temperature=1.0, cfg=2.0, seed=24, max_length=36, prompt=["snow"]),
    dict(name="an_dyn_lantern", model="anole", tree="dynamic", lantern=True, k=32, delta=0.2, temperature=1.0, cfg=2.0, seed=31, max_length=48, prompt=["a green field"]),
    dict(name="an_static_lambda", model="anole", tree="naive_extend_57", lantern=True, k=10, delta=5.0, temperature=1.0, cfg=2.0, seed=32, max_length=48, prompt=["a house by the sea"]),
    dict(name="an_static_plain", model="anole", tree="mc_sim_7b_63", lantern=False, k=10, delta=0.1, temperature=1.0, cfg=2.0, seed=33, max_length=36, prompt=["fog"]),
    dict(name="an_dyn_plain", model="anole", tree="dynamic", lantern=False, k=10, delta=0.1, temperature=1.0, cfg=2.0, seed=34, max_length=40, prompt=["a bridge at night"]),
    # greedy decoding (temperature 0: no processor list; evaluate_posterior's greedy / TVD branch, ea_model_llamagen.py:789-905): round 5
    dict(name="lg_dyn_greedy", model="llamagen", tree="dynamic", lantern=True, k=32, delta=0.2, temperature=0.0, cfg=2.0, seed=41, max_length=44, prompt=["a red bird"]),
    dict(name="lg_static_greedy", model="llamagen", tree="naive_extend_57", lantern=True, k=16, delta=0.3, temperature=0.0, cfg=2.0, seed=42, max_length=40, prompt=["a boat"]),
    dict(name="an_dyn_greedy", model="anole", tree="dynamic", lantern=False, k=10, delta=0.1, temperature=0.0, cfg=2.0, seed=43, max_length=40, prompt=["a green field"]),
    dict(name="an_static_greedy", model="anole", tree="mc_sim_7b_63", lantern=True, k=10, delta=0.2, temperature=0.0, cfg=2.0, seed=44, max_length=40, prompt=["fog"]),
]
TOP_K, TOP_P = 2000, 1.0

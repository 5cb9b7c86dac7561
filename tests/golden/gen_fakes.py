"""Scripted stand-ins for the target model and the drafter, shared by make_golden_generate.py (which drives the REFERENCE's
own EaLumina_mGPT.generate with them on CPU, in the build container) and tests/test_gpu_generate_ref.py (which drives the
lantern_amd mirror with the same objects on the GPU).  Everything the two sides exchange is a table lookup of numpy-seeded
values, so both see bit-identical logits / drafter distributions on any device.  Test infrastructure only."""
import types

import numpy as np
import torch

V, H, HKV, DH, SMAX, M = 16384, 64, 2, 32, 512, 97
IMG_LO, IMG_HI = 4, 8196
TABLE_COLS = 64


def tables():
    rs = np.random.RandomState(20240521)
    tgt = (3.0 * rs.standard_normal((M, V))).astype(np.float32)                   # target logits by (token, position) key
    dl = tgt[:, IMG_LO:IMG_HI].astype(np.float64) + 1.5 * rs.standard_normal((M, IMG_HI - IMG_LO))
    kth = np.sort(dl, axis=1)[:, -200][:, None]
    dl = np.where(dl < kth, -np.inf, dl)
    e = np.exp(dl - dl.max(1, keepdims=True))
    op = np.zeros((M, V), np.float32)
    op[:, IMG_LO:IMG_HI] = (e / e.sum(1, keepdims=True)).astype(np.float32)       # drafter distribution (top-200 of the image range)
    gum = -np.log(-np.log(rs.uniform(1e-9, 1 - 1e-9, (M, V))))
    with np.errstate(divide="ignore"):
        key = np.where(op > 0, np.log(op.astype(np.float64)) + gum, -np.inf)
    tok = np.argsort(-key, axis=1)[:, :10].astype(np.int64)                       # 10 draws without replacement (Gumbel top-k), fixed
    prob = np.take_along_axis(op, tok, 1)
    nb = ((np.arange(8192)[:, None] + 1 + 37 * np.arange(TABLE_COLS)[None, :]) % 8192).astype(np.int64)   # neighbour table [8192, 64]
    return dict(tgt=tgt, op=op, tok=tok, prob=prob, nb=nb)


def decode(h):          # hidden / KV row -> (token, position)
    return (h[..., 0].float() + 128 * h[..., 2].float()).long(), (h[..., 1].float() + 128 * h[..., 3].float()).long()


class Head:
    """lm_head: float32 logits looked up from the (token, position) digits in the hidden state.  The unconditional pass sees
    the same tokens at positions shifted by the prompt length, hence other rows of the table."""

    def __init__(self, T, dev):
        self.weight = torch.zeros(V, H, device=dev, dtype=torch.float32)
        self.tgt = torch.from_numpy(T["tgt"]).to(dev)

    def __call__(self, hidden):
        tok, pos = decode(hidden)
        return self.tgt[(tok * 7 + pos * 13) % M]


class Inner:
    def __init__(self, dev, n_layers=2):
        lin = types.SimpleNamespace(weight=torch.zeros(1, device=dev))
        self.layers = [types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin)) for _ in range(n_layers)]
        self.tree_mask, self.tree_mode, self.dev = None, None, dev

    def __call__(self, input_ids=None, attention_mask=None, past_key_values=None, position_ids=None):
        B, T = input_ids.shape
        cur = int(past_key_values[0][0].current_length)
        if position_ids is None:
            position_ids = torch.arange(cur, cur + T, device=self.dev)[None].expand(B, T)
        position_ids = position_ids.reshape(-1, T).expand(B, T)
        hidden = torch.zeros(B, T, H, device=self.dev, dtype=torch.float32)
        hidden[..., 0] = (input_ids % 128).float()
        hidden[..., 2] = (input_ids // 128).float()
        hidden[..., 1] = (position_ids % 128).float()
        hidden[..., 3] = (position_ids // 128).float()
        kv = hidden[:, None, :, :DH].expand(B, HKV, T, DH).contiguous()
        for layer in past_key_values:
            for c in layer:
                c.cat(kv.to(c.data.dtype), dim=2)
        return (hidden,)


def level_parents(tree_choices):
    """Drafter rows of a static tree, level by level: the root, then every node that has children (sorted, as the tree buffers order them)."""
    out, d = [[()]], 1
    while True:
        parents = sorted({tuple(c[:-1]) for c in tree_choices if len(c) == d + 1})
        if not parents:
            return out
        out.append(parents)
        d += 1


class Drafter:
    """Static-tree (EAGLE-1) drafter with the reference's interface: topK_generate returns (ss_token [R,10], ss_prob [R,10],
    [original_prob of every level]).  The row of a parent node is the target's row for that node (same (token, position) key)
    plus noise, so drafted tokens are plausible under the target and the walk accepts some of them."""

    def __init__(self, T, dev):
        self.dev, self.cfg_scale = dev, 3.0
        self.op, self.prob = torch.from_numpy(T["op"]).to(dev), torch.from_numpy(T["prob"]).to(dev)
        self.tok_np, self.tok = T["tok"], torch.from_numpy(T["tok"]).to(dev)
        self.calls = []

    def reset_kv(self):
        pass

    def init_tree(self, tree=None):
        if tree is not None:
            self.levels = level_parents(tree)

    def topK_generate(self, hidden_states, uncond_hidden_states, input_ids, attention_mask, head, logits_processors, tree_type="static"):
        assert tree_type == "static"
        last_tok, pos = int(input_ids[0, -1]), int(input_ids.shape[1])
        self.calls.append((last_tok, pos, tuple(decode(hidden_states)[0].reshape(-1).tolist())))
        tok_of = {(): last_tok}
        toks, probs, ops_ = [], [], []
        for lvl, parents in enumerate(self.levels):
            keys = [(tok_of[p] * 7 + (pos - 1 + lvl) * 13) % M for p in parents]
            for p, k in zip(parents, keys):
                for slot in range(10):
                    tok_of[p + (slot,)] = int(self.tok_np[k, slot])
            key = torch.tensor(keys, device=self.dev)
            toks.append(self.tok[key])
            probs.append(self.prob[key])
            ops_.append(self.op[key])
        return torch.cat(toks), torch.cat(probs), ops_


class DetDraws:
    """torch.multinomial / torch.rand replacement: inverse CDF over a recorded uniform list (the reference draws the first
    token and every bonus token with torch.multinomial, whose device RNG no other device reproduces)."""

    def __init__(self, us):
        self.us, self.n = [float(u) for u in us], 0

    def next(self):
        u = self.us[self.n]
        self.n += 1
        return u

    def multinomial(self, probs, num_samples, replacement=False, generator=None):
        assert num_samples == 1
        cs = probs.double().cumsum(-1)
        u = self.next()
        return (cs > u * cs[..., -1:]).to(torch.int64).argmax(-1, keepdim=True)

    def rand(self, *size, dtype=None, device=None, **kw):
        """torch.rand(n): the next n recorded uniforms (0.5 behind the end of the record: a block may reach past the last step's draw)."""
        n = int(size[0]) if size else 1
        vals = []
        for _ in range(n):
            vals.append(self.next() if self.n < len(self.us) else 0.5)
        return torch.tensor(vals, dtype=dtype or torch.float32, device=device)


def make_base(T, dev):
    cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=HKV, max_position_embeddings=SMAX, hidden_size=HKV * DH,
                                num_attention_heads=HKV)
    return types.SimpleNamespace(model=Inner(dev), lm_head=Head(T, dev), config=cfg, dtype=torch.float32)


CASES = [dict(name="seq_delta", cfg_mode="sequential", lantern=True, k=32, delta=0.1, seed=11, max_new=70),
         dict(name="par_lambda", cfg_mode="parallel", lantern=True, k=16, delta=5.0, seed=12, max_new=60),
         dict(name="seq_plain", cfg_mode="sequential", lantern=False, k=32, delta=0.1, seed=13, max_new=40)]
PROMPT = [1, 9000, 9100, 9200, 9300]

#!/usr/bin/env python3
"""Golden vectors for the Lumina-mGPT drafter loop (Lumina column): the reference's own
`Model.topK_generate` (models/drafters/cnets_lumina_mgpt.py:1148-1393) run unbound on a scripted stand-in whose forward only
RECORDS its arguments (token ids, cond / uncond position ids from the zero-padded attention mask, the mask itself, the tree mask)
and whose head returns scripted cond / uncond logits; the reference's `MultiModalLogitsProcessor` / `InterleavedTopKLogitsWarper`
(models/ea_model_lumina_mgpt.py:25-112) shape the drafted rows, at the real vocabulary (65536, image ids 4..8195, newline 8803,
end of image 8196) with the uncond stream placed so that the tree depths straddle a newline position.  Dynamic tree: calls and
outputs, first call and the call on top of the drafter cache.  Static tree (its draws come from torch's CPU generator): the calls
only.  Runs only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_lumina_drafter.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

V, LO, HI, NL, EOS, TOPK, H = 65536, 4, 8196, 8803, 8196, 10, 4


class Recorder:
    def __init__(self, script, total_tokens, depth, cfg_scale):
        self.script, self.calls, self.seen = script, 0, []
        self.total_tokens, self.depth, self.top_k, self.cfg_scale = total_tokens, depth, TOPK, cfg_scale
        self.logsoftmax = torch.nn.LogSoftmax(dim=-1)
        self.embed_tokens = types.SimpleNamespace(weight=torch.zeros(1))
        self.tree_mask_init = torch.eye(TOPK)[None, None]
        self.position_ids = torch.zeros(TOPK, dtype=torch.long)
        self.stable_kv = None
        self.tree_mask = None

    def reset(self):
        self.tree_mask = None

    def __call__(self, hidden_states, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, use_cache=True):
        T = input_ids.shape[1]
        past = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        self.seen.append(dict(ids=input_ids.clone(), pos=position_ids.clone(), past=past, attn=attention_mask.clone(),
                              hid_shape=tuple(hidden_states.shape), tree=None if self.tree_mask is None else self.tree_mask.clone()))
        return torch.zeros(2, T, H), ((torch.zeros(2, 1, past + T, 1),),)

    def head(self, hidden):
        blk = torch.from_numpy(self.script[self.calls])
        self.calls += 1
        return torch.stack([blk, 0.5 * blk])            # cond, uncond


def script_for(seed, blocks):
    rs = np.random.RandomState(seed)
    out = []
    for shape in blocks:
        out.append((3.0 * rs.standard_normal(shape)).astype(np.float32))
    return out


def processors(R):
    proc = R.lum.MultiModalLogitsProcessor.__new__(R.lum.MultiModalLogitsProcessor)
    proc.image_next_line_token_id, proc.image_end_token_id = NL, EOS
    supp = torch.ones(V, dtype=torch.bool)
    supp[LO:HI] = False
    proc.suppress_token_mask = supp
    return [proc, R.lum.InterleavedTopKLogitsWarper(image_top_k=300)]


def store_calls(out, pre, seen, attn_full):
    out[pre + "n_calls"] = np.int64(len(seen))
    for j, s in enumerate(seen):
        out[pre + f"call{j}.ids"] = s["ids"].numpy()
        out[pre + f"call{j}.pos"] = s["pos"].numpy()
        out[pre + f"call{j}.past"] = np.int64(s["past"])
        out[pre + f"call{j}.attn"] = s["attn"].numpy()
        out[pre + f"call{j}.hid_shape"] = np.asarray(s["hid_shape"], np.int64)
        out[pre + f"call{j}.tree"] = np.zeros(0, np.float32) if s["tree"] is None else s["tree"].numpy()


def main():
    R = import_reference()
    M = R.clu.Model
    procs = processors(R)
    out = {}
    depth, total = 4, 59
    # ---- dynamic tree: prompt of 12 tokens, image start at index 7 -> uncond stream of 5 + generated tokens
    for ci, (L, Lu, extra) in enumerate([(12 + 44, 5 + 44, 3), (12 + 96, 5 + 96, 2)]):
        blocks = ([(V,)] + [(TOPK, V)] * depth) * 2
        rec = Recorder(script_for(70 + ci, blocks), total - 1, depth, 3.0)
        attn = torch.ones(2, L, dtype=torch.bool)
        attn[1, :L - Lu] = False
        rs = np.random.RandomState(ci)
        ids1 = torch.from_numpy(rs.randint(LO, HI, size=(1, L + 1)))
        d1 = M.topK_generate(rec, torch.zeros(1, L, H), torch.zeros(1, Lu, H), ids1, rec.head, procs, attention_mask=attn, tree_type="dynamic")
        n_first = len(rec.seen)
        ids2 = torch.cat([ids1, torch.from_numpy(rs.randint(LO, HI, size=(1, extra)))], dim=1)
        d2 = M.topK_generate(rec, torch.zeros(1, extra, H), torch.zeros(1, extra, H), ids2, rec.head, procs, attention_mask=attn, tree_type="dynamic")
        pre = f"dyn{ci}."
        out[pre + "L"], out[pre + "Lu"], out[pre + "extra"], out[pre + "n_first"] = np.int64(L), np.int64(Lu), np.int64(extra), np.int64(n_first)
        out[pre + "attn"], out[pre + "ids1"], out[pre + "ids2"], out[pre + "seed"] = attn.numpy(), ids1.numpy(), ids2.numpy(), np.int64(70 + ci)
        store_calls(out, pre, rec.seen, attn)
        for tag, d in (("out1", d1), ("out2", d2)):
            out[pre + tag + ".draft"], out[pre + tag + ".retrieve"] = d[0].numpy(), d[1].numpy()
            out[pre + tag + ".mask"], out[pre + tag + ".pos"] = d[2].numpy(), d[3].numpy()
    # ---- static tree (mc_sim_7b_63): calls only
    tbuf = R.uc.generate_tree_buffers(R.ch.mc_sim_7b_63, device="cpu")
    n_levels = len(tbuf["tree_indices"])
    blocks = [(V,)] + [(len(tbuf["tree_indices"][i]), V) for i in range(n_levels)]
    rec = Recorder(script_for(90, blocks), total - 1, depth, 3.0)
    rec.tree_buffer = tbuf
    L, Lu = 12 + 30, 5 + 30
    attn = torch.ones(2, L, dtype=torch.bool)
    attn[1, :L - Lu] = False
    ids1 = torch.from_numpy(np.random.RandomState(9).randint(LO, HI, size=(1, L + 1)))
    torch.manual_seed(0)
    s1 = M.topK_generate(rec, torch.zeros(1, L, H), torch.zeros(1, Lu, H), ids1, rec.head, procs, attention_mask=attn, tree_type="static")
    pre = "sta0."
    out[pre + "L"], out[pre + "Lu"], out[pre + "attn"], out[pre + "ids1"], out[pre + "seed"] = np.int64(L), np.int64(Lu), attn.numpy(), ids1.numpy(), np.int64(90)
    store_calls(out, pre, rec.seen, attn)
    out[pre + "ss_token_shape"] = np.asarray(s1[0].shape, np.int64)
    out[pre + "n_op"] = np.int64(len(s1[2]))
    out["dims"] = np.array([V, LO, HI, NL, EOS, TOPK, depth, total], np.int64)
    np.savez_compressed(os.path.join(HERE, "lumina_drafter.npz"), **out)
    print("lumina_drafter.npz ok:", int(out["dyn0.n_calls"]), "dynamic calls;", int(out["sta0.n_calls"]), "static calls;",
          "pos of call 0:", out["dyn0.call0.pos"][:, -3:].tolist(), "call 1:", out["dyn0.call1.pos"].tolist())


if __name__ == "__main__":
    main()

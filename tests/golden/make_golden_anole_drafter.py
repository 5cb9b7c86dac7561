#!/usr/bin/env python3
"""Golden vectors for the Anole calling convention of the drafter (Anole column of 8a-bis): the
reference's own `Model.topK_genrate` / `Model.topK_genrate_v1` of models/drafters/cnets_anole.py run unbound on a scripted
stand-in whose forward only RECORDS what it is called with (token ids, the [2,T] cond / uncond position ids built from
`input_position_diff`, the attention mask) and whose head returns scripted logits.  Two consecutive calls per case: the first
without a drafter cache, the second on top of the cache the first one left (`stable_kv` branch).  Stored: the recorded calls
and the method's outputs.  Runs only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_anole_drafter.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

V, LO, HI, TOPK, H = 9000, 4, 8196, 10, 4


class Recorder:
    def __init__(self, script, total_tokens, depth):
        self.script, self.calls, self.seen = script, 0, []
        self.total_tokens, self.depth, self.top_k = total_tokens, depth, TOPK
        self.logsoftmax = torch.nn.LogSoftmax(dim=-1)
        self.embed_tokens = types.SimpleNamespace(weight=torch.zeros(1))
        self.tree_mask_init = torch.eye(TOPK)[None, None]
        self.position_ids = torch.zeros(TOPK, dtype=torch.long)
        self.non_image_tokens = torch.tensor(list(range(0, LO)) + list(range(HI, V)))
        self.stable_kv = None
        self.tree_mask = None
        self.kv_len = 0

    def reset(self):
        self.tree_mask = None

    def __call__(self, hidden_states, input_ids=None, past_key_values=None, position_ids=None, use_cache=True, attention_mask=None):
        T = input_ids.shape[1]
        past = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        self.seen.append(dict(ids=input_ids.clone(), pos=position_ids.clone(), past=past,
                              attn=None if attention_mask is None else attention_mask.clone(),
                              tree=None if self.tree_mask is None else self.tree_mask.clone()))
        return torch.zeros(2, T, H), ((torch.zeros(2, 1, past + T, 1),),)

    def head(self, hidden):
        blk = torch.from_numpy(self.script[self.calls])
        self.calls += 1
        return torch.stack([blk, blk])


def script_for(seed, depth, n_calls):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n_calls):
        out.append((4.0 * rs.standard_normal(V)).astype(np.float32))
        for _ in range(depth):
            out.append((4.0 * rs.standard_normal((TOPK, V))).astype(np.float32))
    return out


def main():
    R = import_reference()
    import models.drafters.cnets_anole as ca
    proc = R.ut.prepare_logits_processor(temperature=1.0, top_p=1.0, top_k=300)
    out = {}
    depth, total = 4, 59
    for ci, (L0, pad) in enumerate([(9, 0), (12, 3)]):
        rec = Recorder(script_for(40 + ci, depth, 2), total - 1, depth)
        diff = L0 - 2                                   # max_input_length - 2 (ea_model_anole.py:1040)
        attn = torch.ones(2, L0, dtype=torch.bool)
        attn[1, :L0 - 2] = False
        if pad:
            attn[0, :pad] = False
        rs = np.random.RandomState(ci)
        ids1 = torch.from_numpy(rs.randint(LO, HI, size=(1, L0 + 1))).repeat(2, 1)
        d1 = ca.Model.topK_genrate(rec, torch.zeros(2, L0, H), ids1, rec.head, proc, 3.0, diff, attn)
        n_first = len(rec.seen)
        ids2 = torch.cat([ids1, torch.from_numpy(rs.randint(LO, HI, size=(1, 3))).repeat(2, 1)], dim=1)
        d2 = ca.Model.topK_genrate(rec, torch.zeros(2, 3, H), ids2, rec.head, proc, 3.0, diff, attn)
        pre = f"c{ci}."
        out[pre + "L0"], out[pre + "diff"] = np.int64(L0), np.int64(diff)
        out[pre + "attn"] = attn.numpy()
        out[pre + "ids1"], out[pre + "ids2"] = ids1.numpy(), ids2.numpy()
        out[pre + "n_first"], out[pre + "n_calls"] = np.int64(n_first), np.int64(len(rec.seen))
        for j, s in enumerate(rec.seen):
            out[pre + f"call{j}.ids"] = s["ids"].numpy()
            out[pre + f"call{j}.pos"] = s["pos"].numpy()
            out[pre + f"call{j}.past"] = np.int64(s["past"])
            out[pre + f"call{j}.attn_same"] = np.int64(s["attn"] is not None and torch.equal(s["attn"], attn))
        for tag, d in (("out1", d1), ("out2", d2)):
            out[pre + tag + ".draft"] = d[0].numpy()
            out[pre + tag + ".retrieve"] = d[1].numpy()
            out[pre + tag + ".mask"] = d[2].numpy()
            out[pre + tag + ".pos"] = d[3].numpy()
        out[pre + "seed"] = np.int64(40 + ci)
    out["n_cases"] = np.int64(2)
    out["dims"] = np.array([V, LO, HI, TOPK, depth, total], np.int64)
    np.savez_compressed(os.path.join(HERE, "anole_drafter.npz"), **out)
    print("anole_drafter.npz ok", int(out["c0.n_calls"]), out["c0.call0.pos"], out["c0.call1.pos"][:, :3], out["c0.call5.pos"])


if __name__ == "__main__":
    main()

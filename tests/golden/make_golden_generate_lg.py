#!/usr/bin/env python3
"""tests/golden/generate_lg.npz: what the REFERENCE's own models/ea_model_llamagen.EaModel.generate and
models/ea_model_anole.EaModel.generate produce when driven, on CPU in the build container, by the scripted target models and
drafters of gen_fakes_lg.py -- dynamic (EAGLE-2) and static trees, LANTERN on and off: the generated token ids, the mean accept
length, the per-step accept lengths, the KV length, the drafter's and the target's call logs, and how many uniforms each stream gave.
tests/test_gpu_generate_lg.py holds the lantern_amd mirrors to them.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_generate_lg.py

The reference objects are built with __new__ (their constructors need CUDA and checkpoints); random.random runs from random.seed,
torch.multinomial is replaced by an inverse CDF over recorded uniforms (gen_fakes.DetDraws)."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fakes as F  # noqa: E402
import gen_fakes_lg as G  # noqa: E402
from make_golden import import_reference  # noqa: E402


def build_reference_model(R, case, T, dev):
    mod = R.lg if case["model"] == "llamagen" else R.an
    base, drafter = G.make_base(T, dev, case["model"]), G.Drafter(T, dev)
    m = mod.EaModel.__new__(mod.EaModel)
    torch.nn.Module.__init__(m)
    m.base_model, m.config = base, base.config
    object.__setattr__(m, "ea_layer", drafter)
    m.nearest_latents = torch.from_numpy(T["nb"])
    if case["model"] == "anole":
        m.tokenizer = G.Tokenizer()
        m.image_token_offset = 4
        m.non_image_tokens = torch.tensor(list(range(0, 4)) + list(range(8196, G.V)))
    return m, base, drafter


def main():
    R = import_reference()
    dev = torch.device("cpu")
    out = {}
    tabs = {}
    for case in G.CASES:
        T = tabs.setdefault(case["model"], G.tables(case["model"]))
        m, base, drafter = build_reference_model(R, case, T, dev)
        rs = np.random.RandomState(2000 + case["seed"])
        us = rs.random_sample(256)
        draws = F.DetDraws(us)
        random.seed(case["seed"])
        # per-step accept lengths: the reference only returns their mean -- record what evaluate_posterior[_v1] hands back
        steps = []
        for name in ("evaluate_posterior", "evaluate_posterior_v1"):
            fn = getattr(m, name)

            def wrapped(*a, _fn=fn, **kw):
                r = _fn(*a, **kw)
                steps.append((int(r[0]), int(r[1])))
                return r
            setattr(m, name, wrapped)
        old = torch.multinomial
        torch.multinomial = draws.multinomial
        static = case["tree"] != "dynamic"
        try:
            ids, mean_alen, _t = m.generate(prompt=case["prompt"], max_length=case["max_length"], temperature=case["temperature"], top_k=G.TOP_K,
                                            top_p=G.TOP_P, cfg=case["cfg"], lantern=case["lantern"], lantern_k=case["k"], lantern_delta=case["delta"],
                                            static_tree=static, tree_choices=getattr(R.ch, case["tree"]) if static else None)
        finally:
            torch.multinomial = old
        st = random.getstate()
        n_uniform = None
        random.seed(case["seed"])
        for n in range(200000):
            if random.getstate() == st:
                n_uniform = n
                break
            random.random()
        pre = case["name"] + "."
        out[pre + "ids"] = ids.numpy().astype(np.int64)
        out[pre + "mean_accept"] = np.float64(mean_alen)
        out[pre + "steps"] = np.asarray(steps, np.int64)                   # (best candidate, accept length) per step
        out[pre + "bonus_uniforms"] = us
        out[pre + "n_bonus_draws"] = np.int64(draws.n)
        out[pre + "n_accept_uniforms"] = np.int64(n_uniform)
        out[pre + "kv_len"] = np.int64(int(base.current_length_data[0]))
        out[pre + "drafter_calls"] = np.asarray([(t, p) for t, p, _ in drafter.calls], np.int64)
        out[pre + "target_calls"] = np.asarray(base.model.calls, np.int64)
        print(case["name"], "tokens", ids.shape[-1], "steps", len(steps), "mean accept", float(mean_alen), "uniforms", n_uniform, "bonus draws", draws.n)
    np.savez_compressed(os.path.join(HERE, "generate_lg.npz"), **out)
    print("wrote generate_lg.npz", os.path.getsize(os.path.join(HERE, "generate_lg.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

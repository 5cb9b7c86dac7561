#!/usr/bin/env python3
"""tests/golden/generate.npz: what the REFERENCE's own EaLumina_mGPT.generate / initialize_tree produce when driven, on CPU in
the build container, by the scripted target model and drafter of gen_fakes.py -- the accepted token ids, the accept-length list,
the KV lengths and the drafter's call log.  tests/test_gpu_generate_ref.py holds the lantern_amd mirror to them.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_generate.py

The reference object is built with __new__ (its constructor needs CUDA and checkpoints); random.random runs from random.seed,
torch.multinomial is replaced by an inverse CDF over recorded uniforms (gen_fakes.DetDraws)."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fakes as F  # noqa: E402
from make_golden import import_reference  # noqa: E402


def main():
    R = import_reference()
    T = F.tables()
    dev = torch.device("cpu")
    out = {}
    for case in F.CASES:
        base = F.make_base(T, dev)
        drafter = F.Drafter(T, dev)
        m = R.lum.EaLumina_mGPT.__new__(R.lum.EaLumina_mGPT)
        torch.nn.Module.__init__(m)
        m.base_model, m.config, m.dtype = base, base.config, torch.float32
        m.cfg_mode, m.eagle_version = case["cfg_mode"], 1
        object.__setattr__(m, "ea_layer", drafter)
        m.nearest_latents = torch.from_numpy(T["nb"])
        m.image_token_offset = 4
        m.image_tokens = torch.arange(4, 8196)
        m.image_syntax_tokens = torch.tensor([8196, 8197, 8803, 8828])
        m.image_start_token_id = 8197
        proc = R.lum.MultiModalLogitsProcessor.__new__(R.lum.MultiModalLogitsProcessor)
        proc.image_next_line_token_id, proc.image_end_token_id = 8803, 8196
        vocab = torch.arange(F.V)
        proc.suppress_token_mask = (vocab < 4) | (vocab > 8195)
        m.internal_logits_processors = [proc]
        m.drafter_logits_processors = [proc]
        rs = np.random.RandomState(1000 + case["seed"])
        us = rs.random_sample(256)
        draws = F.DetDraws(us)
        random.seed(case["seed"])
        old = torch.multinomial
        torch.multinomial = draws.multinomial
        try:
            ids, alens = m.generate(torch.tensor([F.PROMPT]), max_new_tokens=case["max_new"], cfg_scale=3.0, top_k=2000, logits_processors=[None],
                                    lantern=case["lantern"], lantern_k=case["k"], lantern_delta=case["delta"], tree_choices=R.ch.mc_sim_7b_63)
        finally:
            torch.multinomial = old
        n_uniform = None
        st = random.getstate()
        # how many random.random() draws the run consumed: replay the seed until the state matches
        random.seed(case["seed"])
        for n in range(100000):
            if random.getstate() == st:
                n_uniform = n
                break
            random.random()
        pre = case["name"] + "."
        out[pre + "ids"] = ids[0].numpy().astype(np.int64) if ids.dim() == 2 else ids.numpy()
        out[pre + "accept_lengths"] = np.asarray(alens, np.int64)
        out[pre + "bonus_uniforms"] = us
        out[pre + "n_bonus_draws"] = np.int64(draws.n)
        out[pre + "n_accept_uniforms"] = np.int64(n_uniform)
        cl = m.current_length_data
        out[pre + "kv_len"] = np.asarray([int(cl[k][0]) for k in ("cond", "uncond")] if isinstance(cl, dict) else [int(cl[0])], np.int64)
        out[pre + "drafter_calls"] = np.asarray([(t, p) for t, p, _ in drafter.calls], np.int64)
        print(case["name"], "tokens", ids.shape[-1], "steps", len(alens), "mean accept", float(np.mean(alens)), "uniforms", n_uniform, "bonus draws", draws.n)
    np.savez_compressed(os.path.join(HERE, "generate.npz"), **out)


if __name__ == "__main__":
    main()

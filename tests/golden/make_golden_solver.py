#!/usr/bin/env python3
"""Golden vectors for the HF-signature logits processors of the Lumina solver, produced by RUNNING the
reference's own `MultiModalLogitsProcessor.__call__` and `InterleavedTopKLogitsWarper.__call__`
(models/base_models/lumina_mgpt/eagle_inference_solver.py:100-232).  The module itself cannot be imported here (its item
processor needs the absent `xllmx` package and the constructor allocates on "cuda"), so the two class definitions are compiled
straight from the reference file at generation time and `__call__` runs on an instance whose constructor state is filled in on
the CPU.  Nothing of the reference's text is stored: the fixture holds input token rows, the score seed, and for every case the
bit mask of finite outputs + the forced value.  Runs only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_solver.py
"""
import ast
import io
import math
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/models/base_models/lumina_mgpt/eagle_inference_solver.py"
V, BOI, EOI, NL = 65536, 8197, 8196, 8803


def reference_classes():
    from transformers.generation.logits_process import LogitsProcessor
    tree = ast.parse(open(REF_FILE).read())
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("MultiModalLogitsProcessor", "InterleavedTopKLogitsWarper")]
    ns = {"torch": torch, "math": math, "LogitsProcessor": LogitsProcessor, "LogitsWarper": LogitsProcessor}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF_FILE, "exec"), ns)
    return ns["MultiModalLogitsProcessor"], ns["InterleavedTopKLogitsWarper"]


def scores_for(seed):
    return torch.from_numpy((4.0 * np.random.RandomState(seed).standard_normal((1, V))).astype(np.float32))


def cases():
    rs = np.random.RandomState(7)
    prompt = rs.randint(9000, 12000, size=9).tolist()
    img = lambda n: rs.randint(4, 8196, size=n).tolist()          # noqa: E731
    h, w = 4, 6                                                    # 2x3 grids -> 4 x 6 latent tokens (+ newline per row)
    head = prompt + [BOI, 8804 + h // 2, 8804 + w // 2]
    row = lambda: img(w) + [NL]                                    # noqa: E731
    full = head + sum((row() for _ in range(h)), [])
    out = {
        "text_only": prompt,
        "after_boi": prompt + [BOI],                               # fewer than 2 tokens after <boi>: untouched
        "first_image_token": head,
        "mid_row": head + img(3),
        "row_end": head + img(w),                                  # next token must be the newline
        "second_row": head + row() + img(1),
        "last_row_end": full[:-1],                                 # newline of the last row
        "image_end": full,                                         # next token must be <eoi>
        "closed": full + [EOI] + prompt[:2],                       # image closed again: text rules
    }
    return out


def main():
    MM, TK = reference_classes()
    vocab = torch.arange(V)
    out = {}
    names = []
    for ci, (name, ids) in enumerate(cases().items()):
        mm = object.__new__(MM)
        mm.image_start_token_id, mm.image_end_token_id, mm.image_next_line_token_id = BOI, EOI, NL
        mm.image_start_token_id_index = mm.h_latent_dim = mm.w_latent_dim = None
        mm.suppress_token_mask = (vocab < 4) | (vocab > 8195)
        tk = TK(image_top_k=2000, text_top_k=10, image_start_token_id=BOI, image_end_token_id=EOI)
        input_ids = torch.tensor([ids], dtype=torch.long)
        s = scores_for(100 + ci)
        with redirect_stdout(io.StringIO()):
            a = mm(input_ids, s.clone())
            b = tk(input_ids, a.clone())
        for tag, t in (("mm", a), ("tk", b)):
            fin = torch.isfinite(t[0]).numpy()
            out[f"{name}.{tag}.finite"] = np.packbits(fin)
            changed = fin & (t[0].numpy() != s[0].numpy())
            out[f"{name}.{tag}.forced_idx"] = np.nonzero(changed)[0].astype(np.int64)
            out[f"{name}.{tag}.forced_val"] = t[0].numpy()[changed]
        out[f"{name}.ids"] = np.asarray(ids, dtype=np.int64)
        out[f"{name}.seed"] = np.int64(100 + ci)
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "solver.npz"), **out)
    print("wrote solver.npz:", names)


if __name__ == "__main__":
    sys.exit(main())

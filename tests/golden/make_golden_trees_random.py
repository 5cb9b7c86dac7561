#!/usr/bin/env python3
"""Golden tree buffers for RANDOM tree shapes: the reference's own `generate_tree_buffers` (target side,
models/ea_model_lumina_mgpt.py:140-277) and the drafter-side builder (models/drafters/utils_c.py:100-179) run on 40 random
prefix-closed choice lists -- various widths, depths up to 7, given in shuffled order (the builders sort them) -- so that the
orderings the six trees of choices.py never hit (wide levels, lone deep chains, equal-length ties) are pinned too.  Same key
layout as trees.npz.  Runs only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_trees_random.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, nested_b  # noqa: E402


def random_tree(rs, n_nodes, max_depth, top_k=10):
    """Prefix-closed set of child-index paths; a child index is < top_k and children of a node need not be contiguous."""
    nodes = {()}
    paths = []
    guard = 0
    while len(paths) < n_nodes and guard < 10000:
        guard += 1
        parent = list(nodes)[rs.randint(len(nodes))]
        if len(parent) >= max_depth:
            continue
        child = parent + (int(rs.randint(top_k if rs.rand() < 0.3 else 4)),)
        if child in nodes:
            continue
        nodes.add(child)
        paths.append(list(child))
    rs.shuffle(paths)
    return paths


def main():
    R = import_reference()
    rs = np.random.RandomState(4242)
    data, names = {}, []
    t = 0
    while len(names) < 40:
        t += 1
        choices = random_tree(rs, int(rs.randint(3, 62)), int(rs.randint(1, 8)))
        if len(choices) < 2:
            continue
        tb = R.lum.generate_tree_buffers(choices, device="cpu")
        if tb["retrieve_indices"].shape[0] > 64:
            continue
        try:
            db = R.uc.generate_tree_buffers(choices, device="cpu")
        except IndexError:          # a tree without an inner node below the root: the reference's drafter builder indexes an empty list
            continue
        nm = f"rand{len(names):02d}"
        boff, bidx = nested_b(tb["b_indices"])
        flat = [x for c in choices for x in c]
        coff = np.cumsum([0] + [len(c) for c in choices])
        data[f"{nm}.choices"] = np.asarray(flat, np.int32)
        data[f"{nm}.choice_off"] = np.asarray(coff, np.int32)
        data[f"{nm}.mask"] = tb["tree_attn_mask"][0, 0].numpy()
        data[f"{nm}.tree_indices"] = tb["tree_indices"].numpy()
        data[f"{nm}.pos"] = tb["tree_position_ids"].numpy()
        data[f"{nm}.retrieve"] = tb["retrieve_indices"].numpy()
        data[f"{nm}.p_indices"] = np.asarray(tb["p_indices"], np.int32)
        data[f"{nm}.b_off"] = boff
        data[f"{nm}.b_idx"] = bidx
        data[f"{nm}.d_levels"] = np.asarray([len(db["tree_indices"])], np.int32)
        for l in range(len(db["tree_indices"])):
            data[f"{nm}.d_mask{l}"] = db["attn_mask"][l][0, 0].numpy()
            data[f"{nm}.d_ti{l}"] = db["tree_indices"][l].numpy()
            data[f"{nm}.d_rep{l}"] = np.asarray(db["repeat_nums"][l], np.int32)
        names.append(nm)
    data["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, "trees_random.npz"), **data)
    print("trees_random.npz:", len(names), "trees; sizes", sorted(len(data[f"{n}.tree_indices"]) for n in names))


if __name__ == "__main__":
    main()

"""GPU (-m gpu): the LlamaGen and Anole mirrors' generate() against what the REFERENCE's own models/ea_model_llamagen.EaModel.generate /
models/ea_model_anole.EaModel.generate produced for the same scripted target models and drafters (tests/golden/generate_lg.npz, made by
make_golden_generate_lg.py in the build container): token ids, mean accept length, every step's (best path, accept length), the KV
length, the drafter's and the target's call logs, and the number of uniforms drawn from `random` and for the bonus tokens -- exact.
Dynamic (EAGLE-2) and static trees, LANTERN on and off, both kernel sets."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_fakes as F  # noqa: E402
import gen_fakes_lg as G  # noqa: E402
from lantern_amd.drafters import choices  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "generate_lg.npz"))
_T = {}


def tables(model):
    if model not in _T:
        _T[model] = G.tables(model)
    return _T[model]


def build(case, kernel_set):
    dev = torch.device("cuda")
    T = tables(case["model"])
    base, drafter = G.make_base(T, dev, case["model"]), G.Drafter(T, dev)
    table = torch.from_numpy(T["nb"].astype(np.uint16).view(np.int16)).to(dev)
    if case["model"] == "llamagen":
        from lantern_amd.ea_model_llamagen import EaModel
        mdl = EaModel(base, drafter, table)
    else:
        from lantern_amd.ea_model_anole import EaModel
        mdl = EaModel(base, drafter, table, tokenizer=G.Tokenizer())
    mdl.kernel_set = kernel_set
    mdl.uniform_window = 64                        # small window: the refill path runs too
    return mdl, base, drafter


def run_case(case, kernel_set="window", mdl_parts=None, native=True):
    mdl, base, drafter = mdl_parts or build(case, kernel_set)
    mdl.native_step = native                       # True: a static-tree step through ONE lantern_verify_step call; False: a ctypes call per kernel
    g = lambda k: GOLD[case["name"] + "." + k]
    draws = F.DetDraws(g("bonus_uniforms"))
    random.seed(case["seed"])
    old_m, old_r = torch.multinomial, torch.rand
    torch.multinomial, torch.rand = draws.multinomial, draws.rand
    static = case["tree"] != "dynamic"
    try:
        ids, mean_alen, _t = mdl.generate(prompt=case["prompt"], max_length=case["max_length"], temperature=case["temperature"], top_k=G.TOP_K,
                                          top_p=G.TOP_P, cfg=case["cfg"], lantern=case["lantern"], lantern_k=case["k"], lantern_delta=case["delta"],
                                          static_tree=static, tree_choices=getattr(choices, case["tree"]) if static else None)
    finally:
        torch.multinomial, torch.rand = old_m, old_r
    return mdl, base, drafter, draws, ids, mean_alen


@pytest.mark.parametrize("case", G.CASES, ids=[c["name"] for c in G.CASES])
@pytest.mark.parametrize("kernel_set,native", [("window", True), ("window", False), ("dense", False)], ids=["window_one_call_step", "window", "dense"])
def test_generate_reproduces_the_reference_run(case, kernel_set, native, monkeypatch):
    calls = []
    static = case["tree"] != "dynamic"
    if native:          # the one-call step really is the path taken: sampled runs and -- round 5 -- greedy runs (lantern_step_greedy), static and EAGLE-2 trees
        from lantern_amd import ops
        fns = ("evaluate_posterior_window", "update_inference_inputs", "cfg_mask_topk_window") if case["temperature"] > 1e-5 else \
              ("evaluate_posterior_greedy", "update_inference_inputs", "accept_gather")          # (cfg_mask_topk: the prefill picks the first token with it)
        for fn in fns:
            real = getattr(ops, fn)
            monkeypatch.setattr(ops, fn, (lambda real, fn: (lambda *a, **k: (calls.append(fn), real(*a, **k))[1]))(real, fn))
    mdl, base, drafter, draws, ids, mean_alen = run_case(case, kernel_set, native=native)
    assert not calls, calls
    g = lambda k: GOLD[case["name"] + "." + k]
    assert ids.cpu().numpy().tolist() == g("ids").tolist()
    assert mean_alen == float(g("mean_accept"))
    assert [tuple(x) for x in mdl.last_steps] == [tuple(x) for x in g("steps").tolist()]
    assert draws.n == int(g("n_bonus_draws"))
    assert int(base.current_length_data[0]) == int(g("kv_len"))
    assert [(t, p) for t, p, _ in drafter.calls] == [tuple(x) for x in g("drafter_calls").tolist()]
    assert [tuple(x) for x in base.model.calls] == [tuple(x) for x in g("target_calls").tolist()]
    # the module-level generator ends where the reference's own random.random() calls left it
    st = random.getstate()
    random.seed(case["seed"])
    for _ in range(int(g("n_accept_uniforms"))):
        random.random()
    assert random.getstate() == st


def test_generate_reads_the_device_once_per_step():
    """The decode loop keeps a step's results in HBM (verdict -> KV rows, accepted hidden rows, bonus token) and reads ONE packed
    record per step.  torch's sync-debug mode reports every synchronising call (host reads AND pageable host-to-device copies); each
    is attributed to the innermost frame of this package or of the test doubles, and only the package's own are counted."""
    import collections
    import traceback
    import warnings
    for case in (G.CASES[0], G.CASES[2]):          # a dynamic-tree and a static-tree run
        parts = build(case, "window")
        parts[0].uniform_window = 4096             # the default staging window: one refill per few hundred steps
        run_case(case, "window", parts)            # first run: lazy initialisation (tree buffers, packed table, KV cache) out of the way
        parts[2].calls.clear()
        parts[1].model.calls.clear()
        torch.cuda.synchronize()
        own = collections.Counter()

        def show(message, category, filename, lineno, file=None, line=None):
            if "synchroniz" not in str(message).lower():
                return
            for fr in reversed(traceback.extract_stack()[:-1]):
                if os.sep + "lantern_amd" + os.sep in fr.filename:
                    own[(os.path.basename(fr.filename), fr.lineno)] += 1
                    return
                if os.sep + "tests" + os.sep in fr.filename and "gen_fakes" in fr.filename:
                    return
        old_show = warnings.showwarning
        with warnings.catch_warnings():
            warnings.simplefilter("always")
            warnings.showwarning = show
            torch.cuda.set_sync_debug_mode("warn")
            try:
                mdl, base, drafter, draws, ids, mean_alen = run_case(case, "window", parts)
            finally:
                torch.cuda.set_sync_debug_mode("default")
                warnings.showwarning = old_show
        n_steps = len(mdl.last_steps)
        per_step = [k for k, v in own.items() if v >= n_steps]
        assert len(per_step) == 1 and own[per_step[0]] == n_steps, (case["name"], n_steps, own)      # the packed read, once per step
        assert sum(own.values()) <= n_steps + 10, (case["name"], n_steps, own)                       # + set-up: prompt, staging, slab pointers
        assert ids.cpu().numpy().tolist() == GOLD[case["name"] + ".ids"].tolist()

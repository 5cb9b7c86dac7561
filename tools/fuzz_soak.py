"""Extended soak of tests/test_gpu_fuzz.py::test_static_batches_vs_oracle over many more seeds (development aid, not part of the suite):
    python tools/fuzz_soak.py <seeds>   ->  8 parameter sets x <seeds> batches of 32 sequences, both kernel sets vs the oracle."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import test_gpu_fuzz as F
params = [("lumina", "mc_sim_7b_63", True, 100, 0.1, 1.0), ("lumina", "mc_sim_7b_63", True, 300, 5.0, 2.0),
          ("lumina", "naive_extend_57", True, 10, 0.3, 0.5), ("llamagen", "naive_extend_57", True, 50, 0.1, 1.0),
          ("llamagen", "mc_sim_7b_63", True, 200, 10.0, 2.0), ("anole", "naive_extend_57", True, 10, 5.0, 1.0),
          ("anole", "naive_extend_57", True, 5, 20.0, 3.0), ("anole", "mc_sim_7b_63", False, 1, 0.1, 0.5)]
t0 = time.time(); n = 0; fails = 0
for seed in range(100, 100 + (int(sys.argv[1]) if len(sys.argv) < 3 else 0)):
    for p in params:
        try:
            F.test_static_batches_vs_oracle(*p, seed)
        except AssertionError as e:
            fails += 1
            print("FAIL", p, seed, str(e)[:300], flush=True)
        n += 1
if len(sys.argv) < 3:
    print(f"cases={n} batches x 32 sequences, fails={fails}, {time.time() - t0:.0f}s")


# ---------------------------------------------------------------------------------------------- dynamic trees
# `python tools/fuzz_soak.py <seeds> dynamic`: EAGLE-2 trees (N = 59, random shapes per sequence) built by the oracle from
# random drafter scores, target rows that make the drafted tokens plausible, both kernel sets in ragged batches of 16 sequences
# (per-sequence row maps, -1 padded paths) against the oracle.
def dynamic_soak(n_seeds):
    import numpy as np
    import torch
    import cases as CS
    import helpers as H
    import oracle
    from lantern_amd import ops
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    B = 16
    sets = [("lumina", True, 100, 0.1), ("lumina", True, 300, 5.0), ("anole", True, 10, 5.0), ("anole", False, 1, 0.1),
            ("llamagen", True, 50, 0.1), ("llamagen", True, 200, 10.0)]
    n = fails = 0
    t0 = time.time()
    tables = {}
    for seed in range(1000, 1000 + n_seeds):
        for model, lantern, k, delta in sets:
            m = CS.MODELS[model]
            V, lo = m["V"], (m["img_lo"] if model != "llamagen" else 0)
            W = (m["img_hi"] - m["img_lo"]) if model != "llamagen" else V
            if m["K"] not in tables:
                tables[m["K"]] = CS.build_table(m["K"])
            table = tables[m["K"]]
            mk = {"lumina": lambda E: E.lumina(False, lantern=lantern, k=k, delta=delta),
                  "anole": lambda E: E.anole(False, lantern=lantern, k=k, delta=delta),
                  "llamagen": lambda E: E.llamagen(False, lantern=lantern, k=k, delta=delta)}[model]
            co, ch = mk(oracle.EpConfig), mk(ops.EpConfig)
            for c in (co, ch):
                c.img_lo, c.img_hi, c.tok_offset = m["img_lo"], m["img_hi"], m["off"]
                if model == "lumina":
                    c.syntax = tuple(m["syntax"])
                else:
                    c.temperature, c.top_k = 0.9, 150
            seqs = []
            for b in range(B):
                g = CS.gen_dynamic(seed * 100 + b, model, sigma=float(1 + (seed + b) % 3))
                draft, ret, mask, pos = oracle.tree_dynamic_finalize(g["scores"], g["tokens"], g["parents"], CS.TOPK, g["total_tokens"],
                                                                     g["sample_token"])
                N = len(draft)
                rs = np.random.RandomState(seed * 100 + b + 7)
                nl = (4.0 * rs.standard_normal((N, V))).astype(np.float32)
                if model in ("lumina", "anole"):
                    nl[:, :m["img_lo"]] = -np.inf
                    nl[:, m["img_hi"]:] = -np.inf
                if model == "lumina":
                    nl = CS.topk_filter(nl, 200)
                for p in range(ret.shape[0]):
                    for d in range(1, ret.shape[1]):
                        if ret[p, d] >= 0:
                            par, tok = ret[p, d - 1], draft[ret[p, d]]
                            nl[par, tok] = np.max(nl[par][np.isfinite(nl[par])]) - rs.uniform(0.0, 3.0)
                cand = np.where(ret >= 0, draft[np.maximum(ret, 0)], -1)
                seqs.append(dict(nl=nl, cand=cand, ri=H.row_index_from_retrieve(ret, N), uni=rs.random_sample(64)))
            Pm, Dm = max(s["cand"].shape[0] for s in seqs), max(s["cand"].shape[1] for s in seqs)
            N = seqs[0]["nl"].shape[0]
            cand = np.full((B, Pm, Dm), -1, np.int64)
            ri = np.zeros((B, Pm, Dm), np.int32)
            for b, s in enumerate(seqs):
                P, D = s["cand"].shape
                cand[b, :P, :D], ri[b, :P, :D] = s["cand"], s["ri"]
            nl = np.stack([s["nl"] for s in seqs])
            uni = np.stack([s["uni"] for s in seqs])
            tab = dev(table.view(np.int16)) if lantern else None
            dense = ops.evaluate_posterior(ch, dev(nl), dev(ri), dev(cand), dev(uni), table=tab)
            # windowed set: LlamaGen / Anole rows go through the processors in O8 (LANTERN_ROWS_LOGITS, the reference's order)
            win = ops.evaluate_posterior_window(ch, V, dev(np.ascontiguousarray(nl[:, :, lo:lo + W])), lo, dev(ri), dev(cand), dev(uni), table=tab,
                                                want_dense=True)
            for b, s in enumerate(seqs):
                ob, oa, osp, ocnt = oracle.evaluate_posterior(co, s["nl"], s["ri"], s["cand"], s["uni"], table=table if lantern else None)
                for name, best, alen, sp, cnt in (("dense", dense[0], dense[1], dense[2], dense[3]),
                                                  ("window", win["best"], win["accept_len"], win["sample_p"], win["counters"])):
                    ok = (int(cnt[b, 5]) == 0 and (int(best[b]), int(alen[b])) == (ob, oa) and np.array_equal(cnt[b, :5].cpu().numpy(), ocnt[:5])
                          and np.abs(sp[b].cpu().numpy() - osp).max() <= 1e-5)
                    if not ok:
                        fails += 1
                        print("FAIL", name, model, lantern, k, delta, seed, b, int(cnt[b, 5]), (int(best[b]), int(alen[b])), (ob, oa), flush=True)
                n += 1
    print(f"dynamic: sequences={n}, fails={fails}, {time.time() - t0:.0f}s")


if len(sys.argv) > 2 and sys.argv[2] == "dynamic":
    dynamic_soak(int(sys.argv[1]))

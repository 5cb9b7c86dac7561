"""Extended soak of tests/test_gpu_fuzz.py::test_static_batches_vs_oracle over many more seeds (development aid, not part of the suite):
    python tests/fuzz_soak.py <seeds>   ->  8 parameter sets x <seeds> batches of 32 sequences, both kernel sets vs the oracle."""
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
SEED0 = int(os.environ.get("FUZZ_SEED0", "100"))          # (another range of seeds: FUZZ_SEED0=1000 python tests/fuzz_soak.py 100)
for seed in range(SEED0, SEED0 + (int(sys.argv[1]) if len(sys.argv) < 3 else 0)):
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
# `python tests/fuzz_soak.py <seeds> dynamic`: EAGLE-2 trees (N = 59, random shapes per sequence) built by the oracle from
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
                # O4 on the device for the same scores (one sequence per launch here; batched launches are in the test-suite)
                gd, gm, gp, gr, gnl, gmd = ops.tree_dynamic_finalize(dev(g["scores"])[None], dev(g["tokens"])[None], dev(g["parents"])[None],
                                                                     torch.tensor([g["sample_token"]], device="cuda"), CS.TOPK, g["total_tokens"])
                nl_, md_ = int(gnl[0]), int(gmd[0])
                if not (np.array_equal(gd[0].cpu().numpy(), draft) and np.array_equal(gr[0, :nl_, :md_].cpu().numpy(), ret)
                        and np.array_equal(gm[0].cpu().numpy(), mask) and np.array_equal(gp[0].cpu().numpy(), pos)):
                    fails += 1
                    print("FAIL finalize", model, seed, b, flush=True)
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


# ---------------------------------------------------------------------------------------------- O7
# `python tests/fuzz_soak.py <iters> o7`: lantern_cfg_mask_topk_window (and the dense lantern_cfg_mask_topk) on random shapes --
# vocabulary, window, dtype, CFG scale, top-k, model mask, Lumina grid positions incl. newline / end-of-image rows -- against the
# oracle's restatement: processed logits bit for bit, probability rows within 1e-7.
def o7_soak(iters):
    import numpy as np
    import torch
    import cases as CS
    import oracle
    from lantern_amd import ops
    rs = np.random.RandomState(2024)
    fails = n = 0
    t0 = time.time()
    for it in range(iters):
        model = [ops.MODEL_PLAIN, ops.MODEL_ANOLE, ops.MODEL_LUMINA][it % 3]
        bf16 = bool(rs.randint(2))
        if model == ops.MODEL_PLAIN:
            V = int(rs.choice([1024, 2048, 4096, 16384]))
            lo, W, img_lo, img_hi = 0, V, 0, V
        else:
            V = int(rs.choice([2048, 16384, 65536]))
            W = {2048: 1024, 16384: 4096, 65536: 8192}[V]
            lo, img_lo, img_hi = 4, 4, 4 + W
        rows = int(rs.randint(1, 40))
        top_k = int(rs.choice([0, 1, 7, W // 4, W - 1, W, min(V, 2000)]))
        cfg = float(rs.choice([1.0, 3.0, 4.5, 7.5]))
        w_lat, h_lat = int(rs.choice([4, 6, 48])), int(rs.choice([3, 48]))
        cond = (float(rs.uniform(1, 5)) * rs.standard_normal((rows, V))).astype(np.float32)
        unc = rs.standard_normal((rows, V)).astype(np.float32)
        ct, ut = torch.from_numpy(cond), torch.from_numpy(unc)
        if bf16:
            ct, ut = ct.to(torch.bfloat16), ut.to(torch.bfloat16)
            cb, ub = ct.view(torch.int16).numpy().view(np.uint16), ut.view(torch.int16).numpy().view(np.uint16)
        else:
            cb, ub = cond, unc
        pos = rs.randint(0, (w_lat + 1) * h_lat + 3, size=rows).astype(np.int64) + 20
        kw = dict(model=model, pos_ids=torch.from_numpy(pos).cuda() if model == ops.MODEL_LUMINA else None, pos_base=20, w=w_lat, h=h_lat,
                  img_lo=img_lo, img_hi=img_hi, newline_id=img_hi + 7, eos_id=img_hi, top_k=top_k)
        exp = oracle.cfg_mask_topk(cb, ub, cfg, model={ops.MODEL_PLAIN: oracle.MODEL_PLAIN, ops.MODEL_ANOLE: oracle.MODEL_ANOLE,
                                                       ops.MODEL_LUMINA: oracle.MODEL_LUMINA}[model],
                                   pos_ids=pos, pos_base=20, w=w_lat, h=h_lat, img_lo=img_lo, img_hi=img_hi, newline_id=img_hi + 7, eos_id=img_hi,
                                   top_k=top_k if model == ops.MODEL_LUMINA else 0, bf16=bf16)
        if model != ops.MODEL_LUMINA and 0 < top_k < V:
            # the oracle's O7 carries the reference's order (only Lumina filters in tree_decoding); the kernels accept top_k for
            # every model (LlamaGen / Anole windowed set: processors applied where the rows are produced)
            exp = CS.topk_filter(exp, top_k)
        dense = ops.cfg_mask_topk(ct.cuda(), ut.cuda(), cfg, **kw).cpu().numpy()
        win, hot = ops.cfg_mask_topk_window(ct.cuda(), ut.cuda(), cfg, lo, W, **kw)
        pw, hot2 = ops.cfg_mask_topk_window(ct.cuda(), ut.cuda(), cfg, lo, W, probs=True, **kw)
        win, hot, pw = win.cpu().numpy(), hot.cpu().numpy(), pw.cpu().numpy()
        why = []
        if not np.array_equal(dense, exp):
            why.append(f"dense != oracle at {int((dense != exp).sum())} entries")
        if not np.array_equal(hot, hot2.cpu().numpy()):
            why.append("row_hot differs between logits and probs mode")
        for r in range(rows):
            fin = np.nonzero(np.isfinite(exp[r]))[0]
            if hot[r] >= 0:
                if not (len(fin) == 1 and fin[0] == hot[r]):
                    why.append(f"row {r}: hot {hot[r]} vs oracle finite {fin[:4]}")
            else:
                if not np.array_equal(win[r], exp[r, lo:lo + W]):
                    why.append(f"row {r}: window != oracle at {int((win[r] != exp[r, lo:lo + W]).sum())}")
                if model != ops.MODEL_ANOLE and np.isfinite(np.delete(exp[r], np.s_[lo:lo + W])).any():
                    why.append(f"row {r}: oracle finite outside the window")
                ref = CS.softmax64(exp[r, lo:lo + W][None])[0]
                if float(np.abs(pw[r] - ref).max()) > 1e-7:
                    why.append(f"row {r}: probs off by {float(np.abs(pw[r] - ref).max()):.2e}")
        ok = not why
        n += 1
        if not ok:
            fails += 1
            print("FAIL o7", it, dict(model=model, bf16=bf16, V=V, W=W, rows=rows, top_k=top_k, cfg=cfg, w=w_lat, h=h_lat), why[:3], flush=True)
    print(f"o7: cases={n}, fails={fails}, {time.time() - t0:.0f}s")


if len(sys.argv) > 2 and sys.argv[2] == "o7":
    o7_soak(int(sys.argv[1]))


# ---------------------------------------------------------------------------------------------- O3
# `python tests/fuzz_soak.py <iters> o3`: lantern_expand_dynamic (log-softmax + top-10 per row + merge) against the oracle on random
# rows: vocabularies 1024..65536, -inf masked regions (down to fewer than ten finite entries), heavy ties (few distinct values:
# the candidate list overflows and the plain path runs), first level (no incoming scores) and deeper levels.
def o3_soak(iters):
    import numpy as np
    import torch
    import oracle
    from lantern_amd import ops
    rs = np.random.RandomState(77)
    fails = 0
    t0 = time.time()
    for it in range(iters):
        V = int(rs.choice([1024, 4096, 8192, 65536]))
        first = it % 4 == 0
        R = 1 if first else 10
        x = (float(rs.uniform(0.5, 6)) * rs.standard_normal((R, V))).astype(np.float32)
        mode = it % 5
        if mode == 1:                                   # image-window style mask
            lo = int(rs.randint(0, V // 2)); hi = lo + int(rs.randint(16, V // 2))
            x[:, :lo] = -np.inf; x[:, hi:] = -np.inf
        elif mode == 2:                                 # ties everywhere
            x = np.round(x * float(rs.choice([0.5, 2, 8]))) / 8
        elif mode == 3:                                 # almost everything masked: 3..40 finite entries per row
            keep = int(rs.randint(3, 40))
            for r in range(R):
                idx = rs.choice(V, size=V - keep, replace=False)
                x[r, idx] = -np.inf
        elif mode == 4:                                 # top-k filtered row (2000 finite)
            kth = np.sort(x, axis=-1)[:, -min(2000, V)][:, None]
            x[x < kth] = -np.inf
        sc = None if first else (rs.standard_normal(R).astype(np.float32) - 3)
        oti, ocu, oci, osc = oracle.expand_dynamic(x, sc, 10)
        ti, cu, ci, so = ops.expand_dynamic(torch.from_numpy(x).cuda()[None], None if first else torch.from_numpy(sc).cuda()[None], 10)
        ok = np.array_equal(ti[0].cpu().numpy(), oti) and np.array_equal(ci[0].cpu().numpy(), oci)
        a, b = cu[0].cpu().numpy(), ocu
        fin = np.isfinite(b)
        ok = ok and np.array_equal(np.isfinite(a), fin) and (not fin.any() or float(np.abs(a[fin] - b[fin]).max()) <= 1e-5)
        ok = ok and float(np.abs(np.nan_to_num(so[0].cpu().numpy(), neginf=0) - np.nan_to_num(osc, neginf=0)).max()) <= 1e-5
        if not ok:
            fails += 1
            print("FAIL o3", it, dict(V=V, R=R, mode=mode), flush=True)
    print(f"o3: cases={iters}, fails={fails}, {time.time() - t0:.0f}s")


if len(sys.argv) > 2 and sys.argv[2] == "o3":
    o3_soak(int(sys.argv[1]))


# ---------------------------------------------------------------------------------------------- node kernels
# `python tests/fuzz_soak.py <seeds> nodes`: tests/test_gpu_nodes.py::test_nodes_static_batches_vs_oracle over many more seeds -- the chain
# kernel and the node-parallel kernels on the same 32-sequence batches, both against the oracle and each other.
def nodes_soak(n_seeds):
    import test_gpu_nodes as TN
    sets = [("lumina", "mc_sim_7b_63", True, 100, 0.1, 1.0, True), ("lumina", "mc_sim_7b_63", True, 300, 5.0, 2.0, False),
            ("lumina", "naive_extend_57", True, 10, 0.3, 0.5, True), ("lumina", "mc_sim_7b_63", False, 1, 0.1, 3.0, False),
            ("llamagen", "naive_extend_57", True, 50, 0.1, 1.0, True), ("llamagen", "mc_sim_7b_63", True, 200, 10.0, 2.0, False),
            ("anole", "naive_extend_57", True, 10, 5.0, 1.0, True), ("anole", "mc_sim_7b_63", False, 1, 0.1, 0.5, False),
            ("lumina", "rand07", True, 60, 0.2, 2.5, True), ("lumina", "rand21", True, 500, 0.1, 4.0, True)]
    fn = getattr(TN.test_nodes_static_batches_vs_oracle, "__wrapped__", TN.test_nodes_static_batches_vs_oracle)
    n = fails = 0
    t0 = time.time()
    for seed in range(200, 200 + n_seeds):
        for model, tree, lantern, k, delta, sigma, packed in sets:
            try:
                fn(model, tree, lantern, k, delta, sigma, seed, packed)
            except AssertionError as e:
                fails += 1
                print("FAIL", (model, tree, lantern, k, delta, sigma, seed, packed), str(e)[:300], flush=True)
            n += 1
        if (seed - 200) % 5 == 4:
            print(f"  seed {seed}: {n} batches, fails={fails}, {time.time() - t0:.0f}s", flush=True)
    print(f"nodes soak: cases={n} batches x 32 sequences x 2 kernels, fails={fails}, {time.time() - t0:.0f}s")


if len(sys.argv) > 2 and sys.argv[2] == "nodes":
    nodes_soak(int(sys.argv[1]))


# ---------------------------------------------------------------------------------------------- stream-K GEMM
# `python tests/fuzz_soak.py <shapes> streamk`: lantern_linear_rows_streamk on random shapes (rows 1..32, K a multiple of 16 or 64, ragged column
# counts, the three epilogues, row-major and packed weights), every shape launched `REPEAT` times back to back while a second stream keeps the
# GPU busy with copies: every repeat must be BIT-identical to the first (the partial tiles of a split tile meet through device-coherent stores
# and a counter: a visibility bug shows as a sporadic difference) and within the bf16 tolerance of the f64 product.
def streamk_soak(n_shapes):
    import numpy as np
    import torch
    from lantern_amd import ops
    dev, bf = torch.device("cuda"), torch.bfloat16
    rs = np.random.RandomState(4242)
    REPEAT = 60
    side = torch.cuda.Stream()
    noise_a, noise_b = torch.empty(64 << 20, dtype=torch.uint8, device=dev), torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    fails = n = 0
    t0 = time.time()
    for it in range(n_shapes):
        M = int(rs.randint(1, 33))
        packed = bool(rs.randint(0, 2))
        K = int(rs.randint(1, 180)) * (64 if packed else 16)
        N = int(rs.choice([int(rs.randint(1, 300)), int(rs.randint(300, 13000)), 4096, 11008]))
        epi = int(rs.randint(0, 3))
        if (2 if epi == 2 else 1) * N * K > 120e6:
            N = max(1, int(120e6 / K / (2 if epi == 2 else 1)))
        gen = torch.Generator(device="cuda").manual_seed(it)
        x = torch.randn(M, K, device=dev, dtype=bf, generator=gen)
        rows = 2 * N if epi == 2 else N
        w = (torch.randn(rows, K, device=dev, generator=gen) / K ** 0.5).to(bf)
        b = (0.1 * torch.randn(rows, device=dev, generator=gen)).to(bf) if rs.rand() < 0.7 else None
        res = torch.randn(M, N, device=dev, dtype=bf, generator=gen)
        xd, wd = x.double(), w.double()
        bd = b.double() if b is not None else torch.zeros(rows, dtype=torch.float64, device=dev)
        if epi == 0:
            want, kw = xd @ wd.T + bd, {}
        elif epi == 1:
            want, kw = (xd @ wd.T + bd) + res.double(), dict(residual=res)
        else:
            want, kw = torch.nn.functional.silu(xd @ wd[:N].T + bd[:N]) * (xd @ wd[N:].T + bd[N:]), dict(pair_rows=N)
        wt = ops.pack_linear_weight(w, N if epi == 2 else 0) if packed else w
        with torch.cuda.stream(side):
            for _ in range(8):
                noise_b.copy_(noise_a, non_blocking=True)
        outs = [ops.linear_rows_streamk(x, wt, epi, bias=b, **kw) for _ in range(REPEAT)]
        torch.cuda.synchronize()
        scale = max(want.abs().max().item(), 1e-6)
        bad = sum(int(not torch.equal(o, outs[0])) for o in outs[1:])
        err = (outs[0].double() - want).abs().max().item() / scale
        n += 1
        if bad or err > 2e-2:
            fails += 1
            print("FAIL", dict(M=M, K=K, N=N, epi=epi, packed=packed, differing_repeats=bad, rel_err=err), flush=True)
        if it % 25 == 24:
            print(f"  shape {it}: fails={fails}, {time.time() - t0:.0f}s", flush=True)
    print(f"streamk soak: shapes={n} x {REPEAT} launches, fails={fails}, {time.time() - t0:.0f}s")


if len(sys.argv) > 2 and sys.argv[2] == "streamk":
    streamk_soak(int(sys.argv[1]))


# ---------------------------------------------------------------------------------------------- top-p inside the dense kernel
# `python tests/fuzz_soak.py <seeds> top_p`: the parameter sets of test_static_batches_with_top_p_vs_oracle over more seeds.
def top_p_soak(n_seeds):
    sets = [("llamagen", "naive_extend_57", True, 50, 0.1, 1.0, 0.9, 150), ("llamagen", "mc_sim_7b_63", False, 1, 0.1, 2.0, 0.6, 0),
            ("anole", "naive_extend_57", True, 10, 5.0, 1.0, 0.95, 150), ("anole", "mc_sim_7b_63", True, 20, 0.2, 3.0, 0.3, 40),
            ("llamagen", "mc_sim_7b_63", True, 200, 10.0, 0.5, 0.99, 0)]
    t0 = time.time(); n = fails = 0
    for seed in range(300, 300 + n_seeds):
        for (model, tree, lantern, k, delta, sigma, top_p, top_k) in sets:
            try:
                F._static_batches(model, tree, lantern, k, delta, sigma, seed, top_p, top_k)
            except AssertionError as e:
                fails += 1
                print("FAIL", (model, tree, lantern, k, delta, sigma, top_p, top_k), seed, str(e)[:300], flush=True)
            n += 1
    print(f"top-p soak: cases={n} batches x 32 sequences (dense kernel, TopPLogitsWarper per visited row), fails={fails}, {time.time() - t0:.0f}s")


if len(sys.argv) > 2 and sys.argv[2] == "top_p":
    top_p_soak(int(sys.argv[1]))


# ---------------------------------------------------------------------------------------------- the static drafter's head stage (round 5)
# `python tests/fuzz_soak.py <iters> draws`: lantern_head_sample on random heads / hidden rows -- peaked, flat and nearly empty rows (top-k filter from 3 to
# the whole window), random uniforms with 0 and 1 - 2^-53 sprinkled in -- against the oracle's successive inverse-CDF draws taken on the kernel's own
# distribution (exact tokens, exact conditional probabilities); and lantern_mask_left_padding on random masks against torch's reductions.
def draws_soak(iters):
    import numpy as np
    import torch
    import oracle
    from lantern_amd import ops
    t0 = time.time(); n = fails = rows = 0
    g = torch.Generator(device="cuda").manual_seed(11)
    for it in range(iters):
        model = ("lumina", "anole", "llamagen")[it % 3]
        V, lo, W = (16384, 0, 16384) if model == "llamagen" else (65536, 4, 8192)
        mid = ops.MODEL_PLAIN if model == "llamagen" else ops.MODEL_ANOLE
        nr, K, k = 1 + it % 4, (64, 128, 256)[it % 3], (10, 4, 16)[(it // 3) % 3]
        sharp = (0.05, 0.5, 3.0)[(it // 9) % 3]
        A = (sharp * torch.randn(2 * nr, K, device="cuda", generator=g)).to(torch.bfloat16)
        Wt = (0.3 * torch.randn(V, K, device="cuda", generator=g)).to(torch.bfloat16)
        tk = (3, 12, 300, 2000, W)[it % 5]
        u = torch.rand((nr, k), dtype=torch.float64, device="cuda", generator=g)
        edge = torch.rand((nr, k), device="cuda", generator=g)
        u = torch.where(edge < 0.05, torch.zeros_like(u), torch.where(edge > 0.95, torch.full_like(u, 1.0 - 2.0 ** -53), u))
        pk = ops.pack_linear_weight(Wt[lo:lo + W].contiguous())
        probs, tok, prob = ops.head_sample(A, Wt, lo, W, 3.0, model=mid, top_k_filter=min(tk, V), n_draw=k, draw_u=u, packed=pk)
        got, tok_h, prob_h, u_h = probs.cpu().numpy(), tok.cpu().numpy(), prob.cpu().numpy(), u.cpu().numpy()
        for r in range(nr):
            rows += 1
            live = min(k, int((got[r] > 0).sum()))
            idx, cp = oracle.sample_draws(got[r], u_h[r][:live])
            ok = abs(float(got[r].sum()) - 1) <= 1e-5
            if np.array_equal(tok_h[r, :live], idx):
                ok = ok and np.array_equal(prob_h[r, :live], cp[:live])
            else:
                # a uniform within 2^-53 of 1 puts the crossing where the f64 running sum no longer moves: which of the trailing ~1e-17-mass entries
                # "crosses" depends on the order of the f64 additions (the kernel sums 32-id segments, the oracle runs sequentially).  Accept any draw
                # whose sequential cumulative mass brackets u * total to 1e-12, on the distribution with the kernel's earlier draws removed.
                q = got[r].astype(np.float64).copy()
                for j in range(live):
                    t, tot = int(tok_h[r, j]), q.sum()
                    cs = np.cumsum(q)
                    tgt = u_h[r][j] * tot
                    near_one = u_h[r][j] > 1.0 - 1e-15
                    exact = int(oracle.sample_draws(q.astype(np.float32), u_h[r][j:j + 1])[0][0])
                    if t != exact and not (near_one and q[t] > 0 and cs[t] >= tgt * (1 - 1e-12) and cs[t] - q[t] <= tgt * (1 + 1e-12)):
                        ok = False
                    q[t] = 0.0
            if live < k:
                ok = ok and len(set(tok_h[r].tolist())) == k and bool((prob_h[r, live:] == 0).all()) and bool(((tok_h[r, live:] >= lo) & (tok_h[r, live:] < lo + W)).all())
            if not ok:
                fails += 1
                print("FAIL draws", dict(model=model, it=it, row=r, tk=tk, k=k, got=tok_h[r].tolist(), want=idx.tolist()), flush=True)
        S = int(torch.randint(1, 3000, (1,)).item())
        m = (torch.rand((1 + it % 3, S), device="cuda", generator=g) < (0.02, 0.5, 0.98)[it % 3]).to(torch.int64)
        if it % 2:
            pad = torch.randint(0, S + 1, (m.shape[0], 1), device="cuda")
            m = (torch.arange(S, device="cuda")[None] >= pad).to(torch.int64)
        out = ops.mask_left_padding(m.to((torch.bool, torch.uint8, torch.int64)[it % 3]))
        if not (torch.equal(out[0], m.argmax(1)) and torch.equal(out[1], m.sum(1)) and torch.equal(out[2], (m.cummax(1).values != m).any(1).to(torch.int64))):
            fails += 1
            print("FAIL mask", it, flush=True)
        n += 1
        if it % 100 == 99:
            print(f"  iter {it}: rows={rows} fails={fails}, {time.time() - t0:.0f}s", flush=True)
    print(f"draws soak: launches={n}, rows={rows} (head_sample draws vs the oracle, mask_left_padding vs torch), fails={fails}, {time.time() - t0:.0f}s")


if len(sys.argv) > 2 and sys.argv[2] == "draws":
    draws_soak(int(sys.argv[1]))

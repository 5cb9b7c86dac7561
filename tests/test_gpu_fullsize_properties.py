"""GPU (-m gpu): size-independent properties of the hot path AT THE BENCH'S FULL LAUNCH SIZE (64 sequences of the Lumina
768x768 workload: V = 65536, 8192-wide image window, k = 1000, static tree mc_sim_7b_63, 7B KV geometry), where the
oracle is too slow to replay every sequence.  The oracle pins the same kernels bit for bit at smaller sizes elsewhere
(test_gpu_parity / test_gpu_window / test_gpu_loop); here the domain's own invariants are checked on the full batch:

* every processed row is a probability vector with at least top_k survivors inside the image window;
* the accept walk's counters obey the algorithm's bookkeeping (one uniform per tried candidate, accept length against
  levels visited, path / depth bounds), and the bonus token is a legal id;
* the result of a sequence does not depend on which other sequences share its launch (the sharding property of
  SURVEY 8e: groups of 16 == one launch of 64, step by step);
* the KV gather is the identity on an in-place path, touches nothing outside prev..prev+a, and conserves the moved rows.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

B_FULL, STEPS = 64, 24


def _run(groups, path="window"):
    from lantern_amd import harness as HN
    cfg = HN.WorkloadConfig(n_seq=B_FULL, pool_steps=3, with_kv=False, max_steps=STEPS + 8, n_groups=groups, path=path)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    for _ in range(STEPS):
        wl.step()
    torch.cuda.synchronize()
    wl.check_status(0, STEPS)
    return HN, wl


def test_full_batch_rows_and_counters():
    HN, wl = _run(1)
    c = wl.cfg
    # ---- O7: the rows of the last step (probabilities over the image window)
    proc, hot = wl.proc, wl.row_hot
    assert proc.shape == (B_FULL, wl.N, wl.W) and wl.W == HN.IMG_HI - HN.IMG_LO
    live = hot < 0
    assert int(live.sum()) > 0
    rows = proc[live].double()
    assert float(rows.min()) >= 0.0
    assert float((rows.sum(-1) - 1.0).abs().max()) <= 1e-5           # north_star: probabilities within 1e-5
    nz = (rows > 0).sum(-1)
    assert int(nz.min()) >= min(c.top_k, wl.W) and int(nz.max()) <= wl.W          # ties at the k-th value are kept, like torch.topk's threshold
    hv = hot[~live]
    assert bool(((hv == HN.NEWLINE) | (hv == HN.EOS)).all())          # one-hot rows only where the grammar forces a token
    # ---- O8: bookkeeping of every logged step
    cnt = wl.log_cnt[:STEPS].long()
    alen, best, tok = wl.log_alen[:STEPS].long(), wl.log_best[:STEPS].long(), wl.log_token[:STEPS].long()
    levels, tried, rej, used, from_res, status = (cnt[..., i] for i in range(6))
    assert int(status.abs().sum()) == 0
    assert bool((used == tried).all())                                # one uniform per tried candidate
    assert bool((rej <= tried).all()) and bool((tried >= rej + alen).all())
    assert bool(((levels == alen) | (levels == alen + 1)).all())      # the last visited level either accepted or rejected everything
    assert bool((alen >= 0).all()) and bool((alen <= wl.D - 1).all()) and bool((levels <= wl.D - 1).all())
    assert bool((best >= 0).all()) and bool((best < wl.P).all())
    # the bonus token comes from the residual only when the last visited level rejected something and accepted nothing
    # (a level without candidates -- the accepted node is a leaf -- also ends the walk, with a fresh row)
    assert bool(((from_res == 0) | ((levels == alen + 1) & (rej > 0))).all())
    assert bool((from_res[(levels == alen) & (alen == wl.D - 1)] == 0).all())
    assert int(from_res.sum()) > 0 and int((from_res == 0).sum()) > 0
    legal = ((tok >= HN.IMG_LO) & (tok < HN.IMG_HI)) | (tok == HN.NEWLINE) | (tok == HN.EOS)
    assert bool(legal.all())
    # the accepted prefix is a real path of the tree: no padding (-1) token inside it
    cand = wl.cand                                                     # [B, P, D] of the last step
    lb, la = wl.log_best[STEPS - 1].long(), wl.log_alen[STEPS - 1].long()
    path = cand[torch.arange(B_FULL, device=cand.device), lb]          # [B, D]
    depth = torch.arange(wl.D, device=cand.device)[None, :]
    assert bool((path[depth <= la[:, None]] >= 0).all())
    # accept lengths of the synthetic recipe (sigma frozen): the batch is neither all-reject nor all-accept
    mean_alen = float((alen.float() + 1).mean())
    assert 1.5 < mean_alen < 4.5, mean_alen


def test_result_does_not_depend_on_the_launch_grouping():
    """64 sequences in one launch per kernel == the same sequences in 4 stream groups of 16 (a rank's shard is just
    another grouping): identical (best, accept length, bonus token, counters) for every step of every sequence."""
    _, a = _run(1)
    la = [t[:STEPS].clone() for t in (a.log_best, a.log_alen, a.log_token, a.log_cnt)]
    del a
    torch.cuda.empty_cache()
    _, b = _run(4)
    assert b.Bg == B_FULL // 4
    for x, y in zip(la, (b.log_best, b.log_alen, b.log_token, b.log_cnt)):
        assert torch.equal(x, y[:STEPS])


def test_dense_and_windowed_kernel_sets_agree_on_the_full_batch():
    """Two independent implementations of O7 + O8 (full-vocabulary rows in registers / HBM vs 8192-wide window rows with the
    residual in LDS) walk the same 64 sequences: identical decisions, bonus tokens and counters in every step."""
    _, a = _run(1, "window")
    la = [t[:STEPS].clone() for t in (a.log_best, a.log_alen, a.log_token)]
    ca = a.log_cnt[:STEPS, :, :5].clone()
    del a
    torch.cuda.empty_cache()
    _, b = _run(1, "dense")
    for x, y in zip(la, (b.log_best, b.log_alen, b.log_token)):
        assert torch.equal(x, y[:STEPS])
    assert torch.equal(ca, b.log_cnt[:STEPS, :, :5])


def test_kv_gather_identity_conservation_and_bounds_full_geometry():
    """7B slab geometry [64 (layer x K/V), 1, 32 heads, S, 128] bf16, 8 slabs (4 sequences x cond/uncond)."""
    from lantern_amd import ops
    from lantern_amd.drafters import choices
    tb = ops.tree_static_build(choices.mc_sim_7b_63)
    ret = torch.from_numpy(np.ascontiguousarray(tb["retrieve_indices"])).cuda()
    P, D = ret.shape
    S, n_seq = 160, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    slabs = [torch.randint(-32768, 32767, (64, 1, 32, S, 128), dtype=torch.int16, device="cuda", generator=g) for _ in range(2 * n_seq)]
    seq = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], dtype=torch.int32, device="cuda")
    prev = torch.tensor([70, 3, 100, 33, 64, 5, 90, 21], dtype=torch.int64, device="cuda")
    # ---- identity: the first-child chain 0,1,2,... of path 0 is already in place -> nothing may change
    first_chain = 0
    while first_chain + 1 < D and int(ret[0, first_chain + 1]) == first_chain + 1:
        first_chain += 1
    before = [s.clone() for s in slabs]
    best0 = torch.zeros(n_seq, dtype=torch.int32, device="cuda")
    alen0 = torch.full((n_seq,), first_chain, dtype=torch.int32, device="cuda")
    nl = ops.kv_gather(slabs, seq, prev, ret, best0, alen0)
    assert all(torch.equal(x, y) for x, y in zip(slabs, before))
    assert torch.equal(nl, prev + first_chain + 1)
    # ---- a real move: the deepest path whose nodes are NOT in place
    lens = (ret >= 0).sum(-1)
    cands = [p for p in range(P) if int(lens[p]) >= 3 and int(ret[p, 1]) != 1]
    assert cands
    pth = max(cands, key=lambda p: int(lens[p]))
    n = int(lens[pth])
    best = torch.tensor([pth, 0, pth, 1], dtype=torch.int32, device="cuda")
    alen = torch.tensor([n - 1, 0, 1, min(2, int(lens[1]) - 1)], dtype=torch.int32, device="cuda")
    nl = ops.kv_gather(slabs, seq, prev, ret, best, alen)
    for si, (s, b0) in enumerate(zip(slabs, before)):
        q = int(seq[si])
        p0, m = int(prev[si]), int(alen[q]) + 1
        assert int(nl[si]) == p0 + m
        src = ret[int(best[q]), :m] + p0
        assert torch.equal(s[..., p0:p0 + m, :], b0[..., src, :])                      # the moved rows are conserved bit for bit
        keep = torch.ones(S, dtype=torch.bool, device="cuda")
        keep[p0:p0 + m] = False
        assert torch.equal(s[..., keep, :], b0[..., keep, :])                          # nothing outside prev..prev+a is touched
    # ---- idempotence: the accepted rows now sit at prev..prev+a; gathering the in-place chain from there changes nothing
    after = [s.clone() for s in slabs]
    ops.kv_gather(slabs, seq, prev, ret, best0, alen0)
    assert all(torch.equal(x, y) for x, y in zip(slabs, after))


def test_dynamic_tree_batch_invariants():
    """O4 on a batch of 32 different EAGLE-2 trees (N = 59): the outputs must describe a rooted tree -- ancestor sets closed
    under ancestry, one parent per node, positions = depths, retrieve rows = the root-to-leaf chains, one row per leaf -- and
    the batched launch must equal 32 single-sequence launches (the oracle pins single trees bit for bit in test_gpu_parity
    and the soak tool; this is the batch-independence and shape bookkeeping at full N)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases as CS
    from lantern_amd import ops
    B = 32
    gens = [CS.gen_dynamic(7000 + b, "llamagen", sigma=float(1 + b % 3)) for b in range(B)]
    tt = gens[0]["total_tokens"]
    assert all(g["total_tokens"] == tt for g in gens)
    dev = lambda key, dt: torch.from_numpy(np.stack([np.asarray(g[key]) for g in gens]).astype(dt)).cuda()      # noqa: E731
    scores, tokens, parents = dev("scores", np.float32), dev("tokens", np.int64), dev("parents", np.int64)
    st = torch.tensor([int(g["sample_token"]) for g in gens], dtype=torch.int64, device="cuda")
    draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(scores, tokens, parents, st, CS.TOPK, tt)
    N = tt + 1
    assert draft.shape == (B, N) and mask.shape == (B, N, N)
    m = mask.cpu().numpy()
    assert set(np.unique(m)) <= {0.0, 1.0}
    mi = m.astype(np.int64)
    eye = np.eye(N, dtype=np.int64)
    assert (mi * eye[None] == eye[None]).all()                         # every node attends to itself
    assert (mi[:, :, 0] == 1).all()                                    # ... and to the root
    assert (np.triu(mi, 1) == 0).all()                                 # ancestors come first in the node order
    assert (((mi @ mi) > 0).astype(np.int64) == mi).all()              # an ancestor's ancestors are ancestors
    p = pos.cpu().numpy()
    assert (p == mi.sum(-1) - 1).all()                                 # position id = depth = number of proper ancestors
    assert (draft[:, 0].cpu() == st.cpu()).all()
    r, nl_, md_ = ret.cpu().numpy(), nl.cpu().numpy(), md.cpu().numpy()
    for b in range(B):
        # one parent per node: the ancestor one level up carries exactly the node's ancestor set minus the node
        for i in range(1, N):
            anc = np.nonzero(mi[b, i])[0]
            par = [a for a in anc if p[b, a] == p[b, i] - 1]
            assert len(par) == 1 and (mi[b, i] == mi[b, par[0]] + eye[i]).all()
        leaves = [i for i in range(N) if mi[b, :, i].sum() == 1]
        assert nl_[b] == len(leaves) and md_[b] == p[b].max() + 1
        rows = r[b, :nl_[b], :md_[b]]
        ends = []
        for row in rows:
            k = int((row >= 0).sum())
            assert (row[:k] >= 0).all() and (row[k:] == -1).all() and row[0] == 0
            for d in range(1, k):
                assert (mi[b, row[d]] == mi[b, row[d - 1]] + eye[row[d]]).all()     # each step goes from a node to one of its children
            ends.append(int(row[k - 1]))
        assert sorted(ends) == leaves                                   # one retrieve row per leaf, every leaf once
    # batch independence: sequence b alone gives the same outputs
    for b in (0, 13, B - 1):
        d1, m1, p1, r1, n1, x1 = ops.tree_dynamic_finalize(scores[b:b + 1], tokens[b:b + 1], parents[b:b + 1], st[b:b + 1], CS.TOPK, tt)
        assert torch.equal(d1[0], draft[b]) and torch.equal(m1[0], mask[b]) and torch.equal(p1[0], pos[b])
        assert int(n1[0]) == int(nl[b]) and int(x1[0]) == int(md[b])
        assert torch.equal(r1[0, :int(n1[0]), :int(x1[0])], ret[b, :int(nl[b]), :int(md[b])])

"""GPU (-m gpu): loop-level parity (level 4).  The synthetic Lumina verify loop (O6 -> O7 -> O8 -> O9 -> O10,
device-resident state, no host round trip) against the oracle loop on the same pools and the same MT19937 uniform
streams: identical (best path, accept length, bonus token) for every step of every sequence, windowed and dense kernel
sets, eager launches and hipGraph replay."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path,graph,groups", [("window", False, 1), ("window", True, 1), ("dense", False, 1), ("window", False, 3),
                                               ("window", True, 2)])
def test_harness_loop_matches_oracle_loop(path, graph, groups):
    import bench
    from lantern_amd import harness as HN
    steps = 36
    cfg = HN.WorkloadConfig(n_seq=6, pool_steps=4, path=path, use_graph=graph, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 4,
                            sigma=5.0, n_groups=groups)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    wl.check_status(0, steps)
    gb, ga, gt = wl.log_best[:steps].cpu().numpy(), wl.log_alen[:steps].cpu().numpy(), wl.log_token[:steps].cpu().numpy()
    stream = [[(int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) for b in range(cfg.n_seq)] for i in range(steps)]
    res = bench.cpu_baseline(wl, steps_budget_s=1e9, n_seq=cfg.n_seq, gpu_tokens_by_seq=stream)
    assert res["matches_gpu_token_stream"], res
    assert f"x {steps} verify steps" in res["sample"]
    # lengths advanced exactly by the accepted tokens (cond slabs: prompt + 3 header tokens + generated)
    gen = (ga.astype("int64") + 1).sum(0)
    assert (wl.cond_lens(steps & 1).cpu().numpy() == cfg.prompt_len + 3 + gen).all()
    assert (wl.uncond_lens(steps & 1).cpu().numpy() == 3 + gen).all()
    # a newline row was crossed by at least one sequence (position-dependent one-hot rows are in play)
    assert int((torch.as_tensor(gt) == HN.NEWLINE).sum()) > 0


def test_kv_rows_follow_the_accepted_path():
    """After a step the slab rows prev..prev+a hold what the tree rows retrieve[best,:a+1]+prev held before it."""
    from lantern_amd import harness as HN
    cfg = HN.WorkloadConfig(n_seq=3, pool_steps=2, kv_layers=2, kv_heads=4, kv_smax=256, max_steps=8, sigma=2.0)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    for s in wl.slabs:
        s.copy_(torch.randn(s.shape, device="cuda").to(torch.bfloat16))
    before = [s.clone() for s in wl.slabs]
    prev = wl.lens[0].clone()
    wl.step()
    torch.cuda.synchronize()
    best, alen = wl.log_best[0].cpu(), wl.log_alen[0].cpu()
    ret = wl.d_retrieve.cpu()
    for si, (s, b0) in enumerate(zip(wl.slabs, before)):
        seq = si % cfg.n_seq
        p, n = int(prev[si]), int(alen[seq]) + 1
        sel = ret[int(best[seq]), :n] + p
        assert torch.equal(s[..., p:p + n, :], b0[..., sel.cuda(), :])
        keep = torch.ones(s.shape[-2], dtype=torch.bool)
        keep[p:p + n] = False
        assert torch.equal(s[..., keep.cuda(), :], b0[..., keep.cuda(), :])


@pytest.mark.parametrize("fuse", [False, True], ids=["o7_launch", "raw_rows"])
def test_dynamic_tree_loop_matches_oracle_loop(fuse):
    """The device-resident EAGLE-2 loop (O4 -> O6 dynamic -> O7 -> O8 dynamic -> O9 + O10, a different tree per sequence and
    step) against the oracle's loop over the same pools / uniforms: identical (best path, accept length, bonus token), every
    step, every sequence; KV lengths advance by exactly the accepted tokens."""
    import numpy as np
    import oracle
    from lantern_amd import harness as HN
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    steps = 8
    cfg = HN.DynamicConfig(n_seq=3, pool_steps=2, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 2, lantern_k=300, fuse_o7=fuse)
    wl = HN.DynamicVerifyWorkload(cfg, torch.device("cuda"))
    assert wl.fused_o7 == fuse
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    wl.check_status(0, steps)
    gb, ga, gt = wl.log_best[:steps].cpu().numpy(), wl.log_alen[:steps].cpu().numpy(), wl.log_token[:steps].cpu().numpy()
    table = wl.table_full.cpu().numpy().view(np.uint16)
    uni, ub = wl.uniforms.cpu().numpy(), wl.u_bonus.cpu().numpy()
    ocfg = oracle.EpConfig.lumina(False, lantern=True, k=cfg.lantern_k, delta=cfg.lantern_delta)
    N = wl.N
    n_acc = 0
    for b in range(cfg.n_seq):
        tok, cursor, lens = int(wl.first_token[b]), 0, [cfg.prompt_len + 3, 3]
        for i in range(steps):
            p = wl.pools[i % cfg.pool_steps]
            draft, ret, mask, pos = oracle.tree_dynamic_finalize(p["scores"][b].cpu().numpy(), p["tokens"][b].cpu().numpy(), p["parents"][b].cpu().numpy(),
                                                                 cfg.top_k, cfg.total_tokens, tok)
            cand = np.where(ret >= 0, draft[np.clip(ret, 0, None)], -1)
            proc = oracle.cfg_mask_topk(p["cond"][b].cpu().view(torch.int16).numpy().view(np.uint16), p["unc"][b].cpu().view(torch.int16).numpy().view(np.uint16),
                                        cfg.cfg_scale, model=oracle.MODEL_LUMINA, pos_ids=pos + 1 + lens[0], pos_base=cfg.prompt_len + 3, w=HN.W_LATENT,
                                        h=HN.H_LATENT, img_lo=HN.IMG_LO, img_hi=HN.IMG_HI, newline_id=HN.NEWLINE, eos_id=HN.EOS, top_k=cfg.logit_top_k, bf16=True)
            best, alen, sp, cnt = oracle.evaluate_posterior(ocfg, proc, H.row_index_from_retrieve(ret, N), cand, uni[b, cursor:cursor + 64], table=table)
            cursor += int(cnt[3])
            tok = oracle.sample_inverse_cdf(sp, float(ub[i, b]))
            assert (int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) == (best, alen, tok), (b, i)
            lens = [l + alen + 1 for l in lens]
            n_acc += alen
    assert n_acc > 0                                   # the walk really accepts drafted tokens
    gen = (ga.astype("int64") + 1).sum(0)
    assert (wl.lens[steps & 1][:cfg.n_seq].cpu().numpy() == cfg.prompt_len + 3 + gen).all()
    assert (wl.lens[steps & 1][cfg.n_seq:].cpu().numpy() == 3 + gen).all()


@pytest.mark.parametrize("groups,spec", [(1, 0), (2, 0), (1, 5), (2, 1), (1, 26)])
def test_fused_o7_loop_matches_oracle_loop(groups, spec):
    """LANTERN_ROWS_RAW_BF16: no cfg_mask_topk launch -- the chain kernel post-processes (CFG, top-k, softmax) the rows its walk
    visits from the raw cond / uncond logits.  Same oracle loop, same token stream, and every step identical to the unfused run."""
    import bench
    from lantern_amd import harness as HN
    steps = 60
    outs = []
    for fuse in (True, False):
        cfg = HN.WorkloadConfig(n_seq=6, pool_steps=4, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 4, sigma=5.0, n_groups=groups,
                                ep_kernel="chain", fuse_o7=fuse, spec_rows=spec)
        wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
        assert wl.fused_o7 == fuse and wl.n_spec == (spec if fuse else 0)
        for _ in range(steps):
            wl.step()
        wl.join()
        torch.cuda.synchronize()
        wl.check_status(0, steps)
        outs.append((wl.log_best[:steps].clone(), wl.log_alen[:steps].clone(), wl.log_token[:steps].clone(), wl.log_cnt[:steps].clone()))
        if fuse:
            gb, ga, gt = [x.cpu().numpy() for x in outs[0][:3]]
            stream = [[(int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) for b in range(cfg.n_seq)] for i in range(steps)]
            res = bench.cpu_baseline(wl, steps_budget_s=1e9, n_seq=cfg.n_seq, gpu_tokens_by_seq=stream)
            assert res["matches_gpu_token_stream"], res
            assert int((torch.as_tensor(gt) == HN.NEWLINE).sum()) > 0          # forced newline rows were crossed
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("groups,threads,ep,fuse", [(6, 3, "chain", True), (4, 4, "nodes", False), (1, 1, "chain", True)])
def test_worker_thread_launches_give_the_same_stream(groups, threads, ep, fuse):
    """lantern_step_launcher: the step's launches enqueued by worker threads (one stream always fed by the same worker, argument
    blocks copied at submit) -- every step's verdicts, tokens and counters equal the calling-thread run, KV rows included."""
    from lantern_amd import harness as HN
    steps = 80
    outs = []
    for thr in (threads, 0):
        cfg = HN.WorkloadConfig(n_seq=12, pool_steps=4, kv_layers=2, kv_heads=4, kv_smax=1024, max_steps=steps + 4, sigma=5.0, n_groups=groups,
                                ep_kernel=ep, fuse_o7=fuse, spec_rows=3 if fuse else 0, launch_threads=thr)
        wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
        assert (wl._launcher is not None) == (thr > 0)
        for _ in range(steps):
            wl.step()
        wl.sync()
        wl.check_status(0, steps)
        outs.append((wl.log_best[:steps].clone(), wl.log_alen[:steps].clone(), wl.log_token[:steps].clone(), wl.log_cnt[:steps].clone(),
                     wl.lens[steps & 1].clone(), torch.stack([s.clone() for s in wl.slabs[:4]])))
        wl.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_worker_thread_launch_error_reaches_the_caller():
    """An enqueue that fails on a worker thread comes back from wait() with the kernel's own message."""
    import ctypes as C
    from lantern_amd import _lib
    from lantern_amd import harness as HN
    cfg = HN.WorkloadConfig(n_seq=4, pool_steps=2, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=16, n_groups=2, ep_kernel="chain",
                            fuse_o7=True, spec_rows=3, launch_threads=2)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    wl.step()
    wl.sync()
    keep = wl._steps[(0, 0)][1].ep.k
    for arr in wl._steps.values():
        arr[1].ep.k = -5                 # an argument the kernel's host side refuses
    try:
        with pytest.raises(_lib.LanternError, match="worker thread"):
            wl.step()
            wl.sync()
    finally:
        for arr in wl._steps.values():
            arr[1].ep.k = keep
    wl.step()                            # the launcher keeps working after the error was collected
    wl.sync()
    wl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("groups,fuse,workers", [(1, True, 0), (2, True, 3), (1, False, 1), (3, True, 40)])
def test_fused_accept_launch_gives_the_same_stream_and_kv_rows(groups, fuse, workers):
    """lantern_verify_accept: evaluate_posterior + update_inference_inputs in one launch, pipelined per sequence through a work queue
    (chains on the first B workgroups, copy workers behind them).  Verdicts, tokens, counters, lengths, every KV slab, the accepted
    hidden rows and tokens equal the two-launch run; the queue is left empty."""
    from lantern_amd import harness as HN
    steps = 80
    outs = []
    for fa in (True, False):
        cfg = HN.WorkloadConfig(n_seq=12, pool_steps=4, kv_layers=2, kv_heads=4, kv_smax=1024, max_steps=steps + 4, sigma=5.0, n_groups=groups,
                                ep_kernel="chain", fuse_o7=fuse, spec_rows=3 if fuse else 0, fused_accept=fa, fused_workers=workers)
        wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
        assert (wl.fused_ws is not None) == fa
        for _ in range(steps):
            wl.step()
        wl.sync()
        wl.check_status(0, steps)
        if fa:
            for w in wl.fused_ws:
                assert int(w.abs().sum()) == 0
        outs.append((wl.log_best[:steps].clone(), wl.log_alen[:steps].clone(), wl.log_token[:steps].clone(), wl.log_cnt[:steps].clone(),
                     wl.lens[steps & 1].clone(), torch.stack([s.clone() for s in wl.slabs]), wl.out_hidden.clone(), wl.acc_tokens.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 2])
def test_serial_node_kernel_loop_matches_chain_loop(groups):
    """ep_kernel = "walk" (one workgroup per sequence running the node routine at every stop of the walk, lantern_ep_nodes.serial): the
    whole step loop gives the chain kernel's verdicts, tokens, counters and KV rows, and the oracle's token stream."""
    import bench
    from lantern_amd import harness as HN
    steps = 60
    outs = []
    for ep in ("walk", "chain"):
        cfg = HN.WorkloadConfig(n_seq=8, pool_steps=4, kv_layers=2, kv_heads=4, kv_smax=1024, max_steps=steps + 4, sigma=5.0, n_groups=groups,
                                ep_kernel=ep)
        wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
        for _ in range(steps):
            wl.step()
        wl.sync()
        wl.check_status(0, steps)
        outs.append((wl.log_best[:steps].clone(), wl.log_alen[:steps].clone(), wl.log_token[:steps].clone(), wl.log_cnt[:steps].clone(),
                     wl.lens[steps & 1].clone(), torch.stack([s.clone() for s in wl.slabs])))
        if ep == "walk":
            gb, ga, gt = [x.cpu().numpy() for x in outs[0][:3]]
            stream = [[(int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) for b in range(cfg.n_seq)] for i in range(steps)]
            res = bench.cpu_baseline(wl, steps_budget_s=1e9, n_seq=cfg.n_seq, gpu_tokens_by_seq=stream)
            assert res["matches_gpu_token_stream"], res
    for a, b in zip(*outs):
        assert torch.equal(a, b)

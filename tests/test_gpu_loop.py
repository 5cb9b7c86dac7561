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


@pytest.mark.parametrize("path,graph,groups,n_seq,steps", [("window", False, 1, 6, 36), ("window", True, 1, 6, 36), ("dense", False, 1, 6, 36), ("window", False, 3, 6, 36),
                                                           ("window", True, 2, 6, 36), ("window", False, 1, 264, 4)],
                         ids=["window", "window_graph", "dense", "window_3_groups", "window_graph_2_groups", "throughput"])
def test_harness_loop_matches_oracle_loop(path, graph, groups, n_seq, steps):
    """(`throughput`: 264 sequences in one launch -- more than CUs -- run the chain kernel's throughput instance, on probability rows.)"""
    import bench
    from lantern_amd import harness as HN
    cfg = HN.WorkloadConfig(n_seq=n_seq, pool_steps=4 if n_seq < 64 else 2, path=path, use_graph=graph, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 4,
                            sigma=5.0, n_groups=groups, **({} if n_seq < 64 else {"ep_kernel": "chain"}))
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
    assert steps < 20 or int((torch.as_tensor(gt) == HN.NEWLINE).sum()) > 0


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


@pytest.mark.parametrize("fuse,groups,native,spec", [(False, 1, True, 0), (True, 1, True, 0), (True, 1, True, 2), (False, 3, True, 0), (True, 2, True, 1),
                                                     (True, 3, True, 2), (False, 2, False, 0), (True, 2, False, 2)],
                         ids=["o7_launch", "raw_rows", "raw_rows_2_prepared", "o7_launch_3_groups", "raw_rows_2_groups_root_prepared",
                              "raw_rows_3_groups_2_prepared", "per_kernel_calls_2_groups", "per_kernel_calls_raw_rows_2_prepared"])
def test_dynamic_tree_loop_matches_oracle_loop(fuse, groups, native, spec):
    _dynamic_tree_loop(fuse, groups, native, spec, 3 * groups, 8, 1)


def test_dynamic_tree_loop_throughput_instance():
    """More sequences per launch than CUs: the throughput form of the chain kernel's Lumina dynamic-tree instance (256 threads, three workgroups per
    CU) on probability rows, every 13th sequence held to the oracle's loop."""
    _dynamic_tree_loop(False, 1, True, 0, 264, 2, 13)


def _dynamic_tree_loop(fuse, groups, native, spec, n_seq, steps, every):
    """The device-resident EAGLE-2 loop (O4 -> O6 dynamic -> O7 -> O8 dynamic -> O9 + O10, a different tree per sequence and
    step) against the oracle's loop over the same pools / uniforms: identical (best path, accept length, bonus token), every
    step, every sequence; KV lengths advance by exactly the accepted tokens."""
    import numpy as np
    import oracle
    from lantern_amd import harness as HN
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    cfg = HN.DynamicConfig(n_seq=n_seq, pool_steps=2, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 2, lantern_k=300, fuse_o7=fuse,
                           n_groups=groups, native_step=native, spec_rows=spec)
    wl = HN.DynamicVerifyWorkload(cfg, torch.device("cuda"))
    assert wl.fused_o7 == fuse and wl.G == groups and wl.n_spec == (spec if fuse else 0)
    for _ in range(steps):
        wl.step()
    wl.sync()
    wl.check_status(0, steps)
    gb, ga, gt = wl.log_best[:steps].cpu().numpy(), wl.log_alen[:steps].cpu().numpy(), wl.log_token[:steps].cpu().numpy()
    table = wl.table_full.cpu().numpy().view(np.uint16)
    uni, ub = wl.uniforms.cpu().numpy(), wl.u_bonus.cpu().numpy()
    ocfg = oracle.EpConfig.lumina(False, lantern=True, k=cfg.lantern_k, delta=cfg.lantern_delta)
    N = wl.N
    n_acc = 0
    for b in range(0, cfg.n_seq, every):
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
    len_c, len_u = wl.lengths(steps & 1)
    assert (len_c.cpu().numpy() == cfg.prompt_len + 3 + gen).all() and (len_u.cpu().numpy() == 3 + gen).all()
    # the accepted tokens and the KV rows of every slab: the tree slot of accepted token t moved to row prev + t (checked through the
    # lengths above and, for the tokens, against the verdicts)
    acc = wl.acc_tokens.cpu().numpy()
    last = steps - 1
    for b in range(cfg.n_seq):
        assert int(acc[b, int(ga[last, b])]) >= 0 and (acc[b, int(ga[last, b]) + 1:] == -1).all()


@pytest.mark.parametrize("fuse,groups,spec,lam,k", [(False, 1, 0, 5.0, 10), (True, 1, 0, 5.0, 10), (True, 2, 3, 10.0, 5), (True, 1, 5, 0.3, 40)],
                         ids=["o7_launch", "raw_rows", "raw_rows_2_groups_3_prepared", "raw_rows_delta_mode_5_prepared"])
def test_anole_static_loop_matches_oracle_loop(fuse, groups, spec, lam, k):
    _anole_static_loop(fuse, groups, spec, lam, k, 3 * groups, 24, 1)


@pytest.mark.parametrize("groups,spec,lam,k", [(1, 1, 5.0, 10), (2, 3, 10.0, 5), (1, 4, 0.3, 40)], ids=["no_helper", "2_groups_2_helper_rows", "delta_mode_3_helper_rows"])
def test_anole_static_loop_with_the_prepare_stage_inside_the_chain_launch(groups, spec, lam, k):
    """BASELINE config 4 on the two-launch step (LANTERN_STEP_FUSED_PREPARE, the Anole instance epw_kernel_fused<512, 4, 2, 1, true, true, 4, 0>): the oracle's loop."""
    _anole_static_loop(True, groups, spec, lam, k, 3 * groups, 24, 1, fused_prepare=True)


@pytest.mark.parametrize("fuse,spec", [(False, 0), (True, 0), (True, 3)], ids=["o7_launch", "raw_rows", "raw_rows_3_prepared"])
def test_anole_static_loop_with_top_p(fuse, spec):
    """generate(top_p = 0.9): TopPLogitsWarper in front of the top-k (drafters/utils.py:36-52; the reference applies the list per visited row inside
    evaluate_posterior) -- in O7 over all rows, in the rows lantern_prepare_step produces and in the rows the chain kernel post-processes on demand
    (LANTERN_ROWS_RAW_BF16) -- against the oracle's loop with the same processors."""
    _anole_static_loop(fuse, 1, spec, 5.0, 10, 3, 16, 1, top_p=0.9)


@pytest.mark.parametrize("fuse,spec", [(False, 0), (True, 0), (True, 3)], ids=["o7_launch", "raw_rows", "raw_rows_3_prepared"])
def test_anole_static_loop_with_top_p_and_tied_logits(fuse, spec):
    """the tie order at the nucleus boundary.  The bf16 row kernels hold a row as 8-id chunks (two float4 per chunk), the f32 ones as
    4-id chunks; TopPLogitsWarper removes a PREFIX of the stable ascending sort, so among equal logits that straddle 1 - top_p the lower ids go --
    top_p_tile ranks them by the tile's true index order (CHUNK8).  Heavy ties forced into every row; O7 over all rows, the rows prepared beside the
    candidate assembly and the rows the chain kernel post-processes on demand against the oracle's sort-based loop."""
    _anole_static_loop(fuse, 1, spec, 5.0, 10, 3, 16, 1, top_p=0.9, ties=True)


@pytest.mark.parametrize("model,tree", [("anole", "naive_extend_57"), ("lumina", "naive_extend_57")])
def test_static_loop_throughput_instances(model, tree):
    """More sequences per launch than CUs: the throughput forms (256 threads, three workgroups per CU) of the chain kernel's Anole static-tree
    instance and of the Lumina static-tree instance without the fixed default tree, on probability rows; every 13th sequence held to the oracle's loop
    (the Lumina default tree's form: test_harness_loop_matches_oracle_loop[throughput], tests/test_gpu_configs.py)."""
    if model == "anole":
        _anole_static_loop(False, 1, 0, 5.0, 10, 264, 2, 13)
    else:
        _lumina_static_loop_big(tree, 264, 2, 13)


@pytest.mark.parametrize("tp_raw", ["256", "512"])
@pytest.mark.parametrize("form", ["lumina_default_tree", "lumina_static", "anole_static", "lumina_dynamic"])
def test_raw_row_throughput_instances(form, tp_raw, monkeypatch):
    """Round 5: the throughput forms of the chain kernel on RAW cond / uncond bf16 rows (more sequences per launch than CUs; the row post-process of the
    visited rows inside the kernel): 256 threads x 8 float4 with four 16-byte chunks per operand and thread (the default) and 512 threads at 128 VGPRs
    (lantern_tuning_set("epw_tp_raw", 512): an explicit call, the library reads no environment variable), for the four fixed configurations; every 13th
    sequence against the oracle's loop."""
    from lantern_amd import _lib
    _lib.set_tuning("epw_tp_raw", int(tp_raw))
    try:
        _raw_throughput(form)
    finally:
        _lib.set_tuning("epw_tp_raw", 256)


def _raw_throughput(form):
    if form == "lumina_default_tree":
        _lumina_static_loop_big("mc_sim_7b_63", 264, 2, 13, fuse_o7=True, spec_rows=2)
    elif form == "lumina_static":
        _lumina_static_loop_big("naive_extend_57", 264, 2, 13, fuse_o7=True, spec_rows=0)
    elif form == "anole_static":
        _anole_static_loop(True, 1, 3, 5.0, 10, 264, 2, 13)
    else:
        _dynamic_tree_loop(True, 1, True, 2, 264, 2, 13)


def _lumina_static_loop_big(tree, n_seq, steps, every, fuse_o7=False, spec_rows=0, lantern_delta=0.1):
    import numpy as np
    import oracle
    from lantern_amd import harness as HN
    cfg = HN.WorkloadConfig(tree=tree, n_seq=n_seq, pool_steps=2, with_kv=False, max_steps=steps + 4, sigma=5.0, n_groups=1, ep_kernel="chain", fuse_o7=fuse_o7,
                            spec_rows=spec_rows, lantern_delta=lantern_delta)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    wl.check_status(0, steps)
    gb, ga, gt = wl.log_best[:steps].cpu().numpy(), wl.log_alen[:steps].cpu().numpy(), wl.log_token[:steps].cpu().numpy()
    tb, N = wl.tb, wl.N
    ri = tb["retrieve_indices"].copy()
    ri[ri < 0] += N
    ri = ri.astype(np.int32)
    table = wl.table_full.cpu().numpy().view(np.uint16)
    u16 = lambda t: t.cpu().view(torch.int16).numpy().view(np.uint16)
    ub, first, op_off = wl.u_bonus.cpu().numpy(), wl.first_token.cpu().numpy(), wl.d_op_off.cpu().numpy()
    ocfg = oracle.EpConfig.lumina(True, lantern=True, k=cfg.lantern_k, delta=cfg.lantern_delta)
    pos1 = tb["tree_position_ids"] + 1
    n_acc = n_rej = 0
    for b in range(0, n_seq, every):
        tok, cursor, ln = int(first[b]), 0, cfg.prompt_len + 3
        for i in range(steps):
            s_ = i % cfg.pool_steps
            orig = wl.orig_prob[s_, b].cpu().numpy()
            dense = np.zeros(orig.shape[:-1] + (HN.V,), np.float32)
            dense[..., HN.IMG_LO:HN.IMG_HI] = orig
            cand, cp, tc = oracle.gather_candidates(wl.ss_token[s_, b].cpu().numpy(), wl.ss_prob[s_, b].cpu().numpy(), tok, tb["tree_indices"], tb["retrieve_indices"])
            proc = oracle.cfg_mask_topk(u16(wl.cond[s_, b]), u16(wl.uncond[s_, b]), cfg.cfg_scale, model=oracle.MODEL_LUMINA, pos_ids=pos1 + ln,
                                        pos_base=cfg.prompt_len + 3, w=HN.W_LATENT, h=HN.H_LATENT, img_lo=HN.IMG_LO, img_hi=HN.IMG_HI, newline_id=HN.NEWLINE,
                                        eos_id=HN.EOS, top_k=cfg.top_k, bf16=True)
            aux = oracle.StaticAux(cart_prob=cp, orig_prob=dense, op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"], tree_cand=tc)
            best, alen, sp, cnt = oracle.evaluate_posterior(ocfg, proc, ri, cand, wl.uniforms_host[b, cursor:cursor + 64], table=table, aux=aux)
            cursor += int(cnt[3])
            tok = oracle.sample_inverse_cdf(sp, float(ub[i, b]))
            assert (int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) == (best, alen, tok), (b, i)
            ln += alen + 1
            n_acc += alen
            n_rej += int(cnt[2])
    assert n_acc > 0 and n_rej > 0


def _anole_static_loop(fuse, groups, spec, lam, k, n_seq, steps, every, top_p=1.0, ties=False, fused_prepare=False):
    """BASELINE config 4 (Anole, LANTERN++ static tree naive_extend_57: neighbours zeroed in the drafter's row, no syntax shortcut, no grammar rows)
    through the device-resident step loop -- O7 over all rows, and the raw rows post-processed inside evaluate_posterior with the likeliest rows
    prepared beside O6 -- against the oracle's loop over the same pools / uniforms: identical (best path, accept length, bonus token) at every step."""
    import numpy as np
    import oracle
    from lantern_amd import harness as HN
    cfg = HN.WorkloadConfig(model="anole", tree="naive_extend_57", n_seq=n_seq, pool_steps=4 if n_seq < 64 else 2, kv_layers=2, kv_heads=4, kv_smax=512,
                            max_steps=steps + 4, sigma=5.0, n_groups=groups, ep_kernel="chain", fuse_o7=fuse, spec_rows=spec, lantern_k=k, lantern_delta=lam,
                            top_p=top_p, fused_prepare=fused_prepare)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    assert wl.anole and wl.fused_o7 == fuse and wl.n_spec == (spec if fuse else 0) and wl.fused_prepare == fused_prepare
    if ties:
        # logits on a grid of 0.5 (bf16-exact, and so is every CFG mix of them): ~130 distinct values among 8192 ids, so the 1 - top_p boundary
        # falls INSIDE a group of equal logits in every row -- the stable-sort order of the tied entries (index order) decides which of them go
        for t in (wl.cond, wl.uncond):
            t.copy_(((t.float() * 2).round() / 2).to(torch.bfloat16))
    for _ in range(steps):
        wl.step()
    wl.join()
    torch.cuda.synchronize()
    wl.check_status(0, steps)
    gb, ga, gt = wl.log_best[:steps].cpu().numpy(), wl.log_alen[:steps].cpu().numpy(), wl.log_token[:steps].cpu().numpy()
    tb, N = wl.tb, wl.N
    ri = tb["retrieve_indices"].copy()
    ri[ri < 0] += N
    ri = ri.astype(np.int32)
    table = wl.table_full.cpu().numpy().view(np.uint16)
    u16 = lambda t: t.cpu().view(torch.int16).numpy().view(np.uint16)
    sst, ssp = wl.ss_token.cpu().numpy(), wl.ss_prob.cpu().numpy()
    ub, first, op_off = wl.u_bonus.cpu().numpy(), wl.first_token.cpu().numpy(), wl.d_op_off.cpu().numpy()

    def dense_of(s_, b):          # (per checked sequence: the pools of a 264-sequence run do not fit the host as dense rows)
        orig = wl.orig_prob[s_, b].cpu().numpy()
        d = np.zeros(orig.shape[:-1] + (HN.V,), np.float32)
        d[..., HN.IMG_LO:HN.IMG_HI] = orig
        return d
    # (Anole / LlamaGen: the reference applies the HF processors -- here T = 1, top_k -- inside evaluate_posterior, per visited row; O7 / the raw-row
    # path apply them where the rows are produced: the same distribution)
    ocfg = oracle.EpConfig.anole(True, lantern=True, k=k, delta=lam, temperature=1.0, top_p=top_p, top_k=cfg.top_k)
    n_acc = n_rej = 0
    for b in range(0, cfg.n_seq, every):
        tok, cursor = int(first[b]), 0
        for i in range(steps):
            s_ = i % cfg.pool_steps
            cand, cp, tc = oracle.gather_candidates(sst[s_, b], ssp[s_, b], tok, tb["tree_indices"], tb["retrieve_indices"])
            proc = oracle.cfg_mask_topk(u16(wl.cond[s_, b]), u16(wl.uncond[s_, b]), cfg.cfg_scale, model=oracle.MODEL_ANOLE, img_lo=HN.IMG_LO, img_hi=HN.IMG_HI, bf16=True)
            aux = oracle.StaticAux(cart_prob=cp, orig_prob=dense_of(s_, b), op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"], b_idx=tb["b_idx"], tree_cand=tc)
            best, alen, sp, cnt = oracle.evaluate_posterior(ocfg, proc, ri, cand, wl.uniforms_host[b, cursor:cursor + 64], table=table, aux=aux)
            cursor += int(cnt[3])
            tok = oracle.sample_inverse_cdf(sp, float(ub[i, b]))
            assert (int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) == (best, alen, tok), (b, i)
            n_acc += alen
            n_rej += int(cnt[2])
    assert n_acc > 0 and n_rej > 0


@pytest.mark.parametrize("fuse,groups,spec", [(False, 1, 0), (True, 1, 0), (True, 2, 0), (True, 1, 2), (True, 2, 1)],
                         ids=["o7_launch", "raw_rows", "raw_rows_2_groups", "raw_rows_2_prepared", "raw_rows_2_groups_root_prepared"])
def test_llamagen_dynamic_loop_matches_oracle_loop(fuse, groups, spec):
    _llamagen_dynamic_loop(fuse, groups, spec, 1.0)


@pytest.mark.parametrize("fuse,spec", [(False, 0), (True, 0), (True, 2)], ids=["o7_launch", "raw_rows", "raw_rows_2_prepared"])
def test_llamagen_dynamic_loop_with_top_p(fuse, spec):
    """BASELINE config 2 with generate(top_p = 0.8): nucleus filtering in front of the top-k on the 16384-id rows -- O7 over all rows, the prepared rows
    beside the tree build, the rows the chain kernel post-processes on demand -- against the oracle's loop with the same processors."""
    _llamagen_dynamic_loop(fuse, 1, spec, 0.8)


def test_llamagen_dynamic_loop_throughput_instance():
    """More sequences per launch than CUs: LlamaGen's two-per-CU instance of the chain kernel (epw_kernel<512, 8, 1, 4, true, false, 5, ..>, EwSharedLite) on
    probability rows inside the device-resident loop, every 11th sequence held to the oracle's loop."""
    _llamagen_dynamic_loop(False, 1, 0, 1.0, n_seq=264, steps=3, every=11)


def _llamagen_dynamic_loop(fuse, groups, spec, top_p, n_seq=None, steps=8, every=1):
    """BASELINE config 2 (LlamaGen + EAGLE, standard verify: V = 16384 = the window, LANTERN off, HF processors T = 1 / top_k 2000) through the
    device-resident dynamic loop, with O7 over all rows and with the raw cond / uncond rows post-processed inside evaluate_posterior (the
    1024-thread raw-row instance): the oracle's loop over the same pools / uniforms gives the same (best path, accept length, bonus token) for
    every step and sequence."""
    import numpy as np
    import oracle
    from lantern_amd import harness as HN
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    cfg = HN.DynamicConfig(model="llamagen", n_seq=n_seq or 3 * groups, pool_steps=2, depth=4, kv_layers=2, kv_heads=4, kv_dim=64, kv_smax=512, max_steps=steps + 2,
                           fuse_o7=fuse, n_groups=groups, spec_rows=spec, top_p=top_p)
    wl = HN.DynamicVerifyWorkload(cfg, torch.device("cuda"))
    assert wl.fused_o7 == fuse and wl.lg and wl.n_spec == (spec if fuse else 0)
    for _ in range(steps):
        wl.step()
    wl.sync()
    wl.check_status(0, steps)
    gb, ga, gt = wl.log_best[:steps].cpu().numpy(), wl.log_alen[:steps].cpu().numpy(), wl.log_token[:steps].cpu().numpy()
    uni, ub = wl.uniforms.cpu().numpy(), wl.u_bonus.cpu().numpy()
    ocfg = oracle.EpConfig.llamagen(False, lantern=False, temperature=1.0, top_p=top_p, top_k=cfg.logit_top_k)      # the HF processors run inside evaluate_posterior
    N = wl.N
    n_acc = 0
    for b in range(0, cfg.n_seq, every):
        tok, cursor = int(wl.first_token[b]), 0
        for i in range(steps):
            p = wl.pools[i % cfg.pool_steps]
            draft, ret, mask, pos = oracle.tree_dynamic_finalize(p["scores"][b].cpu().numpy(), p["tokens"][b].cpu().numpy(), p["parents"][b].cpu().numpy(),
                                                                 cfg.top_k, cfg.total_tokens, tok)
            cand = np.where(ret >= 0, draft[np.clip(ret, 0, None)], -1)
            proc = oracle.cfg_mask_topk(p["cond"][b].cpu().view(torch.int16).numpy().view(np.uint16), p["unc"][b].cpu().view(torch.int16).numpy().view(np.uint16),
                                        cfg.cfg_scale, model=oracle.MODEL_PLAIN, bf16=True)
            best, alen, sp, cnt = oracle.evaluate_posterior(ocfg, proc, H.row_index_from_retrieve(ret, N), cand, uni[b, cursor:cursor + 64])
            cursor += int(cnt[3])
            tok = oracle.sample_inverse_cdf(sp, float(ub[i, b]))
            assert (int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) == (best, alen, tok), (b, i)
            n_acc += alen
    assert n_acc > 0
    gen = (ga.astype("int64") + 1).sum(0)
    len_c, len_u = wl.lengths(steps & 1)
    assert (len_c.cpu().numpy() == cfg.prompt_len + 3 + gen).all() and (len_u.cpu().numpy() == 3 + gen).all()


@pytest.mark.parametrize("model", ["lumina", "llamagen"])
def test_dynamic_prepared_rows_are_the_o7_rows(model):
    """lantern_prepare_step with dynamic trees: the rows it leaves for the root and node 1 (beside the tree build, in the same launch) are, bit for
    bit, the rows lantern_cfg_mask_topk_window computes for those nodes -- on the 8192-id image window (Lumina) and on LlamaGen's 16384 ids."""
    from lantern_amd import harness as HN
    from lantern_amd import ops
    cfg = HN.DynamicConfig(model=model, n_seq=4, pool_steps=1, depth=4, kv_layers=2, kv_heads=4, kv_dim=64, kv_smax=512, max_steps=4, fuse_o7=True, n_groups=2,
                           spec_rows=2)
    wl = HN.DynamicVerifyWorkload(cfg, torch.device("cuda"))
    assert wl.n_spec == 2
    wl.win.fill_(float("nan"))
    wl.step()
    wl.sync()
    pool = wl.pools[0]
    B, N = cfg.n_seq, wl.N
    lg = model == "llamagen"
    pos = wl.pos_abs.view(B * N)
    full = ops.cfg_mask_topk_window(pool["cond"].view(B * N, wl.V), pool["unc"].view(B * N, wl.V), cfg.cfg_scale, wl.lo, wl.W,
                                    model=ops.MODEL_PLAIN if lg else ops.MODEL_LUMINA, pos_ids=None if lg else pos, pos_base=cfg.prompt_len + 3,
                                    top_k=cfg.logit_top_k, probs=True)
    rows = full[0] if isinstance(full, tuple) else full
    rows = rows.view(B, N, wl.W)
    for node in (0, 1):
        assert torch.equal(wl.win[:, node], rows[:, node]), node
    assert torch.isnan(wl.win[:, 2:]).all()                       # nothing else was written


@pytest.mark.parametrize("fuse,spec", [(False, 0), (True, 0), (True, 2)], ids=["o7_launch", "raw_rows", "raw_rows_2_prepared"])
def test_dynamic_step_one_call_equals_per_kernel_calls(fuse, spec):
    """lantern_verify_step with dynamic groups (O4 + O6-dynamic in one launch, lantern_tree_dynamic_candidates) against the same step as
    one call per entry point (lantern_tree_dynamic_finalize, then lantern_gather_candidates_dynamic, ...): every tree buffer, candidate
    table, verdict, length, KV slab and accepted row identical after every step."""
    from lantern_amd import harness as HN
    steps = 6
    wls = []
    for native, groups in ((True, 2), (False, 1)):
        # (the per-kernel run keeps every row on demand: prepared rows must not change a bit)
        cfg = HN.DynamicConfig(n_seq=4, pool_steps=2, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 2, lantern_k=300, fuse_o7=fuse,
                               n_groups=groups, native_step=native, spec_rows=spec if native else 0)
        wls.append(HN.DynamicVerifyWorkload(cfg, torch.device("cuda")))
    for wl in wls:              # the same random rows in every sequence's slabs, whatever the slab order
        for g in range(wl.G):
            for half in range(2):
                for q in range(wl.Bg):
                    gen = torch.Generator(device="cuda").manual_seed(1000 * half + g * wl.Bg + q)
                    sl = wl.slabs[g * 2 * wl.Bg + half * wl.Bg + q]
                    sl.copy_(torch.randn(sl.shape, generator=gen, device="cuda").to(sl.dtype))
    for i in range(steps):
        for wl in wls:
            wl.step()
            wl.sync()
        a, b = wls
        for name in ("draft", "mask", "pos", "ret", "nleaf", "mdepth", "cand", "ret_pd", "row_index", "pos_abs", "out_hidden", "acc_tokens", "cursor"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (i, name)
        for name in ("log_best", "log_alen", "log_token", "log_cnt"):
            assert torch.equal(getattr(a, name)[:i + 1], getattr(b, name)[:i + 1]), (i, name)
        for x, y in zip(a.lengths((i + 1) & 1), b.lengths((i + 1) & 1)):
            assert torch.equal(x, y)
    # slabs: a's order is [group][cond | uncond][sequence in group], b's is [cond | uncond][sequence]
    a, b = wls
    for g in range(a.G):
        for half in range(2):
            for q in range(a.Bg):
                assert torch.equal(a.slabs[g * 2 * a.Bg + half * a.Bg + q], b.slabs[half * b.Bg + g * a.Bg + q]), (g, half, q)


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
def test_step_error_names_the_group_and_the_stage():
    """An entry point that refuses its arguments inside lantern_verify_step comes back with the group and the stage in front of the
    kernel's own message, and the next step works."""
    from lantern_amd import _lib
    from lantern_amd import harness as HN
    cfg = HN.WorkloadConfig(n_seq=4, pool_steps=2, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=16, n_groups=2, ep_kernel="chain",
                            fuse_o7=True, spec_rows=3)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    wl.step()
    wl.sync()
    keep = wl._steps[(0, 0)][1].ep.k
    for arr in wl._steps.values():
        arr[1].ep.k = -5                 # an argument the kernel's host side refuses
    try:
        with pytest.raises(_lib.LanternError, match="group 1, evaluate_posterior"):
            wl.step()
    finally:
        for arr in wl._steps.values():
            arr[1].ep.k = keep
    wl.sync()
    wl.step()
    wl.sync()


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 2])
def test_node_kernel_loop_matches_chain_loop(groups):
    """ep_kernel = "nodes" (one workgroup per internal tree node, then the walk): the whole step loop gives the chain kernel's verdicts,
    tokens, counters and KV rows, and the oracle's token stream."""
    import bench
    from lantern_amd import harness as HN
    steps = 60
    outs = []
    for ep in ("nodes", "chain"):
        cfg = HN.WorkloadConfig(n_seq=8, pool_steps=4, kv_layers=2, kv_heads=4, kv_smax=1024, max_steps=steps + 4, sigma=5.0, n_groups=groups,
                                ep_kernel=ep)
        wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
        for _ in range(steps):
            wl.step()
        wl.sync()
        wl.check_status(0, steps)
        outs.append((wl.log_best[:steps].clone(), wl.log_alen[:steps].clone(), wl.log_token[:steps].clone(), wl.log_cnt[:steps].clone(),
                     wl.lens[steps & 1].clone(), torch.stack([s.clone() for s in wl.slabs])))
        if ep == "nodes":
            gb, ga, gt = [x.cpu().numpy() for x in outs[0][:3]]
            stream = [[(int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) for b in range(cfg.n_seq)] for i in range(steps)]
            res = bench.cpu_baseline(wl, steps_budget_s=1e9, n_seq=cfg.n_seq, gpu_tokens_by_seq=stream)
            assert res["matches_gpu_token_stream"], res
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("geom", ["slab_blocks", "tiled"])
def test_verify_step_commits_nothing_where_the_walk_reported_a_status(geom):
    """lantern_verify_step's commit gate: a sequence whose evaluate_posterior walk ends in a status (here LANTERN_ST_UNIFORMS: its
    uniform stream is exhausted) must move no KV row, keep its lengths, list no token (accepted_tokens = -1) and leave zero-filled hidden rows -- the
    mirrors' retry paths rewind the cursor, redo the step on the dense kernel and commit from the host, so a second commit here would move KV rows twice.
    Both commit kernels: `slab_blocks` (small slabs: update_inputs_slabs_kernel, several slabs per workgroup) and `tiled` (update_inputs_kernel).  The
    other sequences of the same launches commit exactly what an undisturbed twin workload commits."""
    from lantern_amd import harness as HN
    kv = dict(kv_layers=2, kv_heads=4) if geom == "slab_blocks" else dict(kv_layers=16, kv_heads=32)      # outer x 16 chunks: 256 (<= 4096) / 16384
    mk = lambda: HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=4, pool_steps=2, kv_smax=256, max_steps=8, sigma=2.0, n_groups=2, **kv), torch.device("cuda"))
    wl, twin = mk(), mk()
    assert wl._steps, "the one-call step (lantern_verify_step) is what is under test"
    gen = torch.Generator(device="cuda").manual_seed(5)
    for a, b in zip(wl.slabs, twin.slabs):
        a.copy_(torch.randn(a.shape, device="cuda", generator=gen).to(torch.bfloat16))
        b.copy_(a)
    for w in (wl, twin):
        w.step()
        w.join()
    torch.cuda.synchronize()
    forced = [1, 2]                                         # one sequence of each stream group (groups hold sequences [0, 1] and [2, 3])
    for b in forced:
        wl.cursor[b] = wl.n_uniforms                        # nothing left of its random.random() stream
    before = [s.clone() for s in wl.slabs]
    prev = wl.lens[1].clone()                               # lengths after step 0 (parity 1 is step 1's input)
    wl.out_hidden.fill_(7.0)
    for w in (wl, twin):
        w.step()
        w.join()
    torch.cuda.synchronize()
    cnt = wl.log_cnt[1].cpu()
    assert [int(cnt[b, 5]) for b in range(4)] == [2 if b in forced else 0 for b in range(4)]       # LANTERN_ST_UNIFORMS = 2 (include/lantern_hip.h)
    with pytest.raises(Exception):
        wl.check_status(0, 2)
    new = wl.lens[0]
    acc, tacc = wl.acc_tokens.cpu(), twin.acc_tokens.cpu()
    for b in range(4):
        g, l = divmod(b, wl.Bg)
        for j in range(2):
            si = g * 2 * wl.Bg + j * wl.Bg + l
            if b in forced:
                assert torch.equal(wl.slabs[si], before[si]), (b, j)                 # byte-identical slabs
                assert int(new[si]) == int(prev[si])                                  # new_len == prev
            else:
                assert torch.equal(wl.slabs[si], twin.slabs[si]), (b, j)
                assert int(new[si]) == int(twin.lens[0][si]) == int(prev[si]) + int(wl.log_alen[1, b]) + 1
        if b in forced:
            assert (acc[b] == -1).all()
            assert float(wl.out_hidden[b].float().abs().max()) == 0.0               # zero-filled, not "as the forward left them"
        else:
            assert torch.equal(acc[b], tacc[b]) and torch.equal(wl.out_hidden[b], twin.out_hidden[b])
            assert int(wl.log_best[1, b]) == int(twin.log_best[1, b]) and int(wl.log_token[1, b]) == int(twin.log_token[1, b])


@pytest.mark.gpu
@pytest.mark.parametrize("groups,window,spec,geom", [(3, 1, 2, "slab_blocks"), (4, 1, 3, "tiled"), (4, 2, 3, "tiled"), (2, 1, 0, "slab_blocks")])
def test_commit_turn_taking_changes_the_schedule_not_the_results(groups, window, spec, geom):
    """lantern_step_group.turn (round 6): the stream groups take turns moving their KV rows -- the chain kernel of a group ends when `turn[0]` says it is
    the group's turn, its commit launch releases the turn -- without any cross-stream event.  Every step's verdict, the lengths and every KV slab equal
    the free-running loop's; the counters say every commit launch was counted exactly once (both commit kernel forms: slab blocks and tiles)."""
    from lantern_amd import harness as HN
    steps, n_seq = 40, 4 * groups
    kv = dict(kv_layers=2, kv_heads=4, kv_smax=512) if geom == "slab_blocks" else dict(kv_layers=6, kv_heads=32, kv_smax=512)          # (tiled: outer x chunks > 4096)
    outs = []
    for cw in (window, 0):
        cfg = HN.WorkloadConfig(n_seq=n_seq, pool_steps=4, max_steps=steps + 4, sigma=5.0, n_groups=groups, ep_kernel="chain", fuse_o7=True, spec_rows=spec,
                                commit_window=cw, **kv)
        wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
        for _ in range(steps):
            wl.step()
        wl.join()
        torch.cuda.synchronize()
        wl.check_status(0, steps)
        outs.append((wl.log_best[:steps].clone(), wl.log_alen[:steps].clone(), wl.log_token[:steps].clone(), wl.log_cnt[:steps].clone(),
                     wl.cond_lens(steps & 1).clone(), torch.stack([s_.clone() for s_ in wl.slabs])))
        if cw:
            t = wl._turn.cpu().numpy()
            assert int(t[0]) == steps * groups, t[:64]                   # every commit launch released its turn exactly once
            assert not t[1:].any(), "every completion counter is back at zero once its launch is done"
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 2])
def test_dense_kernel_set_through_verify_step(groups):
    """lantern_step_group.dense (missing 4): O6 -> O7 over all rows at the full vocabulary -> the dense evaluate_posterior -> the bonus draw ->
    the commit launch in ONE call, against the same kernels called one by one (lantern_gather_candidates, lantern_cfg_mask_topk, lantern_evaluate_posterior,
    lantern_accept_gather, lantern_kv_gather): verdicts, bonus tokens, distributions, KV slabs, lengths, accepted hidden rows and token lists, step after step."""
    from lantern_amd import harness as HN
    mk = lambda one: HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=4, pool_steps=2, path="dense", kv_layers=2, kv_heads=4, kv_smax=256, max_steps=12, sigma=4.0,
                                                               n_groups=groups, dense_one_call=one), torch.device("cuda"))
    a, b = mk(True), mk(False)
    gen = torch.Generator(device="cuda").manual_seed(11)
    for x, y in zip(a.slabs, b.slabs):
        x.copy_(torch.randn(x.shape, device="cuda", generator=gen).to(torch.bfloat16))
        y.copy_(x)
    for i in range(8):
        for w in (a, b):
            w.step()
            w.join()
        torch.cuda.synchronize()
        assert torch.equal(a.sample_p, b.sample_p) and torch.equal(a.out_hidden, b.out_hidden) and torch.equal(a.acc_tokens, b.acc_tokens), i
        for x, y in zip(a.lens, b.lens):
            assert torch.equal(x, y), i
    a.check_status(0, 8)
    for k in ("log_best", "log_alen", "log_cnt", "log_token"):
        assert torch.equal(getattr(a, k)[:8], getattr(b, k)[:8]), k
    for x, y in zip(a.slabs, b.slabs):
        assert torch.equal(x, y)
    assert int(a.log_alen[:8].sum()) > 0 and a._densecache and not getattr(b, "_densecache", None)



@pytest.mark.gpu
@pytest.mark.parametrize("groups,spec,window,geom", [(1, 3, 0, "slab_blocks"), (2, 2, 0, "slab_blocks"), (4, 3, 1, "tiled"), (1, 5, 0, "tiled"), (4, 1, 1, "tiled"), (2, 1, 0, "slab_blocks")])
def test_prepare_stage_inside_the_chain_launch(groups, spec, window, geom):
    """LANTERN_STEP_FUSED_PREPARE: the candidate assembly and the likely rows ride in the chain launch (helper workgroups publish each row by storing the step's epoch
    behind its agent-scope row stores; a sequence takes a row when it is published, else post-processes it itself; spec 1 = the root alone: no helper, bench.py's
    default) -- two launches per group and step.  Against the oracle's loop, and every
    verdict, length, KV slab, accepted hidden row and token list identical to the three-launch form over 60 steps, with and without commit turn-taking."""
    import bench
    from lantern_amd import harness as HN
    kv = dict(kv_layers=2, kv_heads=4) if geom == "slab_blocks" else dict(kv_layers=16, kv_heads=32)
    steps = 60
    mk = lambda fused: HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=4 * groups, pool_steps=4, kv_smax=512, max_steps=steps + 4, sigma=5.0, n_groups=groups, ep_kernel="chain",
                                                                 fuse_o7=True, spec_rows=spec, fused_prepare=fused, commit_window=window, **kv), torch.device("cuda"))
    a, b = mk(True), mk(False)
    gen = torch.Generator(device="cuda").manual_seed(9)
    for x, y in zip(a.slabs, b.slabs):
        x.copy_(torch.randn(x.shape, device="cuda", generator=gen).to(torch.bfloat16))
        y.copy_(x)
    for w in (a, b):
        for _ in range(steps):
            w.step()
        w.join()
    torch.cuda.synchronize()
    a.check_status(0, steps)
    assert a.fused_prepare and not b.fused_prepare and not hasattr(b, "_row_ready")
    assert int(a._row_ready.max()) == (a._row_epoch if spec >= 2 else 0) and a._row_epoch == steps          # the helpers published rows in the last step (spec 1: the root alone, no helper)
    for k in ("log_best", "log_alen", "log_cnt", "log_token"):
        assert torch.equal(getattr(a, k)[:steps], getattr(b, k)[:steps]), k
    for x, y in zip(a.slabs, b.slabs):
        assert torch.equal(x, y)
    for x, y in zip(a.lens, b.lens):
        assert torch.equal(x, y)
    assert torch.equal(a.out_hidden, b.out_hidden) and torch.equal(a.acc_tokens, b.acc_tokens)
    assert torch.equal(a.cand2[0], b.cand2[0]) and torch.equal(a.cand2[1], b.cand2[1]) and torch.equal(a.tree_cand, b.tree_cand) and torch.equal(a.cart_prob, b.cart_prob)
    gb, ga, gt = [x[:steps].cpu().numpy() for x in (a.log_best, a.log_alen, a.log_token)]
    stream = [[(int(gb[i, s]), int(ga[i, s]), int(gt[i, s])) for s in range(a.cfg.n_seq)] for i in range(steps)]
    res = bench.cpu_baseline(a, steps_budget_s=1e9, n_seq=a.cfg.n_seq, gpu_tokens_by_seq=stream)
    assert res["matches_gpu_token_stream"], res
    # a second run on the same buffers (reset_state restarts the step index, not the epochs: no word of the first run reads as "published")
    if spec >= 2:
        a.reset_state(); b.reset_state()
        for w in (a, b):
            for _ in range(12):
                w.step()
            w.join()
        torch.cuda.synchronize()
        assert a._row_epoch == steps + 12
        for k in ("log_best", "log_alen", "log_cnt", "log_token"):
            assert torch.equal(getattr(a, k)[:12], getattr(b, k)[:12]), k

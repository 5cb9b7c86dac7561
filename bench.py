#!/usr/bin/env python3
"""bench.py -- accepted image-tokens/s of the LANTERN verify/accept hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1: one rank per GPU, launched by torch.distributed.run (ranks from the environment) or, when that environment is absent, by
  bench.py itself (spawn_ranks).  Ranks verify DISJOINT sequences with no collective on the accept path (the only torch.distributed
  calls are the timing barrier and the max/sum of the per-rank timing scalars the contract asks for).  Weak scaling: per-GPU work fixed.

Workload = BASELINE.json config C3 (Lumina-mGPT-7B-768 + LANTERN relaxed accept, k=1000, delta=0.1, static tree mc_sim_7b_63) on
synthetic 768x768 image-token sequences: a "step" is one verify step (O6 -> O7 -> O8 -> O9 -> O10) over the --seqs-per-gpu sequences
resident on the GPU; inputs are HBM-resident before the timed region.  `value` = accepted tokens of all ranks / max-over-ranks wall time.
Default launch: 64 sequences in 4 stream groups, one lantern_verify_step call per step, per group TWO launches in the order a real decode loop can issue them (a step's
rows exist only after the previous step's commit + the drafter and target forwards): evaluate_posterior_window on raw bf16 rows with its prepare stage inside the launch
(LANTERN_STEP_FUSED_PREPARE: O6 by the sequence workgroups, O8 + the rows of O7 the walk visits, on demand) -> update_inference_inputs (O9 + O10).
`--fused-prepare 0 --spec-rows 3` is the three-launch form (lantern_prepare_step = O6 + the 3 most likely rows of O7 as its own launch in front).  The round-5 variant that hid
the prepare stage in the PREVIOUS step's commit launch is only possible on pre-generated pools: a harness-only extra of the three-launch form, never `value`.
`--groups 1 --no-fuse-o7 --spec-rows 0` is the four-launch step (every row through cfg_mask_topk first).
N > 1 prints BOTH scaling forms in the one line: `value` = weak scaling (--seqs-per-gpu on every rank) and `c5_strong` = BASELINE config C5 / BASELINE.md section 2
(64 sequences in ALL, split evenly over the ranks: run.sh:76-91), each with its own barrier-bracketed K-step timed region.

Extra objects on the JSON line:
  roofline      evaluate_posterior of the timed configuration: algorithmic bytes (SURVEY 8d contract formula, from the kernel's own
                counters) / its mean launch duration (HIP events recorded by the launch itself); peak 8000 GB/s.  `windowed_kernel`: the bytes
                the windowed design really has to move; `traffic`: PMC bytes from profiles/r06_ep_traffic.json (refused when measured on other kernel
                sources); `saturating`: the same kernel alone at the largest batch of the sweep, on rotating inputs.
  kernels       the same for the row post-process launch (prepare_step / cfg_mask_topk) and update_inference_inputs.
  per_kernel_single_group  one stream, every stage its own launch: each kernel against its SURVEY 8d roofline at the full 64-sequence
                launch size, with the chain and the node-parallel evaluate_posterior.
  ep_batch_sweep  evaluate_posterior alone over {1, 8, 64, 256, 512, 4096} sequences per launch (three forms), after the timed region.
                `frac` there = bytes really moved / time / 8 TB/s.
  lambda_mode   the same loop with lantern_delta = 5 (LANTERN++: tau = 4 p(x)), BASELINE.md run B.
  dynamic_tree  the EAGLE-2 half of C3 (a different 59-node tree per sequence and step), raw rows; and with O7 over all rows.
  drafter_layer one drafting call of the drafter's decoder layer at 7B size (tools/layer_bench.py in a child process).
  drafter_cycle one whole drafting cycle (prefill + depth x lantern_draft_depth + finalisation) of the three model families (tools/draft_bench.py).
  mirror_generate  the drop-in API (EaLumina_mGPT.generate) with stand-in forwards: us per verify step of the mirror's loop body (tools/mirror_bench.py).
  step_latency_us  whole-step wall time at 1 and 8 sequences (the reference's own batch is 1), three evaluate_posterior forms.
  other_groupings  the default kernels with 1 and 2 stream groups.
  cpu_baseline  the oracle (C port of the reference path, pthreads) timed on this host's cores over a bounded sample of the same
                pools / uniforms; it must reproduce the GPU's accepted-token stream.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def _groups_from_argv(argv, default=4):
    for i, a in enumerate(argv):
        if a == "--groups" and i + 1 < len(argv) and argv[i + 1].isdigit():
            return int(argv[i + 1])
        if a.startswith("--groups=") and a[9:].isdigit():
            return int(a[9:])
    return default


# The HIP runtime maps streams to GPU_MAX_HW_QUEUES (default 4) hardware queues: the current stream + 4 or more group streams would share
# queues and serialise.  Must be set before the runtime starts (i.e. before torch is imported); left alone for <= 3 groups (measured: no
# effect on the default run, and streams created later in the process land on worse queues with 8).
_ENV_SET_HERE = [k for k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_INTERRUPT") if k not in os.environ]          # (a caller's own settings stay, also for the child tools)
if _groups_from_argv(sys.argv) > 3:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Host waits by polling instead of interrupts (ROCr reads this when the runtime starts): the timed region ends in a barrier + synchronize, and an
# interrupt wake-up costs tens of microseconds -- 1.5-3 us per step of the driver's 20-step run (measured, alternating runs on one box: 75.1 / 76.0 /
# 76.1 us per step with interrupts, 74.5 / 74.2 / 72.9 polling).  A caller's own setting wins.
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seqs-per-gpu", type=int, default=64, help="sequences resident per GPU (KV slabs: 4.3 GB each at 4096 rows; 64 -> 276e9 of the 309e9 bytes): BASELINE's 64, in 4 stream groups of 16")
    ap.add_argument("--total-seqs", type=int, default=0, help="BASELINE config 5 (strong scaling): this many sequences in ALL, split evenly over the --gpus ranks "
                    "(64 over 8 GPUs = 8 per GPU, run.sh:76-91); 0 = --seqs-per-gpu on every rank (weak scaling, the default)")
    ap.add_argument("--pool-steps", type=int, default=16)
    ap.add_argument("--tree", type=str, default="mc_sim_7b_63", help="static tree (a name in lantern_amd/drafters/choices.py), e.g. naive_extend_57 (N=58,P=33,D=6)")
    ap.add_argument("--lantern-k", type=int, default=1000)
    ap.add_argument("--lantern-delta", type=float, default=0.1)
    ap.add_argument("--sigma", type=float, default=5.0, help="drafter noise; frozen at 5.0: mean accepted tokens/step ~2.6")
    ap.add_argument("--path", choices=["window", "dense"], default="window",
                    help="window: v2 kernels (32 KB image-window rows, LDS-resident residual); dense: v1 kernels (full-V rows)")
    ap.add_argument("--ep", choices=["auto", "nodes", "chain"], default="auto",
                    help="windowed evaluate_posterior: nodes = one workgroup per internal tree node + the walk (lantern_evaluate_posterior_nodes); "
                         "chain = one serial chain per sequence (lantern_evaluate_posterior_window); auto (default) = nodes when the rank holds <= 16 "
                         "sequences (C5's 8 per GPU: 40 vs 49-52 us per step, profiles/r04_c5_share.json), chain above")
    ap.add_argument("--no-fuse-o7", dest="fuse_o7", action="store_false", help="every stage its own launch: cfg_mask_topk for ALL rows, then evaluate_posterior on probability rows "
                    "(default: the chain kernel takes the raw logits -- LANTERN_ROWS_RAW_BF16 -- and post-processes the rows its walk visits)")
    ap.add_argument("--spec-rows", type=int, default=-1, help="with --fuse-o7: rows of the K most likely tree nodes are post-processed up front, in the candidate-assembly launch (lantern_prepare_step) -- or, "
                    "with --fused-prepare, by helper workgroups of the chain launch (all but the root's); the others on demand.  -1 (default): 3 with a prepare launch, 1 (the root, by "
                    "its own sequence: no helper) with --fused-prepare")
    ap.add_argument("--python-launch", action="store_true", help="launch every kernel of the step from Python (4 ctypes calls per group) instead of one lantern_verify_step call")
    ap.add_argument("--no-kv", action="store_true", help="skip the KV slabs (debug only; invalid as a headline)")
    ap.add_argument("--kv-smax", type=int, default=4096, help="rows per KV slab (BASELINE.md: 4096 = max_position_embeddings; a 768x768 image needs 2481)")
    ap.add_argument("--kv-pad-rows", type=int, default=None, help="extra rows per (layer, head) group of a KV slab (row stride = kv_smax + pad; harness default 16)")
    ap.add_argument("--spin-up", type=float, default=0.5, help="seconds of untimed setup launches before the warm-up steps (GPU out of its idle power state)")
    ap.add_argument("--no-events", action="store_true", help="skip the eager per-kernel timing pass")
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per (pool slot, group) instead of launching eagerly")
    ap.add_argument("--side-stream", action="store_true", help="launch O6 beside O7 and O10 beside O9 on a second HIP stream (measured slower: event waits)")
    ap.add_argument("--groups", type=int, default=4, help="split the GPU's sequences into this many groups, each on its own HIP stream (independent sequences: one group's latency-bound evaluate_posterior overlaps the others' bandwidth-bound kernels)")
    ap.add_argument("--fused-prepare", type=int, default=-1, help="1: the prepare stage inside the chain launch (LANTERN_STEP_FUSED_PREPARE: two launches per group and step); 0: its own "
                    "launch; -1 (default): on for the chain kernel on raw rows with >= 2 prepared rows")
    ap.add_argument("--commit-window", type=int, default=-1, help="commit turn-taking between the stream groups (lantern_step_group.turn): at most this many groups move their KV rows at the "
                    "same time, ordered by in-kernel tickets instead of cross-stream events; 0 = the groups run free and fall into lock-step; -1 (default) = 1 with four "
                    "stream groups on the chain kernel (68.2 - 68.9 us per step against 71.4 - 72.2 free-running), else 0")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra runs after the timed region (lambda mode, B=1 / B=8 step latency, two stream groups)")
    ap.add_argument("--ep-sweep", type=str, default="1,8,64,256,512,4096",
                    help="e.g. 256,2048: batch sizes for the evaluate_posterior-only roofline sweep (BASELINE.md section 2: the 60 %% target is "
                         "stated 'at the saturating batch size'), run by rank 0 AFTER the timed region and the event pass, KV slabs released first; "
                         "Pass '' under rocprofv3 so that every "
                         "kernel is launched on one homogeneous workload (its averages then agree with the HIP-event averages)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline budget; 0 disables")
    ap.add_argument("--cpu-seqs", type=int, default=0, help="sequences in the CPU sample (0 = host cores)")
    ap.add_argument("--extras-out", type=str, default=os.path.join(ROOT, "bench_extras.json"),
                    help="file that receives the FULL report (the compact headline + every extra run: per_kernel_single_group, configs, dynamic_tree, "
                         "ep_batch_sweep, drafter_*, mirror_generate, ...); the same object goes to stderr as one line.  stdout carries exactly ONE "
                         "line: the compact headline (< 4 KB) the driver parses.  '' = no file")
    ap.add_argument("--tuning", type=str, default="", help="measurement runs only: name=value[,name=value...] passed to lantern_tuning_set before anything is launched "
                    "(kernel-instance / launch-shape choices, include/lantern_hip.h; the library reads no environment variable).  The line's config says what was set")
    ap.add_argument("--dist-backend", choices=["auto", "nccl", "gloo"], default="auto",
                    help="process group of the N > 1 run (used for the timing barrier and three scalars only: the accept path has no collective).  "
                         "auto = nccl (RCCL), falling back to gloo on host tensors when the RCCL group cannot be brought up -- decided before any kernel is launched")
    return ap.parse_args()


# ----------------------------------------------------------------------------- CPU baseline

def effective_cpus() -> int:
    """Cores this process can really use: the affinity mask cut by the cgroup CPU quota (a GPU box hands a container a share of its
    host -- os.cpu_count() said 256 where ~16 cores' worth of time was available, and 63 baseline threads ran 4.6x slower each than one
    alone).  LANTERN_CPU_THREADS overrides."""
    if os.environ.get("LANTERN_CPU_THREADS"):
        return max(1, int(os.environ["LANTERN_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(p)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())               # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(round(q / p))))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(wl, steps_budget_s: float, n_seq: int, gpu_tokens_by_seq, python_budget_s=None):
    """The oracle (C restatement of the reference's Python path) over the first n_seq sequences and as
    many steps as fit the budget, one host thread per sequence.  Checker AND timed baseline: the
    accepted-token stream must equal the GPU's for the same steps."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    from lantern_amd import harness as HN

    c = wl.cfg
    S = c.pool_steps
    N, P, D, R = wl.N, wl.P, wl.D, wl.R
    tb = wl.tb
    ri = tb["retrieve_indices"].copy()
    ri[ri < 0] += N
    ri = ri.astype(np.int32)
    table = wl.table_full.cpu().numpy().view(np.uint16)      # the oracle reads the reference's [K, K-1] layout
    cond = wl.cond[:, :n_seq].cpu().view(torch.int16).numpy().view(np.uint16)
    uncond = wl.uncond[:, :n_seq].cpu().view(torch.int16).numpy().view(np.uint16)
    orig = wl.orig_prob[:, :n_seq].cpu().numpy()
    orig_win = orig if wl.windowed else np.ascontiguousarray(orig[..., HN.IMG_LO:HN.IMG_HI])
    if wl.windowed:     # the oracle takes the reference's dense [R,V] drafter rows: expand once, outside the timed region
        dense = np.zeros(orig.shape[:-1] + (HN.V,), np.float32)
        dense[..., HN.IMG_LO:HN.IMG_HI] = orig
        orig = dense
    sst = wl.ss_token[:, :n_seq].cpu().numpy()
    ssp = wl.ss_prob[:, :n_seq].cpu().numpy()
    hid = wl.hidden[:, :n_seq].cpu().view(torch.int16).numpy()
    u_bonus = np.ascontiguousarray(wl.u_bonus[:, :n_seq].cpu().numpy())
    first = wl.first_token[:n_seq].cpu().numpy()
    op_off = wl.d_op_off.cpu().numpy()
    cfg = oracle.EpConfig.lumina(True, lantern=True, k=c.lantern_k, delta=c.lantern_delta)
    pos1 = tb["tree_position_ids"] + 1
    # Host KV slabs: same (layer, head, head_dim) geometry; S_max cut to what the sample can reach so the
    # slabs can be pre-touched (first-touch page zeroing of 2x2.1 GB per sequence would otherwise
    # dominate the CPU timing).  Bytes moved per step do not depend on S_max.
    kv_smax_cpu = min(c.kv_smax, 1024)
    kv_shape = (2 * c.kv_layers, 1, c.kv_heads, kv_smax_cpu, c.kv_dim)

    def run_seq(b, n_steps, out):
        slabs = kv_slabs.get(b) if c.with_kv else None
        lens = [c.prompt_len + 3, 3]
        cursor, tok, toks = 0, int(first[b]), []
        for i in range(n_steps):
            s = i % S
            cand, cp, tc = oracle.gather_candidates(sst[s, b], ssp[s, b], tok, tb["tree_indices"], tb["retrieve_indices"])
            proc = oracle.cfg_mask_topk(cond[s, b], uncond[s, b], c.cfg_scale, model=oracle.MODEL_LUMINA, pos_ids=pos1 + lens[0],
                                        pos_base=c.prompt_len + 3, w=HN.W_LATENT, h=HN.H_LATENT, img_lo=HN.IMG_LO, img_hi=HN.IMG_HI,
                                        newline_id=HN.NEWLINE, eos_id=HN.EOS, top_k=c.top_k, bf16=True)
            aux = oracle.StaticAux(cart_prob=cp, orig_prob=orig[s, b], op_off=op_off, p_idx=tb["p_indices"], b_off=tb["b_off"],
                                   b_idx=tb["b_idx"], tree_cand=tc)
            best, alen, sp, cnt = oracle.evaluate_posterior(cfg, proc, ri, cand, wl.uniforms_host[b, cursor:cursor + 64],
                                                            table=table, aux=aux)
            cursor += int(cnt[3])
            row = tb["retrieve_indices"][best]
            if slabs is not None:
                for j in range(2):
                    oracle.kv_gather(slabs[j], row, alen + 1, lens[j])
            oracle.hidden_gather(hid[s, b], row, alen + 1)
            tok = oracle.sample_inverse_cdf(sp, float(u_bonus[i, b]))
            lens = [l + alen + 1 for l in lens]
            if lens[1] - 3 >= HN.TOKENS_PER_IMAGE:
                lens = [c.prompt_len + 3, 3]
            toks.append((best, alen, tok))
        out[b] = toks

    cores = effective_cpus()
    threads = min(cores, n_seq)
    if python_budget_s is None:
        python_budget_s = steps_budget_s
    kv_slabs = {b: [np.ones(kv_shape, np.uint16), np.ones(kv_shape, np.uint16)] for b in range(n_seq)} if c.with_kv else {}
    max_cpu_steps = (kv_smax_cpu - c.prompt_len - 3 - 32) // 6 if c.with_kv else 10 ** 9     # host slabs are cut to 1024 positions
    # calibrate on one step of one sequence, then size the sample to the budget
    t0 = time.perf_counter()
    tmp = {}
    run_seq(0, 1, tmp)
    t_step = time.perf_counter() - t0
    n_steps = int(max(2, min(len(gpu_tokens_by_seq), max_cpu_steps, steps_budget_s / max(t_step, 1e-4) * threads / n_seq)))
    # ---- C leg: the same loop in ONE call of the oracle on `threads` pthreads (no interpreter between the kernels' CPU counterparts)
    t0 = time.perf_counter()
    cb, ca, ct = oracle.verify_loop_mt(cfg, tb, op_off, dict(ss_token=sst, ss_prob=ssp, cond=cond, uncond=uncond, orig_win=orig_win, hidden=hid),
                                       wl.uniforms_host[:n_seq], u_bonus, first, table, n_steps, threads, c.cfg_scale, c.prompt_len,
                                       HN.TOKENS_PER_IMAGE, c.top_k, slabs=[kv_slabs[b][j] for b in range(n_seq) for j in range(2)] if c.with_kv else None)
    dt_c = time.perf_counter() - t0
    c_mis = sum(int((int(cb[i, b]), int(ca[i, b]), int(ct[i, b])) != tuple(gpu_tokens_by_seq[i][b])) for i in range(n_steps) for b in range(n_seq))
    c_leg = dict(value=float((ca.astype(np.int64) + 1).sum()) / dt_c, unit="accepted_tokens/s", cores=threads,
                 sample=f"{n_seq} sequences x {n_steps} verify steps, one lo_verify_loop_mt call on {threads} pthreads (oracle/lantern_oracle.c)",
                 ms_per_seq_step=1e3 * dt_c * threads / (n_seq * n_steps), matches_gpu_token_stream=(c_mis == 0), mismatches=c_mis)
    # ---- the same call with every sequence's 26 tree rows of the logit post-process shared over a team of OpenMP threads: the
    # host has more cores than the GPU has sequences (a 256-core host: 63 sequences x 4 threads)
    team = max(1, min(8, cores // max(threads, 1)))
    all_leg = None
    if team > 1:
        if c.with_kv:
            for b in range(n_seq):
                for j in range(2):
                    kv_slabs[b][j][...] = 1
        oracle.set_row_threads(team)
        try:
            t0 = time.perf_counter()
            cb2, ca2, ct2 = oracle.verify_loop_mt(cfg, tb, op_off, dict(ss_token=sst, ss_prob=ssp, cond=cond, uncond=uncond, orig_win=orig_win, hidden=hid),
                                                  wl.uniforms_host[:n_seq], u_bonus, first, table, n_steps, threads, c.cfg_scale, c.prompt_len,
                                                  HN.TOKENS_PER_IMAGE, c.top_k, slabs=[kv_slabs[b][j] for b in range(n_seq) for j in range(2)] if c.with_kv else None)
            dt_a = time.perf_counter() - t0
        finally:
            oracle.set_row_threads(1)
        a_mis = int((cb2 != cb).sum() + (ca2 != ca).sum() + (ct2 != ct).sum())
        all_leg = dict(value=float((ca2.astype(np.int64) + 1).sum()) / dt_a, unit="accepted_tokens/s", cores=threads * team,
                       sample=f"{n_seq} sequences x {n_steps} verify steps on {threads} pthreads x {team} OpenMP threads each (the tree rows of a sequence's logit post-process in parallel)",
                       ms_per_seq_step=1e3 * dt_a * threads / (n_seq * n_steps), matches_gpu_token_stream=(c_mis == 0 and a_mis == 0), mismatches=c_mis + a_mis)
        c_mis += a_mis
    if c.with_kv:
        for b in range(n_seq):          # the Python leg below replays from step 0 on fresh slabs
            for j in range(2):
                kv_slabs[b][j][...] = 1
    n_steps_py = int(max(2, min(n_steps, python_budget_s / max(t_step, 1e-4) * threads / n_seq)))
    n_steps_c, n_steps = n_steps, n_steps_py
    out = {}
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda b: run_seq(b, n_steps, out), range(n_seq)))
    dt = time.perf_counter() - t0
    accepted, mismatches = 0, 0
    for b in range(n_seq):
        for i, (best, alen, tok) in enumerate(out[b]):
            accepted += alen + 1
            gb, ga, gt = gpu_tokens_by_seq[i][b]
            mismatches += int((best, alen, tok) != (gb, ga, gt))
    # one sequence alone on one thread (BASELINE.md section 2 asks for the 1-thread figure too): a few seconds, same checker
    one, n1 = {}, int(max(2, min(n_steps, 3.0 / max(t_step, 1e-4))))
    t1 = time.perf_counter()
    run_seq(0, n1, one)
    dt1 = time.perf_counter() - t1
    single = dict(value=sum(a + 1 for _, a, _ in one[0]) / dt1, unit="accepted_tokens/s", cores=1,
                  sample=f"sequence 0 x {n1} verify steps", ms_per_seq_step=1e3 * dt1 / n1)
    py_leg = dict(value=accepted / dt, unit="accepted_tokens/s", cores=threads,
                  sample=f"{n_seq} sequences x {n_steps} verify steps through the oracle's Python wrappers, {threads} Python threads (one per sequence)",
                  ms_per_seq_step=1e3 * dt * threads / (n_seq * n_steps), matches_gpu_token_stream=(mismatches == 0), mismatches=mismatches)
    # the headline CPU figure is the fastest C leg (the Python-threaded one is throttled by the interpreter, not by the algorithm);
    # the others ride along: one thread per sequence, one thread in all, the Python-threaded loop
    head = all_leg if (all_leg is not None and all_leg["value"] > c_leg["value"]) else c_leg
    return dict(head, kind="port", host_cores=os.cpu_count() or 1, usable_cores=cores, sequences_over_threads=c_leg, all_cores=all_leg, python_threads=py_leg,
                single_thread=single,
                matches_gpu_token_stream=(c_mis == 0 and mismatches == 0), mismatches=c_mis + mismatches)


# ------------------------------------------------------------------------- EP-only batch sweep

def ep_batch_sweep(batches, device, base_cfg, iters=24, kernels=("chain", "nodes"), min_rotation_bytes=3 << 29):
    """evaluate_posterior alone over a batch sweep ({1, 8, 64, 512, 4096} sequences per launch), both forms: the
    per-sequence chain kernel and the node-parallel kernel (+ its walk).  Same synthetic recipe, no KV slabs.

    ROTATING inputs: the real O6 / O7 kernels produce the inputs of R different verify steps (pool slots cycled, the sequences' state
    advancing from step to step); each step's evaluate_posterior inputs -- probability rows, drafter rows, candidates, their tables -- are
    kept in their OWN allocations and launch i runs on set i mod R, so that a launch never finds its rows in the L2 / Infinity Cache
    (256 MiB) from the launch before: R sets hold >= `min_rotation_bytes` (1.5 GiB) whenever the sets are big enough for that
    (`rotation_bytes`, `cache_resident` say what a point got).  Nothing but the kernel sits between a launch's two HIP events (no cursor
    reset: the launches read the uniform stream at offset 0 and do not advance it).
    `frac` = bytes the windowed kernels really have to move / time / 8 TB/s; the SURVEY 8d contract formula (dense V-wide rows, which no
    windowed kernel moves) is kept as `contract_equivalent_GBps`, never as a fraction."""
    import ctypes as C
    from lantern_amd import harness as HN
    from lantern_amd._lib import check
    out = []
    for Bs in batches:
        S = 4 if Bs <= 512 else 1          # pool steps with their own random rows (a 4096-sequence pool step is 27 GB of raw logits + f32 temporaries)
        need = Bs * (S * 2 * 26 * 65536 * 2 + 6 * 26 * 65536 * 4 + 11 * 65536 * 4 * 3) + (6 << 30)   # pools + setup temporaries + the rotation sets
        free, _ = torch.cuda.mem_get_info(device)
        if need > free:
            out.append({"sequences_per_launch": Bs, "skipped": f"needs {need >> 30} GiB, {free >> 30} GiB free"})
            continue
        cfg = HN.WorkloadConfig(n_seq=Bs, pool_steps=S, use_graph=False, lantern_k=base_cfg.lantern_k, lantern_delta=base_cfg.lantern_delta,
                                sigma=base_cfg.sigma, with_kv=False, max_steps=64, path=base_cfg.path, seed_base=base_cfg.seed_base + 500,
                                ep_kernel="nodes", tree=base_cfg.tree, native_step=False)
        wl = HN.LuminaVerifyWorkload(cfg, device)
        L = wl._L
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        set_bytes = (wl.proc.numel() + wl.orig_prob[0].numel() + wl.cart_prob.numel()) * 4 + (wl.cand.numel() + wl.tree_cand.numel()) * 8
        R = int(min(48, max(4, -(-min_rotation_bytes // max(set_bytes, 1)))))
        sets = []
        for r in range(R):
            wl.step()                      # O6 + O7 (+ one O8 / O9-less bookkeeping) of verify step r: cand / proc / row_hot of that step
            torch.cuda.synchronize(device)
            slot = r % S
            keep = dict(proc=wl.proc.clone(), row_hot=wl.row_hot.clone(), cand=wl.cand.clone(), cart=wl.cart_prob.clone(), tcand=wl.tree_cand.clone(),
                        orig=wl.orig_prob[slot].clone(), best=torch.zeros_like(wl.st_best), alen=torch.zeros_like(wl.st_alen),
                        cnt=torch.zeros_like(wl.st_cnt), tok=torch.zeros_like(wl.st_token), otok=torch.zeros_like(wl.out_tok), omass=torch.zeros_like(wl.out_mass))
            buf = wl.ep_buffers(slot, 0)
            buf.logits, buf.cand, buf.cart_prob, buf.tree_cand, buf.orig_prob = (keep["proc"].data_ptr(), keep["cand"].data_ptr(), keep["cart"].data_ptr(),
                                                                              keep["tcand"].data_ptr(), keep["orig"].data_ptr())
            buf.cursor = None              # read the uniforms at offset 0, leave no cursor behind: every launch on a set repeats the same walk
            buf.best, buf.accept_len, buf.counters = keep["best"].data_ptr(), keep["alen"].data_ptr(), keep["cnt"].data_ptr()
            win = wl.ep_window(0)
            win.row_hot, win.token, win.out_tok, win.out_mass = keep["row_hot"].data_ptr(), keep["tok"].data_ptr(), keep["otok"].data_ptr(), keep["omass"].data_ptr()
            sets.append((keep, buf, win))
        row = {"sequences_per_launch": Bs, "rotation_sets": R, "rotation_bytes": R * set_bytes, "pool_steps": S,
               "cache_resident": bool(R * set_bytes < (512 << 20))}
        for kern in (kernels if wl.windowed else ("dense",)):
            def launch(r):
                _k, buf, win = sets[r % R]
                if kern == "nodes":
                    check(L.lantern_evaluate_posterior_nodes(C.byref(wl._ep_prm), C.byref(buf), C.byref(win), C.byref(wl.ep_nodes[0]), st), "ep")
                elif kern == "chain":
                    check(L.lantern_evaluate_posterior_window(C.byref(wl._ep_prm), C.byref(buf), C.byref(win), st), "ep")
                else:
                    check(L.lantern_evaluate_posterior(C.byref(wl._ep_prm), C.byref(buf), st), "ep")
            for r in range(R):             # every set once, untimed (code objects, first touch of the output buffers)
                launch(r)
            torch.cuda.synchronize(device)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            for i, (e0, e1) in enumerate(evs):
                e0.record()
                launch(i)
                e1.record()
            torch.cuda.synchronize(device)
            ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
            # the same launches back to back (whole-loop wall time / launches: what a caller that keeps the queue full gets)
            t0 = time.perf_counter()
            for i in range(iters):
                launch(i)
            torch.cuda.synchronize(device)
            loop_ms = 1e3 * (time.perf_counter() - t0) / iters
            cnts = [k_["cnt"] for k_, _b, _w in sets]
            if any(int(c_[:, 5].abs().sum()) != 0 for c_ in cnts):
                raise RuntimeError("evaluate_posterior reported a per-sequence error in the sweep")
            # bytes the launch has to bring in from HBM: rows of the visited levels, the candidates' neighbour ids, and the drafter row ONCE per
            # level with a rejection (the candidates of a level share their parent's row; later rejections re-read it from L2) -- the levels are
            # replayed from each set's inputs and verdict and checked against the kernel's rejection counters (harness.static_rejection_levels)
            if wl.windowed:
                lv = [HN.static_rejection_levels(k_["cand"], k_["cart"], k_["best"], k_["alen"], k_["cnt"]) for k_, _b, _w in sets]
                nbytes = float(np.mean([wl.ep_window_bytes_from(c_, l_) for c_, l_ in zip(cnts, lv)]))
                per_rej = float(np.mean([wl.ep_window_bytes_from(c_) for c_ in cnts]))
            else:
                nbytes = per_rej = float(np.mean([wl.ep_algorithmic_bytes_from(c_) for c_ in cnts]))
            dense = float(np.mean([wl.ep_algorithmic_bytes_from(c_) for c_ in cnts]))
            row[kern] = {"launch_ms": ms, "back_to_back_ms": loop_ms, "us_per_sequence": 1e3 * ms / Bs, "hbm_bytes_needed_per_launch": nbytes,
                         "achieved_GBps": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0,
                         "bytes_if_every_rejection_read_its_row_from_hbm": per_rej,
                         "contract_equivalent_GBps": dense / (ms * 1e-3) / 1e9}
        row["accepted_tokens_per_launch"] = float(np.mean([float((k_["alen"].float() + 1).sum()) for k_, _b, _w in sets]))
        out.append(row)
        del wl, sets
        torch.cuda.empty_cache()
    return out


def lg_batch_sweep(batches, device, iters=10, variants=(1,), reps=1):
    """evaluate_posterior alone for LlamaGen's standard verify (BASELINE config 2: V = window = 16384 ids, EAGLE-2 trees of 59 nodes, LANTERN off) at batches beyond the
    CU count: the two-per-CU throughput instance `epw_kernel<512, 8, 1, 4, true, false, 5, ..>` (lantern_tuning_set("epw_tp_lg", 1), the default: rows by LDS-DMA) and,
    on request, its variants (2: + second LDS pass for the residual, 3: rows through registers, 4: + raised priority) and the generic one-per-CU instance (0), alternating
    inside one process.  Probability rows (O7 over all 59 rows of every sequence, its own launch) -- 3.9 MB per sequence and step, so one step's rows (16 GB at 4096
    sequences) never sit in a cache.  HIP events around the launch (lantern_profile_next_launch); `frac` = (visited levels + fresh final rows) x 64 KB / time / 8 TB/s."""
    from lantern_amd import _lib as _L
    from lantern_amd import harness as HN
    out = []
    for B in batches:
        need = B * (2 * 2 * 59 * 16384 * 2 + 59 * 16384 * 4 * 3) + (6 << 30)
        free, _ = torch.cuda.mem_get_info(device)
        if need > free:
            out.append({"sequences_per_launch": B, "skipped": f"needs {need >> 30} GiB, {free >> 30} GiB free"})
            continue
        dc = HN.DynamicConfig(model="llamagen", n_seq=B, depth=4, total_tokens=58, kv_layers=12, kv_heads=12, kv_dim=64, with_kv=False, pool_steps=2,
                              max_steps=reps * len(variants) * (iters + 2) + 16, plausible=8.0, n_groups=1, fuse_o7=False, native_step=False)
        wl = HN.DynamicVerifyWorkload(dc, device)
        for _ in range(2):
            wl.step()
        torch.cuda.synchronize(device)
        names = event_names(wl)
        row = {"sequences_per_launch": B, "window": wl.W, "nodes": wl.N, "variants": {}}
        step = 2
        try:
            for _rep in range(reps):
                for v in variants:
                    _L.set_tuning("epw_tp_lg", v)
                    wl.step()
                    step += 1
                    evs = make_events(names, iters, device)
                    for i in range(iters):
                        wl.step(evs[i])
                    torch.cuda.synchronize(device)
                    cnt = wl.log_cnt[step:step + iters, :wl.Bg].double()
                    step += iters
                    ms = float(np.median([e["evaluate_posterior"][0].elapsed_time(e["evaluate_posterior"][1]) for e in evs]))
                    needed = float(((cnt[..., 0] + (1.0 - cnt[..., 4])) * wl.W * 4).sum() / iters)
                    o7 = float(np.median([e["cfg_mask_topk"][0].elapsed_time(e["cfg_mask_topk"][1]) for e in evs]))
                    row["variants"].setdefault(str(v), []).append({"launch_us": 1e3 * ms, "needed_bytes_per_launch": needed, "achieved_GBps": needed / (ms * 1e-3) / 1e9,
                                                                   "frac": needed / (ms * 1e-3) / 1e9 / 8000.0, "levels_per_sequence": float(cnt[..., 0].mean()),
                                                                   "cfg_mask_topk_us": 1e3 * o7, "cfg_mask_topk_frac": B * wl.N * wl.W * (2 * 2 + 4) / (o7 * 1e-3) / 8e12})
        finally:
            _L.set_tuning("epw_tp_lg", 1)
        wl.check_status(0, step)
        out.append(row)
        del wl
        torch.cuda.empty_cache()
    return out


# the sources of the kernels bench.py times (the verify step: candidate assembly / row post-process, the windowed evaluate_posterior kernels, the KV /
# hidden commit, the one-call sequencing); the dense evaluate_posterior (evaluate_posterior.hip: never on the timed path) and the drafter-side files
# (drafter_fc, draft_depth, tree_attention, vq_table, greedy, tree_static) are not part
VERIFY_PATH_SOURCES = ("common.h", "window_dev.h", "epw_body.h", "window_kernels.hip", "epw_generic.hip", "epw_throughput.hip", "node_kernels.hip", "logits_post.hip", "prep_dev.h", "gather_dev.h",
                       "gather_ops.hip", "tree_dynamic.hip", "tree_dynamic_dev.h", "verify_step.cpp")


def kernel_sources_sha() -> str:
    """Fingerprint of the verify path's kernel sources (lantern_amd/csrc/VERIFY_PATH_SOURCES): profiles/*_traffic.json carry the fingerprint they were
    measured at, and a file measured on other kernels is refused instead of quoted (there is no .git on the GPU box to compare commits with)."""
    import hashlib
    h = hashlib.sha256()
    for name in VERIFY_PATH_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(ROOT, "lantern_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_ep_traffic.json")


def traffic_entry(section: str, key: str):
    """(entry, note) of profiles/r06_ep_traffic.json[section][key]; entry None with the reason when the file is missing, lacks the key or was
    measured on other kernel sources."""
    if not os.path.exists(TRAFFIC_FILE):
        return None, "no PMC file (profiles/r06_ep_traffic.json)"
    tj = json.load(open(TRAFFIC_FILE))
    if tj.get("kernel_sources_sha") != kernel_sources_sha():
        return None, f"profiles/r06_ep_traffic.json was measured on other kernel sources (commit {tj.get('commit')}): refused as stale"
    t = tj.get(section, {}).get(key)
    if not t:
        return None, f"profiles/r06_ep_traffic.json has no {section}/{key}"
    return t, ("profiles/r06_ep_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 note), "
               f"measured at commit {tj.get('commit')}, same kernel sources")


def saturating_report(sweep):
    """roofline.saturating: evaluate_posterior at the largest batch of the sweep (the 60 % target is assessed at the saturating batch),
    on rotating inputs; needed bytes (what the windowed kernel has to move) / the median launch duration / 8 TB/s; `traffic` = PMC bytes per
    launch of the same kernel at the same batch (a separate rocprofv3 run of tools/ep_sweep.py)."""
    rows = [r for r in sweep if r.get("chain")]
    if not rows:
        return None
    r = max(rows, key=lambda r_: r_["sequences_per_launch"])
    c = r["chain"]
    t, note = traffic_entry("saturating", f"chain_B{r['sequences_per_launch']}")
    return {"kernel": "epw_kernel (evaluate_posterior, windowed chain, probability rows; the throughput instance: 256 threads per sequence)",
            "sequences_per_launch": r["sequences_per_launch"], "avg_launch_ms": c["launch_ms"], "back_to_back_ms": c["back_to_back_ms"],
            "needed_bytes": c["hbm_bytes_needed_per_launch"], "achieved": c["achieved_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": c["frac"],
            "needed_bytes_definition": "(L + fresh) x W x 4 row bytes + T x k x 2 neighbour ids + W x 4 per LEVEL with a rejection (one drafter row per level: "
                                       "its later rejections hit L2), from the kernel's counters and a host replay of the walks",
            "traffic": None if t is None else t["hbm_bytes"], "traffic_over_needed": None if t is None else t["hbm_bytes"] / c["hbm_bytes_needed_per_launch"],
            "traffic_source": note, "traffic_note": _short(note, 120),
            "inputs": f"{r['rotation_sets']} rotating input sets, {r['rotation_bytes'] / 2**30:.1f} GiB between re-reads"}


def kernel_report(wl, evs, E0, E1, KT):
    """`roofline` (evaluate_posterior, the north-star kernel) and `kernels` (the others) from the HIP events of steps [E0, E1):
    algorithmic bytes per launch (SURVEY 8d formulas on the kernel's own counters, group 0's launches) / mean launch duration."""
    cfg = wl.cfg

    def mean_ms(n):
        return float(np.mean([e[n][0].elapsed_time(e[n][1]) for e in evs]))
    ep_ms = mean_ms("evaluate_posterior")
    # SURVEY 8d contract figure: L*V*4 + T*k*6 + R*(k+1)*4 + V*4 (+V*4), from the kernel's own counters
    contract_bytes = wl.ep_algorithmic_bytes(E0, E1, group=0) / KT      # events bracket group 0's launches
    ach = contract_bytes / (ep_ms * 1e-3) / 1e9
    if not wl.windowed:
        kname = "ep_kernel (evaluate_posterior)"
    elif wl.ep_nodes is not None:
        kname = "epn_kernel + epn_walk_kernel (evaluate_posterior, node-parallel)"
    elif getattr(wl, "fused_prepare", False):
        kname = "epw_kernel_fused<raw rows> (candidate assembly + evaluate_posterior + the tree_decoding post-process of the rows it visits, one launch)"
    elif wl.fused_o7:
        kname = "epw_kernel<raw rows> (evaluate_posterior + the tree_decoding post-process of the rows it visits)"
    else:
        kname = "epw_kernel (evaluate_posterior, windowed chain)"
    rl = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
          "sequences_per_launch": wl.Bg, "algorithmic_bytes_per_launch": contract_bytes, "avg_launch_ms": ep_ms,
          "algorithmic_bytes_definition": "SURVEY 8d: L*V*4 + T*k*6 + R*(k+1)*4 + V*4 (+V*4 if the last level had no rejection), "
                                          "V=65536, summed over the launch's sequences"}
    if wl.windowed:
        # what the windowed kernel actually has to move (rows are 8192-wide windows; gathers/zeroing/scan/bonus draw in LDS)
        wb = wl.ep_window_bytes(E0, E1, group=0) / KT
        # both fractions at the top level: `frac` prices the SURVEY 8d contract bytes (dense V-wide rows that the windowed design never
        # moves: an equivalent rate, it can exceed what the memory system does), `frac_needed` the bytes this kernel has to move -- the
        # physical HBM fraction
        rl["frac_contract"] = rl["frac"]
        rl["needed_bytes_per_launch"] = wb
        rl["achieved_needed"] = wb / (ep_ms * 1e-3) / 1e9
        rl["frac_needed"] = wb / (ep_ms * 1e-3) / 1e9 / 8000.0
        rl["windowed_kernel"] = {"hbm_bytes_needed_per_launch": wb, "achieved": wb / (ep_ms * 1e-3) / 1e9,
                                 "frac": wb / (ep_ms * 1e-3) / 1e9 / 8000.0,
                                 "definition": "(L+fresh)*W*4 + T*k*2 + R*W*4, W=8192 (DESIGN.md 4)" +
                                               ("; raw rows: a visited row is 2 x W bf16 = the same W*4 bytes" if wl.fused_o7 else "")}
    if wl.windowed:
        key = ("raw" if wl.fused_o7 else ("nodes" if wl.ep_nodes is not None else "chain")) + f"_B{wl.Bg}"
        t, note = traffic_entry("per_launch", key)      # PMC passes are separate rocprofv3 runs of the same kernel / launch size (tools/run/prof_default.sh)
        rl["traffic"] = None if t is None else t["hbm_bytes"]
        rl["traffic_source"] = note
        rl["traffic_note"] = _short(note, 120)
    ks = {}
    if "cfg_mask_topk" in evs[0]:
        o7_ms = mean_ms("cfg_mask_topk")
        o7_b = wl.o7_algorithmic_bytes(1, group=0) * ((wl.n_spec / wl.N) if wl.fused_o7 else 1.0)
        ks["cfg_mask_topk"] = {"avg_launch_ms": o7_ms, "algorithmic_bytes_per_launch": o7_b, "achieved": o7_b / (o7_ms * 1e-3) / 1e9,
                               "frac": o7_b / (o7_ms * 1e-3) / 1e9 / 8000.0,
                               "kernel": "prep_rows_kernel (candidate assembly + the %d most likely rows)" % wl.n_spec if wl.fused_o7 else "cfg_window_bf16_kernel"}
    if cfg.with_kv:
        kv_ms = mean_ms("kv_gather")
        kv_b = wl.kv_algorithmic_bytes(E0, E1, group=0) / KT
        kv_m = wl.kv_moved_bytes(E0, E1, group=0) / KT        # rows already in place are not copied
        # `achieved` counts the bytes the kernel really moves (rows already in place are skipped); the contract figure of
        # SURVEY 8d (every accepted row read + written) is kept beside it as an equivalent rate
        ks["kv_gather"] = {"avg_launch_ms": kv_ms, "algorithmic_bytes_per_launch": kv_m,
                           "achieved": kv_m / (kv_ms * 1e-3) / 1e9, "frac": kv_m / (kv_ms * 1e-3) / 1e9 / 8000.0,
                           "contract_bytes_per_launch": kv_b, "contract_equivalent_GBps": kv_b / (kv_ms * 1e-3) / 1e9,
                           "includes": "accepted-hidden copy (O10) in the same launch" if (wl.windowed and cfg.fuse_update) else None}
    return rl, ks


def event_names(wl):
    names = ("cfg_mask_topk", "evaluate_posterior", "kv_gather") if wl.cfg.with_kv else ("cfg_mask_topk", "evaluate_posterior")
    if getattr(wl, "fused_o7", False) and (not getattr(wl, "n_spec", 0) or getattr(wl, "fused_prepare", False)):          # (no launch of its own in front of the walk)
        names = tuple(n for n in names if n != "cfg_mask_topk")
    return names


def make_events(names, n, device):
    evs = [{k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for k in names} for _ in range(n)]
    for d in evs:                      # create the hipEvent_t handles (torch makes them on the first record)
        for e0, e1 in d.values():
            e0.record()
            e1.record()
    torch.cuda.synchronize(device)
    return evs


def per_kernel_run(device, base_cfg, steps=60, **over):
    """The same workload as ONE stream group with every stage its own launch (cfg_mask_topk for all rows, evaluate_posterior on
    probability rows, update_inference_inputs): each kernel against its own SURVEY 8d roofline at the full launch size."""
    import dataclasses
    from lantern_amd import harness as HN
    over.setdefault("n_seq", base_cfg.n_seq)
    cfg = dataclasses.replace(base_cfg, n_groups=1, fuse_o7=False, spec_rows=0, max_steps=max(base_cfg.pool_steps, 2 * steps + 20) + 8, **over)
    wl = HN.LuminaVerifyWorkload(cfg, device)
    wl.prime()
    for _ in range(10):
        wl.step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    evs = make_events(event_names(wl), steps, device)
    for i in range(steps):
        wl.step(evs[i], serial=True)
    torch.cuda.synchronize(device)
    wl.check_status(0, 2 * steps + 10)
    rl, ks = kernel_report(wl, evs, steps + 10, 2 * steps + 10, steps)
    toks = wl.accepted_tokens(10, 10 + steps)
    r = {"value": toks / dt, "unit": "accepted_tokens/s", "ms_per_step": 1e3 * dt / steps, "sequences_per_launch": wl.Bg,
         "evaluate_posterior_kernel": cfg.ep_kernel, "roofline": rl, "kernels": ks}
    wl.release_kv()
    del wl
    torch.cuda.empty_cache()
    return r


def step_latency(device, base_cfg, batches=(1, 8), steps=60):
    """Whole verify step (O6 -> O7 -> O8 -> O9 + O10, KV slabs of the bench geometry) at the reference's own batch sizes:
    wall-clock microseconds per step for both evaluate_posterior forms (BASELINE.md section 2 'Reported')."""
    from lantern_amd import harness as HN
    res = {}
    for Bs in batches:
        for kern in ("nodes", "chain"):
            cfg = HN.WorkloadConfig(n_seq=Bs, pool_steps=4, lantern_k=base_cfg.lantern_k, lantern_delta=base_cfg.lantern_delta, sigma=base_cfg.sigma,
                                    with_kv=base_cfg.with_kv, kv_smax=base_cfg.kv_smax, kv_pad_rows=base_cfg.kv_pad_rows, max_steps=steps + 32,
                                    seed_base=base_cfg.seed_base + 900, ep_kernel=kern, tree=base_cfg.tree)
            wl = HN.LuminaVerifyWorkload(cfg, device)
            wl.prime()
            for _ in range(10):
                wl.step()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                wl.step()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            wl.check_status(0, steps + 10)
            res[f"B{Bs}_{kern}"] = {"us_per_step": 1e6 * dt / steps, "accepted_tokens_per_s": wl.accepted_tokens(10, 10 + steps) / dt}
            del wl
            torch.cuda.empty_cache()
    return res


def dynamic_run(device, base_cfg, steps, n_seq, fuse_o7=False, groups=None, spec_rows=2):
    """The dynamic-tree half of C3 (eagle_version 2: top_k 10, depth 5, 59 nodes, a different tree per sequence and step) on the
    clock: O4 -> O6 -> O7 -> O8 -> O9 + O10 through lantern_verify_step, device-resident, same KV geometry and stream groups as the
    headline run; kernel_ms: the per-kernel pass (one ctypes call per kernel, events around group 0's launches)."""
    from lantern_amd import harness as HN
    if groups is None:
        groups = base_cfg.n_groups if n_seq % max(1, base_cfg.n_groups) == 0 else 1
    cfg = HN.DynamicConfig(n_seq=n_seq, lantern_k=base_cfg.lantern_k, lantern_delta=base_cfg.lantern_delta, with_kv=base_cfg.with_kv,
                           kv_smax=base_cfg.kv_smax, kv_pad_rows=base_cfg.kv_pad_rows, max_steps=2 * steps + 32, fuse_o7=fuse_o7, n_groups=groups, spec_rows=spec_rows,
                           commit_window=int(os.environ.get("LANTERN_BENCH_DYN_WINDOW", "0")))          # (turn-taking measured slower on per-sequence trees: 1.32 vs 1.50 M tokens/s)
    wl = HN.DynamicVerifyWorkload(cfg, device)
    for _ in range(10):
        wl.step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    names = tuple(n for n in event_names(wl) if not (fuse_o7 and n == "cfg_mask_topk"))
    KE = min(steps, 20)
    evs = [{n: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for n in names} for _ in range(KE)]
    for d in evs:
        for e0, e1 in d.values():
            e0.record(); e1.record()
    torch.cuda.synchronize(device)
    for i in range(KE):
        wl.step(evs[i])
    torch.cuda.synchronize(device)
    wl.check_status(0, steps + 10 + KE)
    toks = wl.accepted_tokens(10, 10 + steps)
    cnt = wl.log_cnt[10:10 + steps].float()
    r = {"workload": f"C3 dynamic tree (EAGLE-2): top_k {cfg.top_k}, depth {cfg.depth}, N={wl.N} nodes, {n_seq} sequences in {groups} stream groups", "value": toks / dt,
         "stream_groups": groups, "sequences_per_launch": wl.Bg,
         "tree_decoding_rows": (f"raw bf16 logits post-processed inside evaluate_posterior, {wl.n_spec} rows per sequence up front beside the tree build "
                                "(lantern_prepare_step)") if fuse_o7 else "cfg_mask_topk over all N rows",
         "unit": "accepted_tokens/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "mean_accept_length": toks / (steps * n_seq),
         "per_step": {"levels": float(cnt[..., 0].mean()), "tried": float(cnt[..., 1].mean()), "rejected": float(cnt[..., 2].mean())},
         "kernel_ms": {n: float(np.mean([e[n][0].elapsed_time(e[n][1]) for e in evs])) for n in names}}
    wl.release_kv()
    del wl
    torch.cuda.empty_cache()
    return r


def other_configs(device, base_cfg, steps, n_seq, only=None):
    """BASELINE configs 2 and 4 on the clock (same GPU, after the headline run; rank 0, N = 1).
    C2  LlamaGen + EAGLE, standard (non-relaxed) verify: V = 16384 (window = vocabulary), dynamic EAGLE-2 tree N = 59 (top_k 10, depth 4),
        lantern off, HF processors T = 1 / top_k 2000, LlamaGen-B KV geometry (12 layers x 12 heads x 64, 2 slabs per sequence);
        O4 + O6 + the root's and node 1's rows -> O8 on the raw cond / uncond rows (CFG, top-k and softmax for the other rows the walk visits) -> O9 + O10 per step
        (ea_model_llamagen.py:709-787, :930, :1137-1163); `all_rows_by_cfg_mask_topk`: the same with O7 over all 59 rows first.
    C4  Anole-7B 512x512, LANTERN++ static tree naive_extend_57 (N = 58, P = 33, D = 6), the reference's settings (lambda, k) in
        {(5, 10), (10, 5), (20, 5)} (run.sh:76-91): O6 + the 3 likeliest rows -> O8 on raw rows (chain kernel, neighbours zeroed in the drafter's
        row: ea_model_anole.py:597-669) -> O9 + O10, 7B KV geometry, the stream groups of the headline run; `all_rows_by_cfg_mask_topk`: O7 over all 58 rows first;
        plus the one-group per-kernel pass (all rows + chain on probability rows) for the roofline."""
    import dataclasses
    from lantern_amd import harness as HN
    res = {}
    c2_spec = int(os.environ.get("LANTERN_C2_SPEC_ROWS", "2"))       # tuning knob (diagnostic): rows prepared beside the tree build
    # ---- C2
    def c2_run(fuse):
        dc = HN.DynamicConfig(model="llamagen", n_seq=n_seq, depth=4, total_tokens=58, kv_layers=12, kv_heads=12, kv_dim=64, kv_smax=base_cfg.kv_smax,
                              kv_pad_rows=base_cfg.kv_pad_rows, with_kv=base_cfg.with_kv, max_steps=2 * steps + 32, plausible=8.0,
                              n_groups=_side_groups(n_seq), fuse_o7=fuse, spec_rows=c2_spec)
        wl = HN.DynamicVerifyWorkload(dc, device)
        for _ in range(10):
            wl.step()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        names = event_names(wl)
        KE = min(steps, 20)
        evs = make_events(names, KE, device)
        for i in range(KE):
            wl.step(evs[i])
        torch.cuda.synchronize(device)
        wl.check_status(0, steps + 10 + KE)
        toks = wl.accepted_tokens(10, 10 + steps)
        cnt = wl.log_cnt[10 + steps:10 + steps + KE, :wl.Bg].double()          # the events bracket group 0's launches
        ep_ms = float(np.mean([e["evaluate_posterior"][0].elapsed_time(e["evaluate_posterior"][1]) for e in evs]))
        # bytes evaluate_posterior has to move per launch: one W-wide row per visited level (+ the final row when it is a fresh softmax);
        # a raw row is cond + uncond bf16 = the same W * 4 bytes as an f32 probability row
        needed = float(((cnt[..., 0] + (1.0 - cnt[..., 4])) * wl.W * 4).sum() / KE)
        r = {"workload": f"C2: LlamaGen + EAGLE standard verify, V=16384, dynamic tree N={wl.N} (top_k 10, depth 4), lantern off, processors T=1/top_k=2000, "
                         f"{n_seq} sequences in {dc.n_groups} stream groups, KV [24,1,12,{dc.kv_smax}(+{dc.kv_pad_rows}),64] bf16 x2 per sequence",
             "tree_decoding_rows": ((f"raw bf16 logits post-processed inside evaluate_posterior (LANTERN_ROWS_RAW_BF16), {wl.n_spec} rows per sequence up front beside the "
                                     "tree build (lantern_prepare_step)" if wl.n_spec else
                                     "raw bf16 logits post-processed inside evaluate_posterior (LANTERN_ROWS_RAW_BF16, every row on demand)") if fuse
                                    else "cfg_mask_topk over all N rows"),
             "value": toks / dt, "unit": "accepted_tokens/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "mean_accept_length": toks / (steps * n_seq),
             "kernel_ms": {n: float(np.mean([e[n][0].elapsed_time(e[n][1]) for e in evs])) for n in names},
             "evaluate_posterior": {"avg_launch_ms": ep_ms, "sequences_per_launch": wl.Bg, "needed_bytes_per_launch": needed, "achieved_GBps": needed / (ep_ms * 1e-3) / 1e9,
                                    "frac_needed": needed / (ep_ms * 1e-3) / 1e9 / 8000.0}}
        wl.release_kv()
        del wl
        torch.cuda.empty_cache()
        return r
    res["C2"] = c2_run(True)
    res["C2"]["all_rows_by_cfg_mask_topk"] = {k_: v_ for k_, v_ in c2_run(False).items() if k_ in ("value", "ms_per_step", "kernel_ms", "mean_accept_length")}
    # evaluate_posterior alone at the saturating batch on LlamaGen's 16384-id window (the two-per-CU instance; round 6)
    try:
        sat = lg_batch_sweep([4096], device, iters=8, variants=(1,))[0]
        if "variants" in sat:
            v = sat["variants"]["1"][0]
            res["C2"]["evaluate_posterior_saturating"] = {"kernel": "epw_kernel<512,8,1,4,true,false,5,..> (LlamaGen window, two workgroups per CU, probability rows)",
                                                         "sequences_per_launch": 4096, "avg_launch_ms": 1e-3 * v["launch_us"], "needed_bytes_per_launch": v["needed_bytes_per_launch"],
                                                         "achieved_GBps": v["achieved_GBps"], "frac": v["frac"], "cfg_mask_topk_us": v["cfg_mask_topk_us"],
                                                         "cfg_mask_topk_frac": v["cfg_mask_topk_frac"]}
        else:
            res["C2"]["evaluate_posterior_saturating"] = sat
    except Exception as e:          # (an extra: never fails the line)
        res["C2"]["evaluate_posterior_saturating"] = {"error": repr(e)[:200]}
    if only == "C2":              # (tools/run/c2_check.sh)
        return res
    # ---- C4
    if base_cfg.with_kv:          # as many sequences as the slabs (2 x 2.1 GiB each at 4096 rows) + pools leave room for, in whole stream groups
        free, _ = torch.cuda.mem_get_info(device)
        per_seq = 2 * (2 * 32 * 32 * (base_cfg.kv_smax + base_cfg.kv_pad_rows) * 128 * 2) + base_cfg.pool_steps * 58 * (2 * 65536 * 2 + 2 * 4096 * 2 + 8192 * 4)
        n_seq = max(4, min(n_seq, int((free - (28 << 30)) // per_seq)))
        n_seq -= n_seq % 4
    groups = _side_groups(n_seq)
    res["C4"] = []
    for lam, kk in ((5.0, 10), (10.0, 5), (20.0, 5)):
        over = dict(model="anole", tree="naive_extend_57", lantern_k=kk, lantern_delta=lam, fuse_o7=True, spec_rows=3, ep_kernel="chain",
                    seed_base=4000)
        grouped = side_run(device, base_cfg, steps, n_groups=groups, n_seq=n_seq, **dict(over, fused_prepare=bool(base_cfg.fused_prepare)))
        all_rows = side_run(device, base_cfg, steps, n_groups=groups, n_seq=n_seq, **dict(over, fuse_o7=False, spec_rows=0))
        one = per_kernel_run(device, base_cfg, min(steps, 40), n_seq=n_seq, **{k_: v_ for k_, v_ in over.items() if k_ not in ("fuse_o7", "spec_rows")})
        rl = one["roofline"]
        res["C4"].append({"workload": f"C4: Anole-7B 512x512 LANTERN++ static tree naive_extend_57 (N=58,P=33,D=6), lambda={lam:g}, k={kk}, {n_seq} sequences in "
                                      f"{groups} stream groups, 7B KV geometry",
                          "lantern_delta": lam, "lantern_k": kk, "value": grouped["value"], "unit": "accepted_tokens/s", "ms_per_step": grouped["ms_per_step"],
                          "mean_accept_length": grouped["mean_accept_length"], "one_group_ms_per_step": one["ms_per_step"],
                          "tree_decoding_rows": "raw bf16 logits post-processed inside evaluate_posterior, 3 rows per sequence up front with the candidate assembly",
                          "all_rows_by_cfg_mask_topk": {"value": all_rows["value"], "ms_per_step": all_rows["ms_per_step"]},
                          "evaluate_posterior": {k_: rl.get(k_) for k_ in ("kernel", "avg_launch_ms", "achieved", "frac", "needed_bytes_per_launch", "frac_needed", "unit")},
                          "kernels": one["kernels"]})
    return res


def _side_groups(n_seq: int) -> int:
    """Stream groups of the side configurations: the headline's four when the sequences split evenly, else three, two, one."""
    return next(g for g in (4, 3, 2, 1) if n_seq % g == 0)


def side_run(device, base_cfg, steps, **over):
    """A second, shorter timed run of the same workload with some knobs changed (stream groups, tree, ...): whole-job rate."""
    import dataclasses
    from lantern_amd import harness as HN
    cfg = dataclasses.replace(base_cfg, max_steps=max(base_cfg.pool_steps, steps + 20) + 8, **over)
    wl = HN.LuminaVerifyWorkload(cfg, device)
    wl.prime()
    for _ in range(10):
        wl.step()
    torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
    dt = time.perf_counter() - t0
    wl.check_status(0, steps + 10)
    toks = wl.accepted_tokens(10, 10 + steps)
    r = {"value": toks / dt, "unit": "accepted_tokens/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
         "mean_accept_length": toks / (steps * cfg.n_seq), "sequences_per_launch": wl.Bg}
    wl.release_kv()
    del wl
    torch.cuda.empty_cache()
    return r


def drafter_layer_run():
    """SURVEY 8f-2 beside the path: one drafting call of the drafter's decoder layer at 7B size (2 x 10 rows against 1200 cached positions: stream-K
    GEMMs on packed weights, lantern_tree_attention, in-place cache), timed by tools/layer_bench.py in a child process (its own 1 GB of weights)."""
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "layer_bench.py")
    try:
        r = subprocess.run([sys.executable, tool, "10", "1200"], env=dict(os.environ, LAYER_BENCH_ONLY="tree"), capture_output=True, text=True, timeout=240)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode or not lines:
            return {"error": (r.stderr or r.stdout)[-300:]}
        d = json.loads(lines[-1])
        return {"workload": "drafter decoder layer, one drafting call: " + d["shape"], "us_per_call": d["hip_tree_attention_inplace_cache_us"],
                "weight_bytes": d["weight_bytes"], "weight_stream_GBps": d["weight_stream_GBps"], "frac_hbm_peak": d["weight_stream_GBps"] / 8000.0}
    except Exception as e:          # an extra object: never the reason the line is missing
        return {"error": repr(e)[:300]}


def _tool_json(tool, args, timeout=300, env=None):
    """Last JSON line a tools/ script prints, run in a child process (its own weights / pools), or {"error": ...}."""
    import subprocess
    try:
        # (the child runs as a user of the package would: without the hardware-queue / polling settings this process took for its own stream groups --
        # GPU_MAX_HW_QUEUES=8 alone costs the mirror's one-stream step 7-40 us, measured)
        child_env = {k: v for k, v in os.environ.items() if k not in _ENV_SET_HERE}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args], env=dict(child_env, **(env or {})),
                           capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode or not lines:
            return {"error": (r.stderr or r.stdout)[-300:]}
        return json.loads(lines[-1])
    except Exception as e:          # an extra object: never the reason the line is missing
        return {"error": repr(e)[:300]}


def mirror_generate_run():
    """The drop-in API on the clock (tools/mirror_bench.py): EaLumina_mGPT.generate -- what generate_images.py:240 reaches through the solver -- with
    stand-in target / drafter forwards, full vocabulary, default tree, 7B KV geometry: microseconds per verify step of the mirror's own loop body."""
    return _tool_json("mirror_bench.py", [500])


def drafter_cycle_run():
    """One drafting cycle of the EAGLE-2 drafter at model size (tools/draft_bench.py): prefill of the accepted tokens + `depth` tree steps, each ONE
    lantern_draft_depth call (input stage, decoder layer, fused head expansion, next-depth inputs), + the tree finalisation; wall microseconds."""
    return {m: _tool_json("draft_bench.py", [m, 1200, 30]) for m in ("lumina", "anole", "llamagen", "lumina_static", "anole_static", "llamagen_static")}


def plan_sequences(total_seqs: int, seqs_per_gpu: int, world: int, groups: int):
    """(sequences per rank, stream groups, scaling).  --total-seqs T: T / world sequences per rank (T must divide: every prompt of the
    reference's batch is generated exactly once), "strong"; else --seqs-per-gpu on every rank, "weak".  The group count is the largest
    one <= the requested that divides the rank's sequences -- no sequence is dropped to make the groups equal."""
    if total_seqs > 0:
        if total_seqs % world:
            raise SystemExit(f"bench.py: --total-seqs {total_seqs} does not split evenly over {world} ranks")
        n, scaling = total_seqs // world, "strong"
    else:
        n, scaling = seqs_per_gpu, "weak"
    if n < 1:
        raise SystemExit("bench.py: no sequence per rank")
    g = max(1, min(groups, n))
    while n % g:
        g -= 1
    return n, g, scaling



# BASELINE.json configs[4] / BASELINE.md section 2: "64 sequences split evenly".  (LANTERN_BENCH_C5_TOTAL: test knob for the one-device rehearsal of the
# N > 1 control flow, where every rank's share has to fit the ONE device beside the other ranks'.)
C5_TOTAL = int(os.environ.get("LANTERN_BENCH_C5_TOTAL", "64"))


def c5_strong_plan(world: int, groups: int):
    """(sequences per rank, stream groups, evaluate_posterior form) of the strong-scaling leg, or None when 64 does not split over `world` ranks."""
    if world < 1 or C5_TOTAL % world:
        return None
    n = C5_TOTAL // world
    g = max(1, min(groups if n > 16 else min(groups, 2), n))
    while n % g:
        g -= 1
    return n, g, ("nodes" if n <= 16 else "chain")


def c5_strong_run(args, base_cfg, world, rank, device, dist, group, red_device):
    """BASELINE's OWN scaling form on the clock (C5; the reference's run.sh:76-91 + generate_images.py:185-192 `--slice`): 64 sequences in all, 64 / N per
    rank, same kernels / KV geometry / pools as the headline, its own warm-up and barrier-bracketed K-step timed region, MAX over ranks.  Every rank calls
    this (the barrier is collective); returns the report on every rank."""
    import dataclasses
    from lantern_amd import harness as HN
    from lantern_amd.sharding import reduce_timing
    plan = c5_strong_plan(world, args.groups if args.groups > 0 else 4)
    if plan is None:
        return {"skipped": f"{C5_TOTAL} sequences do not split evenly over {world} ranks"}
    n, g, ep = plan
    fuse = ep == "chain" and base_cfg.fuse_o7
    K, W = args.steps, args.warmup
    cfg = dataclasses.replace(base_cfg, n_seq=n, n_groups=g, ep_kernel=ep, fuse_o7=fuse, spec_rows=(base_cfg.spec_rows if fuse else 0),
                              seed_base=base_cfg.seed_base, max_steps=max(base_cfg.pool_steps, K + W, 80) + 8)
    wl = HN.LuminaVerifyWorkload(cfg, device, rank=rank)
    wl.prime(0.0)
    for _ in range(W):
        wl.step()

    def barrier():
        if dist is not None:
            dist.barrier(group=group)
        torch.cuda.synchronize(device)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        wl.step()
    barrier()
    dt = time.perf_counter() - t0
    wl.check_status(0, W + K)
    toks = float(wl.accepted_tokens(W, W + K))
    dt_all, toks_all = reduce_timing(dist, dt, toks, device=red_device or device, group=group)
    wl.release_kv()
    del wl
    torch.cuda.empty_cache()
    return {"value": toks_all / dt_all, "unit": "accepted_tokens/s", "scaling": "strong", "ms_per_step": 1e3 * dt_all / K, "steps": K, "warmup": W,
            "total_sequences": C5_TOTAL, "sequences_per_rank": n, "stream_groups": g, "evaluate_posterior_kernel": ep, "n_gpus": world,
            "mean_accept_length": toks_all / (K * C5_TOTAL),
            "workload": f"C5: {C5_TOTAL} sequences in all, {n} per GPU (run.sh:76-91), no collective on the accept path"}


# ---------------------------------------------------------------------------- the line the driver parses

HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "mean_accept_length", "per_step", "ranks_seen", "backend", "backend_note", "tokens", "sequences_per_rank", "groups", "c5_strong", "host_waits")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
                 "sequences_per_launch", "needed_bytes_per_launch", "frac_needed", "traffic_note")
SATURATING_KEYS = ("kernel", "sequences_per_launch", "avg_launch_ms", "needed_bytes", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_needed",
                   "traffic_note", "inputs")
CPU_KEYS = ("value", "unit", "cores", "host_cores", "usable_cores", "kind", "sample", "matches_gpu_token_stream", "mismatches")


def _short(v, n=220):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def _pick(d, keys, n=220):
    return {k: _short(d[k], n) for k in keys if isinstance(d, dict) and k in d}


def compact_line(out: dict) -> dict:
    """The headline object for stdout: the contract's keys, `roofline` (+ `saturating`), `cpu_baseline`, the per-kernel one-liners and a
    few scalars of the extra runs -- every string clipped, < 4 KB in all (the driver keeps an 8 KB tail of stdout and could no longer parse
    round 4's 21 KB line).  Everything else lives in the --extras-out file / on stderr."""
    c = _pick(out, HEADLINE_KEYS, 700)
    if isinstance(c.get("config"), dict):
        c["config"] = {k: _short(v, 200 if k == "workload" else 150) for k, v in c["config"].items() if k not in ("side_stream_for_O6_O10", "host_waits", "launch", "kv_cache", "drafter_sigma")}
    if "roofline" in out:
        rl = _pick(out["roofline"], ROOFLINE_KEYS)
        if isinstance(out["roofline"].get("saturating"), dict):
            rl["saturating"] = _pick(out["roofline"]["saturating"], SATURATING_KEYS)
        for blk in (rl, rl.get("saturating") or {}):          # (the notes name the file; the full text is in the extras file)
            if isinstance(blk.get("traffic_note"), str):
                blk["traffic_note"] = _short(blk["traffic_note"], 110)
        c["roofline"] = rl
    if "kernels" in out:
        c["kernels"] = {k: _pick(v, ("kernel", "avg_launch_ms", "algorithmic_bytes_per_launch", "achieved", "frac"), 90) for k, v in out["kernels"].items()}
    if "cpu_baseline" in out:
        c["cpu_baseline"] = _pick(out["cpu_baseline"], CPU_KEYS)
    ex = {}
    mg = out.get("mirror_generate")
    if isinstance(mg, dict) and "us_per_verify_step" in mg:
        ex["mirror_generate_us_per_verify_step"] = mg["us_per_verify_step"]
    dc = out.get("drafter_cycle")
    if isinstance(dc, dict):
        ex["drafter_us_per_depth_wall"] = {k: v.get("us_per_depth_wall") for k, v in dc.items() if isinstance(v, dict)}
        ex["drafter_us_per_cycle_wall"] = {k: v.get("us_per_cycle_wall") for k, v in dc.items() if isinstance(v, dict)}
    if isinstance(out.get("lambda_mode"), dict):
        ex["lambda_mode_value"] = out["lambda_mode"].get("value")
    mp = out.get("merged_prepare_harness_only")
    if isinstance(mp, dict):
        ex["merged_prepare_harness_only_us_per_step"] = 1e3 * mp.get("ms_per_step", 0.0)
    if isinstance(out.get("dynamic_tree"), dict):
        ex["dynamic_tree_value"] = out["dynamic_tree"].get("value")
    cf = out.get("configs")
    if isinstance(cf, dict):
        if isinstance(cf.get("C2"), dict):
            ex["C2_value"] = cf["C2"].get("value")
        if isinstance(cf.get("C4"), list) and cf["C4"]:
            ex["C4_values"] = [r_.get("value") for r_ in cf["C4"] if isinstance(r_, dict)]
    sl = out.get("step_latency_us")
    if isinstance(sl, dict):
        ex["step_latency_us"] = {k: v.get("us_per_step") for k, v in sl.items() if isinstance(v, dict)}
    # evaluate_posterior by itself at the full 64 sequences per launch (the per-kernel single-group pass: probability rows, every stage its own launch):
    # the headline's `roofline` is the same kernel family at 16 per launch, where a launch is as long as its slowest chain whatever it holds
    pk = out.get("per_kernel_single_group")
    if isinstance(pk, dict) and isinstance(pk.get("chain"), dict) and isinstance(pk["chain"].get("roofline"), dict):
        r_ = pk["chain"]["roofline"]
        ex["evaluate_posterior_64_per_launch"] = {"avg_launch_ms": r_.get("avg_launch_ms"), "frac": r_.get("frac"), "frac_needed": r_.get("frac_needed")}
    if ex:
        c["extras"] = ex
    if out.get("extras_file"):
        c["extras_file"] = out["extras_file"]

    def rnd(o):          # six significant digits are plenty for everything but the headline pair (the --extras-out file keeps full precision)
        if isinstance(o, float):
            return float(f"{o:.6g}")
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return o
    keep = {k: c[k] for k in ("value", "ms_per_step") if k in c}
    c = rnd(c)
    c.update(keep)
    return c


def emit(out: dict, args) -> None:
    """Full report -> --extras-out and stderr (one line); compact headline -> the ONE stdout line, printed last."""
    path = getattr(args, "extras_out", "") or ""
    if path:
        try:
            with open(path, "w") as f:
                json.dump(out, f)
            out["extras_file"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError as e:
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    print("bench.py full report: " + json.dumps(out), file=sys.stderr)
    sys.stderr.flush()
    c = compact_line(out)
    line = json.dumps(c)
    for k in ("extras", "kernels", "per_step", "backend_note"):          # never again an unparseable line: under 6 KB whatever the report holds (the driver keeps
        if len(line) < 6144:                                               # an 8 KB tail) -- optional blocks go first, the contract's keys, roofline and cpu_baseline stay
            break
        c.pop(k, None)
        line = json.dumps(c)
    print(line)
    sys.stdout.flush()


# ---------------------------------------------------------------------------- N > 1: the process group

def open_process_group(world, rank, local_rank, want="auto", one_device=False):
    """The group the contract's barrier and the three timing scalars run on -- the accept path itself has NO collective.
    Every rank first joins a gloo group (host tensors; it cannot fail for GPU reasons).  Unless `want` is gloo, the ranks then try to bring
    up an RCCL ("nccl") group -- creation + one all-reduce of a device scalar, in a helper thread bounded by LANTERN_NCCL_PROBE_TIMEOUT seconds
    -- and agree over gloo (MIN of the ok flags): RCCL only if it came up on EVERY rank, else gloo for all.  This happens before any kernel
    of the workload is launched, so an RCCL that cannot initialise on a node (pinned ranks that do not see each other's device, an IPC
    restriction, ...) costs the fallback, not the run.  Returns (torch.distributed, {"group", "backend", "device", "note"})."""
    import datetime
    import threading
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")      # a stuck probe must not let the watchdog end the process
    # gloo's C++ side announces its connections on fd 1: keep stdout for the one JSON line (fd 1 -> fd 2 while the group comes up)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
        dist.barrier()
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    pg = {"group": None, "backend": "gloo", "device": torch.device("cpu"), "note": None}
    if want == "gloo" or one_device:
        pg["note"] = "gloo requested" if want == "gloo" else "one-device rehearsal: gloo"
        return dist, pg
    res = {}
    limit = float(os.environ.get("LANTERN_NCCL_PROBE_TIMEOUT", "120"))

    def probe():
        try:
            if os.environ.get("LANTERN_BENCH_FAIL_NCCL") == "1":          # test knob: an RCCL that does not come up
                raise RuntimeError("LANTERN_BENCH_FAIL_NCCL=1")
            dev = torch.device("cuda", local_rank)
            torch.cuda.set_device(dev)
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=limit), device_id=dev)
            t = torch.ones(1, device=dev)
            dist.all_reduce(t, group=g)
            torch.cuda.synchronize(dev)
            if int(t.item()) != world:
                raise RuntimeError(f"RCCL all-reduce returned {t.item()} for {world} ranks")
            res["group"] = g
        except BaseException as e:          # noqa: BLE001 -- whatever the reason, the answer is gloo
            res["err"] = repr(e)[:200]
    th = threading.Thread(target=probe, daemon=True)
    th.start()
    th.join(limit + 30.0)
    ok = torch.tensor([1 if "group" in res else 0], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok[0]) == 1:
        pg.update(group=res["group"], backend="nccl", device=torch.device("cuda", local_rank))
    else:
        why = res.get("err") or ("RCCL bring-up did not finish in time on this rank" if "group" not in res else "RCCL failed on another rank")
        pg["note"] = f"RCCL group not usable ({why}): barrier and timing scalars over gloo"
        pg["hung_probe"] = th.is_alive()
        if want == "nccl":
            raise SystemExit(f"bench.py: --dist-backend nccl: {pg['note']}")
    return dist, pg


def count_ranks(dist, pg) -> int:
    """Ranks that reached the end of the timed region (SUM of ones over the group): the line's `ranks_seen`."""
    if dist is None:
        return 1
    t = torch.ones(1, dtype=torch.float64, device=pg["device"])
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg["group"])
    return int(round(float(t[0])))


def close_process_group(dist, pg) -> None:
    if pg.get("hung_probe"):
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)                        # a helper thread is still stuck inside RCCL: do not wait for it in the interpreter's teardown
    dist.destroy_process_group()


def cap_groups(args, world: int) -> int:
    """A small per-rank share (C5: 64 prompts over 8 GPUs = 8 per GPU) runs the node-parallel kernels in at most two stream groups
    (profiles/r04_c5_share.json); the default four are for a full GPU of 64 sequences."""
    share = (args.total_seqs // max(world, 1)) if args.total_seqs > 0 else args.seqs_per_gpu
    return min(args.groups, 2) if share <= 16 else args.groups


# ---------------------------------------------------------------------------- N > 1: self-launch

def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N rank processes (one per GPU, LOCAL_RANK = device
    index, RCCL rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU, relay rank 0's JSON line and
    fail if any rank fails.  Same layout as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` and as
    the reference's one-process-per-GPU driver (run.sh:76-91).  The parent never initialises HIP and never exec()s."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if os.environ.get("LANTERN_BENCH_NO_PIN") != "1" and os.environ.get("LANTERN_BENCH_STUB") != "1":
            # one process per GPU, pinned the way the reference's driver pins them (run.sh:76-91: CUDA_VISIBLE_DEVICES=$device):
            # the rank sees exactly its own device, as index 0
            vis = os.environ.get("HIP_VISIBLE_DEVICES")
            devs = [d for d in vis.split(",") if d] if vis else [str(i) for i in range(n)]
            if len(devs) >= n:
                env["HIP_VISIBLE_DEVICES"] = devs[r]
                env["LANTERN_BENCH_DEVICE_INDEX"] = "0"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    deadline = time.time() + float(os.environ.get("LANTERN_BENCH_SPAWN_TIMEOUT", "1500"))
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            failed = True                  # one rank died (or the job hung): the others would wait in the rendezvous for ever
            break
        time.sleep(0.1)
    for p in procs:
        if p.poll() is None:
            p.kill()                       # the exact PIDs this function started
    codes = [p.wait() for p in procs]
    rd.join(timeout=10)
    out0 = buf[0] if buf else ""
    if failed or any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    line = [l for l in out0.splitlines() if l.startswith("{")]
    rest = [l for l in out0.splitlines() if not l.startswith("{")]
    if rest:
        print("\n".join(rest), file=sys.stderr)          # stdout carries the one JSON line and nothing else
    if not line or json.loads(line[-1]).get("n_gpus") != n:
        print(f"bench.py: rank 0 did not report n_gpus={n}", file=sys.stderr)
        return 1
    sys.stdout.write(line[-1] + "\n")
    sys.stdout.flush()
    return 0


def stub_rank(args, world, rank):
    """LANTERN_BENCH_STUB=1 (tests/test_multiproc_cpu.py, no GPU in the build container): the rank body with the kernels
    replaced by a sleep, so that the self-launch, the gloo rendezvous, the barrier-bracketed timing, the MAX / SUM
    reductions and rank 0's JSON line run on CPU.  The line says `"data": "stub"`: it is not a measurement."""
    from lantern_amd.sharding import reduce_timing
    dist, pg = None, None
    if world > 1:
        dist, pg = open_process_group(world, rank, 0, args.dist_backend, one_device=False)     # no GPU here: "auto" must land on gloo
    n_seq, _groups, scaling = plan_sequences(args.total_seqs, args.seqs_per_gpu, world, cap_groups(args, world))
    K, per_step = args.steps, n_seq * 2
    if world > 1:
        dist.barrier(group=pg["group"])
    t0 = time.perf_counter()
    for _ in range(K):
        time.sleep(0.001)
    if world > 1:
        dist.barrier(group=pg["group"])
    dt = time.perf_counter() - t0
    dt_all, tokens_all = reduce_timing(dist if world > 1 else None, dt, float(K * per_step), group=pg["group"] if pg else None)
    ranks_seen = count_ranks(dist, pg)
    c5 = None
    if world > 1 and args.total_seqs == 0:          # the second leg of an N > 1 run: 64 sequences in all (same control flow as main())
        plan = c5_strong_plan(world, cap_groups(args, world))
        if plan is None:
            c5 = {"skipped": f"{C5_TOTAL} sequences do not split evenly over {world} ranks"}
        else:
            dist.barrier(group=pg["group"])
            t1 = time.perf_counter()
            for _ in range(K):
                time.sleep(0.001)
            dist.barrier(group=pg["group"])
            d5, t5 = reduce_timing(dist, time.perf_counter() - t1, float(K * plan[0] * 2), group=pg["group"])
            c5 = {"value": t5 / d5, "unit": "accepted_tokens/s", "scaling": "strong", "ms_per_step": 1e3 * d5 / K, "steps": K, "total_sequences": C5_TOTAL,
                  "sequences_per_rank": plan[0], "stream_groups": plan[1], "evaluate_posterior_kernel": plan[2], "n_gpus": world, "tokens": t5,
                  "ranks_seen": ranks_seen}
    if rank == 0:
        out = {"metric": "stub", "value": tokens_all / dt_all, "unit": "accepted_tokens/s", "n_gpus": world, "steps": K,
               "warmup": args.warmup, "ms_per_step": 1e3 * dt_all / K, "data": "stub", "tokens": tokens_all, "scaling": scaling,
               "sequences_per_rank": n_seq, "groups": _groups, "ranks_seen": ranks_seen, "backend": pg["backend"] if pg else None,
               "backend_note": pg["note"] if pg else None, **({"c5_strong": c5} if c5 is not None else {}),
               # a stand-in for the extras of a real run (tests/test_multiproc_cpu.py: they must stay off the stdout line)
               "roofline": {"bound": "hbm", "achieved": 0.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0, "traffic": None},
               "cpu_baseline": {"value": 0.0, "unit": "accepted_tokens/s", "cores": 1, "kind": "port", "sample": "stub"},
               "ep_batch_sweep": [{"sequences_per_launch": b, "pad": "x" * 512} for b in (1, 8, 64, 256, 512, 4096)],
               "configs": {"pad": ["y" * 256] * 40}}
        emit(out, args)
    if world > 1:
        dist.barrier(group=pg["group"])
        close_process_group(dist, pg)


# ------------------------------------------------------------------------------------ main

def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: every rank of the job must be started (or let "
                         "`python bench.py --gpus N` start them itself)")
    exit_code, stream_mismatch = 0, 0
    if os.environ.get("LANTERN_BENCH_STUB") == "1":
        return stub_rank(args, world, rank)
    # Test knob for a 1-GPU box: LANTERN_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 over gloo (RCCL refuses two ranks on one
    # device), so that the N > 1 control flow -- rank seeds, the MIN / MAX / SUM reductions, rank-0 output -- can be exercised.
    one_device = os.environ.get("LANTERN_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if "LANTERN_BENCH_DEVICE_INDEX" in os.environ:      # a rank pinned by spawn_ranks (HIP_VISIBLE_DEVICES = its own device)
        local_rank = int(os.environ["LANTERN_BENCH_DEVICE_INDEX"])
    dist, pg = None, None
    if world > 1 or os.environ.get("LANTERN_BENCH_FORCE_DIST") == "1":       # the env knob exercises the RCCL path on a 1-GPU box
        dist, pg = open_process_group(world, rank, local_rank, args.dist_backend, one_device)
        if pg["note"] and rank == 0:
            print("bench.py: " + pg["note"], file=sys.stderr)
    red_device = pg["device"] if pg else None                              # gloo reduces host tensors, RCCL device tensors
    group = pg["group"] if pg else None
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from lantern_amd import harness as HN
    from lantern_amd import _lib as _LL
    tuning_set = {}
    for kv in [x for x in args.tuning.split(",") if x]:
        name, _, val = kv.partition("=")
        _LL.set_tuning(name.strip(), int(val))
        tuning_set[name.strip()] = int(val)

    # Never ask for more resident sequences than this GPU can hold: KV slabs (2 per sequence) + pools + 16 GiB of head-room.
    # On the MI355X the default fits (309e9 bytes); a smaller or partly occupied device gets fewer sequences, not a failed run
    # (with more than one rank every rank takes the minimum so that the per-GPU work stays identical).
    args.groups = cap_groups(args, world)
    n_seq, groups, scaling = plan_sequences(args.total_seqs, args.seqs_per_gpu, world, args.groups)
    if groups != args.groups and rank == 0:
        print(f"bench.py: {n_seq} sequences per rank in {groups} stream groups (--groups {args.groups} does not divide them)", file=sys.stderr)
    args.groups = groups
    if args.ep == "auto":
        args.ep = "nodes" if n_seq <= 16 else "chain"          # (measured at 8 sequences per GPU in 1 / 2 groups; larger shares keep the chain on raw rows)
    if args.ep != "chain":
        args.fuse_o7 = False
    if not args.fuse_o7:
        args.spec_rows = 0
    if not args.no_kv:
        free, _tot = torch.cuda.mem_get_info(device)
        from lantern_amd import ops as _ops
        from lantern_amd.drafters import choices as _choices
        _tb = _ops.tree_static_build(getattr(_choices, args.tree))
        _N = len(_tb["tree_indices"])
        _R = int(((_tb["tree_indices"][1:] - 1) // 10).max()) + 1
        pool_step = _N * (2 * 65536 * 2 + 2 * 4096 * 2) + _R * (8192 * 4 + 10 * 12)       # cond + uncond logits, hidden, drafter rows
        per_seq = 2 * (2 * 32 * 32 * (args.kv_smax + 16) * 128 * 2) + args.pool_steps * pool_step
        fit = int((free - (16 << 30)) // per_seq)
        if dist is not None and dist.get_world_size() > 1:
            t = torch.tensor([fit], dtype=torch.int64, device=red_device or device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            fit = int(t[0])
        if fit < n_seq:
            if fit < max(1, args.groups):
                raise SystemExit(f"bench.py: {free / 2**30:.0f} GiB free on {device}: not even one sequence's KV slabs fit")
            if args.total_seqs > 0:
                raise SystemExit(f"bench.py: {free / 2**30:.0f} GiB free on {device}: the {n_seq} sequences of this rank's share of --total-seqs do not fit")
            n_seq, args.groups, _ = plan_sequences(0, fit, world, args.groups)
            if rank == 0:
                print(f"bench.py: {free / 2**30:.0f} GiB free: running {n_seq} sequences per rank in {args.groups} stream groups", file=sys.stderr)
    if args.fused_prepare < 0:          # the prepare stage inside the chain launch: 65.1-65.7 us per step against 67.7-68.3 with its own launch (alternating, two boxes)
        args.fused_prepare = 1 if (args.ep == "chain" and args.fuse_o7 and args.path == "window" and not args.python_launch and not args.graph and not args.side_stream) else 0
    if args.spec_rows < 0:
        args.spec_rows = 1 if args.fused_prepare else 3
    if args.commit_window < 0:
        args.commit_window = 1 if (args.groups == 4 and args.ep == "chain" and not args.no_kv and not args.python_launch and not args.graph) else 0
    cfg = HN.WorkloadConfig(n_seq=n_seq, pool_steps=args.pool_steps, tree=args.tree, lantern_k=args.lantern_k,
                            lantern_delta=args.lantern_delta, sigma=args.sigma, with_kv=not args.no_kv, kv_smax=args.kv_smax,
                            path=args.path, ep_kernel=args.ep, fuse_o7=args.fuse_o7, spec_rows=args.spec_rows, commit_window=args.commit_window, fused_prepare=bool(args.fused_prepare), native_step=not args.python_launch, use_graph=args.graph, n_groups=args.groups, side_stream=args.side_stream,
                            max_steps=max(args.pool_steps, args.steps + args.warmup + min(args.steps, 100), 80) + 8,
                            **({} if args.kv_pad_rows is None else {"kv_pad_rows": args.kv_pad_rows}))
    wl = HN.LuminaVerifyWorkload(cfg, device, rank=rank)

    def barrier():
        if dist is not None:
            dist.barrier(group=group)
        torch.cuda.synchronize(device)

    K, W = args.steps, args.warmup
    wl.prime(args.spin_up)   # setup: every pool slot launched once, state reset (a short --warmup must not leave first-touch costs in the timed
                             # loop), repeated for --spin-up seconds so that the GPU leaves its idle clocks before the W warm-up steps
    for _ in range(W):
        wl.step()
    names = event_names(wl)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        wl.step()
    barrier()
    dt = time.perf_counter() - t0
    # ---- per-kernel durations: the same loop continues eagerly with HIP events on the launch stream for the three
    # HBM-heavy kernels (recorded at kernel begin / end by the launch itself: lantern_profile_next_launch), kept out of the
    # timed region so that `value` carries no event overhead.  rocprofv3 of this command sees both passes.
    evs = None
    KT = min(K, 100)
    if not args.no_events:
        evs = make_events(names, KT, device)
        for i in range(KT):
            wl.step(evs[i], serial=True)     # groups one after the other on one stream: undisturbed kernel durations
        torch.cuda.synchronize(device)

    n_logged = W + K + (KT if evs else 0)
    wl.check_status(0, n_logged)
    tokens = wl.accepted_tokens(W, W + K)
    from lantern_amd.sharding import reduce_timing
    dt_all, tokens_all = reduce_timing(dist, dt, float(tokens), device=red_device or device, group=group)
    ranks_seen = count_ranks(dist, pg)

    out = None
    if rank == 0:
        alen = wl.log_alen[W:W + K].float() + 1
        cnt = wl.log_cnt[W:W + K].float()
        out = {
            "metric": "accepted image-tokens/sec (Lumina-mGPT-7B 768x768 LANTERN verify/accept loop)",
            "value": tokens_all / dt_all, "unit": "accepted_tokens/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * dt_all / K, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen, "backend": (pg["backend"] if pg else None),
            "backend_note": (pg["note"] if pg else None),
            "host_waits": "polling (HSA_ENABLE_INTERRUPT=0, set by bench.py unless the caller set it)" if os.environ.get("HSA_ENABLE_INTERRUPT") == "0" else "interrupts",
            "config": {"workload": ("C5: Lumina-mGPT-7B-768 LANTERN, %d-prompt batch sharded over %d GPU(s), no collective; " % (args.total_seqs, world) if args.total_seqs > 0 else "") +
                                   f"C3: Lumina-mGPT-7B-768 LANTERN relaxed accept, static tree {cfg.tree} (N={wl.N},P={wl.P},D={wl.D}), "
                                   "V=65536, K=8192, cfg=3.0, top_k=2000, sequential-CFG KV [64,1,32,%d,128] bf16 x2 per sequence (row stride %d)"
                                   % (cfg.kv_smax, cfg.kv_smax + cfg.kv_pad_rows),
                       "deviations_from_BASELINE": ("%d of BASELINE's 64 sequences per GPU (%d stream groups of %d), %d of BASELINE.md's 256 pool steps "
                                                    "(the KV slabs of 64 sequences take 276e9 of the 309e9 bytes; %d steps = %.1f GB of pools, far beyond L2 + "
                                                    "Infinity Cache)") % (cfg.n_seq, cfg.n_groups, wl.Bg, cfg.pool_steps, cfg.pool_steps,
                                                                          cfg.pool_steps * cfg.n_seq * (wl.N * (2 * 65536 * 2 + 2 * 4096 * 2) + wl.R * 8192 * 4) / 1e9),
                       "lantern_k": cfg.lantern_k, "lantern_delta": cfg.lantern_delta, "seqs_per_gpu": cfg.n_seq,
                       "total_sequences": cfg.n_seq * world, "pool_steps": cfg.pool_steps, "drafter_sigma": cfg.sigma,
                       "kv_cache": cfg.with_kv, "kernel_path": cfg.path, "launch": "hipGraph replay" if (cfg.use_graph and wl.graphs) else ("eager, one lantern_verify_step call per step" if wl._steps else "eager, one call per kernel"),
                       "evaluate_posterior_kernel": (cfg.ep_kernel if wl.windowed else "dense"),
                       "tree_decoding_rows": (("raw bf16 logits post-processed inside evaluate_posterior (LANTERN_ROWS_RAW_BF16), %d most likely rows per sequence "
                                               "up front with the candidate assembly (lantern_prepare_step)" % wl.n_spec) if getattr(wl, "fused_o7", False)
                                              else "every row post-processed by cfg_mask_topk before evaluate_posterior"),
                       "tuning": tuning_set or None, "commit_window": cfg.commit_window, "fused_prepare": bool(cfg.fused_prepare), "stream_groups": cfg.n_groups, "host_waits": "polling (HSA_ENABLE_INTERRUPT=0)" if os.environ.get("HSA_ENABLE_INTERRUPT") == "0" else "interrupts", "side_stream_for_O6_O10": cfg.side_stream, "sequences_per_launch": wl.Bg, "parallelism": f"dp{world} (independent sequences, no collective)"},
            "mean_accept_length": float(alen.mean()),
            "per_step": {"levels": float(cnt[..., 0].mean()), "tried": float(cnt[..., 1].mean()), "rejected": float(cnt[..., 2].mean())},
        }
        if evs:
            out["roofline"], out["kernels"] = kernel_report(wl, evs, W + K, W + K + KT, KT)
        # the CPU leg replays the run from step 0 (warm-up included): keep the logs before the extra runs below overwrite them
        gb = wl.log_best[:n_logged].cpu().numpy()
        ga = wl.log_alen[:n_logged].cpu().numpy()
        gt = wl.log_token[:n_logged].cpu().numpy()
    # ---- N > 1: BASELINE's own scaling form (C5: 64 sequences in ALL) beside the weak-scaling headline, in the same line.  Every rank takes part
    # (its barriers are collective); the weak run's KV slabs are released first (64 sequences per GPU fill the device).
    if world > 1 and args.total_seqs == 0 and not args.no_kv and os.environ.get("LANTERN_BENCH_NO_C5") != "1":
        torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
        wl.release_kv()
        c5 = c5_strong_run(args, cfg, world, rank, device, dist, group, red_device)
        if rank == 0:
            c5["ranks_seen"] = ranks_seen
            out["c5_strong"] = c5
    if rank == 0:
        if not args.no_extras and wl.windowed and world == 1:
            # LANTERN++ mode of the same workload (run B of BASELINE.md: lantern_delta = 5 -> tau = 4 * p(x)): same pools, same kernels
            KL = 60
            torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
            wl.set_lantern_delta(5.0)
            wl.reset_state()
            for _ in range(15):
                wl.step()
            torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
            t1 = time.perf_counter()
            for _ in range(KL):
                wl.step()
            torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
            dl = time.perf_counter() - t1
            wl.check_status(0, KL + 15)
            tl = wl.accepted_tokens(15, 15 + KL)
            out["lambda_mode"] = {"lantern_delta": 5.0, "value": tl / dl, "unit": "accepted_tokens/s", "ms_per_step": 1e3 * dl / KL, "steps": KL,
                                  "mean_accept_length": tl / (KL * cfg.n_seq), "note": "rank 0's sequences only"}
            wl.set_lantern_delta(args.lantern_delta)
            if cfg.fuse_o7 and cfg.spec_rows > 0 and cfg.with_kv and wl._steps and not wl.fused_prepare:          # (--fused-prepare 0 --spec-rows 3 measures it)
                # the round-5 headline form, kept as a HARNESS-ONLY extra on the same workload and streams: step s + 1's prepare stage inside step s's
                # commit launch -- only possible because the pools hold step s + 1's rows ahead of time; a real decode loop produces them after
                # commit(s)
                wl.cfg.merge_prepare = True
                wl.reset_state()
                for _ in range(15):
                    wl.step()
                torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
                t1 = time.perf_counter()
                for _ in range(KL):
                    wl.step()
                torch.cuda.synchronize(device)          # (device-wide; NOT join(): cross-stream event waits behind a deep queue slow its drain)
                dm = time.perf_counter() - t1
                wl.check_status(0, KL + 15)
                out["merged_prepare_harness_only"] = {"ms_per_step": 1e3 * dm / KL, "value": wl.accepted_tokens(15, 15 + KL) / dm, "steps": KL,
                                                      "note": "prepare_next: not a form a decode loop can run; never the headline"}
                wl.cfg.merge_prepare = False
        if (args.ep_sweep or not args.no_extras) and world == 1:
            wl.release_kv()      # the extra runs build their own workloads: give the memory back first
        if not args.no_extras and world == 1 and wl.windowed:
            out["per_kernel_single_group"] = {k: per_kernel_run(device, cfg, min(K, 60), ep_kernel=k, n_seq=(args.seqs_per_gpu if n_seq + args.groups > args.seqs_per_gpu else n_seq)) for k in ("chain", "nodes")}
            out["step_latency_us"] = step_latency(device, cfg)
            out["other_groupings"] = {f"groups_{g}": side_run(device, cfg, max(min(K, 100), 60), n_groups=g, commit_window=0, n_seq=args.seqs_per_gpu - args.seqs_per_gpu % g)
                                      for g in (1, 2) if g != cfg.n_groups}
        if not args.no_extras and world == 1 and wl.windowed:
            out["dynamic_tree"] = dynamic_run(device, cfg, min(K, 100), n_seq, fuse_o7=True)
            out["dynamic_tree"]["all_rows_by_cfg_mask_topk"] = {k: v for k, v in dynamic_run(device, cfg, min(K, 100), n_seq, fuse_o7=False).items()
                                                                if k in ("value", "ms_per_step", "kernel_ms")}
        if not args.no_extras and world == 1 and wl.windowed:
            out["configs"] = other_configs(device, cfg, min(K, 60), n_seq)
        if not args.no_extras and world == 1 and wl.windowed:
            out["drafter_layer"] = drafter_layer_run()
            out["drafter_cycle"] = drafter_cycle_run()
            out["mirror_generate"] = mirror_generate_run()
        if args.ep_sweep and world == 1:
            out["ep_batch_sweep"] = ep_batch_sweep([int(x) for x in args.ep_sweep.split(",") if x], device, cfg)
            sat = saturating_report(out["ep_batch_sweep"])
            if sat and "roofline" in out:
                out["roofline"]["saturating"] = sat
        if args.cpu_seconds > 0 and world == 1:      # the CPU baseline is reported at N = 1 only
            n_cpu = args.cpu_seqs or cfg.n_seq
            gpu_stream = [[(int(gb[i, b]), int(ga[i, b]), int(gt[i, b])) for b in range(cfg.n_seq)] for i in range(n_logged)]
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds, n_cpu, gpu_stream, python_budget_s=min(args.cpu_seconds, 8.0))
            if not out["cpu_baseline"]["matches_gpu_token_stream"]:
                stream_mismatch = out["cpu_baseline"]["mismatches"]
        emit(out, args)
        if stream_mismatch:          # the checker disagrees with the kernels: the line above is not a valid measurement
            print(f"bench.py: the CPU oracle's accepted-token stream differs from the GPU's in {stream_mismatch} (step, sequence) cells", file=sys.stderr)
            exit_code = 3
    if dist is not None:
        dist.barrier(group=group)
        if exit_code:
            sys.stdout.flush()
            sys.stderr.flush()
            if pg.get("hung_probe"):
                os._exit(exit_code)
        close_process_group(dist, pg)
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()

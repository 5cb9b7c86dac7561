"""CPU, world_size 2 (gloo): the N>1 layout -- disjoint sequence shards, no data-path collective, timing
reduced with MAX / tokens with SUM -- gives the same accepted-token total as one process running every
sequence.  The per-rank compute is the oracle at a tiny size (no GPU here)."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tokens_for(seq_ids):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases as CS
    import oracle
    from lantern_amd.sharding import sequence_seed
    tb = oracle.tree_static_build(CS.mc_sim_7b_63)
    bufs = dict(tree_indices=tb["tree_indices"], tree_position_ids=tb["tree_position_ids"], tree_attn_mask=tb["tree_attn_mask"],
                retrieve_indices=tb["retrieve_indices"])
    m = CS.MODELS["lumina"]
    N = len(tb["tree_indices"])
    ri = tb["retrieve_indices"].copy()
    ri[ri < 0] += N
    cfg = oracle.EpConfig(mode=oracle.MODE_STATIC_LUMINA, syntax_shortcut=True, tok_offset=4, img_lo=4, img_hi=m["img_hi"],
                          syntax=m["syntax"], lantern=True, k=50, delta=0.1)
    table = CS.build_table(m["K"])
    out = {}
    for sid in seq_ids:
        g = CS.gen_static(sequence_seed(9000, sid), "lumina", bufs, sigma=1.5)
        ssp = CS.ss_prob_from(g["orig_prob"], g["ss_token"])
        cand, cp, tc = oracle.gather_candidates(g["ss_token"], ssp, g["sample_token"], tb["tree_indices"], tb["retrieve_indices"])
        aux = oracle.StaticAux(cart_prob=cp, orig_prob=g["orig_prob"], op_off=g["op_off"], p_idx=tb["p_indices"], b_off=tb["b_off"],
                               b_idx=tb["b_idx"], tree_cand=tc)
        _, alen, _, _ = oracle.evaluate_posterior(cfg, g["node_logits"], ri.astype(np.int32), cand, g["uniforms"], table=table, aux=aux)
        out[sid] = alen + 1
    return out


def _worker(rank, world, port, per_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from lantern_amd.sharding import reduce_timing, sequence_ids
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sequence_ids(rank, world, per_rank)
    toks = _tokens_for(ids)
    dt, total = reduce_timing(dist, seconds=1.0 + rank, tokens=float(sum(toks.values())))
    q.put((rank, ids, toks, dt, total))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_equal_single_process():
    world, per_rank = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _tokens_for(range(world * per_rank))
    seen = {}
    for rank, ids, toks, dt, total in res:
        assert dt == 2.0                                   # MAX over ranks of (1.0, 2.0)
        assert total == float(sum(single.values()))        # SUM over ranks == one process, all sequences
        assert not (set(ids) & set(seen))                  # shards are disjoint
        seen.update(toks)
    assert seen == single


def test_prompt_slices_and_statistics_files(tmp_path):
    from lantern_amd import sharding as sh
    prompts = [f"p{i}" for i in range(10)]
    assert sh.slice_prompts(prompts, "2-5") == ["p2", "p3", "p4"] and sh.slice_prompts(prompts) == prompts
    for bad in ("5-2", "3-3", "a-b", "1_2", "-1-4"):
        with pytest.raises(ValueError):
            sh.parse_slice(bad)
    slices = sh.rank_slices(10, 4)
    assert slices == ["0-3", "3-6", "6-9", "9-10"]
    assert sum((sh.slice_prompts(prompts, s) for s in slices), []) == prompts
    assert sh.rank_slices(2, 8) == ["0-1", "1-2"]
    paths = []
    for s in slices:
        a, b = sh.parse_slice(s)
        entries = {f"prompt_{i}": sh.statistics_entry(prompts[i], 2.0 + i, 0.5 * i) for i in range(a, b)}
        paths.append(sh.write_global_statistics(str(tmp_path), entries, a, b))
    assert os.path.basename(paths[1]) == "global_statistics_3_6.json"
    with open(paths[1]) as f:
        one = json.load(f)
    assert list(one) == ["prompt_3", "prompt_4", "prompt_5"] and set(one["prompt_3"]) == {"prompt", "step_compression", "latency"}
    m = sh.merge_global_statistics(paths)
    assert m["prompts"] == 10 and abs(m["mean_step_compression"] - 6.5) < 1e-12 and abs(m["mean_latency"] - 2.25) < 1e-12


@pytest.mark.timeout(300)
def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two rank processes, gloo rendezvous on 127.0.0.1, rank 0's line says
    n_gpus 2 and carries the SUM of both ranks' tokens (LANTERN_BENCH_STUB=1: the kernels are replaced by a sleep -- there is
    no GPU here; the launch / reduction control flow is what is under test).  A WORLD_SIZE that disagrees with --gpus fails."""
    import subprocess
    env = dict(os.environ, LANTERN_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--seqs-per-gpu", "8"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["tokens"] == 2 * 4 * 8 * 2 and line["steps"] == 4
    # --dist-backend auto: the RCCL bring-up fails here (no GPU) on every rank, the ranks agree on gloo BEFORE the timed loop, and say so
    assert line["backend"] == "gloo" and line["ranks_seen"] == 2 and "RCCL group not usable" in line["backend_note"]
    # both scaling forms in the one line: `value` = weak (--seqs-per-gpu on every rank), `c5_strong` = BASELINE's 64 in all
    c5 = line["c5_strong"]
    assert line["scaling"] == "weak" and c5["scaling"] == "strong" and c5["total_sequences"] == 64 and c5["sequences_per_rank"] == 32
    assert c5["ranks_seen"] == 2 and c5["tokens"] == 2 * 4 * 32 * 2 and c5["value"] > 0 and c5["evaluate_posterior_kernel"] == "chain"
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=dict(env, WORLD_SIZE="1"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_sequence_plan_drops_nothing():
    """bench.py's (sequences per rank, stream groups): --total-seqs splits the batch exactly (BASELINE config 5: 64 prompts over 8 GPUs =
    8 per GPU), the group count is lowered to a divisor instead of sequences being dropped (`--seqs-per-gpu 8 --groups 3` used to
    run 6 silently)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.plan_sequences(64, 63, 8, 3) == (8, 2, "strong")
    assert bench.plan_sequences(0, 63, 1, 3) == (63, 3, "weak")
    assert bench.plan_sequences(0, 8, 1, 3) == (8, 2, "weak")
    assert bench.plan_sequences(0, 64, 4, 3) == (64, 2, "weak")
    assert bench.plan_sequences(64, 63, 1, 4) == (64, 4, "strong")
    with pytest.raises(SystemExit):
        bench.plan_sequences(64, 63, 7, 3)
    for total, world in ((64, 8), (64, 4), (64, 2), (64, 1)):
        n, g, _ = bench.plan_sequences(total, 63, world, 3)
        assert n * world == total and n % g == 0


@pytest.mark.timeout(600)
def test_bench_c5_eight_stub_ranks():
    """BASELINE config 5's launch shape on CPU: `python bench.py --gpus 8 --total-seqs 64` starts eight rank processes (gloo, stub
    kernels), every rank runs 64 / 8 = 8 sequences, rank 0's line carries all 64 (strong scaling)."""
    import subprocess
    env = dict(os.environ, LANTERN_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--total-seqs", "64", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["sequences_per_rank"] == 8 and line["groups"] == 2
    assert line["tokens"] == 8 * 3 * 8 * 2          # ranks x steps x sequences per rank x the stub's 2 tokens
    assert "c5_strong" not in line                  # (--total-seqs IS the strong form: no second leg)


@pytest.mark.timeout(600)
def test_bench_eight_stub_ranks_print_both_scaling_forms():
    """The driver's N = 8 invocation (`bench.py --gpus 8 --steps K --warmup W`, nothing else): `value` is the weak-scaling figure (64 per GPU) and
    `c5_strong` BASELINE's form (64 in all = 8 per GPU, node-parallel evaluate_posterior, <= 2 stream groups), each from its own timed region."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                       env=_stub_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["sequences_per_rank"] == 64 and line["ranks_seen"] == 8
    c5 = line["c5_strong"]
    assert c5["scaling"] == "strong" and c5["total_sequences"] == 64 and c5["sequences_per_rank"] == 8 and c5["stream_groups"] == 2
    assert c5["evaluate_posterior_kernel"] == "nodes" and c5["ranks_seen"] == 8 and c5["tokens"] == 8 * 3 * 8 * 2


def _union_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    from lantern_amd.sharding import reduce_timing, sequence_ids
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per_rank, _, _ = bench.plan_sequences(total, 63, world, 3)
    ids = sequence_ids(rank, world, per_rank)
    toks = _tokens_for(ids)
    dt, tot = reduce_timing(dist, seconds=1.0, tokens=float(sum(toks.values())))
    q.put((rank, ids, toks, tot))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_c5_union_of_eight_shards_equals_single_process(tmp_path):
    """64 sequences over 8 ranks (8 each, contiguous slices): the union of the shards' accepted tokens is what one process gets for
    all 64, no sequence twice or missing; the per-rank statistics files merge to the single process's means."""
    from lantern_amd import sharding as sh
    world, total = 8, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_union_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _tokens_for(range(total))
    seen, paths = {}, []
    for rank, ids, toks, tot in res:
        assert len(ids) == 8 and tot == float(sum(single.values()))
        assert not (set(ids) & set(seen))
        seen.update(toks)
        a, b = ids[0], ids[-1] + 1
        paths.append(sh.write_global_statistics(str(tmp_path), {f"prompt_{i}": sh.statistics_entry(f"p{i}", float(toks[i]), 0.0) for i in ids}, a, b))
    assert seen == single
    m = sh.merge_global_statistics(paths)
    assert m["prompts"] == total and abs(m["mean_step_compression"] - sum(single.values()) / total) < 1e-12


def _stub_env():
    env = dict(os.environ, LANTERN_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(300)
def test_bench_stdout_is_one_compact_parseable_line(tmp_path):
    """The driver parses bench.py's stdout (and keeps an 8 KB tail of it): stdout must be exactly ONE json line, well under 8 KB,
    carrying `roofline` and `cpu_baseline`; the extra runs (here: the stub's padding, ~16 KB) go to --extras-out and stderr."""
    import subprocess
    extras = tmp_path / "extras.json"
    for gpus in ("1", "2"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "3", "--warmup", "1", "--dist-backend", "gloo",
                            "--extras-out", str(extras)], env=_stub_env(), capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and len(lines[0]) < 8192
        line = json.loads(lines[-1])
        assert line["n_gpus"] == int(gpus) and line["roofline"]["bound"] == "hbm" and line["cpu_baseline"]["kind"] == "port"
        assert "ep_batch_sweep" not in line and "configs" not in line
        full = json.load(open(extras))
        assert len(full["ep_batch_sweep"]) == 6 and "configs" in full and len(json.dumps(full)) > 8192
        assert "bench.py full report: " in r.stderr


def test_compact_line_of_a_real_report_stays_small():
    """compact_line on round 4's committed 21 KB report (the one the driver could not parse): < 4 KB, json round-trips, both objects kept."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_driver_invocation_20_5.json")))
    assert len(json.dumps(full)) > 20000
    c = bench.compact_line(full)
    line = json.dumps(c)
    assert len(line) < 4096, len(line)
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert back[k] == full[k] or k == "config"
    assert back["config"]["workload"].startswith("C3") and back["roofline"]["saturating"]["sequences_per_launch"] == 4096
    assert back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["matches_gpu_token_stream"] is True
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]          # the headline pair keeps full precision
    assert abs(back["extras"]["mirror_generate_us_per_verify_step"] / full["mirror_generate"]["us_per_verify_step"] - 1) < 1e-5

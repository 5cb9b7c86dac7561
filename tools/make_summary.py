"""profiles/<round>_summary.md from the round's committed profile files (bench line of the driver invocation, rocprofv3 stats, PMC traffic, layer and GEMM
benches).  usage: python tools/make_summary.py [round=r03]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = lambda n: os.path.join(ROOT, "profiles", n)
d = json.loads(open(P(f"{R}_bench_driver_invocation_20_5.json")).read().strip().splitlines()[-1])
rl, cb, c, dy = d["roofline"], d["cpu_baseline"], d["configs"], d["dynamic_tree"]


def ks(name):
    rows = [r for r in csv.DictReader(open(P(name)))]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    return [(float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"]) for r in rows
            if r["Name"].startswith(("void lantern", "lantern::", "void linear_rows", "linear_rows", "pack_linear"))][:8]


t = json.load(open(P(f"{R}_ep_traffic.json")))
o = [f"# Round {R[1:]} profile summary (one MI355X)\n",
     f"All files in this directory named `{R}_*` were produced on the GPU box by the commands named below; `gpurun_out/` is scratch and not committed.\n",
     f"## Headline: `python bench.py --gpus 1 --steps 20 --warmup 5` (`{R}_bench_driver_invocation_20_5.json`)\n",
     f"* value **{d['value'] / 1e6:.3f} M accepted tokens/s**, {1e3 * d['ms_per_step']:.1f} us per step of {d['config']['seqs_per_gpu']} sequences, mean accept length "
     f"{d['mean_accept_length']:.2f}; CPU stream mismatches: {cb['mismatches']}.",
     f"* `roofline` (dominant kernel: {rl['kernel']}): {1e3 * rl['avg_launch_ms']:.1f} us per launch of {rl['sequences_per_launch']} sequences; contract fraction {rl['frac']:.3f}, "
     f"needed-bytes fraction {rl['frac_needed']:.4f}; PMC traffic {rl['traffic'] / 1e6:.2f} MB per launch ({rl['traffic_source']}).",
     f"* `cpu_baseline` (kind {cb['kind']}): {cb['value']:.0f} tokens/s on {cb['cores']} threads (host shows {cb['host_cores']} cores, usable {cb['usable_cores']}), one thread "
     f"{cb['single_thread']['value']:.0f} tokens/s ({cb['single_thread']['ms_per_seq_step']:.1f} ms per sequence-step).",
     f"* `dynamic_tree`: {dy['value'] / 1e6:.3f} M tokens/s, {1e3 * dy['ms_per_step']:.1f} us per step ({dy['workload']}).",
     *([f"* `drafter_layer`: {d['drafter_layer']['us_per_call']:.1f} us per call, weights at {d['drafter_layer']['weight_stream_GBps'] / 1e3:.2f} TB/s "
        f"({d['drafter_layer']['workload']})."] if "us_per_call" in d.get("drafter_layer", {}) else []),
     f"* `configs.C2` ({c['C2'].get('tree_decoding_rows', '')}): {c['C2']['value'] / 1e6:.3f} M tokens/s, {1e3 * c['C2']['ms_per_step']:.1f} us per step; with O7 over all rows: "
     f"{c['C2'].get('all_rows_by_cfg_mask_topk', {}).get('value', 0) / 1e6:.3f} M tokens/s."]
for x in c["C4"]:
    o.append(f"* `configs.C4` lambda={x['lantern_delta']:g} k={x['lantern_k']}: {x['value'] / 1e6:.3f} M tokens/s, {1e3 * x['ms_per_step']:.1f} us per step "
             f"(O7 over all rows: {x.get('all_rows_by_cfg_mask_topk', {}).get('value', 0) / 1e6:.3f} M); chain on probability rows {1e3 * x['evaluate_posterior']['avg_launch_ms']:.1f} us per launch.")
o.append(f"\n## rocprofv3 `--kernel-trace --stats` of the bench command (`tools/run/prof_default.sh {R}p [flags]`; all three groups in flight, hence longer than the undisturbed event pass)\n")
for key, fn in (("raw rows (default)", f"{R}_raw_kernel_stats.csv"), ("all rows by O7 + chain (`--no-fuse-o7 --spec-rows 0`)", f"{R}_chain_kernel_stats.csv")):
    o += [f"**{key}** (`{fn}`)\n", "| kernel | average us | calls |\n|---|---|---|"]
    o += [f"| `{name[:100]}` | {us:.1f} | {n} |" for us, n, name in ks(fn)[:5]] + [""]
o += [f"## PMC (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes; `{R}_ep_traffic.json`, raw rows in `pmc/{R}_*`)\n",
      f"Measured at commit `{t.get('commit')}`.  FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md.\n", "| launch | HBM bytes per launch | needed | contract |\n|---|---|---|---|"]
o += [f"| {k} (`{v['kernel'][:60]}`) | {v['hbm_bytes'] / 1e6:.2f} MB | {v['algorithmic_window_bytes'] / 1e6:.2f} MB | {v['dense_contract_bytes'] / 1e6:.2f} MB |" for k, v in t["per_launch"].items()]
o += [f"| {k} | {v['hbm_bytes'] / 1e6:.2f} MB | | |" for k, v in t["other_kernels"].items()]
o.append("\n## Drafter decoder layer (`tools/run/layer_prof.sh`, `tools/gemm_bench.py`)\n")
L = json.load(open(P(f"{R}_drafter_layer.json")))
o.append(f"* layer wall time per call (2 x 10 rows, 1200 cached positions, 7B size): tree attention + in-place cache **{L['hip_tree_attention_inplace_cache_us']:.1f} us**, tree attention "
         f"{L['hip_tree_attention_us']:.1f}, in-place cache with SDPA {L['hip_inplace_cache_us']:.1f}, plain {L['hip_skinny_gemm_us']:.1f}; torch bf16 ops {L['torch_linear_us']:.1f}.")
o += [f"* kernels of the drafting call (`{R}_drafter_layer_kernel_stats.csv`):\n", "| kernel | average us | calls |\n|---|---|---|"]
o += [f"| `{name[:90]}` | {us:.1f} | {n} |" for us, n, name in ks(f"{R}_drafter_layer_kernel_stats.csv")[:8]]
for fn, lab in ((f"{R}_skinny_gemm_per_tile_round2.json", "per-tile kernels (round 2)"), (f"{R}_skinny_gemm_streamk_row_major.json", "stream-K, row-major weights"),
                (f"{R}_skinny_gemm_streamk_packed.json", "stream-K, packed weights")):
    g = json.load(open(P(fn)))
    o.append(f"* GEMMs alone, {lab}: {g['all']['us']:.1f} us for {g['all']['weight_MB']:.0f} MB = {g['all']['GBps'] / 1e3:.2f} TB/s ({g['all']['frac_of_8TBps']:.3f} of 8 TB/s) -- per kernel: "
             + "; ".join(f"{k.split(' (')[0]} {v['us']:.1f} us" for k, v in g['kernels'].items()))
o += ["\n## In-kernel phase stamps (`tools/ep_trace.py`, separate `-DEPW_TRACE` build)\n",
      f"`{R}_chain_trace_probs64.txt` / `{R}_chain_trace_raw21.txt`: the chain kernel before the compile-time instances; `*_spec2.txt`: after; `{R}_chain_trace_raw21_final.txt` / `{R}_chain_trace_probs64_final.txt`: the round's final kernels; `{R}_fast_walk_trace_*`: the removed "
      f"fast-walk kernel (v5); `{R}_o7_parts.txt`: O7 with top-k / softmax switched off in turn."]
open(P(f"{R}_summary.md"), "w").write("\n".join(o) + "\n")
print("wrote", P(f"{R}_summary.md"))

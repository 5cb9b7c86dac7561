"""profiles/r06_summary.md from the round's committed profile files (the driver invocation's full report, rocprofv3 stats, PMC traffic).
usage: python tools/make_summary_r06.py"""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)
d = json.load(open(P("r06_bench_driver_invocation_20_5.json")))
t = json.load(open(P("r06_ep_traffic.json")))
rl = d["roofline"]; sat = rl["saturating"]; cb = d["cpu_baseline"]


def ks(fn, n=6):
    rows = [r for r in csv.DictReader(open(P(fn)))]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    return [(float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"]) for r in rows if ("lantern" in r["Name"] or "streamk" in r["Name"])][:n]


o = ["# Round 6 profile summary (one MI355X)\n",
     "Files named `r06_*` in this directory were produced on the GPU box by `tools/run/r06_profiles.sh` (rocprofv3 stats + PMC passes, kernel sources fingerprint "
     f"`{t['kernel_sources_sha']}`, commit `{t['commit']}`) and `tools/run/r06_final.sh` (the driver's invocation with the traffic file in place, the two-rank rehearsals); "
     "`gpurun_out/` is scratch.\n",
     "## Headline: `python bench.py --gpus 1 --steps 20 --warmup 5` (`r06_bench_driver_invocation_20_5.json` = the full report, `r06_bench_line.json` = the stdout line)\n",
     f"* value **{d['value'] / 1e6:.3f} M accepted tokens/s**, {1e3 * d['ms_per_step']:.1f} us per verify step of {d['config']['seqs_per_gpu']} sequences in "
     f"{d['config']['stream_groups']} stream groups, mean accept length {d['mean_accept_length']:.2f}; CPU oracle stream mismatches: {cb['mismatches']}.",
     f"* `roofline` ({rl['kernel']}): {1e3 * rl['avg_launch_ms']:.1f} us per launch of {rl['sequences_per_launch']} sequences (live HIP events); contract bytes "
     f"{rl['algorithmic_bytes_per_launch'] / 1e6:.2f} MB -> frac {rl['frac']:.4f}; needed bytes {rl['needed_bytes_per_launch'] / 1e6:.2f} MB -> {rl['frac_needed']:.4f}; PMC traffic "
     f"{rl['traffic'] / 1e6:.2f} MB per launch = {rl['traffic'] / rl['needed_bytes_per_launch']:.2f} x needed.",
     f"* `roofline.saturating` ({sat['kernel']}): {1e3 * sat['avg_launch_ms']:.1f} us per {sat['sequences_per_launch']} sequences on {sat['inputs']}; needed "
     f"{sat['needed_bytes'] / 1e6:.1f} MB -> {sat['achieved']:.0f} GB/s = **{sat['frac']:.3f}** of 8 TB/s; PMC {sat['traffic'] / 1e6:.1f} MB = {sat['traffic_over_needed']:.3f} x needed.",
     f"* `cpu_baseline` (kind {cb['kind']}): {cb['value']:.0f} tokens/s on {cb['cores']} threads ({cb['sample']}).",
     "* drafting cycle wall (us): " + ", ".join(f"{k} {v.get('us_per_cycle_wall', 0):.0f}" for k, v in d.get("drafter_cycle", {}).items() if isinstance(v, dict)),
     f"* drop-in `EaLumina_mGPT.generate`: {d.get('mirror_generate', {}).get('us_per_verify_step', 0):.1f} us per verify step inside this run (`tools/mirror_bench.py`).",
     "* the same workload with three launches per group (`--fused-prepare 0 --spec-rows 3`), alternating with the default on one box: `r06_two_launch_vs_three_driver_form.txt` "
     "(20-step form) and `r06_two_launch_vs_three_200_steps.txt`.",
     f"* other configurations: lambda mode {d['lambda_mode']['value'] / 1e6:.2f} M tokens/s; EAGLE-2 tree {d['dynamic_tree']['value'] / 1e6:.2f} M ({1e3 * d['dynamic_tree']['ms_per_step']:.1f} us per step); "
     f"C2 {d['configs']['C2']['value'] / 1e6:.2f} M; C4 " + " / ".join(f"{x['value'] / 1e6:.2f}" for x in d["configs"]["C4"]) + " M.",
     "\n## rocprofv3 `--kernel-trace --stats` of the bench command (`r06_raw_kernel_stats.csv`; four groups in flight, and -- commit turn-taking, the default at four groups -- "
     "the chain kernel's duration INCLUDES its wait for the group's turn to commit; `r06_raw_free_kernel_stats.csv` is the same command with `--commit-window 0`: "
     + ", ".join(f"`{name.split('<')[0].replace('void lantern::', '')}` {us:.1f} us" for us, n, name in ks("r06_raw_free_kernel_stats.csv", 3)) + ")\n",
     "| kernel | average us | calls |\n|---|---|---|"]
o += [f"| `{name[:110]}` | {us:.1f} | {n} |" for us, n, name in ks("r06_raw_kernel_stats.csv")]
o += ["\n## evaluate_posterior alone at the saturating batch (`tools/run/ep_sweep_prof.sh`; `r06_ep_sweep_B4096_*`, `r06_ep_sweep_B512_*`)\n", "| kernel | average us | calls |\n|---|---|---|"]
for fn in ("r06_ep_sweep_B4096_kernel_stats.csv", "r06_ep_sweep_B512_kernel_stats.csv"):
    o += [f"| `{name[:110]}` ({fn.split('_')[3]}) | {us:.1f} | {n} |" for us, n, name in ks(fn, 1)]
lg = [json.loads(l) for l in open(P("r06_lg_sweep.json")) if l.strip()]
o += ["\n## LlamaGen's 16384-id window at the saturating batch (`tools/lg_sweep.py`; standard verify on EAGLE-2 trees, probability rows; `r06_lg_sweep.json`, "
      "`r06_lg_sweep_B4096_kernel_stats.csv`, `r06_lg_sweep_B4096_pmc.txt`)\n",
      "| sequences per launch | two-per-CU instance `epw_kernel<512,8,1,4,true,false,5,..>` (us, frac of 8 TB/s on needed bytes) | generic one-per-CU instance `epw_kernel<1024,4,1,1,..>` | `cfg_window` over all 59 rows (us, frac) |\n|---|---|---|---|"]
for r in lg:
    v1, v0 = r["variants"]["1"], r["variants"]["0"]
    o += [f"| {r['sequences_per_launch']} | " + " / ".join(f"{x['launch_us']:.1f}" for x in v1) + f" ({v1[-1]['frac']:.3f}) | " + " / ".join(f"{x['launch_us']:.1f}" for x in v0)
          + f" ({v0[-1]['frac']:.3f}) | {v1[-1]['cfg_mask_topk_us']:.0f} ({v1[-1]['cfg_mask_topk_frac']:.3f}) |"]
o += [f"| rocprofv3 stats, 4096 | `{name[:80]}` {us:.1f} us x {n} | | |" for us, n, name in ks("r06_lg_sweep_B4096_kernel_stats.csv", 8) if "epw_kernel" in name]
fetch = [float(l.split()[2]) for l in open(P("r06_lg_sweep_B4096_pmc.txt")) if l.startswith("FETCH_SIZE")]
if fetch:
    o += [f"\nFETCH_SIZE of the 4096-sequence launch: {fetch[0]:.0f} KB x 2 (gfx950) = {2 * fetch[0] * 1024 / 1e6:.1f} MB against {lg[-1]['variants']['1'][-1]['needed_bytes_per_launch'] / 1e6:.1f} MB needed "
          f"= {2 * fetch[0] * 1024 / lg[-1]['variants']['1'][-1]['needed_bytes_per_launch']:.2f} x."]
o += ["\n## PMC (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes; `r06_ep_traffic.json`; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)\n",
      "| launch | HBM bytes per launch | needed |\n|---|---|---|"]
o += [f"| {k} | {v['hbm_bytes'] / 1e6:.2f} MB | {v['algorithmic_window_bytes'] / 1e6:.2f} MB |" for k, v in t["per_launch"].items()]
o += [f"| saturating {k} | {v['hbm_bytes'] / 1e6:.1f} MB | {v['needed_bytes'] / 1e6:.1f} MB |" for k, v in t["saturating"].items()]
o += [f"| {k} | {v['hbm_bytes'] / 1e6:.2f} MB | |" for k, v in t["other_kernels"].items()]
o += ["\n## Other records\n",
      "* `r06_two_rank_one_device.json`, `r06_two_rank_one_device_total16.json`: `LANTERN_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 ...` (two ranks on the one device over gloo: the "
      "N > 1 control flow, `ranks_seen` 2).",
      "* `r06_fuzz_soak.txt`: `tools/run/soak_r06.sh` (static / dynamic / O7 / O3 / top-p / draws soaks against the CPU oracle, 0 failures)."]
open(P("r06_summary.md"), "w").write("\n".join(o) + "\n")
print("wrote", P("r06_summary.md"))

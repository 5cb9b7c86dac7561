"""(round 4: + `saturating`: the ep_sweep passes of tools/run/ep_sweep_prof.sh, given as sat=<gpurun_out dir>:<batch>; + the fingerprint of the
kernel sources, which bench.py checks before quoting the file.)
Collect a round's rocprofv3 outputs (tools/run/prof_default.sh <tag> [flags]) into profiles/: per-kernel stats CSVs, the PMC counter sums
(FETCH_SIZE / WRITE_SIZE, separate passes) as profiles/<round>_ep_traffic.json -- stamped with the commit the numbers were taken at, so that
bench.py can tell whether they still describe the kernels it runs -- and the bench line each profiled run printed.
usage: python tools/make_traffic.py [round=r03] [key=dir ...]   (default keys: raw=<round>p chain=<round>p_chain)"""
import csv, glob, json, os, shutil, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
runs = dict(a.split("=", 1) for a in sys.argv[2:]) or {"raw": RND + "p", "chain": RND + "p_chain"}          # traffic key prefix -> gpurun_out/<dir>
sat_runs = [v for k, v in list(runs.items()) if k.startswith("sat")]
runs = {k: v for k, v in runs.items() if not k.startswith("sat")}
kern = {"raw": ["epw_kernel", "epw_kernel_fused"], "chain": ["epw_kernel"], "nodes": ["epn_kernel", "epn_walk_kernel"]}          # (a run holds one of the two raw-row forms)
others = ["prep_rows_kernel", "cfg_window_bf16", "update_inputs"]


def sums(d, pat):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in acc.items()}, max([v[1] for v in acc.values()] + [0])


out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv) on the bench command "
               "itself (`python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep '' --no-extras [flags]`), averaged over "
               "every launch of the kernel in the run.  Units: counter value x 1024 bytes.  gfx950 correction (MI355X_MICROARCH.md, HBM): "
               "FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE as is.",
       "per_launch": {}, "other_kernels": {}}
for key, d in runs.items():
    src = os.path.join(ROOT, "gpurun_out", d)
    full = os.path.join(src, "b_prof_full.json")          # the full report (bench.py --extras-out); stdout holds the compact line only
    line = json.load(open(full)) if os.path.exists(full) else json.loads(open(os.path.join(src, "b_prof.json")).read().strip().splitlines()[-1])
    B = line["roofline"]["sequences_per_launch"]
    fetch = wr = 0.0
    n = 0
    for k in kern[key]:
        f, n_k = sums(os.path.join(src, "pmc_fetch"), k + "<")
        w, _ = sums(os.path.join(src, "pmc_write"), k + "<")
        n = max(n, n_k)
        fetch += f.get("FETCH_SIZE", 0.0); wr += w.get("WRITE_SIZE", 0.0)
    rl = line["roofline"]
    out["per_launch"][f"{key}_B{B}"] = {
        "kernel": rl["kernel"], "flags": d, "launches_averaged": n, "FETCH_SIZE_raw_KB": fetch, "WRITE_SIZE_raw_KB": wr,
        "hbm_bytes": 2 * fetch * 1024 + wr * 1024,
        "algorithmic_window_bytes": rl.get("windowed_kernel", {}).get("hbm_bytes_needed_per_launch") or rl.get("needed_bytes_per_launch"),
        "dense_contract_bytes": rl["algorithmic_bytes_per_launch"]}
    for k in others:
        f, n = sums(os.path.join(src, "pmc_fetch"), k)
        w, _ = sums(os.path.join(src, "pmc_write"), k)
        if n:
            out["other_kernels"][f"{key}_B{B}:{k}"] = {"launches_averaged": n, "FETCH_SIZE_raw_KB": f.get("FETCH_SIZE", 0.0),
                                                      "WRITE_SIZE_raw_KB": w.get("WRITE_SIZE", 0.0),
                                                      "hbm_bytes": 2 * f.get("FETCH_SIZE", 0.0) * 1024 + w.get("WRITE_SIZE", 0.0) * 1024}
    shutil.copy(os.path.join(src, "prof", "default_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{RND}_{key}_kernel_stats.csv"))
    json.dump(line, open(os.path.join(ROOT, "profiles", f"{RND}_{key}_bench_under_rocprof.json"), "w"), indent=1)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(src, "pmc_" + ("fetch" if c == "FETCH_SIZE" else "write"), "*counter_collection.csv"))[0]
        rows = [r for r in csv.DictReader(open(f)) if "lantern::" in r["Kernel_Name"]]
        with open(os.path.join(ROOT, "profiles", "pmc", f"{RND}_{key}_{c}_B{B}.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"], extrasaction="ignore")
            w.writeheader()
            for r in rows:
                r["Kernel_Name"] = r["Kernel_Name"][:80]
                w.writerow(r)
out["saturating"] = {}
for spec in sat_runs:          # <dir>:<batch>: tools/run/ep_sweep_prof.sh <dir> <batch>
    d, B = spec.split(":")
    src = os.path.join(ROOT, "gpurun_out", d)
    f, n = sums(os.path.join(src, "fetch"), "epw_kernel<")
    w, _ = sums(os.path.join(src, "write"), "epw_kernel<")
    sw = json.loads(open(os.path.join(src, "stats.json")).read().strip().splitlines()[-1])["sweep"]
    row = [r for r in sw if r["sequences_per_launch"] == int(B)][0]
    out["saturating"][f"chain_B{B}"] = {"kernel": "epw_kernel (probability rows, throughput instance)", "flags": d, "launches_averaged": n,
                                        "FETCH_SIZE_raw_KB": f.get("FETCH_SIZE", 0.0), "WRITE_SIZE_raw_KB": w.get("WRITE_SIZE", 0.0),
                                        "hbm_bytes": 2 * f.get("FETCH_SIZE", 0.0) * 1024 + w.get("WRITE_SIZE", 0.0) * 1024,
                                        "needed_bytes": row["chain"]["hbm_bytes_needed_per_launch"], "launch_ms_under_rocprof": row["chain"]["launch_ms"]}
    st = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{RND}_ep_sweep_B{B}_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "summary.txt")):
        shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", f"{RND}_ep_sweep_B{B}_pmc.txt"))
import bench
out["kernel_sources_sha"] = bench.kernel_sources_sha()
try:
    # the commit the GPU box ran (tools/run/.commit, written when the snapshot was sent) -- HEAD may have moved on since; the kernel sources it was
    # measured on are pinned by kernel_sources_sha either way
    cf = os.path.join(ROOT, "tools", "run", ".commit")
    out["commit"] = open(cf).read().strip() if os.path.exists(cf) else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    changed = subprocess.check_output(["git", "-C", ROOT, "diff", "--name-only", out["commit"], "--", "lantern_amd/csrc"], text=True).split()
    out["verify_path_sources_changed_since"] = [f for f in changed if os.path.basename(f) in bench.VERIFY_PATH_SOURCES]
except Exception:
    out["commit"] = None
json.dump(out, open(os.path.join(ROOT, "profiles", f"{RND}_ep_traffic.json"), "w"), indent=1)
print(json.dumps(out["per_launch"], indent=1))

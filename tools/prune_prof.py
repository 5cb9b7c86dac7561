"""Shrink rocprofv3 output directories so that a GPU call's gpurun_out/ stays under the 64 MiB that travel back: counter_collection CSVs keep
the four columns the summaries read (kernel names clipped), kernel traces of the PMC passes and agent tables are dropped; *_stats.csv stay.
usage: python tools/prune_prof.py <dir> [<dir> ...]"""
import csv, glob, os, sys

KEEP = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
for root in sys.argv[1:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(f, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=KEEP, extrasaction="ignore")
            w.writeheader()
            for r in rows:
                r["Kernel_Name"] = r["Kernel_Name"][:160]
                w.writerow(r)
    for pat in ("*kernel_trace.csv", "*agent_info.csv", "*.rocpd", "*.db"):
        for f in glob.glob(root + "/**/" + pat, recursive=True):
            if "/prof/" in f or "/stats/" in f:
                continue          # the --stats passes keep their trace
            os.remove(f)

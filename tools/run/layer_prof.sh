#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2layer}
mkdir -p $OUT
timeout -k 10 300 python tools/layer_bench.py 10 1200 $OUT/layer.json > $OUT/layer.txt 2>&1; tail -3 $OUT/layer.txt
# the profile: the drafting call alone (tree attention, in-place cache)
LAYER_BENCH_ONLY=tree timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o layer -- python3 tools/layer_bench.py 10 1200 > $OUT/layer_tree_only.txt 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/prof/layer_kernel_stats.csv"))]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:16]: print(f"{float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']:>5s}  {r['Name'][:100]}")
PY

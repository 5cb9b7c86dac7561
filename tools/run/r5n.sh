#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5n}; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 bench.py --gpus 1 --steps 40 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --extras-out "" > $O/b.json 2> $O/b.err || tail -5 $O/b.err
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
tail -c 400 $O/b.json

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kvonly
mkdir -p $O
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python3 tools/kv_only.py 21 40 > $O/$c.json 2> $O/$c.err || { tail -5 $O/$c.err; exit 1; }
  python3 tools/pmc_sum.py $O/$c update_inputs_kernel
done
tail -1 $O/WRITE_SIZE.json
python3 - <<PY
import csv,glob
# per-dispatch values of the LAST 40 update_inputs launches (the identical ones)
for c in ("WRITE_SIZE","FETCH_SIZE"):
    f=glob.glob("gpurun_out/kvonly/%s/*counter_collection.csv"%c)[0]
    rows=[r for r in csv.DictReader(open(f)) if "update_inputs_kernel" in r["Kernel_Name"]]
    vals=[float(r["Counter_Value"]) for r in rows][-40:]
    print(c, "last 40 launches: mean KB", sum(vals)/len(vals), "min", min(vals), "max", max(vals))
PY

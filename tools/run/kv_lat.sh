#!/bin/bash
O=gpurun_out/kvlat
mkdir -p $O
for st in 3 5 8; do
KV_LAT_STEP=$st timeout -k 10 500 python3 tools/kv_lat.py 50 21:4096 > $O/s$st.json 2> $O/s$st.err || tail -5 $O/s$st.err
echo "step $st"; grep "sequences\|moved MB" $O/s$st.json | grep -v knobs
done

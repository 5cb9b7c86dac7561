#!/bin/bash
# round 5, second GPU call: the whole -m gpu suite on the re-split chain kernel (epw_body.h + three instance files), then the raw-row throughput forms
O=gpurun_out/r5b; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/test.log 2>&1; echo "tests rc=$?" | tee -a $O/test.log; tail -5 $O/test.log
for v in 256 512; do
  LANTERN_EPW_TP_RAW=$v timeout -k 10 400 python tools/raw_sweep.py 512,2048 12 0 > $O/raw_$v.json 2> $O/raw_$v.err || tail -5 $O/raw_$v.err
done
LANTERN_EPW_TP=0 timeout -k 10 400 python tools/raw_sweep.py 512,2048 12 0 > $O/raw_generic.json 2> $O/raw_generic.err || tail -5 $O/raw_generic.err
cat $O/raw_*.json

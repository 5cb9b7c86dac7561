#!/bin/bash
# the round's final records: the driver's invocation (now with profiles/r06_ep_traffic.json in place) and the two-rank rehearsals on the one device
O=gpurun_out/r06final; mkdir -p $O
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras-out $O/bench_full.json > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(len(open('$O/bench.json').read()), d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['saturating']['frac'], d['roofline']['saturating']['traffic'], d['extras'])"
export LANTERN_BENCH_ONE_DEVICE=1 LANTERN_BENCH_C5_TOTAL=16          # (two ranks on the one device: the strong leg as 16 sequences in all)
timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --seqs-per-gpu 16 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out $O/two_rank_full.json > $O/two_rank.json 2> $O/two_rank.err || { tail -20 $O/two_rank.err; exit 1; }
tail -c 600 $O/two_rank.json
timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --total-seqs 16 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out $O/two_rank_total16_full.json > $O/two_rank_total16.json 2> $O/two_rank_total16.err || { tail -20 $O/two_rank_total16.err; exit 1; }
tail -c 600 $O/two_rank_total16.json
unset LANTERN_BENCH_ONE_DEVICE LANTERN_BENCH_C5_TOTAL
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
bash tools/run/soak_r06.sh soak_r06 > $O/soak.log 2>&1; tail -12 $O/soak.log

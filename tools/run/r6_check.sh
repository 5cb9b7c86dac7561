#!/bin/bash
# the whole GPU suite, the driver's invocation, phase traces of the raw latency instance at 16 sequences
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6chk}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras-out $O/bench_full.json > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(len(open('$O/bench.json').read()), d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['saturating']['frac'], d['roofline']['saturating']['avg_launch_ms'], d.get('extras'))"
if [ -f tools/liblantern_trace1.so ]; then EPW_TRACE=1 EPW_B=16 EPW_MODE=raw timeout -k 10 200 python3 tools/ep_trace.py > $O/raw16_l1.txt 2>&1; tail -45 $O/raw16_l1.txt; fi
if [ -f tools/liblantern_trace3.so ]; then EPW_TRACE=3 EPW_B=16 EPW_MODE=raw timeout -k 10 200 python3 tools/ep_trace.py > $O/raw16_l3.txt 2>&1; tail -60 $O/raw16_l3.txt | head -50; fi

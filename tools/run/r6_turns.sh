#!/bin/bash
# commit turn-taking: parity test, then the timed step by commit window (0 = free-running) and stream groups; the latency variant of the chain kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6turn}; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "turn_taking or fused_o7_loop or harness_loop" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for cfg in ${CFGS:-"4 0" "4 1" "4 2" "3 1" "2 1" "4 1" "4 0"}; do set -- $cfg
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --groups $1 --commit-window $2 > $O/cw_$1_$2.json 2> $O/cw_$1_$2.err || { tail -5 $O/cw_$1_$2.err; continue; }
  python3 -c "import json; d=json.load(open('$O/cw_$1_$2.json')); print('groups $1 window $2: us/step %.2f value %.0f' % (1e3*d['ms_per_step'], d['value']), {k: round(1e3*v['avg_launch_ms'],1) for k,v in d.get('kernels',{}).items()}, round(1e3*d['roofline']['avg_launch_ms'],1))"
done

#!/bin/bash
# driver-form A/B (20 timed steps after 5 warm-up steps, fresh process each): commit turn-taking on / off, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6ab}; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "turn_taking" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
for rep in 1 2 3 4; do for cw in 1 0; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --commit-window $cw > $O/d_${cw}_$rep.json 2> $O/d_${cw}_$rep.err || { tail -5 $O/d_${cw}_$rep.err; continue; }
  python3 -c "import json; d=json.load(open('$O/d_${cw}_$rep.json')); print('driver form window $cw rep $rep: us/step %.2f value %.0f' % (1e3*d['ms_per_step'], d['value']))"
done; done
for cw in 1 0; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 100 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --commit-window $cw > $O/m_$cw.json 2> $O/m_$cw.err
  python3 -c "import json; d=json.load(open('$O/m_$cw.json')); print('100 steps window $cw: us/step %.2f' % (1e3*d['ms_per_step']))"
done

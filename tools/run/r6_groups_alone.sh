#!/bin/bash
# one / two / four stream groups of 16 sequences each: how much of the 4-group step is contention between the groups?  + the compact instance's variants
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6ga}; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_loop.py -x -q -m gpu -k "throughput or two_workgroups" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for rep in 1 2; do for tp4 in 1 3 4 2; do
  LANTERN_EPW_TP4=$tp4 timeout -k 10 300 python3 tools/ep_sweep.py 4096 24 chain > $O/tp4_${tp4}_$rep.json 2> $O/tp4_${tp4}_$rep.err || { tail -5 $O/tp4_${tp4}_$rep.err; exit 1; }
  python3 -c "
import json
d = json.load(open('$O/tp4_${tp4}_$rep.json'))
for r in d['sweep']:
    c = r['chain']; print('tp4=$tp4 rep $rep B', r['sequences_per_launch'], 'launch us %.1f  back-to-back us %.1f  frac %.3f' % (1e3 * c['launch_ms'], 1e3 * c['back_to_back_ms'], c['frac']))"
done; done
for cfg in "16 1" "32 2" "48 3" "64 4" "32 1" "64 2" "64 1"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --seqs-per-gpu $1 --groups $2 --ep chain > $O/g_$1_$2.json 2> $O/g_$1_$2.err || { tail -5 $O/g_$1_$2.err; continue; }
  python3 -c "import json; d=json.load(open('$O/g_$1_$2.json')); print('seqs $1 groups $2: us/step %.2f value %.0f' % (1e3*d['ms_per_step'], d['value']), {k: round(1e3*v['avg_launch_ms'],1) for k,v in d.get('kernels',{}).items()}, round(1e3*d['roofline']['avg_launch_ms'],1))"
done
timeout -k 10 100 python3 tools/host_vs_gpu.py 4 chain 1 3 64 2>/dev/null
timeout -k 10 100 python3 tools/host_vs_gpu.py 1 chain 1 3 16 2>/dev/null

#!/bin/bash
# round 6: the compact throughput instance's variants at the saturating batch (same box, alternating), the SIMD placement probe, and the timed step by
# prepared rows per sequence.  usage: r6_tp_variants.sh [tag]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6tp}; mkdir -p $O
./tools/probe/simd_probe > $O/simd.txt 2>&1; cat $O/simd.txt
timeout -k 10 500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "throughput_instance_variants or two_workgroups" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for rep in 1 2; do for tp4 in ${TP4S:-1 2 3 0}; do
  LANTERN_EPW_TP4=$tp4 timeout -k 10 300 python3 tools/ep_sweep.py 4096 24 chain > $O/tp4_${tp4}_$rep.json 2> $O/tp4_${tp4}_$rep.err || { tail -5 $O/tp4_${tp4}_$rep.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$O/tp4_${tp4}_$rep.json"))
for r in d["sweep"]:
    c = r["chain"]; print("tp4=$tp4 rep $rep B", r["sequences_per_launch"], "launch us %.1f  back-to-back us %.1f  frac %.3f" % (1e3 * c["launch_ms"], 1e3 * c["back_to_back_ms"], c["frac"]))
PY
done; done
for sr in ${SRS:-3 4 5 6 8}; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --spec-rows $sr > $O/sr_$sr.json 2> $O/sr_$sr.err || { tail -5 $O/sr_$sr.err; continue; }
  python3 -c "import json; d=json.load(open('$O/sr_$sr.json')); print('spec-rows $sr: us/step %.2f value %.0f' % (1e3*d['ms_per_step'], d['value']), {k: round(1e3*v['avg_launch_ms'],1) for k,v in d.get('kernels',{}).items()}, round(1e3*d['roofline']['avg_launch_ms'],1))"
done

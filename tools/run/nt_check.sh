#!/bin/bash
O=gpurun_out/${1:-ntc}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_drafter.py tests/test_drafter_layer.py tests/test_gpu_more.py tests/test_gpu_generate.py tests/test_gpu_generate_lg.py tests/test_gpu_generate_ref.py tests/test_gpu_mirror.py -m gpu -q -x > $O/test.log 2>&1; echo "tests rc=$?"; tail -3 $O/test.log
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["model"], round(d["us_per_cycle_wall"],1), round(d["us_per_depth_wall"],1))'
for m in lumina_static anole_static llamagen_static lumina anole llamagen; do python tools/draft_bench.py $m 1200 30 2>/dev/null | python -c "$P"; done
python tools/layer_bench.py 2>/dev/null | tail -1 | cut -c1-600

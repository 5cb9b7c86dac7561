#!/bin/bash
O=gpurun_out/${1:-r5t}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_parity.py tests/test_gpu_loop.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_properties.py -m gpu -q -x > $O/test.log 2>&1; echo "tests rc=$?"; tail -3 $O/test.log
python tools/o7_time.py 63 && python tools/o7_time.py 64 && python tools/o7_time.py 21
run() { timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 8 --extras-out "" "$@" 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(1e3*d['ms_per_step'],2), 'ep_us', round(1e3*d['roofline']['avg_launch_ms'],1), d['cpu_baseline']['matches_gpu_token_stream'], {k:(v.get('avg_launch_ms'), v.get('frac')) for k,v in d.get('kernels',{}).items()})"; }
run
run

python -m pytest tests/test_gpu_loop.py -q -x 2>&1 | tail -4
for g in 1 2 4 6; do for gr in "" "--graph"; do
python bench.py --cpu-seconds 0 --groups $g $gr 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('groups $g $gr', round(d['value']), round(d['ms_per_step'],4), 'ep', round(d['roofline']['avg_launch_ms'],4), 'o7', round(d['kernels']['cfg_mask_topk']['avg_launch_ms'],4), 'kv', round(d['kernels']['kv_gather']['avg_launch_ms'],4))"
done; done

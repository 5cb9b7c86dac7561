#!/bin/bash
# the commit launch's tuning values under the two-launch step (the step is the ring of commits now): bench.py --tuning kv_u / kv_variant, 200 timed steps each
O=gpurun_out/r6_fused; mkdir -p $O
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M  commit %.1f us' % (1e3*d['ms_per_step'], d['value']/1e6, 1e3*d['kernels']['kv_gather']['avg_launch_ms']))"; }
echo -n "default: "; run
for t in kv_u=1 kv_u=4 kv_variant=21 kv_variant=22 kv_variant=23 kv_variant=40 kv_variant=43 kv_variant=10 kv_variant=11; do echo -n "$t: "; run --tuning $t; done
echo -n "default: "; run

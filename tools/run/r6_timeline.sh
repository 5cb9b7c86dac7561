#!/bin/bash
# kernel trace of the timed loop (200 steps): per-queue durations, gaps and step period (tools/timeline.py).  usage: r6_timeline.sh [tag] [bench flags]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6tl}; shift; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --extras-out "" "$@" > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/b.json')); print('us/step %.2f value %.0f' % (1e3*d['ms_per_step'], d['value']))"
python3 tools/timeline.py $O/trace 60 | tee $O/timeline.txt
find $O/trace -name "*kernel_trace.csv" -size +20M -delete

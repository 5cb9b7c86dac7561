#!/bin/bash
# kernel trace of drafting cycles through the depth plans (tools/draft_bench.py): usage draft_prof.sh <tag> <model...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-dprof}; shift
for M in "$@"; do
  O=gpurun_out/$TAG/$M; mkdir -p $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o d -- python3 tools/draft_bench.py $M 1200 20 > $O/out.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  cat $O/out.json
  rm -f $O/*agent_info.csv
done

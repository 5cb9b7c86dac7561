#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 1" "2 1" "4 1" "8 1"; do set -- $cfg; export LANTERN_TA_SPLITS=$1 LANTERN_TA_MIN_TILES=$2
  O=gpurun_out/tap2/s$1; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 tools/probe/ta_draft_shape.py 10 1210 > $O/out.txt 2>&1
  echo "splits $1:"; grep "tree_attention" $O/t_kernel_stats.csv | cut -d, -f1,2,4,6,7 | cut -c1-60,100-200
  rm -f $O/t_kernel_trace.csv $O/t_agent_info.csv
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/o7
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/t_all.txt 2>&1 || { tail -30 $O/t_all.txt; exit 1; }
tail -2 $O/t_all.txt
(timeout -k 10 200 python3 tools/o7_parts.py 64; timeout -k 10 200 python3 tools/o7_parts.py 1) > $O/parts.txt 2>&1; cat $O/parts.txt
EPW_TRACE=3 EPW_B=21 EPW_MODE=raw timeout -k 10 200 python3 tools/ep_trace.py > $O/raw21_l3.txt 2>&1; tail -9 $O/raw21_l3.txt

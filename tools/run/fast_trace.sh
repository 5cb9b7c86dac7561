#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/fast
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_nodes.py -x -q -m gpu > $O/t_nodes.txt 2>&1 || { tail -20 $O/t_nodes.txt; exit 1; }
tail -2 $O/t_nodes.txt
EPW_B=64 EPW_MODE=chain timeout -k 10 200 python3 tools/epf_trace.py > $O/trace_probs64.txt 2>&1 &&
EPW_B=21 EPW_MODE=raw timeout -k 10 200 python3 tools/epf_trace.py > $O/trace_raw21.txt 2>&1
tail -12 $O/trace_probs64.txt; tail -10 $O/trace_raw21.txt

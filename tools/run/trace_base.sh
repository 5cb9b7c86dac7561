#!/bin/bash
# baseline phase traces of the chain kernel: probability rows @64, raw rows @21 (trace levels 1 and 3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tr
EPW_TRACE=3 EPW_B=64 EPW_MODE=chain timeout -k 10 200 python3 tools/ep_trace.py > gpurun_out/tr/chain64_l3.txt 2>&1 &&
EPW_TRACE=1 EPW_B=64 EPW_MODE=chain timeout -k 10 200 python3 tools/ep_trace.py > gpurun_out/tr/chain64_l1.txt 2>&1 &&
EPW_TRACE=1 EPW_B=21 EPW_MODE=raw timeout -k 10 200 python3 tools/ep_trace.py > gpurun_out/tr/raw21_l1.txt 2>&1 &&
EPW_TRACE=3 EPW_B=21 EPW_MODE=raw timeout -k 10 200 python3 tools/ep_trace.py > gpurun_out/tr/raw21_l3.txt 2>&1
tail -30 gpurun_out/tr/chain64_l1.txt

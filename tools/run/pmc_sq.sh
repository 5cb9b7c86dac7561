#!/bin/bash
# SQ counters (VALU / LDS activity, LDS conflicts) of O7 alone and of the bench's kernels: two --pmc passes each (kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmcsq}
mkdir -p $O
A="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS_ATOMIC"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/o7_a -o a -- python3 tools/o7_only.py 64 > $O/o7_a.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/o7_b -o b -- python3 tools/o7_only.py 64 > $O/o7_b.txt 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/bench_a -o a -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/bench_a.txt 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/bench_b -o b -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/bench_b.txt 2>&1
{
  echo "# rocprofv3 --pmc SQ counters, per launch (sums over all XCDs / CUs), commit $(cat tools/run/.commit 2>/dev/null)"
  echo "## cfg_window_bf16_kernel, 1664 rows (tools/o7_only.py 64)"; python3 tools/pmc_sum.py $O/o7_a cfg_window_bf16; python3 tools/pmc_sum.py $O/o7_b cfg_window_bf16
  for k in "epw_kernel<" prep_rows_kernel update_inputs_kernel; do echo "## $k (bench.py --steps 20 --warmup 5, 3 groups of 21)"; python3 tools/pmc_sum.py $O/bench_a "$k"; python3 tools/pmc_sum.py $O/bench_b "$k"; done
} > $O/summary.txt
cat $O/summary.txt

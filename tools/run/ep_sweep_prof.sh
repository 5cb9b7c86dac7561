#!/bin/bash
# rocprofv3 passes of evaluate_posterior alone at the saturating batch (tools/ep_sweep.py, rotating inputs): kernel trace + stats, FETCH_SIZE,
# WRITE_SIZE and SQ counter groups -- each its own run, kernel trace only.  usage: ep_sweep_prof.sh <tag> <batches> [LANTERN_EPW_TP]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-epsw}; BATCHES=${2:-4096}; export LANTERN_EPW_TP=${3:-5}
O=gpurun_out/$TAG
mkdir -p $O
[ -f $O/counters.txt ] || rocprofv3 -L > $O/counters.txt 2>&1
run() { # name, counters...
  local n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 tools/ep_sweep.py $BATCHES 12 chain > $O/$n.json 2> $O/$n.err || { echo "pass $n failed"; tail -5 $O/$n.err; return 1; }
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 tools/ep_sweep.py $BATCHES 24 chain > $O/stats.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
run fetch FETCH_SIZE && run write WRITE_SIZE &&
run sqa SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run sqb SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS &&
run sqc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_ATOMIC
{
  echo "# rocprofv3 passes of tools/ep_sweep.py $BATCHES (evaluate_posterior alone, rotating inputs), LANTERN_EPW_TP=$LANTERN_EPW_TP, commit $(cat tools/run/.commit 2>/dev/null)"
  for n in fetch write sqa sqb sqc; do python3 tools/pmc_sum.py $O/$n "epw_kernel"; done
} > $O/summary.txt
cat $O/summary.txt
find $O/stats -name "*kernel_stats*" | head -2

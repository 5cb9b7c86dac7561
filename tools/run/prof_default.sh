#!/bin/bash
# round-2 profiles of the bench command: rocprofv3 kernel trace + stats, then the FETCH_SIZE / WRITE_SIZE passes (each its own run,
# kernel trace only).  usage: prof_default.sh <tag> [extra bench.py flags].  Scratch output under gpurun_out/<tag>/; what is to be
# kept is copied into profiles/ by tools/make_traffic.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2p}
shift
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o default -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out $OUT/b_prof_full.json "$@" > $OUT/b_prof.json 2> $OUT/b_prof.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out $OUT/b_fetch_full.json "$@" > $OUT/b_fetch.json 2> $OUT/b_fetch.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out $OUT/b_write_full.json "$@" > $OUT/b_write.json 2> $OUT/b_write.err &&
for k in epw_kernel epn_kernel epn_walk_kernel prep_rows_kernel cfg_window_bf16 update_inputs; do echo "== $k"; python3 tools/pmc_sum.py $OUT/pmc_fetch $k; python3 tools/pmc_sum.py $OUT/pmc_write $k; done > $OUT/pmc_summary.txt
find $OUT/prof -name "*stats*"
cat $OUT/pmc_summary.txt

#!/bin/bash
# rocprofv3 kernel trace of the bench step loop + a sequences-per-GPU sweep (scratch output under gpurun_out/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2c}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof -o nodes -- python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events > $OUT/b.json 2> $OUT/b.err
for n in 8 32; do python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --seqs-per-gpu $n > $OUT/b_$n.json 2>/dev/null; done
find $OUT/prof -name "*stats*" | head

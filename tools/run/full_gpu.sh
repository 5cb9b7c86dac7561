#!/bin/bash
# the round-end checks on one box: GPU test suite, then smoke (stops at the first failure)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2t}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
rc=$?; echo "smoke exit $rc"; tail -3 $OUT/smoke.log
exit $rc

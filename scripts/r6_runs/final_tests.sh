#!/bin/bash
# round 6: the whole GPU suite at the current commit
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_tests; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests -q -m gpu 2>&1 | tail -25 > $OUT/r6_gpu_tests_full.log
cat $OUT/r6_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > $OUT/smoke.log; cat $OUT/smoke.log

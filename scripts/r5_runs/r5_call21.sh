#!/bin/bash
# round 5, call 21: full GPU suite at the documentation commit
OUT=gpurun_out/r5c21; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -4 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log

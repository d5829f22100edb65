#!/bin/bash
# round 5, call 19: the model nequip's builder produced + enable_NequipAMD_full converted, on the GPU vs the oracle
OUT=gpurun_out/r5c19; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_converted_reference_model.py tests/test_model_parity.py > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log

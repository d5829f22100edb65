#!/bin/bash
OUT=gpurun_out/r5c29; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_edge_pairs.py > $OUT/tests.log 2>&1; tail -12 $OUT/tests.log

cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for ov in 1 0; do echo "NQA_NO_OVERLAP=$ov"; NQA_NO_OVERLAP=$ov bash scripts/r2_quick_bench.sh; done
done
timeout 900 python -m pytest tests/test_model_parity.py tests/test_edge_pairs.py tests/test_model_properties_gpu.py tests/test_training_step.py -x -q -m gpu 2>&1 | tail -3

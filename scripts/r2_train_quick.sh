# training-step A/B helper: ms/step and the HIP-event regions of the train256 workload
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload train256 --steps ${STEPS:-10} --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train256 ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})"

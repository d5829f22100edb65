# quick A/B helper: prints ms/step and the per-kernel HIP-event breakdown of the default workload (no CPU baseline, no PMC)
cd $GRAFT_REPO_ROOT
for wl in ${WORKLOADS:-water10k}; do
timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
done

# idle time between kernels in hipGraph replay of the default workload (rocprofv3 kernel trace, last steps)
OUT=$GRAFT_REPO_ROOT/gpurun_out/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --kernel-steps 0 > $OUT/bench.json 2> $OUT/err.log
python - <<'PY' | tee $OUT/gaps.txt
import csv, glob
f = glob.glob("/tmp/gaps/**/bench_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region = the last ~10 graph replays: take the last 10 * k kernels where k = kernels per step (find the period)
names = [r[2] for r in rows]
last = names[-1]
idx = [i for i, n in enumerate(names) if n == last]
k = idx[-1] - idx[-2]
seg = rows[-5 * k:]
busy_union = 0; cur_s, cur_e = seg[0][0], seg[0][1]; gaps = []
for s, e, n in seg[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, n)); busy_union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy_union += cur_e - cur_s
span = seg[-1][1] - seg[0][0]
print(f"kernels per step {k}; over 5 steps: span {span/5e3:.1f} us/step, busy (union) {busy_union/5e3:.1f} us/step, idle {(span-busy_union)/5e3:.1f} us/step in {len(gaps)/5:.0f} gaps/step")
import collections
c = collections.Counter(); t = collections.Counter()
for g, n in gaps:
    c[n[:70]] += 1; t[n[:70]] += g
for n, tt in t.most_common(15):
    print(f"{tt/5e3:7.1f} us/step idle before {c[n]/5:4.1f}x {n}")
PY

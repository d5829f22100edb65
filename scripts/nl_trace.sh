#!/bin/bash
# per-dispatch durations of the node kernels inside one cfg-3 step (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/nl_prof && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/nl_prof
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nl_prof -o nl -- python $GRAFT_REPO_ROOT/scripts/node_breakdown.py > /dev/null 2>&1
python - <<PY
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/nl_prof/**/nl_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "node_linear" in r["Kernel_Name"] or "gate_" in r["Kernel_Name"]]
tot=0
for r in rows[-21:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    tot+=d
    print(r["Kernel_Name"][:36], r["Grid_Size_X"], r["Grid_Size_Y"], d, r.get("VGPR_Count"), r.get("Accum_VGPR_Count"))
print("total", tot)
PY

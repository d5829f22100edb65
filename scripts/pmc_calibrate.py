#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE (KB) against known byte counts in this repo's access patterns.

    python scripts/pmc_calibrate.py <out.json>

Runs scripts/micro/pmc_calib.out (build: hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib.out) under
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (kernel trace only, as the pool's gpurun requires) and
writes, per micro kernel, bytes_moved / (counter x 1024): the factor a counter reading has to be multiplied with.  The committed
result (profiles/r5_pmc_calibration.json) is read by scripts/summarize_profile.py.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "scripts", "micro", "pmc_calib.out")


def main(out_path):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="nqa_calib_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    meta = None
    vals = defaultdict(dict)
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        d = os.path.join(tmp, name)
        res = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "calib", "--", EXE],
                             cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        if res.returncode != 0:
            raise SystemExit(f"rocprofv3 --pmc {counter} failed:\n{res.stdout[-2000:]}\n{res.stderr[-2000:]}")
        for line in res.stdout.splitlines():
            if line.startswith("{\"bytes_per_kernel\""):
                meta = json.loads(line)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        acc, cnt = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
        for k in acc:
            vals[k][counter + "_KB"] = acc[k] / cnt[k]
    assert meta is not None, "pmc_calib.out printed no byte count"
    b = meta["bytes_per_kernel"]
    out = {"bytes_per_kernel": b, "rows": meta["rows"], "W": meta["W"], "kernels": {},
           "note": "factor = bytes moved / (counter KB x 1024); reads: FETCH_SIZE, writes: WRITE_SIZE (the other counter of each "
                   "kernel is listed as a cross-check: a pure reader should write ~0 and vice versa)"}
    for k, v in sorted(vals.items()):
        rec = dict(v)
        main_counter = "FETCH_SIZE_KB" if "read" in k else "WRITE_SIZE_KB"
        if v.get(main_counter, 0) > 0:
            rec["factor"] = b / (v[main_counter] * 1024.0)
        out["kernels"][k] = rec
    get = lambda sub: next((r.get("factor") for k, r in out["kernels"].items() if sub in k), None)
    out["factors"] = {
        "fetch_16B_per_lane_stream": get("calib_read16"),
        "fetch_4B_per_lane_rows": get("calib_read4_rows"),
        "write_4B_per_lane_rows_nontemporal": get("calib_write4_rows<true>"),
        "write_4B_per_lane_rows_plain": get("calib_write4_rows<false>"),
        "write_16B_per_lane_lines_nontemporal": get("calib_write16_nt"),
    }
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out["factors"], indent=1))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r5_pmc_calibration.json"))

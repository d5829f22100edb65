#!/usr/bin/env python3
"""Per-unit timeline of node_linear_wave_kernel (NQA_NODE_DBG=64): for the cfg-3 linear_2 shape, when does each unit
start, when does its first slab arrive, how long is each stage, when does it end.  Diagnostic."""
import os
import sys

os.environ["NQA_NODE_DBG"] = "64"
import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from nequip_amd.o3.irreps import Irreps  # noqa: E402
from nequip_amd.o3.modules import Linear  # noqa: E402
from nequip_amd.nn.interaction_block import uvu_paths  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10125
dev = torch.device("cuda:0")
F, L = 64, 2
hidden = Irreps([(F, (l, 1 if l % 2 == 0 else -1)) for l in range(L + 1)])
sh = Irreps([(1, (l, 1 if l % 2 == 0 else -1)) for l in range(L + 1)])
conv_out = (Irreps([(F, (0, 1))]) + Irreps([(F, (0, 1))] * L) + Irreps(list(hidden)[1:])).simplify()
mid, _ = uvu_paths(hidden, sh, conv_out)
mod = Linear(mid.simplify(), conv_out).to(dev).eval()
x = torch.randn(N, mod.irreps_in.dim, device=dev)
with torch.no_grad():
    for _ in range(3):
        out = mod(x)
    torch.cuda.synchronize()
raw = out.view(torch.int32).reshape(-1)[: 16 * 200000].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
units = 3 * ((N + 31) // 32) + (N + 9) // 10 + (N + 5) // 6
rec = raw[: units * 16].reshape(units, 16)
nst = rec[:, 0]
t0 = rec[:, 1].astype(np.int64)
t0 = (t0 - t0.min()) & 0xFFFFFFFF
print("units", units, "stamps per unit: min", nst.min(), "max", nst.max())
end = np.array([rec[i, nst[i]] for i in range(units)])
first = rec[:, 2]
print("unit start (cycles after the first unit): p10 %d  p50 %d  p90 %d  max %d" % tuple(np.percentile(t0, [10, 50, 90, 100])))
print("first slab in LDS after start:            p10 %d  p50 %d  p90 %d  max %d" % tuple(np.percentile(first, [10, 50, 90, 100])))
print("unit duration:                            p10 %d  p50 %d  p90 %d  max %d" % tuple(np.percentile(end, [10, 50, 90, 100])))
print("kernel span (last end - first start): %d cycles" % int((t0 + end).max()))
# per-stage durations (between successive 'slab stored' stamps)
for k in range(2, 10):
    sel = nst > k + 1
    if sel.sum() == 0:
        break
    d = rec[sel, k + 1] - rec[sel, k]
    print("stage %d -> %d: p10 %d p50 %d p90 %d  (n=%d)" % ((k - 2, k - 1) + tuple(np.percentile(d, [10, 50, 90])) + (int(sel.sum()),)))
# by chunk kind
b = [0, 3 * ((N + 31) // 32), 3 * ((N + 31) // 32) + (N + 9) // 10, units]
# NOTE chunks are sorted by stage count (8-stage d=3 / d=5 chunks first, then the 6-stage d=1 chunks)
import sys as _s; _s.exit(0)
hw = rec[:, 15]
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
xcc = rec[:, 14] & 0xF
key = xcc * 1000000 + se * 10000 + cu * 100 + simd
uniq, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, cu, simd):", len(uniq), " units per SIMD: min", cnt.min(), "p50", int(np.median(cnt)), "max", cnt.max())
for x in range(2):
    sel = xcc == x
    ts = np.sort(rec[sel, 1].astype(np.int64))
    ts = (ts - ts[0]) & 0xFFFFFFFF
    en = (rec[sel, 1].astype(np.int64) - rec[sel, 1].astype(np.int64).min() + end[sel])
    print(f"xcc {x}: {sel.sum()} units; start spread p50 {int(np.percentile(ts, 50))} p90 {int(np.percentile(ts, 90))} max {int(ts.max())}; last end {int(en.max())} cycles")

"""Per-kernel timing with HIP events on the launching stream (used by bench.py for the roofline object).

Disabled by default (zero overhead beyond one attribute check).  When enabled, every native kernel call site
wraps its launch in a pair of timing events recorded on the *current* stream -- the same stream the kernel is
enqueued on -- and tags it with the kernel's algorithmic byte count (DESIGN.md "Kernels").
"""

from __future__ import annotations

import contextlib
from collections import defaultdict
from typing import Dict, List, Tuple

import torch

enabled = False
# hidden width H of the radial MLP (set by bench.py): lets `summary` report the "fused-algorithmic" byte count of SURVEY.md
# 8(d) next to the boundary one -- the weight rows of a launch ([E, W] operands / results) counted as [E, min(W, H)]
hidden_width = None
_records: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event, float, float, float, int]]] = defaultdict(list)


def enable(flag: bool = True) -> None:
    global enabled
    enabled = flag


def reset() -> None:
    _records.clear()


@contextlib.contextmanager
def region(name: str, algorithmic_bytes: float = 0.0, algorithmic_flops: float = 0.0, weight_bytes: float = 0.0,
           weight_cols: int = 0):
    """``weight_bytes``: the part of ``algorithmic_bytes`` that is edge_weight / grad_weight rows of ``weight_cols`` columns."""
    if not enabled:
        yield
        return
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    start.record()
    try:
        yield
    finally:
        stop.record()
        _records[name].append((start, stop, float(algorithmic_bytes), float(algorithmic_flops), float(weight_bytes),
                               int(weight_cols)))


def summary() -> Dict[str, dict]:
    """name -> {calls, total_ms, avg_ms, bytes_per_call, gbps}; call after torch.cuda.synchronize()."""
    out = {}
    for name, recs in _records.items():
        ms = [r[0].elapsed_time(r[1]) for r in recs]
        nbytes = [r[2] for r in recs]
        nflops = [r[3] for r in recs]
        total = sum(ms)
        out[name] = {
            "calls": len(recs),
            "total_ms": total,
            "avg_ms": total / max(len(recs), 1),
            "bytes_per_call": sum(nbytes) / max(len(recs), 1),
            "gbps": (sum(nbytes) / 1e9) / (total / 1e3) if total > 0 else 0.0,
            "flops_per_call": sum(nflops) / max(len(recs), 1),
            "tflops": (sum(nflops) / 1e12) / (total / 1e3) if total > 0 else 0.0,
        }
        if hidden_width and any(r[5] > 0 for r in recs):
            fused = sum(r[2] - r[4] * (1.0 - min(r[5], hidden_width) / r[5]) if r[5] > 0 else r[2] for r in recs)
            out[name]["fused_bytes_per_call"] = fused / max(len(recs), 1)
    return out

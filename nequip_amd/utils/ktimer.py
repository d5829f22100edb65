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
_records: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event, float, float]]] = defaultdict(list)


def enable(flag: bool = True) -> None:
    global enabled
    enabled = flag


def reset() -> None:
    _records.clear()


@contextlib.contextmanager
def region(name: str, algorithmic_bytes: float = 0.0, algorithmic_flops: float = 0.0):
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
        _records[name].append((start, stop, float(algorithmic_bytes), float(algorithmic_flops)))


def summary() -> Dict[str, dict]:
    """name -> {calls, total_ms, avg_ms, bytes_per_call, gbps}; call after torch.cuda.synchronize()."""
    out = {}
    for name, recs in _records.items():
        ms = [a.elapsed_time(b) for a, b, _, _ in recs]
        nbytes = [c for _, _, c, _ in recs]
        nflops = [f for _, _, _, f in recs]
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
    return out

"""Optional per-kernel device timing with HIP events on the launch stream.

`ops` brackets selected launches with `profiling.span(name, units)`; when profiling is off this
is a no-op.  torch.cuda.Event records on torch's current stream, which is the stream every op of
this package launches on, so the elapsed times are the kernels' own durations.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional

import torch

_spans: Optional[list] = None


def start() -> None:
    global _spans
    _spans = []


def active() -> bool:
    return _spans is not None


@contextlib.contextmanager
def span(name: str, units: float = 0.0):
    if _spans is None:
        yield
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    yield
    e1.record()
    _spans.append((name, units, e0, e1))


def stop() -> Dict[str, Dict[str, float]]:
    """Synchronise and return {name: {"ms": total, "launches": n, "units": total_units}}."""
    global _spans
    spans, _spans = _spans, None
    out: Dict[str, Dict[str, float]] = {}
    if not spans:
        return out
    torch.cuda.synchronize()
    for name, units, e0, e1 in spans:
        d = out.setdefault(name, {"ms": 0.0, "launches": 0, "units": 0.0})
        d["ms"] += e0.elapsed_time(e1)
        d["launches"] += 1
        d["units"] += units
    return out

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


def free_port() -> int:
    """A TCP port free right now on 127.0.0.1 (rendezvous of the multi-process tests)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    return a


def assert_bitexact(got, want, name):
    got = np.asarray(got); want = np.asarray(want)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    if got.dtype == np.bool_:
        got = got.astype(np.uint8)
    if want.dtype == np.bool_:
        want = want.astype(np.uint8)
    bad = bits(got.astype(want.dtype)) != bits(want)
    if bad.any():
        idx = np.argwhere(bad)[:5]
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        raise AssertionError(f"{name}: {int(bad.sum())}/{bad.size} elements differ; max |diff|={diff.max():.3e}; "
                             f"first at {idx.tolist()} got={got[tuple(idx[0])]} want={want[tuple(idx[0])]}")

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns (meta dict, arrays dict).  npz, never pickled."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: z[k] for k in z.files if k != "meta"}


def unpack_mask(bits, W, H, D):
    n = (W + 2) * (H + 2) * (D + 2)
    return np.unpackbits(bits)[:n].astype(bool).reshape(D + 2, H + 2, W + 2)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    u = np.uint32 if a.dtype == np.float32 else np.uint64
    return bool(np.array_equal(a.view(u), b.view(u)))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


def ball_mask(W, H, D, cx, cy, cz, r):
    z, y, x = np.mgrid[0:D + 2, 0:H + 2, 0:W + 2]
    m = ((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) <= r * r
    m[0] = m[-1] = False
    m[:, 0] = m[:, -1] = False
    m[:, :, 0] = m[:, :, -1] = False
    return m


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import cpu_ref
    cpu_ref.build()
    return cpu_ref

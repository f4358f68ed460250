"""Live comparison of the oracle with the compiled, unmodified reference (oracle/_ref/libref.so)
on cases that are NOT among the committed goldens.  Runs wherever the reference shim exists (the
build container; the GPU box when the prebuilt .so travelled) and needs OMP_NUM_THREADS=1 for the
reference to be deterministic, which this file arranges by running the comparison in a child."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import cpu_ref as O

def ball(W, H, D, cx, cy, cz, r):
    z, y, x = np.mgrid[0:D + 2, 0:H + 2, 0:W + 2]
    m = ((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) <= r * r
    m[0] = m[-1] = False; m[:, 0] = m[:, -1] = False; m[:, :, 0] = m[:, :, -1] = False
    return m

rng = np.random.default_rng(11)
for case in range(6):
    W, H, D = (int(v) for v in rng.integers(3, 22, size=3))
    acc = int(rng.integers(0, 9))
    speed = int(rng.integers(1, 40))
    dt = float(rng.choice([0.05, 0.02, 0.1]))
    kw = dict(iter=3, acc=acc, speed=speed, dt=dt, diff=float(rng.choice([2.0e-5, 1.0e-3])))
    r, o = O.Reference(W, H, D, **kw), O.Oracle(W, H, D, solver=O.GS_LEX, threads=1, **kw)
    m = ball(W, H, D, *(rng.uniform(1, n) for n in (W, H, D)), rng.uniform(0.8, 3.0))
    for s in (r, o):
        s.set_mask(m)
        s.add_density(1 + W // 2, 1 + H // 2, 1 + D // 2, 0.7)
        s.set_velocity(1, 1, 1, 0.5, -0.25, 2.0)
    for step in range(3):
        r.run_one(); o.run_one()
        for f in range(11):
            a, b = r.get(f), o.get(f)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (case, W, H, D, acc, step, O.FIELD_NAMES[f])

# voxelizer with fresh arguments; the seed is whatever the reference's thread-id hash is today
for stl, kw in ((%(sphere)r, dict(scale=0.9, rot=(15.0, 0.0, 30.0), translate=(1.0, -2.0, 0.5))),
                (%(plate)r, dict(scale=0.4, rot=(0.0, 45.0, 0.0), translate=(-3.0, 0.0, 2.0)))):
    W, H, D = 28, 20, 24
    r, o = O.Reference(W, H, D), O.Oracle(W, H, D)
    seed = r.thread_seed()
    r.load_stl(stl, **kw)
    o.load_stl(stl, seed=seed, **kw)
    assert np.array_equal(r.get(O.OBS), o.get(O.OBS)), stl
    assert r.get(O.OBS).sum() > 0
print("ok")
'''


def test_oracle_matches_live_reference(oracle_mod):
    if not oracle_mod.have_reference():
        pytest.skip("oracle/_ref/libref.so not built here (needs /root/reference)")
    code = CHILD % {"root": ROOT, "sphere": os.path.join(GOLDEN, "sphere_24x12.stl"),
                    "plate": os.path.join(GOLDEN, "plate_ascii.stl")}
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "ok" in out.stdout.split()   # the reference's own buffered console lines may follow

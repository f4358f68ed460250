import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F
from oracle import cpu_ref as O

def run(W, H, D, acc, fp64, shape, b=2):
    rng = np.random.default_rng(5)
    dt = np.float64 if fp64 else np.float32
    x = rng.standard_normal((D + 2, H + 2, W + 2)).astype(dt)
    r = rng.standard_normal((D + 2, H + 2, W + 2)).astype(dt)
    for a in (x, r):
        a[0, 0, :] = a[0, -1, :] = a[-1, 0, :] = a[-1, -1, :] = 0
        a[0, :, 0] = a[0, :, -1] = a[-1, :, 0] = a[-1, :, -1] = 0
        a[:, 0, 0] = a[:, 0, -1] = a[:, -1, 0] = a[:, -1, -1] = 0
    sim = F.Simulation(W, H, D, 1, acc=acc, quiet=1, precision="fp64" if fp64 else "fp32", pair_small=shape)
    ora = O.Oracle(W, H, D, solver=O.JACOBI, acc=acc, fp64=fp64)
    for s in (sim, ora):
        s.set(F.VY, x); s.set(F.VY_PREV, r)
    sim.linear_solver(b, F.VY, F.VY_PREV, 0.3, 1 + 6 * 0.3)
    ora.linear_solver(b, O.VY, O.VY0, 0.3, 1 + 6 * 0.3)
    g, w = sim.get(F.VY), ora.get(O.VY)
    bad = np.argwhere(g != w)
    print("W%d H%d D%d acc%d fp64=%d shape=%d: %d cells differ" % (W, H, D, acc, fp64, shape, len(bad)), end="")
    if len(bad):
        print("  z in", sorted(set(bad[:, 0]))[:12], " y in", sorted(set(bad[:, 1]))[:30], " x range", bad[:, 2].min(), bad[:, 2].max())
    else:
        print()

for fp64 in (0, 1):
    for shape in (0, 1):
        run(300, 20, 5, 2, fp64, shape)
        run(300, 20, 5, 4, fp64, shape)
        run(24, 20, 5, 2, fp64, shape)
        run(300, 6, 5, 2, fp64, shape)

"""Development check of the three-sweeps-per-pass kernel: runs the same tunnel with sweep_fuse=4
(triple kernel forced) and sweep_fuse=2 (pair kernel) and compares every field bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402


def ball(W, H, D, cx, cy, cz, r):
    z, y, x = np.mgrid[0:D + 2, 0:H + 2, 0:W + 2]
    m = ((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) <= r * r
    m[0] = m[-1] = False
    m[:, 0] = m[:, -1] = False
    m[:, :, 0] = m[:, :, -1] = False
    return m


def run(W, H, D, acc, steps, fuse):
    sim = F.Simulation(W, H, D, steps, acc=acc, quiet=1, dump_every=0)
    sim.set_option("sweep_fuse", str(fuse))
    m = ball(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(W, H, D) / 4.0)
    m[1, 1, 1] = m[D, H, W] = True            # corners next to every wall
    m[D // 2 + 1, 1, W // 2] = True
    sim.set_mask(m)
    t = time.perf_counter()
    for _ in range(steps):
        sim.run_one()
    sim.sync()
    dt = time.perf_counter() - t
    out = [sim.get(f) for f in range(11)]
    plan = sim._geti("triple_plan")
    sim.close()
    return out, dt, plan


bad = 0
cases = [(14, 9, 7, 7, 2), (5, 3, 2, 4, 2), (1, 1, 1, 3, 2), (33, 21, 5, 9, 2), (256, 40, 12, 6, 2), (257, 19, 9, 6, 2),
         (300, 47, 33, 11, 2), (512, 25, 18, 8, 2), (509, 12, 40, 10, 1), (64, 64, 64, 15, 2), (128, 200, 70, 9, 1)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (W, H, D, acc, steps) in cases:
    a, ta, plan = run(W, H, D, acc, steps, 4)
    b, tb, _ = run(W, H, D, acc, steps, 2)
    diff = [F.FIELD_NAMES[f] for f in range(11) if a[f].tobytes() != b[f].tobytes()]
    print("%4dx%4dx%4d acc %2d: triple plan %d, %s   (%.1f vs %.1f ms/step)" % (
        W, H, D, acc, plan, "identical" if not diff else "DIFFERS in %s" % diff, ta / steps * 1e3, tb / steps * 1e3), flush=True)
    if diff:
        bad += 1
        f = F.FIELD_NAMES.index(diff[0]) if isinstance(F.FIELD_NAMES, list) else 0
        d = np.argwhere(a[f].view(np.uint32) != b[f].view(np.uint32))
        print("   first mismatches (z,y,x):", d[:6].tolist(), "count", len(d))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Development soak on one GPU: seeded random grids / parameters with the solver kernels FORCED (three sweeps per pass,
each two-sweep kernel, both precisions, both advection forms), every field against the oracle bit for bit.
python tools/soak_single.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402

O.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n):
    # (256 and 512 exactly: the lane-aligned builds of the three-sweep kernel with its two bodies per group of iterations)
    W = int(rng.choice([rng.integers(1, 70), rng.integers(250, 262), rng.integers(505, 520), rng.integers(760, 775), rng.integers(1015, 1030), 256, 512],
                       p=[0.3, 0.1, 0.1, 0.05, 0.1, 0.15, 0.2]))
    H, D = (int(rng.integers(1, 70)), int(rng.integers(1, 60))) if W in (256, 512) else (int(rng.integers(1, 36)), int(rng.integers(1, 28)))
    acc, steps = int(rng.integers(0, 10)), int(rng.integers(1, 3))
    fp64 = bool(rng.random() < 0.35)
    opts = {"sweep_fuse": str(rng.choice([2, 3, 4, 4])), "two_sweep_kernel": str(rng.choice(["auto", "pair", "fused"])),
            "advect_kernels": str(rng.choice(["cell", "row", "celltab", "tile"])), "fuse_advect": str(rng.choice([0, 1])), "wall_free": str(rng.choice(["auto", "0", "1"]))}
    speed = int(rng.choice([30, 30, 3, -10]))
    sim = F.Simulation(W, H, D, steps, speed=speed, acc=acc, quiet=1, precision="fp64" if fp64 else "fp32", **opts)
    ora = O.Oracle(W, H, D, solver=O.JACOBI, fp64=fp64, threads=8, speed=speed, acc=acc)
    mask = np.zeros((D + 2, H + 2, W + 2), dtype=bool)
    mask[1:-1, 1:-1, 1:-1] = rng.random((D, H, W)) < rng.choice([0.0, 0.02, 0.15])
    sim.set_mask(mask)
    ora.set_mask(mask)
    for _ in range(steps):
        sim.run_one()
        ora.run_one()
    u = np.uint64 if fp64 else np.uint32
    diff = [F.FIELD_NAMES[f] for f in range(11) if not np.array_equal(sim.get(f).view(u), ora.get(f).view(u))]
    tag = "ok" if not diff else "DIFF " + ",".join(diff)
    bad += bool(diff)
    print(i, (W, H, D), "acc", acc, "steps", steps, "fp64" if fp64 else "fp32", opts, "speed", speed,
          "plans", sim._geti("pair_shape"), sim._geti("triple_plan"), tag, flush=True)
    sim.close()
    ora.close()
print("cases", n, "failed", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""How much of a small-grid step is launch gaps / event overhead (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
acc = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for prof in (0, 1):
    sim = F.Simulation(N, N, N, 1, acc=acc, quiet=1, dump_every=0, profile=prof)
    sim.addObstacle(N // 3, N // 2, N // 2)
    for _ in range(2):
        sim.run_one()
    sim.sync(); sim.reset_timing()
    t0 = time.perf_counter()
    for _ in range(10):
        sim.run_one()
    sim.sync()
    dt = (time.perf_counter() - t0) / 10
    k = sum(sim.timing(f)[0] for f in ("sweep", "sweep_pair", "divergence", "gradient", "advect", "misc")) / 10 if prof else float("nan")
    print("N=%d acc=%d profile=%d: %.3f ms/step wall, %.3f ms/step in kernels (events)" % (N, acc, prof, dt * 1e3, k))
    sim.close()

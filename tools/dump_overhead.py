#!/usr/bin/env python3
"""Cost of the per-step frame dump (simulation.cpp:140-148) with the asynchronous writer vs the
synchronous one vs no dumps (development tool): python tools/dump_overhead.py [N] [steps]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for mode in ("none", "async", "sync"):
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        sim = F.Simulation(N, N, N, steps, acc=40, quiet=1, dump_dir=tmp,
                           dump_every=0 if mode == "none" else 1, dump_async=0 if mode == "sync" else 1)
        sim.addObstacle(N // 3, N // 2, N // 2)
        sim.iter = 1
        sim.run()            # warm-up incl. staging allocation
        sim.iter = steps
        t0 = time.perf_counter()
        sim.run()
        sim.sync()
        dt = time.perf_counter() - t0
        gb = 5 * (N + 2) ** 3 * 4 * steps / 1e9 if mode != "none" else 0.0
        print("%-6s %d^3: %.1f ms/step  (%.2f GB dumped, %.2f GB/s to %s)" % (
            mode, N, dt / steps * 1e3, gb, gb / dt, tmp.split("/")[1]))
        sim.close()

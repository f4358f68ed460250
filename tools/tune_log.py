#!/usr/bin/env python3
"""Prints the host driver's launch-plan timings (FS_TUNE_LOG) for one grid: every candidate of the two-sweep
kernels (plan < 64: jacobi_pair_kernel shape + 8*alt; plan >= 64: jacobi_fused_kernel<NL=2>) and of the
three-sweep kernel.  python tools/tune_log.py W H D [fp32|fp64] [acc]"""
import os
import sys

os.environ["FS_TUNE_LOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

W, H, D = (int(v) for v in sys.argv[1:4])
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
acc = int(sys.argv[5]) if len(sys.argv) > 5 else 8
sim = F.Simulation(W, H, D, 1, acc=acc, precision=prec, quiet=1, dump_every=0)
sim.addObstacle(W // 3, H // 2, D // 2)
sim.run_one()
sim.run_one()
sim.sync()
for kind in ("pair", "fused"):
    sim.set_option("two_sweep_kernel", kind)
    ms = min(sim.time_sweeps(0, F.PRESSURE, F.DIVERGENCE, 1.0, 6.0, 40) for _ in range(3))
    print("two_sweep_kernel=%s: %.4f ms per sweep (40 sweeps, best of 3), plan %d, triple plan %d" % (
        kind, ms, sim._geti("pair_shape"), sim._geti("triple_plan")))
sim.close()

"""Time per step of solver=gs_lex (the reference's own sweep order, bit-identical with the
reference at one thread) next to solver=jacobi.  Usage: python tools/gs_lex_time.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

out = {}
for (W, H, D, acc, steps) in [(128, 64, 64, 15, 10), (256, 256, 256, 40, 2)]:
    row = {}
    for solver in ("gs_lex", "jacobi"):
        sim = F.Simulation(W, H, D, 1, acc=acc, solver=solver, quiet=1)
        sim.add_ball(W // 4, H // 2, D // 2, H / 6.0) if hasattr(sim, "add_ball") else None
        sim.run_one()
        sim.sync()
        t = time.perf_counter()
        for _ in range(steps):
            sim.run_one()
        sim.sync()
        row[solver + "_ms_per_step"] = round((time.perf_counter() - t) / steps * 1e3, 3)
        del sim
    out["%dx%dx%d_acc%d" % (W, H, D, acc)] = row
    print(W, H, D, acc, row, flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

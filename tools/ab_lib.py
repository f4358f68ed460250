#!/usr/bin/env python3
"""A/B of library builds on ONE box: python tools/ab_lib.py libA.so libB.so ... -- W H D [fp32|fp64] [reps]
Each library is loaded in its own child process (FLUIDSIM_LIB) in turn, `reps` rounds interleaved, and times the
solver passes of one developed grid (time_sweeps: 40 sweeps).  Development tool."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import fluid_simulation_amd as F
W, H, D, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sim = F.Simulation(W, H, D, 1, acc=6, precision=prec, quiet=1, dump_every=0)
sim.addObstacle(W // 3, H // 2, D // 2)
sim.run_one(); sim.run_one(); sim.sync()
ms = sorted(sim.time_sweeps(0, F.PRESSURE, F.DIVERGENCE, 1.0, 6.0, 42) for _ in range(5))
print(json.dumps({"ms_per_sweep_best": ms[0], "ms_per_sweep_median": ms[2], "pair_shape": sim._geti("pair_shape"), "triple_plan": sim._geti("triple_plan")}))
''' % ROOT

args = sys.argv[1:]
sep = args.index("--")
libs, rest = args[:sep], args[sep + 1:]
W, H, D = rest[:3]
prec = rest[3] if len(rest) > 3 else "fp32"
reps = int(rest[4]) if len(rest) > 4 else 3
out = {lib: [] for lib in libs}
for _ in range(reps):
    for lib in libs:
        env = dict(os.environ, FLUIDSIM_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", CHILD, W, H, D, prec], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            out[lib].append({"error": r.stderr[-300:]})
        else:
            out[lib].append(json.loads(r.stdout.strip().splitlines()[-1]))
print(json.dumps(out, indent=1))

"""fs_streamlines on the device next to the numpy restatement of the viewer's CPU code
(oracle/streamlines_ref.py), on the reference's stock grid with a ball in the tunnel.
Usage: python tools/streamlines_time.py [out.json]   (development / measurement tool)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_simulation_amd as F  # noqa: E402
from oracle import streamlines_ref as R  # noqa: E402

W, H, D = 128, 64, 64
sim = F.Simulation(W, H, D, 1, acc=15, quiet=1, dump_every=0)
z, y, x = np.mgrid[0:D + 2, 0:H + 2, 0:W + 2]
sim.set_mask(((x - 40) ** 2 + (y - 32) ** 2 + (z - 32) ** 2) <= 10 ** 2)
for _ in range(60):
    sim.run_one()
sim.sync()
sim.streamlines()                       # warm
t = time.perf_counter()
for _ in range(5):
    lines, norm = sim.streamlines()
gpu_ms = (time.perf_counter() - t) / 5 * 1e3
vx, vy, vz, obs = (np.transpose(sim.get(f), (2, 1, 0)) for f in (F.VX, F.VY, F.VZ, F.OBS))
t = time.perf_counter()
want, wnorm = R.generate_streamlines(vx, vy, vz, obs)
cpu_s = time.perf_counter() - t
same = len(want) == len(lines) and all(a.shape == b.shape and np.allclose(a, b, rtol=0, atol=1e-9) for a, b in zip(lines, want))
out = {"grid": [W, H, D], "config": "GUI/config.py defaults: density 30, proximity 2, 100 steps of 0.2, threshold 0.1",
       "lines": len(lines), "points": int(sum(len(l) for l in lines)), "gpu_ms_per_call": gpu_ms,
       "numpy_restatement_s": cpu_s, "same_lines_to_1e-9": bool(same)}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

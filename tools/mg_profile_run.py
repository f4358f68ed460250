#!/usr/bin/env python3
"""Four steps of a benchmark workload under solver=mg, for a profiler to wrap:
  rocprofv3 --kernel-trace --stats -d out -- python3 tools/mg_profile_run.py c3"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import fluid_simulation_amd as F
from bench import WORKLOADS, add_obstacles
cfg = WORKLOADS[sys.argv[1]]
sim = F.Simulation(cfg["W"], cfg["H"], cfg["D"], 1, acc=cfg["acc"], solver="mg", quiet=1, dump_every=0)
with tempfile.TemporaryDirectory() as tmp:
    add_obstacles(F, sim, cfg, tmp)
for _ in range(4):
    sim.run_one()
sim.sync()

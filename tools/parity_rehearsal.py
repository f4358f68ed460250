#!/usr/bin/env python3
"""Development: bench.py's slab_parity_check alone, N rank processes sharing the GPU (gloo control plane), repeated.
python -m torch.distributed.run --nproc-per-node N tools/parity_rehearsal.py [ipc|shm] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
import fluid_simulation_amd as F
from fluid_simulation_amd import dist as fsdist

transport = sys.argv[1] if len(sys.argv) > 1 else "ipc"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
schedules = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else None
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
for i in range(reps):
    r = bench.slab_parity_check(F, fsdist, dist, rank, world, transport, None, schedules)
    if rank == 0:
        print("repeat", i, r["ok"], r["schedule_run_under_overlap"], flush=True)
dist.destroy_process_group()

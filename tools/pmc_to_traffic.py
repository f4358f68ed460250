#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE output directories into per-launch HBM bytes of
the solver kernels (development tool):
    python tools/pmc_to_traffic.py <fetch_dir> <write_dir> <workload> [<bench line of the same build>.json]
Each entry is stamped with the hash of the kernel sources, the grid and -- from the bench line -- the launch
plan it was measured on; bench.py prints roofline.traffic only for a run that matches the stamp.

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and
WRITE_SIZE are in units of 1024 B; FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, workload = sys.argv[1:4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import subprocess  # noqa: E402

from bench import WORKLOADS, kernels_sha  # noqa: E402

stamp = {"kernels_sha": kernels_sha(), "grid": [WORKLOADS[workload.split("_")[0]][k] for k in ("W", "H", "D")]}   # "c3_fp64" -> c3's grid
try:
    stamp["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:  # noqa: BLE001
    stamp["commit"] = "unknown"
if len(sys.argv) > 4:
    line = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
    stamp["pair_shape"] = line["roofline"]["workgroup_shape_id"]
    stamp["triple_plan"] = line["roofline"]["launch_plan_three_sweeps"]
import re  # noqa: E402

# json key -> pattern of the kernel name as rocprofv3 prints it (first match wins)
KEYS = (("jacobi_fused_kernel<NL=3>", r"jacobi_fused_kernel<\w+, 3,"), ("jacobi_fused_kernel<NL=2>", r"jacobi_fused_kernel<\w+, 2,"),
        ("jacobi_pair_kernel", "jacobi_pair_kernel"), ("jacobi_sweep_kernel", "jacobi_sweep_kernel"),
        ("advect_velocity_kernel", "advect_velocity_kernel"), ("advect_kernel", "advect_kernel"),
        ("divergence_march_kernel", "divergence_march_kernel"), ("gradient_march_kernel", "gradient_march_kernel"),
        ("divergence_kernel", "divergence_kernel"), ("gradient_kernel", "gradient_kernel"))


def mean_by_kernel(d, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                for key, pat in KEYS:
                    if re.search(pat, r["Kernel_Name"]):
                        acc[key].append(float(r["Counter_Value"]))
                        break
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


f, nf = mean_by_kernel(fetch_dir, "FETCH_SIZE")
w, nw = mean_by_kernel(write_dir, "WRITE_SIZE")
out_path = os.path.join(root, "profiles", "sweep_traffic.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data.setdefault(workload, {})
for k in sorted(set(f) | set(w)):
    rd = 2.0 * f.get(k, 0.0) * 1024.0
    wr = w.get(k, 0.0) * 1024.0
    data[workload][k] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                         "FETCH_SIZE_raw_mean": f.get(k), "WRITE_SIZE_raw_mean": w.get(k),
                         "dispatches_sampled": [nf.get(k, 0), nw.get(k, 0)],
                         "correction": "read = 2 * FETCH_SIZE * 1024 (gfx950 half-count), write = WRITE_SIZE * 1024",
                         "stamp": stamp}
    print(k, "read %.1f MB write %.1f MB" % (rd / 1e6, wr / 1e6))
json.dump(data, open(out_path, "w"), indent=1)

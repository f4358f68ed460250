#!/usr/bin/env python3
"""Ungated deltas that put the parity gates in context (SURVEY 8d "parity gates", F1):
  * GPU Jacobi vs GPU gs_lex (= the reference at one thread, bit for bit)
  * the reference against itself at 1 vs N threads (needs the compiled reference shim; run this
    in the build container or on a box where oracle/_ref/libref.so travelled)
python tests/report_solver_deltas.py [N] [steps] [acc]   -> one JSON line"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    acc = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    z, y, x = np.mgrid[0:N + 2, 0:N + 2, 0:N + 2]
    mask = ((x - N / 3) ** 2 + (y - N / 2) ** 2 + (z - N / 2) ** 2) <= (N / 8) ** 2
    out = {"grid": [N, N, N], "steps": steps, "acc": acc}
    import fluid_simulation_amd as F
    names = {F.DENS: "dens", F.VX: "v_x", F.VY: "v_y", F.VZ: "v_z"}
    res = {}
    for solver in ("jacobi", "gs_lex"):
        sim = F.Simulation(N, N, N, steps, acc=acc, solver=solver, quiet=1, dump_every=0)
        sim.set_mask(mask)
        for _ in range(steps):
            sim.run_one()
        res[solver] = {n: sim.get(f) for f, n in names.items()}
        sim.close()
    out["gpu_jacobi_vs_gpu_gs_lex"] = {n: rel(res["jacobi"][n], res["gs_lex"][n]) for n in names.values()}
    # tools/ never touches oracle/; this script lives under tests/ and only loads it when this optional leg is requested
    if "--with-reference" in sys.argv:
        from oracle import cpu_ref as O
        if O.have_reference():
            ref = {}
            for t in (1, 8):
                r = O.Reference(N, N, N, threads=t, iter=steps, acc=acc)
                r.set_mask(mask)
                for _ in range(steps):
                    r.run_one()
                ref[t] = {n: r.get(f) for f, n in ((O.DENS, "dens"), (O.VX, "v_x"), (O.VY, "v_y"), (O.VZ, "v_z"))}
            out["reference_1_thread_vs_8_threads"] = {n: rel(ref[8][n], ref[1][n]) for n in names.values()}
            out["gpu_gs_lex_vs_reference_1_thread"] = {n: rel(res["gs_lex"][n], ref[1][n]) for n in names.values()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

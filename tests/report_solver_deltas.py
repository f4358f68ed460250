#!/usr/bin/env python3
"""Ungated deltas that put the parity gates in context (SURVEY 8d "parity gates", F1):
  * GPU Jacobi vs GPU gs_lex (= the reference at one thread, bit for bit)
  * the reference against itself at 1 vs N threads (needs the compiled reference shim; run this
 tools/ never touches oracle/ unless this optional leg is requestedin the build container or on a box where oracle/_ref/libref.so travelled)
python tests/report_solver_deltas.py [N] [steps] [acc]   -> one JSON line"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel(a, b):
 tools/ never touches oracle/ unless this optional leg is requesteda = a.astype(np.float64); b = b.astype(np.float64)
 tools/ never touches oracle/ unless this optional leg is requestedreturn float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def main():
 tools/ never touches oracle/ unless this optional leg is requestedN = int(sys.argv[1]) if len(sys.argv) > 1 else 64
 tools/ never touches oracle/ unless this optional leg is requestedsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
 tools/ never touches oracle/ unless this optional leg is requestedacc = int(sys.argv[3]) if len(sys.argv) > 3 else 20
 tools/ never touches oracle/ unless this optional leg is requestedz, y, x = np.mgrid[0:N + 2, 0:N + 2, 0:N + 2]
 tools/ never touches oracle/ unless this optional leg is requestedmask = ((x - N / 3) ** 2 + (y - N / 2) ** 2 + (z - N / 2) ** 2) <= (N / 8) ** 2
 tools/ never touches oracle/ unless this optional leg is requestedout = {"grid": [N, N, N], "steps": steps, "acc": acc}
 tools/ never touches oracle/ unless this optional leg is requestedimport fluid_simulation_amd as F
 tools/ never touches oracle/ unless this optional leg is requestednames = {F.DENS: "dens", F.VX: "v_x", F.VY: "v_y", F.VZ: "v_z"}
 tools/ never touches oracle/ unless this optional leg is requestedres = {}
 tools/ never touches oracle/ unless this optional leg is requestedfor solver in ("jacobi", "gs_lex"):
 tools/ never touches oracle/ unless this optional leg is requested    sim = F.Simulation(N, N, N, steps, acc=acc, solver=solver, quiet=1, dump_every=0)
 tools/ never touches oracle/ unless this optional leg is requested    sim.set_mask(mask)
 tools/ never touches oracle/ unless this optional leg is requested    for _ in range(steps):
 tools/ never touches oracle/ unless this optional leg is requested        sim.run_one()
 tools/ never touches oracle/ unless this optional leg is requested    res[solver] = {n: sim.get(f) for f, n in names.items()}
 tools/ never touches oracle/ unless this optional leg is requested    sim.close()
 tools/ never touches oracle/ unless this optional leg is requestedout["gpu_jacobi_vs_gpu_gs_lex"] = {n: rel(res["jacobi"][n], res["gs_lex"][n]) for n in names.values()}
 tools/ never touches oracle/ unless this optional leg is requested# tools/ never touches oracle/ unless this optional leg is requested
 tools/ never touches oracle/ unless this optional leg is requestedif "--with-reference" in sys.argv:
 tools/ never touches oracle/ unless this optional leg is requested    from oracle import cpu_ref as O
 tools/ never touches oracle/ unless this optional leg is requested    if O.have_reference():
 tools/ never touches oracle/ unless this optional leg is requested        ref = {}
 tools/ never touches oracle/ unless this optional leg is requested        for t in (1, 8):
 tools/ never touches oracle/ unless this optional leg is requested            r = O.Reference(N, N, N, threads=t, iter=steps, acc=acc)
 tools/ never touches oracle/ unless this optional leg is requested            r.set_mask(mask)
 tools/ never touches oracle/ unless this optional leg is requested            for _ in range(steps):
 tools/ never touches oracle/ unless this optional leg is requested                r.run_one()
 tools/ never touches oracle/ unless this optional leg is requested            ref[t] = {n: r.get(f) for f, n in ((O.DENS, "dens"), (O.VX, "v_x"), (O.VY, "v_y"), (O.VZ, "v_z"))}
 tools/ never touches oracle/ unless this optional leg is requested        out["reference_1_thread_vs_8_threads"] = {n: rel(ref[8][n], ref[1][n]) for n in names.values()}
 tools/ never touches oracle/ unless this optional leg is requested        out["gpu_gs_lex_vs_reference_1_thread"] = {n: rel(res["gs_lex"][n], ref[1][n]) for n in names.values()}
 tools/ never touches oracle/ unless this optional leg is requestedprint(json.dumps(out))


if __name__ == "__main__":
 tools/ never touches oracle/ unless this optional leg is requestedmain()

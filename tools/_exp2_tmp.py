import sys, time
sys.path.insert(0, "/root/repo")
import fluid_simulation_amd as F
W, H, D, acc = (int(v) for v in sys.argv[2:6])
sim = F.Simulation(W, H, D, 1, acc=acc, quiet=1, dump_every=0, profile=1)
sim.set_option("sweep_fuse", sys.argv[1])
sim.run_one(); sim.sync(); sim.reset_timing()
t = time.perf_counter()
for _ in range(5): sim.run_one()
sim.sync()
ms, n = sim.timing("sweep_triple"); ms2, n2 = sim.timing("sweep_pair")
print("%dx%dx%d fuse %s: ms/step %.3f  triple pass %.4f ms (%d)  pair pass %.4f ms (%d) plan %d" % (W, H, D, sys.argv[1], (time.perf_counter() - t) / 5 * 1e3, ms / max(n, 1), n, ms2 / max(n2, 1), n2, sim._geti("triple_plan")), flush=True)

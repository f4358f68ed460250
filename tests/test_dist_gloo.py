"""world_size-2 (and 3) gloo runs on CPU of the host-side N>1 logic: slab bounds, unique-id
hand-out, max-over-ranks timing, and the halo rule -- the latter by running a slab-partitioned
Jacobi solve whose per-slab arithmetic is written out in numpy and whose halos move through
torch.distributed exactly as csrc/comm.h moves them, against the single-domain oracle."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from fluid_simulation_amd import dist as D
from oracle import cpu_ref as O

rank, local, world = D.env_ranks()
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, Dz, sweeps = 10, 8, 12, 4
zoff, dl, lo_wall, hi_wall = D.slab_bounds(Dz, rank, world)

# unique id: rank 0 makes it, everyone gets the same bytes
uid = D.share_unique_id(dist, lambda: bytes(range(128)), rank)
assert uid == bytes(range(128))
assert D.max_over_ranks(dist, 1.0 + rank) == float(world)

# "overlap" = "auto": the ranks' own clocks disagree (each would pick another schedule), the agreed plan does not
local = {0: 1.00 + 0.30 * rank, 1: 1.20 - 0.25 * rank, 2: 1.10 + (0.2 if rank == world - 1 else -0.3)}
mine = min(local, key=local.get)
plan, worst = D.choose_overlap_plan(dist, local)
plans = [None] * world
dist.all_gather_object(plans, (plan, sorted(worst.items()), mine))
assert len({(p, tuple(w)) for p, w, _ in plans}) == 1, plans            # one plan, one set of times, on every rank
assert len({m for _, _, m in plans}) > 1, plans                         # although the local favourites differ
for c in D.OVERLAP_CANDIDATES:
    assert worst[c] == max(({0: 1.00 + 0.30 * r, 1: 1.20 - 0.25 * r, 2: 1.10 + (0.2 if r == world - 1 else -0.3)})[c] for r in range(world))
# hysteresis: a later candidate within 1.5 %% of the best so far does not displace it
same = D.choose_overlap_plan(dist, {1: 1.000, 0: 0.990, 2: 0.991})[0]
assert same == 1, same
assert D.choose_overlap_plan(dist, {1: 1.000, 0: 0.980, 2: 0.979})[0] == 0

# single-domain truth from the oracle (every rank computes it; tiny)
rng = np.random.default_rng(3)
x = rng.standard_normal((Dz + 2, H + 2, W + 2)).astype(np.float32)
rhs = rng.standard_normal((Dz + 2, H + 2, W + 2)).astype(np.float32)
for a in (x, rhs):
    a[0, 0, :] = a[0, -1, :] = a[-1, 0, :] = a[-1, -1, :] = 0
    a[0, :, 0] = a[0, :, -1] = a[-1, :, 0] = a[-1, :, -1] = 0
    a[:, 0, 0] = a[:, 0, -1] = a[:, -1, 0] = a[:, -1, -1] = 0
ora = O.Oracle(W, H, Dz, solver=O.JACOBI, acc=sweeps, threads=1)
ora.set(O.VZ, x); ora.set(O.VZ0, rhs)
ora.linear_solver(3, O.VZ, O.VZ0, 0.7, 1 + 6 * 0.7)
truth = ora.get(O.VZ)

# the same solve on this rank's slab: one sweep at a time, walls only where they are physical
a32, inv = np.float32(0.7), np.float32(1.0) / np.float32(1 + 6 * 0.7)
loc = torch.from_numpy(x[zoff:zoff + dl + 2].copy())
r = rhs[zoff:zoff + dl + 2]
for it in range(sweeps):
    q = loc.numpy()
    n = q.copy()
    nb = (q[1:-1, 1:-1, 2:] + q[1:-1, 1:-1, :-2]) + q[1:-1, 2:, 1:-1]
    nb = nb + q[1:-1, :-2, 1:-1]
    nb = nb + q[2:, 1:-1, 1:-1]
    nb = nb + q[:-2, 1:-1, 1:-1]
    n[1:-1, 1:-1, 1:-1] = (r[1:-1, 1:-1, 1:-1] + a32 * nb) * inv
    n[1:-1, 1:-1, 0] = n[1:-1, 1:-1, 1]; n[1:-1, 1:-1, -1] = n[1:-1, 1:-1, -2]     # x faces (b=3: no flip)
    n[1:-1, 0, 1:-1] = n[1:-1, 1, 1:-1]; n[1:-1, -1, 1:-1] = n[1:-1, -2, 1:-1]     # y faces
    if lo_wall: n[0, 1:-1, 1:-1] = -n[1, 1:-1, 1:-1]                               # z walls flip for b=3
    if hi_wall: n[-1, 1:-1, 1:-1] = -n[-2, 1:-1, 1:-1]
    loc = D.exchange_halo_planes(dist, torch.from_numpy(n), rank, world)
got = loc.numpy()
lo = 0 if lo_wall else 1
hi = dl + 2 if hi_wall else dl + 1
want = truth[zoff:zoff + dl + 2]
assert np.array_equal(got[lo:hi].view(np.uint32), want[lo:hi].view(np.uint32)), "slab %%d differs" %% rank
assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "halo planes of slab %%d differ" %% rank
dist.destroy_process_group()
print("rank", rank, "ok")
'''


# Fused passes across slab boundaries (csrc/sweep_fused.hip, SLAB = true): a pass of `lv` sweeps is computed from
# lv-deep halos -- level j on the planes lv-j beyond the slab on every side that borders another slab -- and followed
# by ONE exchange whose depth is the next pass's level count.  Written out in numpy on zh-deep slabs, moved through
# torch.distributed with the depths fluid_simulation_amd.dist derives, against the single-domain oracle.
WORKER_DEEP = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from fluid_simulation_amd import dist as D
from oracle import cpu_ref as O

rank, local, world = D.env_ranks()
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, Dz, b = 10, 8, 12, 3
zoff, dl, lo_wall, hi_wall = D.slab_bounds(Dz, rank, world)
rng = np.random.default_rng(11)
x = rng.standard_normal((Dz + 2, H + 2, W + 2)).astype(np.float32)
rhs = rng.standard_normal((Dz + 2, H + 2, W + 2)).astype(np.float32)
for a in (x, rhs):
    a[0, 0, :] = a[0, -1, :] = a[-1, 0, :] = a[-1, -1, :] = 0
    a[0, :, 0] = a[0, :, -1] = a[-1, :, 0] = a[-1, :, -1] = 0
    a[:, 0, 0] = a[:, 0, -1] = a[:, -1, 0] = a[:, -1, -1] = 0
a32, inv = np.float32(0.7), np.float32(1.0) / np.float32(1 + 6 * 0.7)

for sweeps, can3, can2, fp64 in ((8, True, True, False), (5, False, True, True), (7, True, True, False), (3, False, False, False)):
    zh = D.halo_depth(fp64, W, dl)
    plan = D.pass_plan(sweeps, can3 and zh >= 3, can2)
    depths = D.exchange_depths(plan, zh)
    assert sum(plan) == sweeps and len(depths) == len(plan) and depths[-1] == zh
    ora = O.Oracle(W, H, Dz, solver=O.JACOBI, acc=sweeps, threads=1)
    ora.set(O.VZ, x); ora.set(O.VZ0, rhs)
    ora.linear_solver(b, O.VZ, O.VZ0, 0.7, 1 + 6 * 0.7)
    truth = ora.get(O.VZ)
    at = lambda z: z + zh - 1
    # local arrays with zh halos, filled from the global state where the planes exist (halos current, as at rest)
    def take(g):
        out = np.zeros((dl + 2 * zh, H + 2, W + 2), dtype=np.float32)
        for z in range(1 - zh, dl + zh + 1):
            gz = zoff + z
            if 0 <= gz <= Dz + 1:
                out[at(z)] = g[gz]
        return out
    loc, r = torch.from_numpy(take(x)), take(rhs)
    for lv, e in zip(plan, depths):
        q = loc.numpy()
        for j in range(1, lv + 1):
            lo = 1 if lo_wall else 1 - (lv - j)
            hi = dl if hi_wall else dl + (lv - j)
            n = q.copy()
            c = slice(at(lo), at(hi) + 1)
            up, dn = slice(at(lo) + 1, at(hi) + 2), slice(at(lo) - 1, at(hi))
            nb = (q[c, 1:-1, 2:] + q[c, 1:-1, :-2]) + q[c, 2:, 1:-1]
            nb = nb + q[c, :-2, 1:-1]
            nb = nb + q[up, 1:-1, 1:-1]
            nb = nb + q[dn, 1:-1, 1:-1]
            n[c, 1:-1, 1:-1] = (r[c, 1:-1, 1:-1] + a32 * nb) * inv
            n[c, 1:-1, 0] = n[c, 1:-1, 1]; n[c, 1:-1, -1] = n[c, 1:-1, -2]          # x faces (b = 3: no flip)
            n[c, 0, 1:-1] = n[c, 1, 1:-1]; n[c, -1, 1:-1] = n[c, -2, 1:-1]          # y faces
            if lo_wall: n[at(0), 1:-1, 1:-1] = -n[at(1), 1:-1, 1:-1]               # z walls flip for b = 3
            if hi_wall: n[at(dl + 1), 1:-1, 1:-1] = -n[at(dl), 1:-1, 1:-1]
            q = n
        loc = D.exchange_halo_planes(dist, torch.from_numpy(q), rank, world, depth=e, zh=zh)
    got = loc.numpy()
    lo = 0 if lo_wall else 1 - zh          # after the last exchange every halo plane is current again
    hi = dl + 1 if hi_wall else dl + zh
    for z in range(lo, hi + 1):
        gz = zoff + z
        assert np.array_equal(got[at(z)].view(np.uint32), truth[gz].view(np.uint32)), (rank, sweeps, plan, z)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_slab_logic_under_gloo(tmp_path, world, oracle_mod):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()[-2000:]


@pytest.mark.parametrize("world", [2, 3])
def test_fused_passes_across_deep_halos_under_gloo(tmp_path, world, oracle_mod):
    script = tmp_path / "worker_deep.py"
    script.write_text(WORKER_DEEP % {"root": ROOT})
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()[-2000:]


def test_pass_plan_and_exchange_depths():
    from fluid_simulation_amd.dist import exchange_depths, halo_depth, pass_plan
    assert pass_plan(80, True, True) == [3] * 26 + [2]
    assert pass_plan(80, False, True) == [2] * 40
    assert pass_plan(7, True, True) == [3, 3, 1] and pass_plan(5, True, True) == [3, 2]
    assert pass_plan(3, False, False) == [1, 1, 1] and pass_plan(0, True, True) == []
    assert pass_plan(4, True, True, rbsor=True) == [2, 2, 2, 2]
    assert exchange_depths([3, 3, 2], 3) == [3, 2, 3] and exchange_depths([2, 2, 1], 2) == [2, 1, 2]
    assert halo_depth(False, 512, 64) == 3 and halo_depth(False, 1024, 64) == 2 and halo_depth(True, 512, 64) == 2
    assert halo_depth(False, 20, 2) == 2


def test_slab_bounds():
    from fluid_simulation_amd.dist import slab_bounds
    assert slab_bounds(512, 0, 1) == (0, 512, True, True)
    assert [slab_bounds(512, r, 8)[:2] for r in range(8)] == [(64 * r, 64) for r in range(8)]
    assert slab_bounds(512, 0, 8)[2:] == (True, False) and slab_bounds(512, 7, 8)[2:] == (False, True)
    with pytest.raises(ValueError):
        slab_bounds(10, 0, 3)
    with pytest.raises(ValueError):
        slab_bounds(8, 2, 2)

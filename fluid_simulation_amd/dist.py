"""Host-side helpers for one-process-per-GPU runs (torch.distributed is plumbing only: the halo
traffic itself goes through RCCL inside libfluidsim.so).  Everything here works with the gloo
backend on CPU, which is how the N>1 host logic is tested without GPUs."""
import os


def env_ranks():
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def slab_bounds(depth, rank, nranks):
    """z-slab of `rank`: (z_offset, local_depth, lo_wall, hi_wall).  Global plane z (1-based)
    is local plane z - z_offset; the physical z walls belong to the first and last rank only.
    Mirrors fs::Comm::local_depth / z_offset (csrc/comm.h) and Engine::init (csrc/fluidsim.cpp)."""
    if nranks < 1 or not 0 <= rank < nranks:
        raise ValueError("bad rank %d of %d" % (rank, nranks))
    if depth % nranks:
        raise ValueError("depth %d does not divide over %d slabs" % (depth, nranks))
    dl = depth // nranks
    return rank * dl, dl, rank == 0, rank == nranks - 1


def share_unique_id(dist, make_id, rank, device=None):
    """Rank 0 creates the 128-byte communicator id, every rank receives it."""
    import torch
    buf = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        raw = make_id()
        buf.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    dist.broadcast(buf, src=0)
    return bytes(buf.cpu().numpy().tobytes())


def max_over_ranks(dist, value, device=None):
    """The slowest rank's time: bench.py reports whole-job throughput against it."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def halo_depth(fp64, width, local_depth):
    """Halo planes a slab keeps per side (Engine::init, csrc/fluidsim.cpp): as many as the deepest fused
    solver pass has levels -- three where the three-sweep kernel exists (fp32, rows up to 512 cells), else two."""
    return 3 if (not fp64 and width <= 512 and local_depth >= 3) else 2


def pass_plan(sweeps, can3, can2, rbsor=False):
    """The passes of one solve (Engine::solve): levels per pass -- three sweeps while at least three remain
    (where that kernel exists), then two, then one; an rbsor iteration is one two-level pass.  Every rank
    derives the same list: it fixes the depth of every halo exchange."""
    plan, left = [], sweeps
    while left > 0:
        if rbsor:
            plan.append(2)
            left -= 1
            continue
        lv = 3 if (can3 and left >= 3) else 2 if (can2 and left >= 2) else 1
        plan.append(lv)
        left -= lv
    return plan


def exchange_depths(plan, zh):
    """Planes exchanged per direction after each pass: what the NEXT pass needs (its level count); after
    the last pass the halos are brought to their full depth, which every other kernel assumes."""
    return [plan[i + 1] if i + 1 < len(plan) else zh for i in range(len(plan))]


def exchange_halo_planes(dist, field, rank, nranks, depth=1, zh=1):
    """Reference statement of the halo rule the C++ transports implement (csrc/comm.h
    exchange_halo), on a (D_local + 2*zh, ...) torch tensor whose index i holds local plane i - zh + 1:
    my planes 1..depth become the lower neighbour's planes D+1..D+depth, my planes D-depth+1..D the upper
    neighbour's planes 1-depth..0; wall planes are left alone."""
    n = field.shape[0] - 2 * zh                      # local depth D
    at = lambda z: z + zh - 1                        # noqa: E731  local plane -> index
    ops = []
    lo = hi = None
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, field[at(1):at(depth) + 1].contiguous(), rank - 1))
        lo = field[at(1 - depth):at(0) + 1].clone()
        ops.append(dist.P2POp(dist.irecv, lo, rank - 1))
    if rank < nranks - 1:
        ops.append(dist.P2POp(dist.isend, field[at(n - depth + 1):at(n) + 1].contiguous(), rank + 1))
        hi = field[at(n + 1):at(n + depth) + 1].clone()
        ops.append(dist.P2POp(dist.irecv, hi, rank + 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if lo is not None:
        field[at(1 - depth):at(0) + 1].copy_(lo)
    if hi is not None:
        field[at(n + 1):at(n + depth) + 1].copy_(hi)
    return field


OVERLAP_CANDIDATES = (1, 0, 2)   # the order Engine::choose_overlap times them in (csrc/fluidsim.cpp)


def choose_overlap_plan(dist, local_ms, device=None, hysteresis=0.985):
    """The rule by which z-slab ranks agree on a communication schedule ("overlap" = "auto",
    Engine::choose_overlap): every candidate's time is the SLOWEST rank's (all-reduce max), candidates are
    compared in a fixed order and a later one has to beat the best so far by 1.5 %.  `local_ms` maps
    candidate -> this rank's milliseconds per pass.  Every rank returns the same (plan, times)."""
    import torch
    t = torch.tensor([float(local_ms[c]) for c in OVERLAP_CANDIDATES], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    worst = [float(v) for v in t.cpu()]
    best, plan = float("inf"), OVERLAP_CANDIDATES[0]
    for c, ms in zip(OVERLAP_CANDIDATES, worst):
        if ms < best * hysteresis:
            best, plan = ms, c
    return plan, dict(zip(OVERLAP_CANDIDATES, worst))

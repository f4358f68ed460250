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


def exchange_halo_planes(dist, field, rank, nranks):
    """Reference statement of the halo rule the C++ transports implement (csrc/comm.h
    exchange_halo), on a (D_local+2, ...) torch tensor: my plane 1 becomes the lower neighbour's
    plane D+1, my plane D the upper neighbour's plane 0; wall planes are left alone."""
    ops = []
    lo = hi = None
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, field[1].contiguous(), rank - 1))
        lo = field[0].clone()
        ops.append(dist.P2POp(dist.irecv, lo, rank - 1))
    if rank < nranks - 1:
        ops.append(dist.P2POp(dist.isend, field[-2].contiguous(), rank + 1))
        hi = field[-1].clone()
        ops.append(dist.P2POp(dist.irecv, hi, rank + 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if lo is not None:
        field[0].copy_(lo)
    if hi is not None:
        field[-1].copy_(hi)
    return field

// comm.h -- z-slab communication over RCCL (xGMI), one process per GPU; plus two development transports for hosts
// with fewer GPUs than ranks: "FSIPC:" (ipc.h: stream-ordered device-to-device copies between rank processes, the
// asynchronous one) and "FSSHM:" (host-staged and fully synchronous).
//
// The reference is single-process ("Distributed communication
// backend: none"); this is new functionality.  z is the slowest memory axis
// (simulation.h:9), so a slab's boundary plane is one contiguous block of `sz` elements
// and a neighbour exchange is a grouped ncclSend/ncclRecv pair per direction on the
// solver's own stream (no host synchronisation inside the sweep loop).
//
// RCCL is loaded lazily with dlopen so that single-GPU users never map the 570 MB
// library.  It is taken from the directory of the HIP runtime this library is bound to (see
// RcclApi::get), so a PyTorch process and a plain process each get a consistent pair.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types and enums only; no link-time dependency

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstring>
#include <vector>
#include <string>

#include "kernels.h"
#include "ipc.h"

struct fs_sim;

namespace fs {

struct RcclApi {
    void* lib = nullptr;
    std::string path;                    // what was dlopen'ed (diagnostics)
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    static RcclApi& get(std::string* err)
    {
        static RcclApi api;
        if (api.lib) return api;
        // Load the RCCL that sits next to the HIP runtime this library is actually bound to.  A
        // process that imported PyTorch first runs on PyTorch's bundled libamdhip64 (same soname
        // as ROCm's) and must then use PyTorch's bundled RCCL; a plain process uses /opt/rocm's.
        // Mixing an RCCL with the other HIP runtime instance would hand it foreign streams.
        std::string dir;
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            dir = info.dli_fname;
            size_t slash = dir.rfind('/');
            dir = (slash == std::string::npos) ? std::string() : dir.substr(0, slash + 1);
        }
        const char* names[] = { "librccl.so.1", "librccl.so" };
        if (!dir.empty())
            for (const char* n : names)
                if ((api.lib = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL))) { api.path = dir + n; break; }
        if (!api.lib)
            for (const char* n : names)
                if ((api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) { api.path = n; break; }
        if (!api.lib) {
            if (err) *err = std::string("cannot load RCCL: ") + dlerror();
            return api;
        }
#define FS_SYM(field, name)                                                        \
        api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name));   \
        if (!api.field) { if (err) *err = std::string("RCCL symbol missing: ") + name; api.lib = nullptr; return api; }
        FS_SYM(GetUniqueId, "ncclGetUniqueId")
        FS_SYM(CommInitRank, "ncclCommInitRank")
        FS_SYM(CommDestroy, "ncclCommDestroy")
        FS_SYM(Send, "ncclSend")
        FS_SYM(Recv, "ncclRecv")
        FS_SYM(GroupStart, "ncclGroupStart")
        FS_SYM(GroupEnd, "ncclGroupEnd")
        FS_SYM(AllGather, "ncclAllGather")
        FS_SYM(AllReduce, "ncclAllReduce")
        FS_SYM(Broadcast, "ncclBroadcast")
        FS_SYM(GetErrorString, "ncclGetErrorString")
#undef FS_SYM
        return api;
    }
};

// Host-staged transport for development and tests: ranks are processes on one host (they may
// even share one GPU, which RCCL refuses), planes travel through a POSIX shared-memory
// segment with a sense-reversing barrier.  Fully synchronous and slow by design; it runs the
// same slab kernels, flags, wall handling and exchange schedule as the RCCL transport, so a
// 1-GPU box can verify that a z-slab run is bit-identical with the single-GPU run.
// Selected by a unique id that starts with "FSSHM:" followed by the segment name.
struct ShmTransport {
    struct Header {
        std::atomic<int> count;
        std::atomic<int> generation;
        std::atomic<int> ready;        // set by rank 0 once the segment is sized
        size_t slot_bytes;             // capacity of one mailbox slot (four fp64 planes); a gathered plane takes a quarter
        int gather_planes;
    };
    std::string name;
    int rank = 0, nranks = 1;
    Header* hdr = nullptr;
    char* base = nullptr;
    size_t total = 0;
    std::vector<char> bounce;

    static size_t layout(size_t slot, int nranks, int gplanes) { return 4096 + slot * ((size_t)2 * nranks + gplanes) + 4096; }
    char* mailbox(int r, int side) { return base + 4096 + hdr->slot_bytes * ((size_t)2 * r + side); }
    char* gather(int plane) { return base + 4096 + hdr->slot_bytes * ((size_t)2 * nranks) + (hdr->slot_bytes / 4) * (size_t)plane; }
    double* stats(int r) { return reinterpret_cast<double*>(base + 4096 + hdr->slot_bytes * ((size_t)2 * nranks + hdr->gather_planes)) + 4 * r; }   // gather area over-reserved

    // every rank calls this with the same sizes before the first exchange
    int ensure(size_t slot_bytes, int gplanes, std::string* err)
    {
        if (base) return 0;
        total = layout(slot_bytes, nranks, gplanes);
        int fd = -1;
        if (rank == 0) {
            fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { *err = "shm_open/ftruncate failed for " + name; return -1; }
        } else {
            for (int tries = 0; tries < 60000; ++tries) {
                fd = shm_open(name.c_str(), O_RDWR, 0600);
                struct stat sb;
                if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= total) break;
                if (fd >= 0) { close(fd); fd = -1; }
                usleep(1000);
            }
            if (fd < 0) { *err = "timed out waiting for shared segment " + name; return -1; }
        }
        void* m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { *err = "mmap failed for " + name; return -1; }
        base = static_cast<char*>(m);
        hdr = reinterpret_cast<Header*>(base);
        if (rank == 0) {
            hdr->slot_bytes = slot_bytes;
            hdr->gather_planes = gplanes;
            hdr->ready.store(1, std::memory_order_release);
        } else {
            while (hdr->ready.load(std::memory_order_acquire) != 1) usleep(200);
        }
        barrier();
        return 0;
    }
    void barrier()
    {
        const int gen = hdr->generation.load(std::memory_order_acquire);
        if (hdr->count.fetch_add(1, std::memory_order_acq_rel) == nranks - 1) {
            hdr->count.store(0, std::memory_order_relaxed);
            hdr->generation.fetch_add(1, std::memory_order_acq_rel);
        } else {
            while (hdr->generation.load(std::memory_order_acquire) == gen) sched_yield();
        }
    }
    void destroy()
    {
        if (base) {
            barrier();
            munmap(base, total);
            if (rank == 0) shm_unlink(name.c_str());
        }
        base = nullptr;
        hdr = nullptr;
    }
};

struct Comm {
    int rank = 0, nranks = 1;
    ncclComm_t comm = nullptr;
    RcclApi* api = nullptr;
    ShmTransport* shm = nullptr;
    IpcTransport* ipc = nullptr;   // "FSIPC:" ids (ipc.h)
    bool null_transport = false;   // "FSNULL:" ids: exchanges are skipped (compute-side timing of one slab; results invalid)
    bool null_push = false;
    std::string err;

    bool active() const { return nranks > 1; }
    const char* transport_name() const
    {
        return null_transport ? "none (FSNULL: timing only)" : shm ? "host shared memory (FSSHM: development)" :
               ipc ? "device-to-device copies between rank processes (FSIPC: hipIpc-mapped arrays, stream-ordered)" :
               api ? api->path.c_str() : "none";
    }
    const char* last_error() const { return err.c_str(); }
    int local_depth(int D) const { return D / nranks; }
    int z_offset(int D) const { return rank * (D / nranks); }

    static int unique_id(void* out128, std::string* e)
    {
        RcclApi& a = RcclApi::get(e);
        if (!a.lib) return -1;
        static_assert(sizeof(ncclUniqueId) == 128, "FS_COMM_ID_BYTES must match ncclUniqueId");
        ncclResult_t r = a.GetUniqueId(reinterpret_cast<ncclUniqueId*>(out128));
        if (r != ncclSuccess) { if (e) *e = a.GetErrorString(r); return -1; }
        return 0;
    }

    int init(int rank_, int nranks_, const void* id128)
    {
        const char* idc = static_cast<const char*>(id128);
        if (strncmp(idc, "FSNULL:", 7) == 0) {
            null_transport = true;
            // "FSNULL:push": the push schedule's kernels can be timed too -- a pass stores its boundary planes into the rank's
            // OWN halo planes (where a neighbour's would go), the handshakes are skipped
            null_push = strncmp(idc + 7, "push", 4) == 0;
            rank = rank_;
            nranks = nranks_;
            return 0;
        }
        if (strncmp(idc, "FSSHM:", 6) == 0) {
            shm = new ShmTransport;
            shm->name = std::string(idc + 6, strnlen(idc + 6, 120));
            shm->rank = rank_;
            shm->nranks = nranks_;
            rank = rank_;
            nranks = nranks_;
            return 0;
        }
        if (strncmp(idc, "FSIPC:", 6) == 0) {
            if (nranks_ > IPC_MAXR) { err = "the FSIPC transport carries at most 8 ranks"; return -1; }
            ipc = new IpcTransport;
            ipc->name = std::string(idc + 6, strnlen(idc + 6, 120));
            ipc->rank = rank_;
            ipc->nranks = nranks_;
            rank = rank_;
            nranks = nranks_;
            return 0;
        }
        RcclApi& a = RcclApi::get(&err);
        if (!a.lib) return -1;
        api = &a;
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        ncclResult_t r = a.CommInitRank(&comm, nranks_, id, rank_);
        if (r != ncclSuccess) { err = a.GetErrorString(r); comm = nullptr; return -1; }
        rank = rank_;
        nranks = nranks_;
        return 0;
    }

    void destroy()
    {
        if (comm && api) api->CommDestroy(comm);
        if (shm) { shm->destroy(); delete shm; shm = nullptr; }
        if (ipc) { ipc->destroy(); delete ipc; ipc = nullptr; }
        comm = nullptr;
        nranks = 1;
    }

#define FS_NCCL(call)                                                         \
    do {                                                                      \
        ncclResult_t r_ = (call);                                             \
        if (r_ != ncclSuccess) { err = api->GetErrorString(r_); return -1; }  \
    } while (0)
#define FS_HIPC(call)                                                         \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) { err = hipGetErrorString(e_); return -1; }     \
    } while (0)

    // Collective, FSIPC only: export allocation `slot` and map the peers' copies (neighbours, or all ranks for gather targets).
    int register_buffer(int slot, void* base, size_t bytes, bool all_ranks)
    {
        if (!ipc || !active()) return 0;
        return ipc->register_buffer(slot, base, bytes, all_ranks, &err);
    }
    // every rank has unmapped its peers' arrays: the owner may free them (FSIPC; call before the arrays are freed)
    void release_buffers()
    {
        if (ipc) { ipc->destroy(); delete ipc; ipc = nullptr; null_transport = true; }
    }
    // FSIPC only: solver passes may store their boundary planes straight into the neighbours' halo planes
    bool can_push() const { return (ipc != nullptr || null_push) && active(); }
    // the PeerPush offsets for a pass that writes the local array `dst` (plane z at dst + z * sz)
    int peer_push(const void* dst, const GridDesc& g, size_t elem, int planes, PeerPush* out)
    {
        *out = PeerPush();
        if (!can_push()) return 0;
        const char* d = static_cast<const char*>(dst);
        const long slab = (long)g.D * g.sz * (long)elem;
        if (null_push) {                                 // timing only: the rank's own halo planes stand in for the neighbours'
            if (rank > 0) out->lo = slab;
            if (rank < nranks - 1) out->hi = -slab;
            out->planes = planes;
            return 0;
        }
        if (rank > 0) {
            const char* p = ipc->peer_address(rank - 1, d, 1, &err);
            if (!p) return -1;
            out->lo = (long)(p - d) + slab;              // my plane z is the lower neighbour's plane D + z
        }
        if (rank < nranks - 1) {
            const char* p = ipc->peer_address(rank + 1, d, 1, &err);
            if (!p) return -1;
            out->hi = (long)(p - d) - slab;              // my plane z is the upper neighbour's plane z - D
        }
        out->planes = planes;
        return 0;
    }
    // both z neighbours have completed everything they queued before this point (stream-ordered, FSIPC)
    int handshake(hipStream_t st)
    {
        if (!can_push() || null_push) return 0;
        int peers[2], n = 0;
        if (rank > 0) peers[n++] = rank - 1;
        if (rank < nranks - 1) peers[n++] = rank + 1;
        return ipc->handshake(st, peers, n, &err);
    }

    // a bounded device-side wait gave up (lost peer): the results since then are void
    int check()
    {
        if (ipc && ipc->device_error()) {
            err = "FSIPC: a device-side wait for a peer timed out at operation " + std::to_string(ipc->device_error());
            return -1;
        }
        return 0;
    }

    int shm_ready(const GridDesc& g, int Dglobal)
    {
        // slots sized for fp64 planes so that one segment serves every field type
        return shm->ensure((size_t)g.sz * 8 * 4, Dglobal + 2, &err);   // a mailbox slot holds up to four planes
    }

    // Refresh the `depth` halo planes on each slab side of `a` (shifted pointer: plane z starts at
    // a + z*sz) from the neighbouring slabs' outermost `depth` interior planes.  Physical wall
    // planes (rank 0 low side, last rank high side) are left alone: the kernels write them.
    int exchange_halo(hipStream_t st, void* a, const GridDesc& g, size_t elem, int Dglobal, int depth)
    {
        if (!active() || null_transport) return 0;
        char* base = static_cast<char*>(a);
        const size_t plane = (size_t)g.sz * elem;
        const size_t bytes = plane * (size_t)depth;
        char* send_lo = base + plane;                                   // planes 1 .. depth
        char* recv_lo = base - (ptrdiff_t)(plane * (size_t)(depth - 1));   // planes 1-depth .. 0
        char* send_hi = base + plane * (size_t)(g.D - depth + 1);       // planes D-depth+1 .. D
        char* recv_hi = base + plane * (size_t)(g.D + 1);               // planes D+1 .. D+depth
        if (shm) {
            if (depth > 4) { err = "shm transport carries at most four planes"; return -1; }
            if (shm_ready(g, Dglobal)) return -1;
            FS_HIPC(hipStreamSynchronize(st));
            if (rank > 0) FS_HIPC(hipMemcpyAsync(shm->mailbox(rank - 1, 1), send_lo, bytes, hipMemcpyDeviceToHost, st));
            if (rank < nranks - 1) FS_HIPC(hipMemcpyAsync(shm->mailbox(rank + 1, 0), send_hi, bytes, hipMemcpyDeviceToHost, st));
            FS_HIPC(hipStreamSynchronize(st));
            shm->barrier();
            if (rank > 0) FS_HIPC(hipMemcpyAsync(recv_lo, shm->mailbox(rank, 0), bytes, hipMemcpyHostToDevice, st));
            if (rank < nranks - 1) FS_HIPC(hipMemcpyAsync(recv_hi, shm->mailbox(rank, 1), bytes, hipMemcpyHostToDevice, st));
            FS_HIPC(hipStreamSynchronize(st));
            shm->barrier();
            return 0;
        }
        if (ipc) {
            IpcTransport::Xfer x[2];
            int peers[2], n = 0;
            // my planes 1..depth land in the lower neighbour's planes D+1..D+depth, i.e. where I keep recv_hi
            if (rank > 0) { x[n] = { rank - 1, send_lo, recv_hi, bytes }; peers[n++] = rank - 1; }
            if (rank < nranks - 1) { x[n] = { rank + 1, send_hi, recv_lo, bytes }; peers[n++] = rank + 1; }
            return ipc->exchange(st, x, n, peers, n, &err);
        }
        FS_NCCL(api->GroupStart());
        if (rank > 0) {
            FS_NCCL(api->Send(send_lo, bytes, ncclInt8, rank - 1, comm, st));
            FS_NCCL(api->Recv(recv_lo, bytes, ncclInt8, rank - 1, comm, st));
        }
        if (rank < nranks - 1) {
            FS_NCCL(api->Send(send_hi, bytes, ncclInt8, rank + 1, comm, st));
            FS_NCCL(api->Recv(recv_hi, bytes, ncclInt8, rank + 1, comm, st));
        }
        FS_NCCL(api->GroupEnd());
        return 0;
    }

    // Assemble the whole global array (planes 0..Dglobal+1) of `src` in `dst` on every
    // rank: owned planes by all-gather, the two physical ghost planes by broadcast.
    int all_gather_planes(hipStream_t st, const void* src, void* dst, const GridDesc& g, int Dglobal, size_t elem)
    {
        const char* s = static_cast<const char*>(src);
        char* d = static_cast<char*>(dst);
        const size_t plane = (size_t)g.sz * elem;
        if (null_transport) return 0;
        if (shm) {
            if (shm_ready(g, Dglobal)) return -1;
            FS_HIPC(hipStreamSynchronize(st));
            const int zoff = z_offset(Dglobal);
            for (int z = 1; z <= g.D; ++z)
                FS_HIPC(hipMemcpyAsync(shm->gather(zoff + z), s + (size_t)z * plane, plane, hipMemcpyDeviceToHost, st));
            if (rank == 0) FS_HIPC(hipMemcpyAsync(shm->gather(0), s, plane, hipMemcpyDeviceToHost, st));
            if (rank == nranks - 1)
                FS_HIPC(hipMemcpyAsync(shm->gather(Dglobal + 1), s + (size_t)(g.D + 1) * plane, plane, hipMemcpyDeviceToHost, st));
            FS_HIPC(hipStreamSynchronize(st));
            shm->barrier();
            for (int z = 0; z <= Dglobal + 1; ++z)
                FS_HIPC(hipMemcpyAsync(d + (size_t)z * plane, shm->gather(z), plane, hipMemcpyHostToDevice, st));
            FS_HIPC(hipStreamSynchronize(st));
            shm->barrier();
            return 0;
        }
        if (ipc) return gather_window(st, src, dst, g, Dglobal, elem, Dglobal);   // a window that covers everything
        FS_NCCL(api->AllGather(s + plane, d + plane, plane * (size_t)g.D, ncclInt8, comm, st));
        FS_NCCL(api->Broadcast(s, d, plane, ncclInt8, 0, comm, st));
        FS_NCCL(api->Broadcast(s + (size_t)(g.D + 1) * plane, d + (size_t)(Dglobal + 1) * plane, plane, ncclInt8,
                               nranks - 1, comm, st));
        return 0;
    }

    // Planes of the global array that rank `i` reads when every back-trace stays within `reach`
    // planes of its own slab, as an inclusive global range.
    void window(int i, int Dglobal, int reach, int& lo, int& hi) const
    {
        const int dl = Dglobal / nranks;
        lo = i * dl + 1 - reach;
        hi = i * dl + dl + reach;
        if (lo < 0) lo = 0;
        if (hi > Dglobal + 1) hi = Dglobal + 1;
    }
    // Global planes rank `j` can provide: its interior planes plus the physical ghost plane it holds.
    void owned(int j, int Dglobal, int& lo, int& hi) const
    {
        const int dl = Dglobal / nranks;
        lo = (j == 0) ? 0 : j * dl + 1;
        hi = (j == nranks - 1) ? Dglobal + 1 : j * dl + dl;
    }

    // Windowed gather of the advection source: `dst` is the global array (planes 0..Dglobal+1);
    // on return it holds, for this rank, every plane within `reach` of its slab.  Each pair of
    // ranks whose window/ownership ranges intersect exchanges exactly that intersection
    // (grouped ncclSend/ncclRecv); the own planes are a device-to-device copy.
    int gather_window(hipStream_t st, const void* src, void* dst, const GridDesc& g, int Dglobal, size_t elem, int reach)
    {
        const char* s = static_cast<const char*>(src);
        char* d = static_cast<char*>(dst);
        const size_t plane = (size_t)g.sz * elem;
        const int zoff = z_offset(Dglobal);
        int mylo, myhi, ownlo, ownhi;
        window(rank, Dglobal, reach, mylo, myhi);
        owned(rank, Dglobal, ownlo, ownhi);
        if (d + (size_t)ownlo * plane != s + (ptrdiff_t)(ownlo - zoff) * (ptrdiff_t)plane)   // (in place: the planes are where they belong)
            FS_HIPC(hipMemcpyAsync(d + (size_t)ownlo * plane, s + (ptrdiff_t)(ownlo - zoff) * (ptrdiff_t)plane,
                                   plane * (size_t)(ownhi - ownlo + 1), hipMemcpyDeviceToDevice, st));
        if (null_transport) return 0;
        if (shm) {
            if (shm_ready(g, Dglobal)) return -1;
            // stage what others need of mine, then pick up what I need of theirs
            for (int j = 0; j < nranks; ++j) {
                if (j == rank) continue;
                int wlo, whi;
                window(j, Dglobal, reach, wlo, whi);
                const int lo = wlo > ownlo ? wlo : ownlo, hi = whi < ownhi ? whi : ownhi;
                for (int z = lo; z <= hi; ++z)
                    FS_HIPC(hipMemcpyAsync(shm->gather(z), s + (ptrdiff_t)(z - zoff) * (ptrdiff_t)plane, plane,
                                           hipMemcpyDeviceToHost, st));
            }
            FS_HIPC(hipStreamSynchronize(st));
            shm->barrier();
            for (int j = 0; j < nranks; ++j) {
                if (j == rank) continue;
                int olo, ohi;
                owned(j, Dglobal, olo, ohi);
                const int lo = mylo > olo ? mylo : olo, hi = myhi < ohi ? myhi : ohi;
                for (int z = lo; z <= hi; ++z)
                    FS_HIPC(hipMemcpyAsync(d + (size_t)z * plane, shm->gather(z), plane, hipMemcpyHostToDevice, st));
            }
            FS_HIPC(hipStreamSynchronize(st));
            shm->barrier();
            return 0;
        }
        if (ipc) {
            // push: what rank j needs of my planes goes straight into j's gathered array (same layout as mine)
            IpcTransport::Xfer x[IPC_MAXR];
            int recv_peers[IPC_MAXR], n = 0, nrecv = 0;
            for (int j = 0; j < nranks; ++j) {
                if (j == rank) continue;
                int wlo, whi, olo, ohi;
                window(j, Dglobal, reach, wlo, whi);
                int lo = wlo > ownlo ? wlo : ownlo, hi = whi < ownhi ? whi : ownhi;
                if (lo <= hi) x[n++] = { j, s + (ptrdiff_t)(lo - zoff) * (ptrdiff_t)plane, d + (size_t)lo * plane, plane * (size_t)(hi - lo + 1) };
                owned(j, Dglobal, olo, ohi);
                lo = mylo > olo ? mylo : olo;
                hi = myhi < ohi ? myhi : ohi;
                if (lo <= hi) recv_peers[nrecv++] = j;
            }
            return ipc->exchange(st, x, n, recv_peers, nrecv, &err);
        }
        FS_NCCL(api->GroupStart());
        for (int j = 0; j < nranks; ++j) {
            if (j == rank) continue;
            int wlo, whi, olo, ohi;
            window(j, Dglobal, reach, wlo, whi);
            int lo = wlo > ownlo ? wlo : ownlo, hi = whi < ownhi ? whi : ownhi;
            if (lo <= hi)
                FS_NCCL(api->Send(s + (ptrdiff_t)(lo - zoff) * (ptrdiff_t)plane, plane * (size_t)(hi - lo + 1), ncclInt8, j,
                                  comm, st));
            owned(j, Dglobal, olo, ohi);
            lo = mylo > olo ? mylo : olo;
            hi = myhi < ohi ? myhi : ohi;
            if (lo <= hi) FS_NCCL(api->Recv(d + (size_t)lo * plane, plane * (size_t)(hi - lo + 1), ncclInt8, j, comm, st));
        }
        FS_NCCL(api->GroupEnd());
        return 0;
    }

    // RCCL plumbing check that needs only one GPU: load the library, create a one-rank
    // communicator and push data through every collective the slab path uses (grouped
    // send/recv to self, all-gather, broadcast, all-reduce sum/min/max).
    static int selftest(hipStream_t st, std::string* e)
    {
        Comm c;
        char id[128];
        if (unique_id(id, e)) return -1;
        if (c.init(0, 1, id)) { *e = c.err; return -1; }
        c.nranks = 1;
        const int n = 4096;
        double *a = nullptr, *b = nullptr;
        std::vector<double> h(n), out(n);
        for (int i = 0; i < n; ++i) h[i] = 0.5 * i - 7.0;
        int rc = -1;
        do {
            if (hipMalloc((void**)&a, n * 8) != hipSuccess || hipMalloc((void**)&b, n * 8) != hipSuccess) { *e = "hipMalloc"; break; }
            if (hipMemcpyAsync(a, h.data(), n * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemsetAsync(b, 0, n * 8, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { *e = "hipMemcpyAsync"; break; }
            auto same = [&](const char* what) {
                if (hipStreamSynchronize(st) != hipSuccess) { *e = std::string("sync after ") + what; return false; }
                if (hipMemcpyAsync(out.data(), b, n * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess) { *e = "hipMemcpy back"; return false; }
                for (int i = 0; i < n; ++i)
                    if (out[i] != h[i]) { *e = std::string("wrong data after ") + what; return false; }
                if (hipMemsetAsync(b, 0, n * 8, st) != hipSuccess) { *e = "hipMemsetAsync"; return false; }
                return true;
            };
            ncclResult_t r;
#define ST(call, what) if ((r = (call)) != ncclSuccess) { *e = std::string(what) + ": " + c.api->GetErrorString(r); break; }
            ST(c.api->GroupStart(), "GroupStart");
            ST(c.api->Send(a, n * 8, ncclInt8, 0, c.comm, st), "Send");
            ST(c.api->Recv(b, n * 8, ncclInt8, 0, c.comm, st), "Recv");
            ST(c.api->GroupEnd(), "GroupEnd");
            if (!same("send/recv")) break;
            ST(c.api->AllGather(a, b, n * 8, ncclInt8, c.comm, st), "AllGather");
            if (!same("all-gather")) break;
            ST(c.api->Broadcast(a, b, n * 8, ncclInt8, 0, c.comm, st), "Broadcast");
            if (!same("broadcast")) break;
            ST(c.api->AllReduce(a, b, n, ncclDouble, ncclSum, c.comm, st), "AllReduce sum");
            if (!same("all-reduce sum")) break;
            ST(c.api->AllReduce(a, b, n, ncclDouble, ncclMin, c.comm, st), "AllReduce min");
            if (!same("all-reduce min")) break;
            ST(c.api->AllReduce(a, b, n, ncclDouble, ncclMax, c.comm, st), "AllReduce max");
            if (!same("all-reduce max")) break;
#undef ST
            rc = 0;
        } while (0);
        if (a) hipFree(a);
        if (b) hipFree(b);
        c.destroy();
        return rc;
    }

    // all ranks have finished everything queued before this call
    int barrier(hipStream_t st, double* d_scratch1)
    {
        if (!active() || null_transport) return 0;
        FS_HIPC(hipStreamSynchronize(st));
        if (shm) {
            if (!shm->base) { err = "shared segment not mapped yet"; return -1; }
            shm->barrier();
            return 0;
        }
        if (ipc) {
            if (ipc->open(&err)) return -1;
            ipc->barrier();
            return check();
        }
        FS_NCCL(api->AllReduce(d_scratch1, d_scratch1, 1, ncclDouble, ncclSum, comm, st));
        FS_HIPC(hipStreamSynchronize(st));
        return 0;
    }

    // in-place reductions of {sum, min, max} held in device memory as three doubles
    int reduce_stats(hipStream_t st, double* d3, const GridDesc& g, int Dglobal)
    {
        if (null_transport) return 0;
        if (shm) {
            if (shm_ready(g, Dglobal)) return -1;
            double loc[3];
            FS_HIPC(hipStreamSynchronize(st));
            FS_HIPC(hipMemcpyAsync(loc, d3, sizeof loc, hipMemcpyDeviceToHost, st));
            FS_HIPC(hipStreamSynchronize(st));
            memcpy(shm->stats(rank), loc, sizeof loc);
            shm->barrier();
            double out[3] = { 0.0, 1e300, -1e300 };
            for (int r = 0; r < nranks; ++r) {
                const double* q = shm->stats(r);
                out[0] += q[0];
                out[1] = q[1] < out[1] ? q[1] : out[1];
                out[2] = q[2] > out[2] ? q[2] : out[2];
            }
            shm->barrier();
            FS_HIPC(hipMemcpyAsync(d3, out, sizeof out, hipMemcpyHostToDevice, st));
            FS_HIPC(hipStreamSynchronize(st));
            return 0;
        }
        if (ipc) return ipc->reduce3(st, d3, &err);
        FS_NCCL(api->AllReduce(d3 + 0, d3 + 0, 1, ncclDouble, ncclSum, comm, st));
        FS_NCCL(api->AllReduce(d3 + 1, d3 + 1, 1, ncclDouble, ncclMin, comm, st));
        FS_NCCL(api->AllReduce(d3 + 2, d3 + 2, 1, ncclDouble, ncclMax, comm, st));
        return 0;
    }
#undef FS_NCCL
#undef FS_HIPC
};

}  // namespace fs

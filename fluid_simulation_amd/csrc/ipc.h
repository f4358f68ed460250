// ipc.h -- stream-ordered device-to-device transport between rank PROCESSES of one host ("FSIPC:" ids).
//
// The reference is single-process; this is new functionality, the second way (next to
// RCCL) a z-slab rank can move its halo planes: every rank exports its field arrays with hipIpcGetMemHandle,
// maps its peers' arrays with hipIpcOpenMemHandle, and an exchange is
//     tiny kernel:  tell my neighbours "my halo planes of this pass may be overwritten" ; wait for theirs
//     hipMemcpyAsync device-to-device of my boundary planes straight into the neighbours' halo planes
//     tiny kernel:  tell my neighbours "your halo planes have landed" ; wait for theirs
// all queued on ONE stream: nothing blocks the host, the compute stream never sees it, and the copies run on
// the copy engines (over xGMI between GPUs), not on CUs.  The handshake words live in a POSIX shared-memory
// segment that every rank registers with HIP (fine-grained host memory: GPU stores and loads at system scope).
// Ranks may share one GPU (RCCL refuses that), so a one-GPU development box can run the asynchronous
// overlap schedules between 2-4 rank processes; between GPUs of one node it is a real alternative to RCCL's
// send/recv kernels.  Every wait is bounded: a wave that waits longer than FS_IPC_TIMEOUT_S (default 30 s)
// records the operation number in the segment and returns, so no grid outlives a lost peer.
#pragma once
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace fs {

constexpr int IPC_MAXR = 8;      // ranks of one host
constexpr int IPC_MAXBUF = 24;   // exported allocations per rank

struct IpcSyncArgs {
    unsigned* store[IPC_MAXR];   // words that receive `seq`
    unsigned* wait[IPC_MAXR];    // words awaited until they reach `seq`
    int nstore, nwait;
    unsigned seq;
    unsigned* err;               // sticky: first operation whose wait timed out
    long long timeout_ticks;     // wall_clock64 ticks (100 MHz)
};

// One wave.  Stores first (so that two ranks that wait for each other both get what they wait for), then waits.
static __global__ void ipc_sync_kernel(IpcSyncArgs a)
{
    const int l = threadIdx.x;
    if (l < a.nstore) __hip_atomic_store(a.store[l], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (l < a.nwait) {
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(a.wait[l], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
            if (wall_clock64() - t0 > a.timeout_ticks) {          // the exit every wave reaches
                __hip_atomic_store(a.err, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

struct IpcReduceArgs {
    double* mine;                // this rank's row of four doubles in the segment
    const double* all;           // row r = rank r's four doubles
    unsigned* flag_mine;         // receives `seq` once `mine` is written
    unsigned* flags;             // word r = rank r's flag
    int nranks;
    unsigned seq;
    unsigned* err;
    long long timeout_ticks;
    double* d3;                  // device: {sum, min, max} in, reduced over the ranks out (rank order, every rank the same bits)
};

static __global__ void ipc_reduce_kernel(IpcReduceArgs a)
{
    const int l = threadIdx.x;
    if (l < 3) __hip_atomic_store(a.mine + l, a.d3[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (l == 0) __hip_atomic_store(a.flag_mine, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (l < a.nranks) {
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(a.flags + l, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
            if (wall_clock64() - t0 > a.timeout_ticks) {
                __hip_atomic_store(a.err, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (l == 0) {
        double s = 0.0, mn = 1e300, mx = -1e300;
        for (int r = 0; r < a.nranks; ++r) {
            const double v0 = __hip_atomic_load(a.all + 4 * r + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const double v1 = __hip_atomic_load(a.all + 4 * r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const double v2 = __hip_atomic_load(a.all + 4 * r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s += v0;
            mn = v1 < mn ? v1 : mn;
            mx = v2 > mx ? v2 : mx;
        }
        a.d3[0] = s;
        a.d3[1] = mn;
        a.d3[2] = mx;
    }
}

struct IpcTransport {
    struct Header {
        std::atomic<int> count, generation, ready, veto;
        unsigned error;                                  // written by the device (first timed-out operation), 0 = none
        hipIpcMemHandle_t handle[IPC_MAXR][IPC_MAXBUF];
        size_t bytes[IPC_MAXR][IPC_MAXBUF];
        // device-visible handshake words: rdy[r][s] -- receiver r is ready for sender s's data of operation `seq`;
        // done[r][s] -- sender s's data of operation `seq` has landed in receiver r
        alignas(64) unsigned rdy[IPC_MAXR][IPC_MAXR];
        alignas(64) unsigned done[IPC_MAXR][IPC_MAXR];
        alignas(64) unsigned red[2][IPC_MAXR];           // reductions: flag of rank r, two alternating rows
        alignas(64) double vals[2][IPC_MAXR][4];
    };
    struct Buf {
        char* base = nullptr;
        size_t bytes = 0;
        char* peer[IPC_MAXR] = {nullptr};
    };

    std::string name;
    int rank = 0, nranks = 1;
    Header* hdr = nullptr;     // host mapping
    Header* dhdr = nullptr;    // the same segment as the device addresses it
    Buf bufs[IPC_MAXBUF];
    unsigned seq = 0;          // operation counter: every rank issues the same operations in the same order
    unsigned nred = 0;         // reductions issued (selects the row of red / vals)
    unsigned nregistered = 0;  // allocations exported so far (every rank registers the same ones in the same order)
    long long timeout_ticks = 0;
    bool registered = false;
    long ops = 0, copies = 0;  // diagnostics

    int open(std::string* err)
    {
        if (hdr) return 0;
        if (nranks > IPC_MAXR) { *err = "the FSIPC transport carries at most 8 ranks"; return -1; }
        const size_t total = (sizeof(Header) + 4095) / 4096 * 4096;
        int fd = -1;
        if (rank == 0) {
            shm_unlink(name.c_str());                    // ids are unique per run; a leftover of that name would carry old flags
            fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
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
        hdr = static_cast<Header*>(m);
        if (rank == 0) hdr->ready.store(1, std::memory_order_release);   // a fresh segment is zero-filled
        else while (hdr->ready.load(std::memory_order_acquire) != 1) usleep(200);
        hipError_t e = hipHostRegister(hdr, total, hipHostRegisterMapped | hipHostRegisterPortable);
        void* d = nullptr;
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d, hdr, 0);
        if (e != hipSuccess) { *err = std::string("hipHostRegister of the handshake segment: ") + hipGetErrorString(e); return -1; }
        dhdr = static_cast<Header*>(d);
        registered = true;
        const char* t = getenv("FS_IPC_TIMEOUT_S");
        const double secs = t ? atof(t) : 30.0;
        timeout_ticks = (long long)((secs > 0.0 ? secs : 30.0) * 1e8);
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
        if (!hdr) return;
        barrier();                                       // nobody still copies into my arrays
        for (Buf& b : bufs)
            for (int r = 0; r < IPC_MAXR; ++r)
                if (b.peer[r]) { hipIpcCloseMemHandle(b.peer[r]); b.peer[r] = nullptr; }
        barrier();                                       // nobody still maps my arrays: the owner may free them
        if (registered) hipHostUnregister(hdr);
        munmap(hdr, (sizeof(Header) + 4095) / 4096 * 4096);
        if (rank == 0) shm_unlink(name.c_str());
        hdr = dhdr = nullptr;
    }
    unsigned device_error() const { return hdr ? *(volatile unsigned*)&hdr->error : 0u; }

    // Collective: every rank exports allocation `slot` (same slot, same size on every rank) and maps the copies
    // of the ranks it may write to -- its z neighbours, or all ranks (gather targets).
    int register_buffer(int slot, void* base, size_t bytes, bool all_ranks, std::string* err)
    {
        if (slot < 0 || slot >= IPC_MAXBUF) { *err = "FSIPC: buffer slot out of range"; return -1; }
        if (open(err)) return -1;
        // The owner marks the first and the last 16 bytes of the allocation with {magic, rank, slot, registration count}
        // before it exports it; whoever maps it reads both marks back through the mapping.  A mapping that points somewhere
        // else (a stale or aliased handle) is refused here instead of corrupting a run later.  The marks are cleared again:
        // exports happen on freshly zeroed allocations.
        if (bytes > ((size_t)2 << 30) - 4096) {
            // an export beyond 2 GiB never returns on the HIP runtime PyTorch bundles (round 3); nobody has entered a barrier yet
            *err = "FSIPC: an allocation of " + std::to_string(bytes >> 20) + " MiB cannot be exported (2 GiB per hipIpc handle); use more ranks, or RCCL";
            return -1;
        }
        ++nregistered;
        const bool trace = getenv("FS_IPC_TRACE") != nullptr;
        if (trace) fprintf(stderr, "fsipc rank %d: exporting allocation %d (%zu MiB)\n", rank, slot, bytes >> 20);
        const unsigned mark[4] = { 0xF51DC0DEu, (unsigned)rank, (unsigned)slot, nregistered };
        hipError_t e = (bytes >= 32) ? hipMemcpy(base, mark, sizeof mark, hipMemcpyHostToDevice) : hipErrorInvalidValue;
        if (e == hipSuccess) e = hipMemcpy(static_cast<char*>(base) + bytes - sizeof mark, mark, sizeof mark, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&hdr->handle[rank][slot], base);
        if (e != hipSuccess) { *err = std::string("exporting an allocation (hipIpcGetMemHandle): ") + hipGetErrorString(e); hdr->bytes[rank][slot] = 0; }
        else hdr->bytes[rank][slot] = bytes;
        barrier();
        int rc = (e == hipSuccess) ? 0 : -1;
        bufs[slot].base = static_cast<char*>(base);
        bufs[slot].bytes = bytes;
        for (int r = 0; r < nranks && !rc; ++r) {
            if (r == rank || !(all_ranks || r == rank - 1 || r == rank + 1)) continue;
            if (hdr->bytes[r][slot] != bytes) { *err = "FSIPC: rank " + std::to_string(r) + " exported a different size (or failed)"; rc = -1; break; }
            void* p = nullptr;
            e = hipIpcOpenMemHandle(&p, hdr->handle[r][slot], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { *err = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); rc = -1; break; }
            bufs[slot].peer[r] = static_cast<char*>(p);
            unsigned got[8] = {0};
            e = hipMemcpy(got, p, 16, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(got + 4, static_cast<char*>(p) + bytes - 16, 16, hipMemcpyDeviceToHost);
            const unsigned want[4] = { 0xF51DC0DEu, (unsigned)r, (unsigned)slot, nregistered };
            if (e != hipSuccess || memcmp(got, want, 16) != 0 || memcmp(got + 4, want, 16) != 0) {
                *err = "FSIPC: the mapping of rank " + std::to_string(r) + "'s allocation " + std::to_string(slot) +
                       " does not show that allocation (stale or aliased hipIpc handle)";
                rc = -1;
                break;
            }
        }
        if (trace) fprintf(stderr, "fsipc rank %d: allocation %d mapped (rc %d)\n", rank, slot, rc);
        barrier();                                       // everybody has looked: the marks go
        if (bytes >= 32) {
            (void)hipMemset(base, 0, 16);
            (void)hipMemset(static_cast<char*>(base) + bytes - 16, 0, 16);
            (void)hipDeviceSynchronize();
        }
        // all ranks or none: a rank whose mapping failed must not leave the others exchanging with it
        if (rc) hdr->veto.fetch_add(1, std::memory_order_acq_rel);
        barrier();
        if (!rc && hdr->veto.load(std::memory_order_acquire) != 0) { *err = "FSIPC: another rank could not map an allocation"; rc = -1; }
        return rc;
    }

    // where rank r keeps what I keep at `local`
    char* peer_address(int r, const void* local, size_t bytes, std::string* err)
    {
        const char* p = static_cast<const char*>(local);
        for (const Buf& b : bufs)
            if (b.base && p >= b.base && p + bytes <= b.base + b.bytes) {
                if (!b.peer[r]) { *err = "FSIPC: the array is not mapped from rank " + std::to_string(r); return nullptr; }
                return b.peer[r] + (p - b.base);
            }
        *err = "FSIPC: exchange on an array that was not exported";
        return nullptr;
    }

    // One exchange = `n` transfers {peer, my source, the address on MY side that corresponds to the peer's
    // destination, bytes}; transfers are symmetric (whoever I send to sends to me) or the peer lists are given
    // explicitly: send_peers = ranks I write to, recv_peers = ranks that write to me.
    struct Xfer { int peer; const void* src; const void* dst_as_local; size_t bytes; };
    int exchange(hipStream_t st, const Xfer* x, int n, const int* recv_peers, int nrecv, std::string* err)
    {
        if (open(err)) return -1;
        ++seq;
        ++ops;
        int send_peers[IPC_MAXR], nsend = 0;
        for (int i = 0; i < n; ++i) {
            bool seen = false;
            for (int k = 0; k < nsend; ++k) seen |= send_peers[k] == x[i].peer;
            if (!seen) send_peers[nsend++] = x[i].peer;
        }
        // ready: tell everyone who writes to me; wait for everyone I write to
        {
            IpcSyncArgs a;
            a.nstore = nrecv;
            a.nwait = nsend;
            a.seq = seq;
            a.err = &dhdr->error;
            a.timeout_ticks = timeout_ticks;
            for (int i = 0; i < nrecv; ++i) a.store[i] = &dhdr->rdy[rank][recv_peers[i]];
            for (int i = 0; i < nsend; ++i) a.wait[i] = &dhdr->rdy[send_peers[i]][rank];
            hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, a);
        }
        for (int i = 0; i < n; ++i) {
            char* dst = peer_address(x[i].peer, x[i].dst_as_local, x[i].bytes, err);
            if (!dst) return -1;
            hipError_t e = hipMemcpyAsync(dst, x[i].src, x[i].bytes, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) { *err = std::string("FSIPC copy: ") + hipGetErrorString(e); return -1; }
            ++copies;
        }
        // done: tell everyone I wrote to; wait for everyone who writes to me
        {
            IpcSyncArgs a;
            a.nstore = nsend;
            a.nwait = nrecv;
            a.seq = seq;
            a.err = &dhdr->error;
            a.timeout_ticks = timeout_ticks;
            for (int i = 0; i < nsend; ++i) a.store[i] = &dhdr->done[send_peers[i]][rank];
            for (int i = 0; i < nrecv; ++i) a.wait[i] = &dhdr->done[rank][recv_peers[i]];
            hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, a);
        }
        if (hipGetLastError() != hipSuccess) { *err = "FSIPC: launching the handshake kernels failed"; return -1; }
        return 0;
    }

    // "Everything I queued before this has completed" to each of `peers`, and wait for the same from them: what separates
    // two solver passes that store their boundary planes straight into the neighbours' halo planes (PeerPush, kernels.h).
    int handshake(hipStream_t st, const int* peers, int n, std::string* err)
    {
        if (open(err)) return -1;
        ++seq;
        ++ops;
        IpcSyncArgs a;
        a.nstore = a.nwait = n;
        a.seq = seq;
        a.err = &dhdr->error;
        a.timeout_ticks = timeout_ticks;
        for (int i = 0; i < n; ++i) {
            a.store[i] = &dhdr->done[peers[i]][rank];
            a.wait[i] = &dhdr->done[rank][peers[i]];
        }
        hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, a);
        if (hipGetLastError() != hipSuccess) { *err = "FSIPC: launching the handshake kernel failed"; return -1; }
        return 0;
    }

    // {sum, min, max} over the ranks, in place in device memory; every rank computes the same bits
    int reduce3(hipStream_t st, double* d3, std::string* err)
    {
        if (open(err)) return -1;
        ++seq;
        ++ops;
        const unsigned row = nred++ & 1u;
        IpcReduceArgs a;
        a.mine = dhdr->vals[row][rank];
        a.all = &dhdr->vals[row][0][0];
        a.flag_mine = &dhdr->red[row][rank];
        a.flags = &dhdr->red[row][0];
        a.nranks = nranks;
        a.seq = seq;
        a.err = &dhdr->error;
        a.timeout_ticks = timeout_ticks;
        a.d3 = d3;
        hipLaunchKernelGGL(ipc_reduce_kernel, dim3(1), dim3(64), 0, st, a);
        if (hipGetLastError() != hipSuccess) { *err = "FSIPC: launching the reduction kernel failed"; return -1; }
        return 0;
    }
};

}  // namespace fs

// comm.h -- z-slab communication over RCCL (xGMI), one process per GPU.
//
// The reference is single-process ("Distributed communication
// backend: none"); this is new functionality.  z is the slowest memory axis
// (simulation.h:9), so a slab's boundary plane is one contiguous block of `sz` elements
// and a neighbour exchange is a grouped ncclSend/ncclRecv pair per direction on the
// solver's own stream (no host synchronisation inside the sweep loop).
//
// RCCL is loaded lazily with dlopen so that single-GPU users never map the 570 MB
// library; if the host process (e.g. torch.distributed) already mapped an RCCL, that
// instance is reused.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types and enums only; no link-time dependency

#include <dlfcn.h>
#include <cstring>
#include <string>

#include "kernels.h"

struct fs_sim;

namespace fs {

struct RcclApi {
    void* lib = nullptr;
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
        const char* names[] = { "librccl.so.1", "librccl.so" };
        for (const char* n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!api.lib)
            for (const char* n : names)
                if ((api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
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

struct Comm {
    int rank = 0, nranks = 1;
    ncclComm_t comm = nullptr;
    RcclApi* api = nullptr;
    std::string err;

    bool active() const { return nranks > 1; }
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
        comm = nullptr;
        nranks = 1;
    }

#define FS_NCCL(call)                                                         \
    do {                                                                      \
        ncclResult_t r_ = (call);                                             \
        if (r_ != ncclSuccess) { err = api->GetErrorString(r_); return -1; }  \
    } while (0)

    // Refresh local planes 0 and D+1 of `a` (LEAD-shifted pointer) from the neighbouring
    // slabs' planes D and 1.  Physical wall planes (rank 0 low side, last rank high side)
    // are left alone: the kernels write them.
    int exchange_halo(hipStream_t st, void* a, const GridDesc& g, size_t elem)
    {
        if (!active()) return 0;
        char* base = static_cast<char*>(a);
        const size_t plane = (size_t)g.sz * elem;
        FS_NCCL(api->GroupStart());
        if (rank > 0) {
            FS_NCCL(api->Send(base + 1 * plane, plane, ncclInt8, rank - 1, comm, st));
            FS_NCCL(api->Recv(base + 0 * plane, plane, ncclInt8, rank - 1, comm, st));
        }
        if (rank < nranks - 1) {
            FS_NCCL(api->Send(base + (size_t)g.D * plane, plane, ncclInt8, rank + 1, comm, st));
            FS_NCCL(api->Recv(base + (size_t)(g.D + 1) * plane, plane, ncclInt8, rank + 1, comm, st));
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
        FS_NCCL(api->AllGather(s + plane, d + plane, plane * (size_t)g.D, ncclInt8, comm, st));
        FS_NCCL(api->Broadcast(s, d, plane, ncclInt8, 0, comm, st));
        FS_NCCL(api->Broadcast(s + (size_t)(g.D + 1) * plane, d + (size_t)(Dglobal + 1) * plane, plane, ncclInt8,
                               nranks - 1, comm, st));
        return 0;
    }

    // in-place reductions of {sum, min, max} held in device memory as three doubles
    int reduce_stats(hipStream_t st, double* d3)
    {
        FS_NCCL(api->AllReduce(d3 + 0, d3 + 0, 1, ncclDouble, ncclSum, comm, st));
        FS_NCCL(api->AllReduce(d3 + 1, d3 + 1, 1, ncclDouble, ncclMin, comm, st));
        FS_NCCL(api->AllReduce(d3 + 2, d3 + 2, 1, ncclDouble, ncclMax, comm, st));
        return 0;
    }
#undef FS_NCCL
};

}  // namespace fs

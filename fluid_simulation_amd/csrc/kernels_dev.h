// kernels_dev.h -- device-side helpers shared by the kernel translation units (kernels.hip,
// sweep_fused.hip).  Internal to libfluidsim.so.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace fs {

template <class T>
struct alignas(16) V4 {
    T e[4];
};

__device__ __forceinline__ long cell(const GridDesc& g, int x, int y, int z)
{
    return (long)x + (long)y * g.sy + (long)z * g.sz;
}

// Blocks are dealt round-robin over the 8 XCDs (block b lands on XCD b % 8, each with its
// own 4 MiB L2).  Remap so that every XCD owns one contiguous range of work items and
// y-adjacent tiles, which share halo rows, hit the same L2.  Affects speed only.
__device__ __forceinline__ int xcd_contiguous(int b, int nblk)
{
    int q = nblk >> 3, r = nblk & 7, k = b & 7;
    return k * q + (k < r ? k : r) + (b >> 3);
}

// Kernel-argument form of EdgeFirst: nblocks leading workgroups are boundary chunks of `planes` planes each, the first
// `nbands` of them for the region starting at `first`, the next `nbands` (if any) for the one starting at `second`.
struct EdgeArgs {
    int nblocks, first, second, planes;
    unsigned* counter;       // device memory: boundary workgroups finished so far (all passes)
    unsigned* signal;        // host-visible signal word a stream waits on (hipStreamWaitValue32)
    unsigned target;         // value of *counter once every boundary workgroup of this launch has finished
};

// A boundary workgroup has stored its planes: make them visible to the device (the halo exchange runs in another
// kernel, possibly while this one is still computing the interior) and count the workgroup in; the workgroup that
// completes the count publishes it in the signal word -- ONE write to host-visible memory per pass (256 atomics on
// the signal word itself cost 270 us per pass).
// Every storing wave drains its stores, the workgroup meets, one lane writes back L2 and counts
// (/opt/skills/guides/MI355X_MICROARCH.md, "Valid forms": plain stores + release fence + relaxed agent atomic).
__device__ __forceinline__ void edge_signal(const EdgeArgs& ea)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned before = __hip_atomic_fetch_add(ea.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == ea.target) __hip_atomic_store(ea.signal, ea.target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace fs

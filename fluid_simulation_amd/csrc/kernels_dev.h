// kernels_dev.h -- device-side helpers shared by the kernel translation units (kernels.hip,
// sweep_fused.hip, multigrid.hip).  Internal to libfluidsim.so.
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

// Ghost-face writes shared by the per-cell kernels: `u` is the un-zeroed new value of
// interior cell (x,y,z) of a field with boundary code b (simulation.cpp:187-215).
template <class T>
__device__ __forceinline__ void write_face_ghosts(const GridDesc& g, const SlabCtx& sc, T* q, long c, int x, int y,
                                                  int z, T u, int b)
{
    if (x == 1) q[c - 1] = (b == 1) ? -u : u;
    if (x == g.W) q[c + 1] = u;
    if (y == 1) q[c - g.sy] = (b == 2) ? -u : u;
    if (y == g.H) q[c + g.sy] = (b == 2) ? -u : u;
    if (z == 1 && sc.lo_wall) q[c - g.sz] = (b == 3) ? -u : u;
    if (z == g.D && sc.hi_wall) q[c + g.sz] = (b == 3) ? -u : u;
}

// A store of a solver pass, and (PUSH) its copies into the neighbours' halo planes: dl / dh = PeerPush::lo / hi where the
// plane being written is one the lower / upper neighbour needs, else 0 (wave-uniform).
template <bool PUSH, class V>
__device__ __forceinline__ void put(V* p, const V& v, long dl, long dh)
{
    *p = v;
    if constexpr (PUSH) {
        if (dl) *reinterpret_cast<V*>(reinterpret_cast<char*>(p) + dl) = v;
        if (dh) *reinterpret_cast<V*>(reinterpret_cast<char*>(p) + dh) = v;
    }
}

// Blocks are dealt round-robin over the 8 XCDs (block b lands on XCD b % 8, each with its
// own 4 MiB L2).  Remap so that every XCD owns one contiguous range of work items and
// y-adjacent tiles, which share halo rows, hit the same L2.  Affects speed only.
__device__ __forceinline__ int xcd_contiguous(int b, int nblk)
{
    int q = nblk >> 3, r = nblk & 7, k = b & 7;
    return k * q + (k < r ? k : r) + (b >> 3);
}

}  // namespace fs

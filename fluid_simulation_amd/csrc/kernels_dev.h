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

}  // namespace fs

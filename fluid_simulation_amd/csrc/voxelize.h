// voxelize.h -- GPU restatement of loadSTLIntoObstacles (object_loader.cpp:270-452).
#pragma once
#include <hip/hip_runtime_api.h>
#include <string>

#include "kernels.h"

namespace fs {

struct VoxelResult {
    long added = 0;          // accepted sample points ("Added N obstacle points", object_loader.cpp:451)
    int* d_cells = nullptr;  // device: `added` packed GLOBAL cell ids x + y*(W+2) + z*(W+2)*(H+2)
    void* d_work = nullptr;  // device scratch owned by the result
    long ntri = 0;
    int ns = 0;
    float resolution = 0;
    std::string error;
};

// Returns 0, FS_EIO (-2) when the STL cannot be read / holds no triangle, or FS_EHIP (-3).
int voxelize_stl(hipStream_t st, const char* path, int W, int H, int D, float scale, float rot_x, float rot_y,
                 float rot_z, float tr_x, float tr_y, float tr_z, unsigned seed, bool quiet, VoxelResult* out);
void voxelize_free(VoxelResult* r);

// obs[cell] = 1 for every listed global cell that falls into this slab
template <class T>
void launch_mark_cells(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* obs, const int* cells,
                       long n);

}  // namespace fs

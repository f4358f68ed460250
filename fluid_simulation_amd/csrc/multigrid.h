// multigrid.h -- the optional multigrid pressure solve (fs_set_option "solver" = "mg").  Internal to libfluidsim.so.
//
// NOT the reference's arithmetic: the reference only relaxes (simulation.cpp:251-273, linearSolver(0, p, div, 1, 6) at :320).
// This solves the same equation -- fluid cell: 6 p - sum of six neighbours = div, solid neighbours count 0 (:219-223),
// wall ghosts mirror (:187-215) -- by V-cycles; level 0 is the simulation grid smoothed by the red-black form of the
// reference's update (jacobi_pair_kernel<.., RB>), the coarser levels live here.  Definition and association order of
// every expression: oracle/cpu_ref_mg.h (the test-side statement of this mode; the two are compared bit for bit).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <vector>

#include "kernels.h"

namespace fs {

// One coarse level: dense padded arrays, element (x,y,z) at x + y*sy + z*sz, 0 <= x <= W+1.
template <class T>
struct MgLevel {
    int W, H, D;
    long sy, sz, n;
    T *wx, *wy, *wz;   // weight of the face on the minus side of a cell (wx[c] couples c-1 and c; x = 1..W+1)
    T *d, *dg;         // Dirichlet term (links to solid cells); diagonal (0 = dead cell: no links at all)
    T *e, *b;          // unknown (correction), right-hand side
    uint8_t* reg;      // 1 = regular cell: all six weights 1, d = 0 (diagonal 6): the kernels then skip the coefficient arrays
};

template <class T>
struct Multigrid {
    std::vector<MgLevel<T>> lv;      // lv[0] is unused (level 0 is the simulation grid); lv[1..] the coarse levels
    T* pool = nullptr;               // one allocation behind all level arrays
    uint8_t* reg_pool = nullptr;     // ... and one behind the regular-cell bytes
    int W0 = 0, H0 = 0, D0 = 0;

    int levels() const { return (int)lv.size(); }          // including level 0
    // (re)allocates for this grid if needed and computes the coefficients of every level from the flag bytes
    hipError_t build(hipStream_t st, const GridDesc& g, const uint8_t* flags);
    void release();
    // One V-cycle below level 0: restrict the level-0 residual of (p, rhs), recurse, and add the interpolated
    // correction to p (ghost faces of p rewritten, solids stay 0).  Level-0 smoothing is the caller's.
    void coarse_correction(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, T* p, const T* rhs,
                           int pre, int post, int coarse_iters);
};

}  // namespace fs

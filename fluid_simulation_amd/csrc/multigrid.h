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
#include <functional>
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
    // z-slab runs: D is this rank's share of the level's planes; plane z is global plane zoff + z (the red-black colour of a
    // cell follows the global z); a side without a physical wall has a halo plane of `e` that the neighbour owns
    int zoff = 0, lo_wall = 1, hi_wall = 1;
};

// What the host driver's transport does for the coarse levels of a z-slab run (null on one GPU).
template <class T>
struct MgHooks {
    // refresh the halo plane on each slab side of level array `a` (element (x, y, z) at a + x + y*sy + z*sz) from the neighbours
    std::function<int(const MgLevel<T>&, T* a)> halo;
    // `a` is an array of the whole (global) level that every rank holds; this rank has just written planes zoff+1 .. zoff+dl:
    // afterwards every rank holds every rank's planes
    std::function<int(const MgLevel<T>& global_level, T* a, int dl, int zoff)> gather;
};

template <class T>
struct Multigrid {
    std::vector<MgLevel<T>> lv;      // lv[0] is unused (level 0 is the simulation grid); lv[1..] the coarse levels
    T* pool = nullptr;               // one allocation behind all level arrays
    uint8_t* reg_pool = nullptr;     // ... and one behind the regular-cell bytes
    int W0 = 0, H0 = 0, D0 = 0;

    // z-slab runs: levels 1 .. first_repl-1 are distributed like level 0 (every rank holds its planes + one halo plane per
    // side, the smoother's colours are followed by a halo exchange); levels first_repl .. are small enough to be held
    // whole by every rank, which all compute them identically (no communication below that level).  One GPU: all "replicated".
    int first_repl = 1;
    int nranks = 1, rank = 0;
    size_t pool_elems = 0;

    int levels() const { return (int)lv.size(); }          // including level 0
    // (re)allocates for this grid if needed and computes the coefficients of every level from the flag bytes.
    // nranks > 1: g is the rank's slab; min_planes = fewest planes per rank a distributed level may have
    int build(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, int nranks_ = 1, int rank_ = 0,
              int min_planes = 4, const MgHooks<T>* hooks = nullptr);
    void release();
    // One V-cycle below level 0: restrict the level-0 residual of (p, rhs), recurse, and add the interpolated
    // correction to p (ghost faces of p rewritten, solids stay 0).  Level-0 smoothing is the caller's, and on a slab
    // the refresh of p's halo planes afterwards.
    int coarse_correction(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, T* p, const T* rhs,
                          int pre, int post, int coarse_iters, const MgHooks<T>* hooks = nullptr);

private:
    MgLevel<T> slab_view(int l, int fine_zoff, int fine_D, const SlabCtx& sc) const;
    void vcycle_replicated(hipStream_t st, int first, int pre, int post, int coarse_iters);
};

}  // namespace fs

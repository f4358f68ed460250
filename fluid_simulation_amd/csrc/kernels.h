// kernels.h -- launch interface between the host driver (fluidsim.cpp) and the gfx950
// kernels (kernels.hip, voxelize.hip).  Internal to libfluidsim.so.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace fs {

// HBM layout of one field (DESIGN.md "Data layout").  Element (x,y,z), 0 <= x <= W+1 etc,
// lives at p[x + y*sy + z*sz] where p is the allocation base shifted by LEAD elements, so
// that the first interior cell of every row (x = 1) is 16-byte aligned and a lane can
// move its four x-consecutive cells with one dwordx4.  sy is a multiple of 4.
struct GridDesc {
    int W, H, D;   // interior extents of THIS slab (D = local planes under z-slab partitioning)
    long sy, sz;   // row and plane pitch in elements
    long n;        // elements per allocation (including `lead` and the tail pad)
    long lead;     // elements between the allocation base and p: LEAD + (zh-1) planes
    int zh;        // halo planes kept on each z side: 1 (single GPU), 2 or 3 (z-slabs: planes 1-zh..D+zh
                   // exist so that two / three fused sweeps can cross a slab boundary)
};
constexpr int LEAD = 3;

// per-cell flag byte, same indexing as a field
enum : unsigned {
    F_SOLID = 1u,        // obs == 1                              (simulation.cpp:222)
    F_NEAR = 2u,         // fluid cell with an in-range solid 6-neighbour (simulation.cpp:232-238)
    F_XP = 4u, F_XM = 8u, F_YP = 16u, F_YM = 32u, F_ZP = 64u, F_ZM = 128u   // neighbour in range and obs == 0 (simulation.cpp:307-312)
};

// z-slab context: global z of local plane 1 is zoff+1; the physical z walls exist only on
// the first / last slab.
struct SlabCtx {
    int zoff;       // global index offset of this slab's planes
    int Dglobal;    // global depth
    int lo_wall;    // 1 if local plane 0 is the physical z=0 ghost plane
    int hi_wall;    // 1 if local plane D+1 is the physical z=Dglobal+1 ghost plane
};

// Tunables of the sweep launch (set through fs_set_option "sweep_ry" / "sweep_zc" /
// "sweep_blocks"; "sweep_abl" selects timing-only ablation builds used by tools/tune_sweep.py).
// One instance per handle (fs_sim::tune): options never leak between handles or host threads.
struct SweepTune {
    int ry = 2;               // rows per wave patch of the single-sweep kernel: 2 or 4
    int zc_len = 0;           // planes per z chunk; 0 = derive from target_blocks
    int target_blocks = 2048; // aim for about this many workgroups per launch
    int abl = 0;
    int cu_slots = 256;       // CUs a launch of this handle can fill (fewer when the compute stream carries a CU mask, z-slabs with
                              // "comm_cus"): what the launchers' z-chunk models divide the workgroups over
    int fuse = 3;             // sweeps fused per pass over memory: 1 never, 2 two-sweep kernels only, 3 (default) also time the
                              // three-sweep kernel per grid and use it where a sweep costs less, 4 use it wherever it exists
    int pair_zc = 0;          // planes per z chunk of the pair kernel; 0 = automatic
    int project_cell = 0;     // 1 = per-cell divergence/gradient kernels instead of the z-marching ones
    int pair_shape = 0;       // >0 forces a pair-kernel workgroup shape (1 = 8, 2 = 10, 3 = 16 waves); 0 = timed choice
    int advect_cell = 1;      // 1 (default) = per-cell advection kernels; 2 = the same with clamp tables; 3 = the tile kernels (clamp
                              // tables staged in LDS, for rough flows; single GPU, else as 2); 0 = the row kernels (four
                              // cells per lane + clamp tables); bit-identical, within 5 % of each other (profiles/r03e_*, r02g_*)
    int advect_window = 24;   // advect_cell == 3 (the tile kernels): rows / planes around a tile whose inlet-table values are staged in
                              // LDS; a trace that ends further away takes the per-cell path (capped by what 64 KB of LDS hold)
    int wall_free = 1;        // three-sweep kernel, whole-domain aligned grids: plane iterations that touch no y / z wall run a wall-free
                              // second body -- 0 never, 1 (default) and 2 always (round 3: selected per group of three iterations)
    int two_kind = 0;         // which two-sweep kernel: 0 = timed choice, 1 = jacobi_pair_kernel only, 2 = jacobi_fused_kernel<NL=2> only
};

// z-slab "push" exchange (FSIPC transport, csrc/ipc.h): a solver pass stores the planes its neighbours need next straight
// into THEIR halo planes (peer-mapped arrays) beside its own -- no boundary launch, no copy, no second stream.
// lo / hi = byte offset from a cell of `dst` to the same cell of the lower / upper neighbour's copy of that plane in the
// neighbour's halo (0 = no neighbour on that side), planes = how many of the slab's outermost planes per side go over.
struct PeerPush {
    long lo = 0, hi = 0;
    int planes = 0;
};

// NOTE: the sweep launchers take the KILL-byte array (launch_build_kill), not the flag bytes.
template <class T>
void launch_jacobi(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* src, const T* rhs,
                   T* dst, const uint8_t* kill, int b, T a, T inv_c, int z_first, int z_last, int second_first = -1,
                   const PeerPush* push = nullptr);
// second_first >= 0: ALSO compute the equally long range starting there, in the same launch (a
// slab's two boundary regions); push: see PeerPush (plain Jacobi passes over one range only)

// Two sweeps in one pass (temporal blocking); same result as two launch_jacobi calls.
// Needs W <= 1024; on a z-slab additionally two halo planes per side (g.zh == 2), current in
// `src`, and one current halo plane of `rhs` and `flags`.
template <class T>
bool pair_supported(const SweepTune& tune, const GridDesc& g, const SlabCtx& sc);
template <class T>
void launch_jacobi_pair(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* src, const T* rhs,
                        T* dst, const uint8_t* flags, int b, T a, T inv_c, int z_first, int z_last, int shape,
                        int second_first = -1, T omega = (T)0, bool damped = false, const PeerPush* push = nullptr);
// omega != 0: one red-black SOR iteration instead; with `damped`, two Jacobi sweeps damped by omega (q + omega*(r - q))
// NL = `levels` (2 or 3) sweeps per pass, register-centred (sweep_fused.hip): fp32 x 3 for rows up to 512
// cells, fp32 x 2 for rows of 513..1024 cells, fp64 x 2 for rows up to 512 cells.  On a z-slab `src` needs
// `levels` current halo planes per side, `rhs` and `flags` levels-1.  plan = workgroup shape
// (0 .. fused_shape_count-1) + 8 * (which of the launcher's three best z-chunk counts); second_first as above.
template <class T>
bool fused_supported(const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int levels);
template <class T>
int fused_shape_count(const GridDesc& g, int levels);
template <class T>
void launch_jacobi_fused(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int levels, const T* src,
                         const T* rhs, T* dst, const uint8_t* flags, int b, T a, T inv_c, int z_first, int z_last, int plan,
                         int second_first = -1, const PeerPush* push = nullptr);
// number of workgroup shapes (0 .. count-1) worth timing for this grid; results do not depend on the shape
template <class T>
int pair_shape_count(const GridDesc& g);

template <class T>
void launch_gs_lex(hipStream_t st, const GridDesc& g, T* q, const T* rhs, const uint8_t* flags, int b, T a, T inv_c,
                   int sweeps);

template <class T>
void launch_set_bounds(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* q, const uint8_t* flags, int b);

template <class T>
void launch_divergence(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* vx, const T* vy,
                       const T* vz, T* div, T* p, const uint8_t* flags, T mhalf_h);

template <class T>
void launch_gradient(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* p, T* vx, T* vy,
                     T* vz, const uint8_t* flags, T h, T two_h);

// `prev_zshift` = element offset added to local indices of `prev` (0 normally; under
// z-slabs `prev` is the all-gathered global array and the shift is zoff planes).
// `kill` = the kill-byte array, `coltab` = scratch for the clamp tables of the row kernels: 6 * (H+2) * (D+2)
// elements, or nullptr (no tables; also ignored on slabs).  tune.advect_cell selects the per-cell kernels.
template <class T>
void launch_advect(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int b, T* field, const T* prev,
                   const T* vx, const T* vy, const T* vz, const uint8_t* flags, const uint8_t* kill, T* coltab, T kx, T ky, T kz,
                   long prev_zshift);

// advect(1,vx,px); advect(2,vy,py); advect(3,vz,pz) in one pass.  On a slab px/py/pz are the
// gathered global arrays and zshift the element offset of this slab's plane 0 in them.
template <class T>
void launch_advect_velocity(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, T* vx, T* vy, T* vz,
                            const T* px, const T* py, const T* pz, const uint8_t* flags, const uint8_t* kill, T* coltab, T kx,
                            T ky, T kz, long zshift);

template <class T>
void launch_build_flags(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const T* obs, uint8_t* flags);

// one kill byte per four cells for the sweep kernels (see kernels.hip); `kill` is shifted so
// that byte (cell + 3) / 4 belongs to the lane group starting at `cell`
void launch_build_kill(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, uint8_t* kill);

template <class T>
void launch_inlet_velocity(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* vx, T* vy, T* vz, T speed);
template <class T>
void launch_inlet_density(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* dens, T amount);

// dense (reference layout, (W+2)(H+2)(D+2), x fastest) <-> pitched device layout, with
// element-type conversion.  zlo..zhi (inclusive, local planes) select the planes moved.
template <class T, class U>
void launch_pack(hipStream_t st, const GridDesc& g, const T* field, U* dense, int zlo, int zhi);
template <class T, class U>
void launch_unpack(hipStream_t st, const GridDesc& g, const U* dense, T* field, int zlo, int zhi);

template <class T>
void launch_copy(hipStream_t st, const GridDesc& g, const T* src, T* dst);

// sum / min / max over the padded box; out = 3 doubles on the device
template <class T>
void launch_stats(hipStream_t st, const GridDesc& g, const T* field, double* out3, double* scratch, int nscratch,
                  int zlo, int zhi);

template <class T>
void launch_point_add(hipStream_t st, T* p, long idx, T amount, int set_instead);

}  // namespace fs

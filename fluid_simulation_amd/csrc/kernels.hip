// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the wind-tunnel step.
//
// Everything here is HBM-bound stencil / gather work (0.67 flop per byte for the sweep), so
// there is no MFMA anywhere: the design points are 16-byte-per-lane coalesced streams,
// register/LDS reuse of the six stencil neighbours, wave shuffles for the x neighbours,
// XCD-aware block order, and boundary handling fused into the producing kernel.
//
// Arithmetic follows the reference expression by expression (cited per kernel) and the
// file is compiled with -ffp-contract=off, so fp32 results are bit-identical with the
// reference's `c++ -O2` build for the same iteration order.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "kernels_dev.h"

namespace fs {

// =====================================================================================
// Jacobi sweep with fused setBounds.
//   linearSolver  simulation.cpp:251-273 (one iteration of the k loop, neighbours read
//                 from the previous iterate `src`), followed by
//   setBounds     simulation.cpp:183-246 applied to the new iterate `dst`.
//
// Work decomposition: a wave owns 256 x-consecutive cells (4 per lane, one dwordx4) by RY
// rows and marches along z keeping planes z-1, z, z+1 of its patch in registers, so each
// value of `src` is fetched from memory once per sweep (plus halo rows/columns, which are
// L1/L2 hits).  x neighbours cross lanes with a wave shuffle; only the wave's edge lanes
// touch memory for them.  A block is four waves stacked in y.
// =====================================================================================
// Everything a wave needs of one z plane when that plane is the stencil centre: its own
// RY x 4 patch, the rows above and below the patch, and the two columns beside it.
template <class T, int RY>
struct PlaneIn {
    T core[RY][4];
    T hb[4], ht[4];      // rows y0-1 and y0+RY
    T eL[RY], eR[RY];    // columns x0-1 and x0+4 (only the wave's edge lanes load them)
};
template <class T, int RY>
struct AuxIn {
    T rhs[RY][4];
    unsigned fl[RY];     // kill byte of the lane's four cells: bits 0-3 solid, bits 4-7 solid-or-near
};

template <class T, int RY, int ABL, bool PUSH>
__global__ __launch_bounds__(256) void jacobi_sweep_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ src,
                                                            const T* __restrict__ rhs, T* __restrict__ dst,
                                                            const uint8_t* __restrict__ flags, int b, T a, T inv_c,
                                                            int z_first, int z_last, int zc_len, int z_stride, int nxw,
                                                            int nybg, int nblk, PeerPush pp)
{
    const int v = xcd_contiguous(blockIdx.x, nblk);
    const int xw = v % nxw;
    const int ybg = (v / nxw) % nybg;
    const int zc = v / (nxw * nybg);

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform, and the compiler may know
    const int W = g.W, H = g.H, D = g.D;
    const int y0 = 1 + (ybg * 4 + wv) * RY;
    if (y0 > H) return;                                  // wave-uniform
    const int x0 = 1 + xw * 256 + lane * 4;
    const bool lane_on = x0 <= W;
    const int zbeg = z_first + zc * z_stride;            // z_stride == zc_len except for the two-range launch
    const int zend = min(z_last, zbeg + zc_len - 1);
    if (zbeg > zend) return;

    const bool full_group = (x0 + 3 <= W);
    const bool edge_l = lane_on && (lane == 0);                                   // x0-1 comes from memory
    const bool edge_r = lane_on && full_group && ((lane == 63) || (x0 + 4 > W));  // x0+4 comes from memory
    const int kill_shift = (b == 0) ? 0 : 4;             // which nibble of the kill byte applies (simulation.cpp:222 vs :240)
    const T zero = (T)0;
    const long row0 = cell(g, x0, y0, 0);                // lane's first cell in plane 0

    auto ld4 = [&](const T* ptr, bool on, T (&out)[4]) {
        V4<T> q = {{zero, zero, zero, zero}};
        if (on) q = *reinterpret_cast<const V4<T>*>(ptr);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = q.e[e];
    };
    // rows up to the ghost row H+1 are fetched: in a partial band it is some row's y+1 neighbour
    auto fetch_core = [&](int z, T (&out)[RY][4]) {
        const T* pz = src + row0 + (long)z * g.sz;
#pragma unroll
        for (int r = 0; r < RY; ++r) ld4(pz + r * g.sy, lane_on && (y0 + r <= H + 1), out[r]);
    };
    auto fetch_plane = [&](int z, PlaneIn<T, RY>& P, bool as_centre) {
        fetch_core(z, P.core);
        const T* pz = src + row0 + (long)z * g.sz;
        const bool halo = as_centre && !(ABL & 2);
        ld4(pz - g.sy, lane_on && halo, P.hb);
        ld4(pz + RY * g.sy, lane_on && halo && (y0 + RY <= H + 1), P.ht);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool row_on = as_centre && (y0 + r <= H);
            P.eL[r] = (edge_l && row_on) ? pz[r * g.sy - 1] : zero;
            P.eR[r] = (edge_r && row_on) ? pz[r * g.sy + 4] : zero;
        }
    };
    auto fetch_aux = [&](int z, AuxIn<T, RY>& X) {
        const long off = row0 + (long)z * g.sz;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool on = lane_on && (y0 + r <= H);
            ld4(rhs + off + r * g.sy, on && !(ABL & 4), X.rhs[r]);
            X.fl[r] = (on && !(ABL & 1)) ? (unsigned)flags[(off + r * g.sy + 3) >> 2] : 0u;
        }
    };

    // Software pipeline: while plane z is computed, the loads of plane z+2 (and of the rhs /
    // flags of plane z+1) are in flight, so a wave waits for memory once per plane instead of
    // once per dependent load.
    T m[RY][4];
    PlaneIn<T, RY> A, B, N;
    AuxIn<T, RY> xa, xn;
    fetch_core(zbeg - 1, m);
    fetch_plane(zbeg, A, true);
    fetch_aux(zbeg, xa);
    fetch_plane(zbeg + 1, B, zbeg + 1 <= zend);

    for (int z = zbeg; z <= zend; ++z) {
        if (z + 1 <= zend) {                              // wave-uniform
            fetch_plane(z + 2, N, z + 2 <= zend);
            fetch_aux(z + 1, xn);
        }

#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = y0 + r;
            // x neighbours across lanes (all lanes execute the shuffles)
            T left = __shfl_up(A.core[r][3], 1);
            T right = __shfl_down(A.core[r][0], 1);
            if (edge_l) left = A.eL[r];
            if (edge_r) right = A.eR[r];
            if (y <= H && lane_on) {                      // y <= H is wave-uniform
                const long base = row0 + (long)z * g.sz + r * g.sy;
                const unsigned fl = xa.fl[r];
                T u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    T xp1 = (e < 3) ? A.core[r][e + 1] : right;
                    T xm1 = (e > 0) ? A.core[r][e - 1] : left;
                    T yp1 = (r < RY - 1) ? A.core[r + 1][e] : A.ht[e];
                    T ym1 = (r > 0) ? A.core[r - 1][e] : A.hb[e];
                    // simulation.cpp:264-269: order x+1, x-1, y+1, y-1, z+1, z-1
                    T nb = xp1 + xm1 + yp1 + ym1 + B.core[r][e] + m[r][e];
                    u[e] = (xa.rhs[r][e] + a * nb) * inv_c;
                }

                // ---- fused setBounds on the new iterate (faces read un-zeroed values) ----
                V4<T> st;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int x = x0 + e;
                    const bool kill = ((fl >> (e + kill_shift)) & 1u) != 0;
                    T ghost_src = (e > 0) ? u[e - 1] : zero;
                    // cells past W: the outflow ghost copies u(W) (:191); row padding stays 0
                    st.e[e] = (x <= W) ? (kill ? zero : u[e]) : ((x == W + 1) ? ghost_src : zero);
                }
                const long dl = (PUSH && z <= pp.planes) ? pp.lo : 0, dh = (PUSH && z > D - pp.planes) ? pp.hi : 0;   // wave-uniform
                if (!(ABL & 8) || st.e[0] == (T)123456789) put<PUSH>(reinterpret_cast<V4<T>*>(dst + base), st, dl, dh);
                if (x0 == 1) put<PUSH>(dst + base - 1, (b == 1) ? -u[0] : u[0], dl, dh);        // :189-190
                if (full_group && x0 + 3 == W) put<PUSH>(dst + base + 4, u[3], dl, dh);         // :191
                const bool yface = (y == 1) || (y == H);                                        // wave-uniform
                const bool zface = (z == 1 && sc.lo_wall) || (z == D && sc.hi_wall);            // wave-uniform
                if (yface || zface) {
                    V4<T> gy, gz;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool in = (x0 + e <= W);
                        gy.e[e] = in ? ((b == 2) ? -u[e] : u[e]) : zero;                        // :198-201
                        gz.e[e] = in ? ((b == 3) ? -u[e] : u[e]) : zero;                        // :208-214
                    }
                    if (y == 1) put<PUSH>(reinterpret_cast<V4<T>*>(dst + base - g.sy), gy, dl, dh);
                    if (y == H) put<PUSH>(reinterpret_cast<V4<T>*>(dst + base + g.sy), gy, dl, dh);
                    if (z == 1 && sc.lo_wall) *reinterpret_cast<V4<T>*>(dst + base - g.sz) = gz;
                    if (z == D && sc.hi_wall) *reinterpret_cast<V4<T>*>(dst + base + g.sz) = gz;
                }
            }
        }

#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) m[r][e] = A.core[r][e];
        A = B;
        B = N;
        xa = xn;
    }
}

template <class T, int RY, int ABL>
static void launch_jacobi_v(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* src,
                            const T* rhs, T* dst, const uint8_t* flags, int b, T a, T inv_c, int z_first, int z_last,
                            int second_first, const PeerPush* push)
{
    const int nxw = (g.W + 255) / 256;
    const int nyb = (g.H + RY - 1) / RY;
    const int nybg = (nyb + 3) / 4;
    const int planes = z_last - z_first + 1;
    const long per_layer = (long)nxw * nybg;
    if (second_first >= 0) {
        // two equally long ranges (the slab's two boundary regions) as two chunks of one launch
        const int last2 = second_first + planes - 1;
        hipLaunchKernelGGL((jacobi_sweep_kernel<T, RY, ABL, false>), dim3((unsigned)(per_layer * 2)), dim3(256), 0, st, g, sc, src,
                           rhs, dst, flags, b, a, inv_c, z_first, last2, planes, second_first - z_first, nxw, nybg,
                           (int)(per_layer * 2), PeerPush());
        return;
    }
    // enough z chunks for ~target_blocks blocks, but chunks of at least 8 planes (each chunk
    // re-reads 2 warm-up planes)
    long want = (tune.target_blocks + per_layer - 1) / per_layer;
    if (want < 1) want = 1;
    int zc_len = (int)((planes + want - 1) / want);
    if (zc_len < 8) zc_len = planes < 8 ? planes : 8;
    if (tune.zc_len > 0) zc_len = tune.zc_len < planes ? tune.zc_len : planes;
    const int nzc = (planes + zc_len - 1) / zc_len;
    const int nblk = (int)(per_layer * nzc);
    if (ABL == 0 && push && (push->lo || push->hi))
        hipLaunchKernelGGL((jacobi_sweep_kernel<T, RY, 0, true>), dim3(nblk), dim3(256), 0, st, g, sc, src, rhs, dst, flags, b,
                           a, inv_c, z_first, z_last, zc_len, zc_len, nxw, nybg, nblk, *push);
    else
        hipLaunchKernelGGL((jacobi_sweep_kernel<T, RY, ABL, false>), dim3(nblk), dim3(256), 0, st, g, sc, src, rhs, dst, flags, b,
                           a, inv_c, z_first, z_last, zc_len, zc_len, nxw, nybg, nblk, PeerPush());
}

template <class T>
void launch_jacobi(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* src, const T* rhs,
                   T* dst, const uint8_t* flags, int b, T a, T inv_c, int z_first, int z_last, int second_first, const PeerPush* push)
{
    if (z_last < z_first) return;
#define FS_GO(RY, ABL) launch_jacobi_v<T, RY, ABL>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, second_first, push)
    if (tune.abl == 0) {
        if (tune.ry == 4) FS_GO(4, 0);
        else FS_GO(2, 0);
    } else {   // timing-only ablations (wrong results by design), RY = 4
        switch (tune.abl) {
            case 1: FS_GO(4, 1); break;
            case 2: FS_GO(4, 2); break;
            case 4: FS_GO(4, 4); break;
            case 8: FS_GO(4, 8); break;
            case 3: FS_GO(4, 3); break;
            default: FS_GO(4, 0); break;
        }
    }
#undef FS_GO
}
template void launch_jacobi<float>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, const float*, const float*,
                                   float*, const uint8_t*, int, float, float, int, int, int, const PeerPush*);
template void launch_jacobi<double>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, const double*,
                                    const double*, double*, const uint8_t*, int, double, double, int, int, int, const PeerPush*);

// =====================================================================================
// Two Jacobi sweeps per pass over memory (temporal blocking), bit-identical with two
// launches of jacobi_sweep_kernel.
//
// One sweep moves 12 B per cell for 8 flops; the only way past the HBM roof is to apply
// several sweeps while the data is on chip.  Level 0 = `src`, level 1 = the iterate after
// one sweep + setBounds (never written to memory), level 2 = `dst`.
//
// A workgroup owns the full row width (NXW waves of 256 cells) times a band of BY = NYW*2
// rows and marches along z.  Per plane zl it (1) computes level 1 of plane zl for its band
// from level-0 planes zl-1..zl+1 held in registers, exactly as the single sweep does,
// including the zeroing of solids and the ghost faces of setBounds; (2) publishes that
// level-1 plane tile (interior rows + the ghost rows/columns setBounds would have written)
// in LDS; (3) after one barrier computes level 2 of plane zl-1 from level-1 planes
// zl-2, zl-1, zl of its own cells (registers) and the in-plane neighbours out of the LDS
// tile, and stores it with the fused setBounds.  Three LDS plane buffers rotate, so one
// barrier per plane suffices.  Bands overlap by two rows and z chunks by two planes: the
// first and last level-1 row of a band have no level-2 output there (their level-1
// neighbour row belongs to the next band); rows next to the walls use the ghost rows.
// =====================================================================================
// MODE: 0 = two Jacobi sweeps; 1 = one red-black iteration (solver=rbsor; coarse-level-style smoothing); 2 = two damped
// Jacobi sweeps, q + omega*(r - q) in every cell (the level-0 smoother of solver=mg)
template <class T, int NXW, int NYW, int MODE, bool PUSH>
__global__ __launch_bounds__(NXW* NYW * 64) void jacobi_pair_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ src,
                                                                     const T* __restrict__ rhs, T* __restrict__ dst,
                                                                     const uint8_t* __restrict__ flags, int b, T a,
                                                                     T inv_c, int z_first, int z_last, int zc_len,
                                                                     int z_stride, int nbands, int nblk, T omega, PeerPush pp)
{
    constexpr int RY = 2, BY = NYW * RY, TW = NXW * 256 + 8;
    // ring of four level-1 plane tiles (plane z lives in slot z & 3); column index = x + 3
    __shared__ T tile[4][BY][TW];

    const int v = xcd_contiguous(blockIdx.x, nblk);
    const int band = v % nbands, zc = v / nbands;
    // readfirstlane: the wave index is the same in all 64 lanes, but only this tells the compiler so -- rows,
    // row pointers and every row test then live in scalar registers and branch as scalars
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wx = wave % NXW, wy = wave / NXW;
    const int W = g.W, H = g.H, D = g.D;
    const int s = band * (BY - 2);                       // tile row t <-> grid row s + t
    const int ty0 = wy * RY, y0 = s + ty0;
    const int x0 = 1 + wx * 256 + lane * 4;
    const bool lane_on = x0 <= W;
    const int zbeg = z_first + zc * z_stride, zend = min(z_last, zbeg + zc_len - 1);  // level-2 output planes
    if (zbeg > zend) return;                             // block-uniform
    // level-1 planes: one beyond the output chunk on each side; beyond a physical wall there is
    // no such plane (its level-1 ghost is derived below), beyond a slab boundary it is the
    // neighbour's plane, recomputed here from the two-deep halo
    const int zl_first = max(sc.lo_wall ? 1 : 0, zbeg - 1), zl_last = min(sc.hi_wall ? D : D + 1, zend + 1);
    const int out_lo = max(1, s + 1), out_hi = min(H, s + BY - 2);        // level-2 output rows

    const bool full_group = (x0 + 3 <= W);
    const bool edge_l = lane_on && (lane == 0);
    const bool edge_r = lane_on && full_group && ((lane == 63) || (x0 + 4 > W));
    const int kill_shift = (b == 0) ? 0 : 4;
    const T zero = (T)0;
    const long row0 = cell(g, x0, y0, 0);

    auto ld4 = [&](const T* ptr, bool on, T (&out)[4]) {
        V4<T> q = {{zero, zero, zero, zero}};
        if (on) q = *reinterpret_cast<const V4<T>*>(ptr);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = q.e[e];
    };
    auto row_in_mem = [&](int y) { return y >= 0 && y <= H + 1; };
    auto fetch_core = [&](int z, T (&out)[RY][4]) {
        const T* pz = src + row0 + (long)z * g.sz;
#pragma unroll
        for (int r = 0; r < RY; ++r) ld4(pz + r * g.sy, lane_on && row_in_mem(y0 + r), out[r]);
    };
    // the rows above/below the patch and the columns beside it are only needed of the plane that is
    // the stencil centre: they are fetched one plane behind the cores (P.core is not used here)
    auto fetch_side = [&](int z, PlaneIn<T, RY>& P) {
        const T* pz = src + row0 + (long)z * g.sz;
        ld4(pz - g.sy, lane_on && row_in_mem(y0 - 1), P.hb);
        ld4(pz + RY * g.sy, lane_on && row_in_mem(y0 + RY), P.ht);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool row_on = (y0 + r >= 1) && (y0 + r <= H);
            P.eL[r] = (edge_l && row_on) ? pz[r * g.sy - 1] : zero;
            P.eR[r] = (edge_r && row_on) ? pz[r * g.sy + 4] : zero;
        }
    };
    auto fetch_aux = [&](int z, AuxIn<T, RY>& X) {
        const long off = row0 + (long)z * g.sz;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool on = lane_on && (y0 + r >= 1) && (y0 + r <= H);
            ld4(rhs + off + r * g.sy, on, X.rhs[r]);
            X.fl[r] = on ? (unsigned)flags[(off + r * g.sy + 3) >> 2] : 0u;
        }
    };

    // one stencil application; simulation.cpp:264-269 order x+1, x-1, y+1, y-1, z+1, z-1
    auto relax4 = [&](const T (&cc)[4], T left, T right, const T (&ym)[4], const T (&yp)[4], const T (&zm)[4],
                      const T (&zp)[4], const T (&rh)[4], T (&u)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            T xp1 = (e < 3) ? cc[e + 1] : right;
            T xm1 = (e > 0) ? cc[e - 1] : left;
            T nb = xp1 + xm1 + yp[e] + ym[e] + zp[e] + zm[e];
            u[e] = (rh[e] + a * nb) * inv_c;
        }
    };
    // what setBounds leaves in memory for the lane's four cells of an interior row
    auto settle4 = [&](const T (&u)[4], unsigned fl, T (&st)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x0 + e;
            const bool kill = ((fl >> (e + kill_shift)) & 1u) != 0;
            T ghost_src = (e > 0) ? u[e - 1] : zero;
            st[e] = (x <= W) ? (kill ? zero : u[e]) : ((x == W + 1) ? ghost_src : zero);
        }
    };
    auto face4 = [&](const T (&u)[4], bool negate, T (&out)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = (x0 + e <= W) ? (negate ? -u[e] : u[e]) : zero;
    };
    // RB (solver=rbsor, not in the reference): the two levels of a pass are the two colours of one
    // red-black SOR iteration -- level 1 moves the cells with even x+y+z (global z), level 2 the odd
    // ones, each from its value q towards the Jacobi update r by q + omega*(r - q); the other colour
    // keeps its value.  setBounds between the halves is what the levels do anyway.
    auto blend4 = [&](T (&u)[4], const T (&old)[4], int y, int z, int colour) {
        const int par = (x0 + y + z + sc.zoff + colour) & 1;          // parity of cell e = 0 relative to the colour
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = (((par + e) & 1) == 0) ? old[e] + omega * (u[e] - old[e]) : old[e];
    };
    auto damp4 = [&](T (&u)[4], const T (&old)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = old[e] + omega * (u[e] - old[e]);
    };
    auto lds_row = [&](int z, int t, T (&out)[4]) {
        V4<T> q = *reinterpret_cast<const V4<T>*>(&tile[z & 3][t][x0 + 3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = q.e[e];
    };
    auto lds_put = [&](int z, int t, const T (&in)[4]) {
        V4<T> q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q.e[e] = in[e];
        *reinterpret_cast<V4<T>*>(&tile[z & 3][t][x0 + 3]) = q;
    };

    // level 2 of plane zo: every level-1 operand comes out of the LDS ring (planes zo-1, zo,
    // zo+1), x neighbours of the lane's own group by wave shuffle
    auto level2 = [&](int zo, const AuxIn<T, RY>& X) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = y0 + r, t = ty0 + r;
            const bool row_out = (y >= out_lo) && (y <= out_hi);     // wave-uniform
            T cc[4] = {zero, zero, zero, zero};
            if (row_out && lane_on) lds_row(zo, t, cc);
            T left = __shfl_up(cc[3], 1);
            T right = __shfl_down(cc[0], 1);
            if (!(row_out && lane_on)) continue;
            if (edge_l) left = tile[zo & 3][t][x0 + 2];
            if (edge_r) right = tile[zo & 3][t][x0 + 7];
            T ym[4], yp[4], zm[4], zp[4];
            lds_row(zo, t - 1, ym);
            lds_row(zo, t + 1, yp);
            lds_row(zo - 1, t, zm);
            lds_row(zo + 1, t, zp);
            T u[4], st[4];
            relax4(cc, left, right, ym, yp, zm, zp, X.rhs[r], u);
            if (MODE == 1) blend4(u, cc, y, zo, 1);
            if (MODE == 2) damp4(u, cc);
            settle4(u, X.fl[r], st);
            const long base = row0 + (long)zo * g.sz + r * g.sy;
            V4<T> q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q.e[e] = st[e];
            const long dl = (PUSH && zo <= pp.planes) ? pp.lo : 0, dh = (PUSH && zo > D - pp.planes) ? pp.hi : 0;   // wave-uniform
            put<PUSH>(reinterpret_cast<V4<T>*>(dst + base), q, dl, dh);
            if (x0 == 1) put<PUSH>(dst + base - 1, (b == 1) ? -u[0] : u[0], dl, dh);            // :189-190
            if (full_group && x0 + 3 == W) put<PUSH>(dst + base + 4, u[3], dl, dh);             // :191
            const bool zlo_face = (zo == 1) && sc.lo_wall, zhi_face = (zo == D) && sc.hi_wall;
            if (y == 1 || y == H || zlo_face || zhi_face) {
                T f[4];
                V4<T> qq;
                face4(u, b == 2, f);
#pragma unroll
                for (int e = 0; e < 4; ++e) qq.e[e] = f[e];
                if (y == 1) put<PUSH>(reinterpret_cast<V4<T>*>(dst + base - g.sy), qq, dl, dh);  // :198-201
                if (y == H) put<PUSH>(reinterpret_cast<V4<T>*>(dst + base + g.sy), qq, dl, dh);
                face4(u, b == 3, f);
#pragma unroll
                for (int e = 0; e < 4; ++e) qq.e[e] = f[e];
                if (zlo_face) *reinterpret_cast<V4<T>*>(dst + base - g.sz) = qq;                // :208-214
                if (zhi_face) *reinterpret_cast<V4<T>*>(dst + base + g.sz) = qq;
            }
        }
    };

    // level-0 stream, software-pipelined one plane ahead of its use (see jacobi_sweep_kernel)
    T m[RY][4], Bc[RY][4], Nc[RY][4];
    PlaneIn<T, RY> A, SN;                                // A: centre plane (core + sides); SN: sides of the next centre
    AuxIn<T, RY> xc, xn, xm;
    fetch_core(zl_first - 1, m);
    fetch_core(zl_first, A.core);
    fetch_side(zl_first, A);
    fetch_aux(zl_first, xc);
    fetch_core(zl_first + 1, Bc);
    xm = xc;

    for (int zl = zl_first; zl <= zl_last; ++zl) {
        if (zl + 1 <= zl_last) {                         // block-uniform
            fetch_core(zl + 2, Nc);                      // zl+2 <= D+1
            fetch_side(zl + 1, SN);
            fetch_aux(zl + 1, xn);
        }

        // ---- level 1 of plane zl for this wave's rows, published with its setBounds ghosts
        T gz[RY][4];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = y0 + r, t = ty0 + r;
            T left = __shfl_up(A.core[r][3], 1);
            T right = __shfl_down(A.core[r][0], 1);
            if (edge_l) left = A.eL[r];
            if (edge_r) right = A.eR[r];
            T u[4] = {zero, zero, zero, zero};
            const bool row_on = (y >= 1) && (y <= H);    // wave-uniform
            if (row_on && lane_on) {
                T ym[4], yp[4], st[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ym[e] = (r > 0) ? A.core[r > 0 ? r - 1 : 0][e] : A.hb[e];
                    yp[e] = (r < RY - 1) ? A.core[r < RY - 1 ? r + 1 : r][e] : A.ht[e];
                }
                relax4(A.core[r], left, right, ym, yp, m[r], Bc[r], xc.rhs[r], u);
                if (MODE == 1) blend4(u, A.core[r], y, zl, 0);
                if (MODE == 2) damp4(u, A.core[r]);
                settle4(u, xc.fl[r], st);
                lds_put(zl, t, st);
                if (x0 == 1) tile[zl & 3][t][3] = (b == 1) ? -u[0] : u[0];                      // ghost column x = 0
                if (full_group && x0 + 3 == W) tile[zl & 3][t][W + 4] = u[3];                  // ghost column x = W+1
                T f[4];
                if (y == 1 && t >= 1) { face4(u, b == 2, f); lds_put(zl, t - 1, f); }           // ghost row y = 0
                if (y == H && t + 1 < BY) { face4(u, b == 2, f); lds_put(zl, t + 1, f); }       // ghost row y = H+1
            }
            face4(u, b == 3, gz[r]);
        }
        if (zl == 1 && sc.lo_wall) {                     // level-1 ghost plane z = 0 (:208-210)
#pragma unroll
            for (int r = 0; r < RY; ++r)
                if (lane_on && y0 + r >= 1 && y0 + r <= H) lds_put(0, ty0 + r, gz[r]);
        }
        __syncthreads();

        // ---- level 2 of plane zl-1
        if (zl - 1 >= zbeg) level2(zl - 1, xm);
        if (zl == D && zend == D && sc.hi_wall) {
            // top wall: level-1 ghost plane z = D+1 is +-u1(D) (:212-214).  Its ring slot still
            // holds plane D-3, which slower waves may be reading: fence both sides.
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RY; ++r)
                if (lane_on && y0 + r >= 1 && y0 + r <= H) lds_put(D + 1, ty0 + r, gz[r]);
            __syncthreads();
            level2(D, xc);
        }

#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[r][e] = A.core[r][e];
                SN.core[r][e] = Bc[r][e];
                Bc[r][e] = Nc[r][e];
            }
        A = SN;
        xm = xc;
        xc = xn;
    }
}

template <class T, int NXW, int NYW>
static void launch_pair_v(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* src,
                          const T* rhs, T* dst, const uint8_t* flags, int b, T a, T inv_c, int z_first, int z_last, int alt,
                          int second_first, T omega, bool damped, const PeerPush* push)
{
    // omega == 0: two Jacobi sweeps; otherwise one red-black SOR iteration with that relaxation factor, or (damped)
    // two Jacobi sweeps damped by it
    const bool pushing = push && (push->lo || push->hi) && omega == (T)0 && second_first < 0;
    const PeerPush pp = pushing ? *push : PeerPush();
    auto kernel = pushing ? jacobi_pair_kernel<T, NXW, NYW, 0, true>
                  : (omega == (T)0) ? jacobi_pair_kernel<T, NXW, NYW, 0, false>
                  : damped ? jacobi_pair_kernel<T, NXW, NYW, 2, false> : jacobi_pair_kernel<T, NXW, NYW, 1, false>;
    constexpr int BY = NYW * 2;
    const int planes = z_last - z_first + 1;
    if (planes <= 0) return;
    const int nbands = (g.H + (BY - 2) - 1) / (BY - 2);
    if (second_first >= 0) {
        // two equally long ranges (the slab's two boundary regions) as two chunks of one launch
        hipLaunchKernelGGL(kernel, dim3(nbands * 2), dim3(NXW * NYW * 64), 0, st, g, sc, src,
                           rhs, dst, flags, b, a, inv_c, z_first, second_first + planes - 1, planes,
                           second_first - z_first, nbands, nbands * 2, omega, pp);
        return;
    }
    // z chunks: each re-reads 4 level-0 planes and recomputes 2 level-1 planes, so keep them
    // long; pick the count that fills the CUs most evenly (one workgroup per CU)
    // model: fraction of CU slots filled x useful fraction of a chunk's planes; `alt` picks the
    // alt-th best chunk count by that model (the host driver times alt = 0, 1, 2 once per grid,
    // because how the block count falls against the 256 CUs matters more than the model knows)
    int cand_nzc[3] = {1, 1, 1};
    double cand_eff[3] = {-1.0, -1.0, -1.0};
    const int slots = tune.cu_slots > 0 ? tune.cu_slots : 256;
    for (int nzc = 1; nzc <= 64 && (nzc == 1 || planes / nzc >= 12); ++nzc) {
        const long blocks = (long)nbands * nzc;
        const long rounds = (blocks + slots - 1) / slots;
        const int len = (planes + nzc - 1) / nzc;
        const double eff = (double)blocks / (double)(rounds * slots) * (double)len / (double)(len + 3);
        for (int k = 0; k < 3; ++k)
            if (eff > cand_eff[k] + 1e-9) {
                for (int j = 2; j > k; --j) { cand_eff[j] = cand_eff[j - 1]; cand_nzc[j] = cand_nzc[j - 1]; }
                cand_eff[k] = eff;
                cand_nzc[k] = nzc;
                break;
            }
    }
    int pick = alt < 0 ? 0 : (alt > 2 ? 2 : alt);
    while (pick > 0 && cand_eff[pick] < 0.0) --pick;
    const int best_nzc = cand_nzc[pick];
    int zc_len = (planes + best_nzc - 1) / best_nzc;
    if (tune.pair_zc > 0) zc_len = tune.pair_zc < planes ? tune.pair_zc : planes;
    const int nzc = (planes + zc_len - 1) / zc_len;
    const int nblk = nbands * nzc;
    hipLaunchKernelGGL(kernel, dim3(nblk), dim3(NXW * NYW * 64), 0, st, g, sc, src, rhs,
                       dst, flags, b, a, inv_c, z_first, z_last, zc_len, zc_len, nbands, nblk, omega, pp);
}

template <class T>
bool pair_supported(const SweepTune& tune, const GridDesc& g, const SlabCtx& sc)
{
    const bool whole = sc.lo_wall && sc.hi_wall;
    return (whole || g.zh >= 2) && g.W <= 1024 && tune.fuse >= 2;
}
template bool pair_supported<float>(const SweepTune&, const GridDesc&, const SlabCtx&);
template bool pair_supported<double>(const SweepTune&, const GridDesc&, const SlabCtx&);

template <>
int pair_shape_count<float>(const GridDesc& g) { return (g.W <= 512) ? 3 : 1; }
template <>
int pair_shape_count<double>(const GridDesc&) { return 1; }

template <>
void launch_jacobi_pair<float>(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const float* src,
                               const float* rhs, float* dst, const uint8_t* flags, int b, float a, float inv_c, int z_first,
                               int z_last, int shape, int second_first, float omega, bool damped, const PeerPush* push)
{
    // shape: 0 = 12 waves (768 threads, <=168 VGPRs), 2 = 10 waves, 1 = 8 waves, 3 = 16 waves (spills;
    // tuning tool only).  All shapes give identical results; the host driver times 0..count-1 once per
    // grid and keeps the fastest (band count vs CU count decides, e.g. 10 waves at 512^3, 12 at 256^3).
    const int nxw = (g.W + 255) / 256;
    if (shape < 0) shape = 0;
    const int alt = shape >> 3;                          // which of the three best chunk counts
    shape &= 7;
    if (tune.pair_shape > 0) shape = tune.pair_shape;
#define FS_PAIR(NX, NY) launch_pair_v<float, NX, NY>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, alt, second_first, omega, damped, push)
    if (nxw == 1) { if (shape == 1) FS_PAIR(1, 8); else if (shape == 2) FS_PAIR(1, 10); else if (shape == 3) FS_PAIR(1, 16); else FS_PAIR(1, 12); }
    else if (nxw == 2) { if (shape == 1) FS_PAIR(2, 4); else if (shape == 2) FS_PAIR(2, 5); else if (shape == 3) FS_PAIR(2, 8); else FS_PAIR(2, 6); }
    else if (nxw == 3) FS_PAIR(3, 4);
    else FS_PAIR(4, 3);
#undef FS_PAIR
}
template <>
void launch_jacobi_pair<double>(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const double* src,
                                const double* rhs, double* dst, const uint8_t* flags, int b, double a, double inv_c,
                                int z_first, int z_last, int shape, int second_first, double omega, bool damped, const PeerPush* push)
{
    const int alt = shape < 0 ? 0 : (shape >> 3);
    const int nxw = (g.W + 255) / 256;   // LDS: 4 * BY * TW * 8 bytes must stay under 160 KB
    if (nxw == 1) launch_pair_v<double, 1, 8>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, alt, second_first, omega, damped, push);
    else if (nxw == 2) launch_pair_v<double, 2, 4>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, alt, second_first, omega, damped, push);
    else if (nxw == 3) launch_pair_v<double, 3, 3>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, alt, second_first, omega, damped, push);
    else launch_pair_v<double, 4, 2>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, alt, second_first, omega, damped, push);
}

// =====================================================================================
// Reference-order in-place sweep (verification mode, single GPU).
//   linearSolver  simulation.cpp:251-273 at one thread: x outermost, z innermost, in place.
// Cells on a hyperplane x+y+z = s only depend on hyperplanes s-1 (already updated) and s+1
// (not yet updated), exactly like the lexicographic sweep, so walking s upward with a
// barrier between hyperplanes reproduces the reference bit for bit.  One workgroup does
// the whole solve (all `sweeps` iterations and their setBounds) so that a workgroup
// barrier is the only synchronisation needed; this mode is for parity, not speed.
// =====================================================================================
template <class T>
__device__ void bounds_in_block(const GridDesc& g, T* q, const uint8_t* flags, int b)
{
    const int W = g.W, H = g.H, D = g.D;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int t = tid; t < H * D; t += nt) {
        int y = 1 + t % H, z = 1 + t / H;
        T in = q[cell(g, 1, y, z)];
        q[cell(g, 0, y, z)] = (b == 1) ? -in : in;
        q[cell(g, W + 1, y, z)] = q[cell(g, W, y, z)];
    }
    for (int t = tid; t < W * D; t += nt) {
        int x = 1 + t % W, z = 1 + t / W;
        T lo = q[cell(g, x, 1, z)], hi = q[cell(g, x, H, z)];
        q[cell(g, x, 0, z)] = (b == 2) ? -lo : lo;
        q[cell(g, x, H + 1, z)] = (b == 2) ? -hi : hi;
    }
    for (int t = tid; t < W * H; t += nt) {
        int x = 1 + t % W, y = 1 + t / W;
        T lo = q[cell(g, x, y, 1)], hi = q[cell(g, x, y, D)];
        q[cell(g, x, y, 0)] = (b == 3) ? -lo : lo;
        q[cell(g, x, y, D + 1)] = (b == 3) ? -hi : hi;
    }
    __syncthreads();
    const unsigned zero_bits = (b == 0) ? F_SOLID : (F_SOLID | F_NEAR);
    for (long t = tid; t < (long)W * H * D; t += nt) {
        int x = 1 + (int)(t % W), y = 1 + (int)((t / W) % H), z = 1 + (int)(t / ((long)W * H));
        long c = cell(g, x, y, z);
        if (flags[c] & zero_bits) q[c] = (T)0;
    }
    __syncthreads();
}

template <class T>
__global__ __launch_bounds__(1024) void gs_lex_kernel(GridDesc g, T* q, const T* rhs, const uint8_t* flags, int b, T a,
                                                       T inv_c, int sweeps)
{
    const int W = g.W, H = g.H, D = g.D;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int it = 0; it < sweeps; ++it) {
        for (int s = 3; s <= W + H + D; ++s) {
            // y in [max(1, s-W-D) .. min(H, s-2)]
            for (int t = tid; t < H * D; t += nt) {
                int y = 1 + t % H, z = 1 + t / H;
                int x = s - y - z;
                if (x >= 1 && x <= W) {
                    long c = cell(g, x, y, z);
                    T nb = q[c + 1] + q[c - 1] + q[c + g.sy] + q[c - g.sy] + q[c + g.sz] + q[c - g.sz];
                    q[c] = (rhs[c] + a * nb) * inv_c;
                }
            }
            __syncthreads();
        }
        bounds_in_block(g, q, flags, b);
    }
}

template <class T>
void launch_set_bounds(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* q, const uint8_t* flags, int b);

// Larger grids: the same order, tile by tile.  Cubic tiles of TS^3 cells are themselves swept in
// hyperplane order I+J+L = k, one launch per k (tiles on one tile-hyperplane only touch tiles on
// k-1, already done, and k+1, not yet started); inside a tile one workgroup walks the cell
// hyperplanes with a barrier in between, as gs_lex_kernel does for the whole grid -- on a copy of
// the tile in LDS, so that a step costs an LDS round trip rather than one through L2.
template <class T, int TS>
__global__ __launch_bounds__(TS* TS) void gs_tile_kernel(GridDesc g, T* q, const T* __restrict__ rhs, T a, T inv_c, int k,
                                                          int nL)
{
    constexpr int E = TS + 2;
    __shared__ T t[E][E][E];                             // the tile with one cell of its surroundings on every side
    const int I = blockIdx.x, J = blockIdx.y, L = k - I - J;
    if (L < 0 || L >= nL) return;                        // block-uniform
    const int bx = I * TS, by = J * TS, bz = L * TS;     // tile index e <-> grid coordinate b + e
    // Surroundings as they are in memory now: the -1 faces belong to tiles of hyperplane k-1 (already
    // swept), the +1 faces to tiles of k+1 (not yet), exactly what the lexicographic order sees.
    for (int i = threadIdx.x; i < E * E * E; i += TS * TS) {
        const int ex = i % E, ey = (i / E) % E, ez = i / (E * E);
        const int gx = bx + ex, gy = by + ey, gz = bz + ez;
        t[ez][ey][ex] = (gx <= g.W + 1 && gy <= g.H + 1 && gz <= g.D + 1) ? q[cell(g, gx, gy, gz)] : (T)0;
    }
    const int ly = threadIdx.x % TS, lz = threadIdx.x / TS;
    const int y = by + 1 + ly, z = bz + 1 + lz;
    const bool yz_on = (y <= g.H) && (z <= g.D);
    T r[TS];
#pragma unroll
    for (int lx = 0; lx < TS; ++lx) r[lx] = (yz_on && bx + 1 + lx <= g.W) ? rhs[cell(g, bx + 1 + lx, y, z)] : (T)0;
    __syncthreads();
    for (int s = 0; s < 3 * TS - 2; ++s) {
        const int lx = s - ly - lz;
        if (yz_on && lx >= 0 && lx < TS && bx + 1 + lx <= g.W) {
            T(*c)[E][E] = t;
            const int ex = lx + 1, ey = ly + 1, ez = lz + 1;
            T nb = c[ez][ey][ex + 1] + c[ez][ey][ex - 1] + c[ez][ey + 1][ex] + c[ez][ey - 1][ex] + c[ez + 1][ey][ex] +
                   c[ez - 1][ey][ex];
            T rv = (T)0;
#pragma unroll
            for (int j = 0; j < TS; ++j) rv = (j == lx) ? r[j] : rv;
            c[ez][ey][ex] = (rv + a * nb) * inv_c;
        }
        __syncthreads();
    }
    if (yz_on) {
#pragma unroll
        for (int lx = 0; lx < TS; ++lx)
            if (bx + 1 + lx <= g.W) q[cell(g, bx + 1 + lx, y, z)] = t[lz + 1][ly + 1][lx + 1];
    }
}

template <class T>
void launch_gs_lex(hipStream_t st, const GridDesc& g, T* q, const T* rhs, const uint8_t* flags, int b, T a, T inv_c,
                   int sweeps)
{
    if ((long)g.W * g.H * g.D <= 32768) {                // small: one workgroup does the whole solve in one launch
        hipLaunchKernelGGL((gs_lex_kernel<T>), dim3(1), dim3(1024), 0, st, g, q, rhs, flags, b, a, inv_c, sweeps);
        return;
    }
    constexpr int TS = 16;
    const int nI = (g.W + TS - 1) / TS, nJ = (g.H + TS - 1) / TS, nL = (g.D + TS - 1) / TS;
    SlabCtx whole = { 0, g.D, 1, 1 };                    // gs_lex is single-GPU only
    for (int it = 0; it < sweeps; ++it) {
        for (int k = 0; k <= nI + nJ + nL - 3; ++k)
            hipLaunchKernelGGL((gs_tile_kernel<T, TS>), dim3(nI, nJ), dim3(TS * TS), 0, st, g, q, rhs, a, inv_c, k, nL);
        launch_set_bounds<T>(st, g, whole, q, flags, b);
    }
}
template void launch_gs_lex<float>(hipStream_t, const GridDesc&, float*, const float*, const uint8_t*, int, float, float,
                                   int);
template void launch_gs_lex<double>(hipStream_t, const GridDesc&, double*, const double*, const uint8_t*, int, double,
                                    double, int);

template <class T>
__device__ __forceinline__ T one_sided_grad(bool fp, bool fm, T pp, T pc, T pm, T h, T two_h);

// =====================================================================================
// z-marching forms of the two projection passes.  Same wave decomposition as the sweep
// (256 x-consecutive cells x RY rows per wave, 16 B per lane, XCD-contiguous tiles): the
// planes z-1 / z+1 a cell needs stay in registers while the wave walks along z, rows y-1 /
// y+1 are L1/L2 hits, x neighbours cross lanes by shuffle, so every input array is read from
// HBM once.  Arithmetic and pass order are those of the per-cell kernels below, which remain
// as the plain statement (and serve fs_set_option "project_kernels"="cell").
// =====================================================================================
template <class T>
struct MarchTile {
    int lane, x0, y0, zbeg, zend;
    bool lane_on, full_group, edge_l, edge_r, live;
    long row0;
};
template <class T, int RY>
__device__ __forceinline__ MarchTile<T> march_tile(const GridDesc& g, int zc_len, int nxw, int nybg, int nblk)
{
    MarchTile<T> t;
    const int v = xcd_contiguous(blockIdx.x, nblk);
    const int xw = v % nxw, ybg = (v / nxw) % nybg, zc = v / (nxw * nybg);
    t.lane = threadIdx.x & 63;
    t.y0 = 1 + (ybg * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * RY;   // wave-uniform row: scalar tests and row pointers
    t.x0 = 1 + xw * 256 + t.lane * 4;
    t.lane_on = t.x0 <= g.W;
    t.zbeg = 1 + zc * zc_len;
    t.zend = min(g.D, t.zbeg + zc_len - 1);
    t.full_group = (t.x0 + 3 <= g.W);
    t.edge_l = t.lane_on && (t.lane == 0);
    t.edge_r = t.lane_on && t.full_group && ((t.lane == 63) || (t.x0 + 4 > g.W));
    t.live = (t.y0 <= g.H) && (t.zbeg <= t.zend);
    t.row0 = cell(g, t.x0, t.y0, 0);
    return t;
}
template <class T>
__device__ __forceinline__ void ld_row(const T* ptr, bool on, T (&out)[4])
{
    V4<T> q = {{(T)0, (T)0, (T)0, (T)0}};
    if (on) q = *reinterpret_cast<const V4<T>*>(ptr);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = q.e[e];
}
// Store the lane's four cells of an interior row of a field with boundary code b, the way
// setBounds leaves them (simulation.cpp:183-246): `u` un-zeroed values, `killmask` bit e set =>
// cell e is zeroed; ghost faces determined by this row are written from the un-zeroed values.
template <class T>
__device__ __forceinline__ void store_row_bounds(const GridDesc& g, const SlabCtx& sc, T* dst, long base, int x0, int y, int z,
                                                 const T (&u)[4], unsigned killmask, int b)
{
    const T zero = (T)0;
    V4<T> st;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int x = x0 + e;
        T ghost_src = (e > 0) ? u[e - 1] : zero;
        st.e[e] = (x <= g.W) ? (((killmask >> e) & 1u) ? zero : u[e]) : ((x == g.W + 1) ? ghost_src : zero);
    }
    *reinterpret_cast<V4<T>*>(dst + base) = st;
    if (x0 == 1) dst[base - 1] = (b == 1) ? -u[0] : u[0];
    if (x0 + 3 == g.W) dst[base + 4] = u[3];
    const bool zlo = (z == 1) && sc.lo_wall, zhi = (z == g.D) && sc.hi_wall;
    if (y == 1 || y == g.H || zlo || zhi) {
        V4<T> gy, gz;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = (x0 + e <= g.W);
            gy.e[e] = in ? ((b == 2) ? -u[e] : u[e]) : zero;
            gz.e[e] = in ? ((b == 3) ? -u[e] : u[e]) : zero;
        }
        if (y == 1) *reinterpret_cast<V4<T>*>(dst + base - g.sy) = gy;
        if (y == g.H) *reinterpret_cast<V4<T>*>(dst + base + g.sy) = gy;
        if (zlo) *reinterpret_cast<V4<T>*>(dst + base - g.sz) = gz;
        if (zhi) *reinterpret_cast<V4<T>*>(dst + base + g.sz) = gz;
    }
}

// divergence + pressure reset + setBounds(0,div) + setBounds(0,p)   simulation.cpp:295-319
template <class T, int RY>
__global__ __launch_bounds__(256) void divergence_march_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ vx,
                                                                const T* __restrict__ vy, const T* __restrict__ vz,
                                                                T* __restrict__ dv, T* __restrict__ p,
                                                                const uint8_t* __restrict__ flags, T mhalf_h, int zc_len,
                                                                int nxw, int nybg, int nblk)
{
    const MarchTile<T> t = march_tile<T, RY>(g, zc_len, nxw, nybg, nblk);
    if (!t.live) return;                                 // wave-uniform
    const int H = g.H;
    const T zero = (T)0;
    T zm[RY][4], zc[RY][4], zp[RY][4];                   // v_z at planes z-1, z, z+1
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const bool on = t.lane_on && (t.y0 + r <= H);
        ld_row(vz + t.row0 + (long)(t.zbeg - 1) * g.sz + r * g.sy, on, zm[r]);
        ld_row(vz + t.row0 + (long)t.zbeg * g.sz + r * g.sy, on, zc[r]);
    }
    for (int z = t.zbeg; z <= t.zend; ++z) {
        const long off = t.row0 + (long)z * g.sz;
        T xr[RY][4], yr[RY + 2][4];
        unsigned fl[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool on = t.lane_on && (t.y0 + r <= H);
            ld_row(vz + off + g.sz + r * g.sy, on, zp[r]);
            ld_row(vx + off + r * g.sy, on, xr[r]);
            fl[r] = on ? *reinterpret_cast<const unsigned*>(flags + off + r * g.sy) : 0u;
        }
#pragma unroll
        for (int r = 0; r < RY + 2; ++r) ld_row(vy + off + (r - 1) * g.sy, t.lane_on && (t.y0 + r - 1 <= H + 1), yr[r]);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = t.y0 + r;
            T left = __shfl_up(xr[r][3], 1);
            T right = __shfl_down(xr[r][0], 1);
            if (!(t.lane_on && y <= H)) continue;        // y <= H is wave-uniform; shuffles are above
            const long base = off + r * g.sy;
            if (t.edge_l) left = vx[base - 1];
            if (t.edge_r) right = vx[base + 4];
            T d[4], zeros[4] = {zero, zero, zero, zero};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned f = (fl[r] >> (8 * e)) & 0xffu;
                T acc = zero;                            // :306-312 in this order
                if (f & F_XP) acc += (e < 3) ? xr[r][e + 1] : right;
                if (f & F_XM) acc -= (e > 0) ? xr[r][e - 1] : left;
                if (f & F_YP) acc += yr[r + 2][e];
                if (f & F_YM) acc -= yr[r][e];
                if (f & F_ZP) acc += zp[r][e];
                if (f & F_ZM) acc -= zm[r][e];
                d[e] = (f & F_SOLID) ? zero : mhalf_h * acc;
            }
            store_row_bounds<T>(g, sc, dv, base, t.x0, y, z, d, 0u, 0);
            store_row_bounds<T>(g, sc, p, base, t.x0, y, z, zeros, 0u, 0);
        }
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zm[r][e] = zc[r][e];
                zc[r][e] = zp[r][e];
            }
    }
}

// v -= grad p, then setBounds(1,vx), (2,vy), (3,vz)   simulation.cpp:322-361
template <class T, int RY>
__global__ __launch_bounds__(256) void gradient_march_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ p,
                                                              T* __restrict__ vx, T* __restrict__ vy, T* __restrict__ vz,
                                                              const uint8_t* __restrict__ flags, T h, T two_h, int zc_len,
                                                              int nxw, int nybg, int nblk)
{
    const MarchTile<T> t = march_tile<T, RY>(g, zc_len, nxw, nybg, nblk);
    if (!t.live) return;
    const int H = g.H;
    T pm[RY][4], pc[RY][4], pp[RY][4];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const bool on = t.lane_on && (t.y0 + r <= H);
        ld_row(p + t.row0 + (long)(t.zbeg - 1) * g.sz + r * g.sy, on, pm[r]);
        ld_row(p + t.row0 + (long)t.zbeg * g.sz + r * g.sy, on, pc[r]);
    }
    for (int z = t.zbeg; z <= t.zend; ++z) {
        const long off = t.row0 + (long)z * g.sz;
        T hb[4], ht[4];
        ld_row(p + off - g.sy, t.lane_on, hb);
        ld_row(p + off + RY * g.sy, t.lane_on && (t.y0 + RY <= H + 1), ht);
        unsigned fl[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool on = t.lane_on && (t.y0 + r <= H);
            ld_row(p + off + g.sz + r * g.sy, on, pp[r]);
            fl[r] = on ? *reinterpret_cast<const unsigned*>(flags + off + r * g.sy) : 0u;
        }
        // in a partial band the row above the last live row is the ghost row H+1, not loaded above
#pragma unroll
        for (int r = 1; r < RY; ++r)
            if (t.y0 + r == H + 1) ld_row(p + off + r * g.sy, t.lane_on, pc[r]);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = t.y0 + r;
            T left = __shfl_up(pc[r][3], 1);
            T right = __shfl_down(pc[r][0], 1);
            if (!(t.lane_on && y <= H)) continue;
            const long base = off + r * g.sy;
            if (t.edge_l) left = p[base - 1];
            if (t.edge_r) right = p[base + 4];
            T ux[4], uy[4], uz[4];
            ld_row(vx + base, true, ux);
            ld_row(vy + base, true, uy);
            ld_row(vz + base, true, uz);
            unsigned killmask = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned f = (fl[r] >> (8 * e)) & 0xffu;
                if (f & (F_SOLID | F_NEAR)) killmask |= 1u << e;
                if (!(f & F_SOLID)) {                    // :326
                    const T c0 = pc[r][e];
                    const T xp1 = (e < 3) ? pc[r][e + 1] : right, xm1 = (e > 0) ? pc[r][e - 1] : left;
                    const T yp1 = (r < RY - 1) ? pc[r < RY - 1 ? r + 1 : r][e] : ht[e];
                    const T ym1 = (r > 0) ? pc[r > 0 ? r - 1 : 0][e] : hb[e];
                    ux[e] -= one_sided_grad<T>(f & F_XP, f & F_XM, xp1, c0, xm1, h, two_h);
                    uy[e] -= one_sided_grad<T>(f & F_YP, f & F_YM, yp1, c0, ym1, h, two_h);
                    uz[e] -= one_sided_grad<T>(f & F_ZP, f & F_ZM, pp[r][e], c0, pm[r][e], h, two_h);
                }
            }
            store_row_bounds<T>(g, sc, vx, base, t.x0, y, z, ux, killmask, 1);
            store_row_bounds<T>(g, sc, vy, base, t.x0, y, z, uy, killmask, 2);
            store_row_bounds<T>(g, sc, vz, base, t.x0, y, z, uz, killmask, 3);
        }
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pm[r][e] = pc[r][e];
                pc[r][e] = pp[r][e];
            }
    }
}

struct MarchLaunch {
    int zc_len, nxw, nybg, nblk;
};
static MarchLaunch march_launch(const GridDesc& g, int RY)
{
    MarchLaunch m;
    m.nxw = (g.W + 255) / 256;
    const int nyb = (g.H + RY - 1) / RY;
    m.nybg = (nyb + 3) / 4;
    const long per_layer = (long)m.nxw * m.nybg;
    long want = (2048 + per_layer - 1) / per_layer;
    if (want < 1) want = 1;
    m.zc_len = (int)((g.D + want - 1) / want);
    if (m.zc_len < 8) m.zc_len = g.D < 8 ? g.D : 8;
    const int nzc = (g.D + m.zc_len - 1) / m.zc_len;
    m.nblk = (int)(per_layer * nzc);
    return m;
}

// =====================================================================================
// Stand-alone setBounds (simulation.cpp:183-246): faces first, then the zeroing passes.
// The hot kernels fuse this; the stand-alone form serves fs_set_bounds and odd callers.
// =====================================================================================
template <class T>
__global__ void bounds_faces_kernel(GridDesc g, SlabCtx sc, T* q, int b)
{
    const int W = g.W, H = g.H, D = g.D;
    const long nx = (long)H * D, ny = (long)W * D, nz = (long)W * H;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < nx + ny + nz; t += (long)gridDim.x * blockDim.x) {
        if (t < nx) {
            int y = 1 + (int)(t % H), z = 1 + (int)(t / H);
            T in = q[cell(g, 1, y, z)];
            q[cell(g, 0, y, z)] = (b == 1) ? -in : in;
            q[cell(g, W + 1, y, z)] = q[cell(g, W, y, z)];
        } else if (t < nx + ny) {
            long u = t - nx;
            int x = 1 + (int)(u % W), z = 1 + (int)(u / W);
            T lo = q[cell(g, x, 1, z)], hi = q[cell(g, x, H, z)];
            q[cell(g, x, 0, z)] = (b == 2) ? -lo : lo;
            q[cell(g, x, H + 1, z)] = (b == 2) ? -hi : hi;
        } else {
            long u = t - nx - ny;
            int x = 1 + (int)(u % W), y = 1 + (int)(u / W);
            if (sc.lo_wall) {
                T lo = q[cell(g, x, y, 1)];
                q[cell(g, x, y, 0)] = (b == 3) ? -lo : lo;
            }
            if (sc.hi_wall) {
                T hi = q[cell(g, x, y, D)];
                q[cell(g, x, y, D + 1)] = (b == 3) ? -hi : hi;
            }
        }
    }
}

template <class T>
__global__ void bounds_zero_kernel(GridDesc g, T* q, const uint8_t* flags, int b)
{
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = 1 + blockIdx.z;
    if (x > g.W || y > g.H) return;
    const unsigned zero_bits = (b == 0) ? F_SOLID : (F_SOLID | F_NEAR);
    long c = cell(g, x, y, z);
    if (flags[c] & zero_bits) q[c] = (T)0;
}

static inline dim3 cell_grid(const GridDesc& g) { return dim3((g.W + 63) / 64, (g.H + 3) / 4, g.D); }
static inline dim3 cell_block() { return dim3(64, 4, 1); }

template <class T>
void launch_set_bounds(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* q, const uint8_t* flags, int b)
{
    long faces = (long)g.H * g.D + (long)g.W * g.D + (long)g.W * g.H;
    int nb = (int)((faces + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL((bounds_faces_kernel<T>), dim3(nb), dim3(256), 0, st, g, sc, q, b);
    hipLaunchKernelGGL((bounds_zero_kernel<T>), cell_grid(g), cell_block(), 0, st, g, q, flags, b);
}
template void launch_set_bounds<float>(hipStream_t, const GridDesc&, const SlabCtx&, float*, const uint8_t*, int);
template void launch_set_bounds<double>(hipStream_t, const GridDesc&, const SlabCtx&, double*, const uint8_t*, int);

// (write_face_ghosts, the ghost-face writes of the per-cell kernels, lives in kernels_dev.h)

// =====================================================================================
// project, part 1: divergence + pressure reset + setBounds(0,div) + setBounds(0,p)
//   simulation.cpp:295-319
// =====================================================================================
template <class T>
__global__ __launch_bounds__(256) void divergence_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ vx,
                                                          const T* __restrict__ vy, const T* __restrict__ vz,
                                                          T* __restrict__ dv, T* __restrict__ p,
                                                          const uint8_t* __restrict__ flags, T mhalf_h)
{
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = 1 + blockIdx.z;
    if (x > g.W || y > g.H) return;
    const long c = cell(g, x, y, z);
    const unsigned f = flags[c];
    T d = (T)0;
    if (!(f & F_SOLID)) {
        T acc = (T)0;                                    // :306-312
        if (f & F_XP) acc += vx[c + 1];
        if (f & F_XM) acc -= vx[c - 1];
        if (f & F_YP) acc += vy[c + g.sy];
        if (f & F_YM) acc -= vy[c - g.sy];
        if (f & F_ZP) acc += vz[c + g.sz];
        if (f & F_ZM) acc -= vz[c - g.sz];
        d = mhalf_h * acc;                               // (-0.5f*h)*div_val, :314
    }
    dv[c] = d;
    p[c] = (T)0;
    write_face_ghosts(g, sc, dv, c, x, y, z, d, 0);      // setBounds(0,div): solid cells already hold 0
    write_face_ghosts(g, sc, p, c, x, y, z, (T)0, 0);    // setBounds(0,p)
}

template <class T>
void launch_divergence(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* vx, const T* vy,
                       const T* vz, T* div, T* p, const uint8_t* flags, T mhalf_h)
{
    if (tune.project_cell) {
        hipLaunchKernelGGL((divergence_kernel<T>), cell_grid(g), cell_block(), 0, st, g, sc, vx, vy, vz, div, p, flags,
                           mhalf_h);
        return;
    }
    constexpr int RY = 2;
    const MarchLaunch m = march_launch(g, RY);
    hipLaunchKernelGGL((divergence_march_kernel<T, RY>), dim3(m.nblk), dim3(256), 0, st, g, sc, vx, vy, vz, div, p, flags,
                       mhalf_h, m.zc_len, m.nxw, m.nybg, m.nblk);
}
template void launch_divergence<float>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, const float*,
                                       const float*, const float*, float*, float*, const uint8_t*, float);
template void launch_divergence<double>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, const double*,
                                        const double*, const double*, double*, double*, const uint8_t*, double);

// =====================================================================================
// project, part 2: v -= grad p, then setBounds(1,vx), (2,vy), (3,vz)
//   simulation.cpp:322-361
// =====================================================================================
template <class T>
__device__ __forceinline__ T one_sided_grad(bool fp, bool fm, T pp, T pc, T pm, T h, T two_h)
{
    if (fp && fm) return (pp - pm) / two_h;              // :329-330
    if (fp) return (pp - pc) / h;                        // :331-332
    if (fm) return (pc - pm) / h;                        // :333-334
    return (T)0;
}

template <class T>
__global__ __launch_bounds__(256) void gradient_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ p,
                                                        T* __restrict__ vx, T* __restrict__ vy, T* __restrict__ vz,
                                                        const uint8_t* __restrict__ flags, T h, T two_h)
{
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = 1 + blockIdx.z;
    if (x > g.W || y > g.H) return;
    const long c = cell(g, x, y, z);
    const unsigned f = flags[c];
    T ux = vx[c], uy = vy[c], uz = vz[c];
    if (!(f & F_SOLID)) {
        const T pc = p[c];
        ux -= one_sided_grad<T>(f & F_XP, f & F_XM, p[c + 1], pc, p[c - 1], h, two_h);
        uy -= one_sided_grad<T>(f & F_YP, f & F_YM, p[c + g.sy], pc, p[c - g.sy], h, two_h);
        uz -= one_sided_grad<T>(f & F_ZP, f & F_ZM, p[c + g.sz], pc, p[c - g.sz], h, two_h);
    }
    const bool kill = (f & (F_SOLID | F_NEAR)) != 0;
    vx[c] = kill ? (T)0 : ux;
    vy[c] = kill ? (T)0 : uy;
    vz[c] = kill ? (T)0 : uz;
    write_face_ghosts(g, sc, vx, c, x, y, z, ux, 1);
    write_face_ghosts(g, sc, vy, c, x, y, z, uy, 2);
    write_face_ghosts(g, sc, vz, c, x, y, z, uz, 3);
}

template <class T>
void launch_gradient(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* p, T* vx, T* vy,
                     T* vz, const uint8_t* flags, T h, T two_h)
{
    if (tune.project_cell) {
        hipLaunchKernelGGL((gradient_kernel<T>), cell_grid(g), cell_block(), 0, st, g, sc, p, vx, vy, vz, flags, h, two_h);
        return;
    }
    constexpr int RY = 2;
    const MarchLaunch m = march_launch(g, RY);
    hipLaunchKernelGGL((gradient_march_kernel<T, RY>), dim3(m.nblk), dim3(256), 0, st, g, sc, p, vx, vy, vz, flags, h, two_h,
                       m.zc_len, m.nxw, m.nybg, m.nblk);
}
template void launch_gradient<float>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, const float*, float*,
                                     float*, float*, const uint8_t*, float, float);
template void launch_gradient<double>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, const double*,
                                      double*, double*, double*, const uint8_t*, double, double);

// =====================================================================================
// Semi-Lagrangian advection + setBounds(b, field)   simulation.cpp:367-424
// The back-trace is clamped in GLOBAL coordinates; `prev` may be the all-gathered global
// array under z-slab partitioning (prev_zshift = zoff planes), see fluidsim.cpp.
// =====================================================================================
template <class T>
__device__ __forceinline__ T clamp_ref(T v, T lo, T hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }   // std::clamp

// The two x-neighbours of a trace's corner as ONE load: an element-aligned 2-vector (global memory takes 8- / 16-byte
// loads at 4- / 8-byte alignment on this part).  A gather is paid per load instruction and distinct cache line, so four
// loads per trace instead of eight is what matters here, not the bytes.
typedef float pair_f __attribute__((ext_vector_type(2), aligned(4)));
typedef double pair_d __attribute__((ext_vector_type(2), aligned(8)));
template <class T> struct PairOf;
template <> struct PairOf<float> { using type = pair_f; };
template <> struct PairOf<double> { using type = pair_d; };
template <class T>
__device__ __forceinline__ typename PairOf<T>::type ld_pair(const T* p) { return *reinterpret_cast<const typename PairOf<T>::type*>(p); }

template <class T>
__device__ __forceinline__ T back_trace_tab(const GridDesc& g, const SlabCtx& sc, const T* __restrict__ src, long zshift,
                                            const T* __restrict__ tab, int x, int y, int z, T ux, T uy, T uz, T kx, T ky, T kz);

// TAB: traces whose x coordinate clamps read the pre-interpolated inlet / outlet column tables (see
// advect_columns_kernel below) instead of the eight scattered values of the big array.
template <class T, bool TAB>
__global__ __launch_bounds__(256) void advect_kernel(GridDesc g, SlabCtx sc, int b, T* __restrict__ field,
                                                      const T* __restrict__ prev, const T* vx, const T* vy,
                                                      const T* vz, const uint8_t* __restrict__ flags, T kx, T ky, T kz,
                                                      long prev_zshift, const T* __restrict__ tab)
{
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y);   // cell_block(): a wave is one row
    const int z = 1 + blockIdx.z;
    if (x > g.W || y > g.H) return;
    const long c = cell(g, x, y, z);
    const unsigned f = flags[c];
    const T one = (T)1, half = (T)0.5;
    T u = (T)0;
    if (!(f & F_SOLID)) {
        // :380-382: the component being advected is carried by its own pre-advection value; the density (b = 0) by the
        // current velocity alone -- its own value is not read at all (4 of 24 streamed bytes per cell)
        T ux, uy, uz;
        if (b == 0) {                                    // uniform
            ux = vx[c]; uy = vy[c]; uz = vz[c];
        } else {                                         // (the array being written is never read: `field` is restrict)
            const T own = prev[c + prev_zshift];
            ux = (b == 1) ? own : vx[c];
            uy = (b == 2) ? own : vy[c];
            uz = (b == 3) ? own : vz[c];
        }
        if constexpr (TAB) {
            u = back_trace_tab<T>(g, sc, prev, prev_zshift, tab, x, y, z, ux, uy, uz, kx, ky, kz);
        } else {
        const int zg = z + sc.zoff;                      // global plane index
        T px = clamp_ref<T>((T)x - kx * ux, half, (T)g.W + half);          // :384-390
        T py = clamp_ref<T>((T)y - ky * uy, half, (T)g.H + half);
        T pz = clamp_ref<T>((T)zg - kz * uz, half, (T)sc.Dglobal + half);
        const int x0 = (int)floor(px), y0 = (int)floor(py), z0 = (int)floor(pz);
        const T tx = px - (T)x0, ty = py - (T)y0, tz = pz - (T)z0;
        const T* s = prev + prev_zshift + cell(g, x0, y0, z0 - sc.zoff);
        const auto q00 = ld_pair(s), q01 = ld_pair(s + g.sz), q10 = ld_pair(s + g.sy), q11 = ld_pair(s + g.sy + g.sz);
        const T a00 = q00.x * (one - tx) + q00.y * tx;                        // :412-415
        const T a01 = q01.x * (one - tx) + q01.y * tx;
        const T a10 = q10.x * (one - tx) + q10.y * tx;
        const T a11 = q11.x * (one - tx) + q11.y * tx;
        const T b0 = a00 * (one - ty) + a10 * ty;                             // :417-418
        const T b1 = a01 * (one - ty) + a11 * ty;
        u = b0 * (one - tz) + b1 * tz;                                        // :420
        }
    }
    const bool kill = (b != 0) && (f & F_NEAR);
    field[c] = kill ? (T)0 : u;
    write_face_ghosts(g, sc, field, c, x, y, z, u, b);
}

// The three velocity advections of a step (simulation.cpp:125-127) in one pass.  advect(2)
// and advect(3) read the already advected v_x / v_y only at their own cell (:380-382), so one
// thread can chain them: trace v_x, then v_y with the new (stored) v_x, then v_z with the new
// v_x and v_y.  Eight array streams instead of fifteen; results are bit-identical.
template <class T>
__device__ __forceinline__ T back_trace(const GridDesc& g, const SlabCtx& sc, const T* __restrict__ src, long zshift,
                                        int x, int y, int z, T ux, T uy, T uz, T kx, T ky, T kz)
{
    // z is the local plane; the trace is clamped in global coordinates and `src + zshift` is indexed
    // with local planes (zshift = zoff planes when src is the gathered global array of a slab)
    const T one = (T)1, half = (T)0.5;
    T px = clamp_ref<T>((T)x - kx * ux, half, (T)g.W + half);          // :384-390
    T py = clamp_ref<T>((T)y - ky * uy, half, (T)g.H + half);
    T pz = clamp_ref<T>((T)(z + sc.zoff) - kz * uz, half, (T)sc.Dglobal + half);
    const int x0 = (int)floor(px), y0 = (int)floor(py), z0 = (int)floor(pz);
    const T tx = px - (T)x0, ty = py - (T)y0, tz = pz - (T)z0;
    const T* s = src + zshift + cell(g, x0, y0, z0 - sc.zoff);
    const auto q00 = ld_pair(s), q01 = ld_pair(s + g.sz), q10 = ld_pair(s + g.sy), q11 = ld_pair(s + g.sy + g.sz);
    const T a00 = q00.x * (one - tx) + q00.y * tx;                        // :412-415
    const T a01 = q01.x * (one - tx) + q01.y * tx;
    const T a10 = q10.x * (one - tx) + q10.y * tx;
    const T a11 = q11.x * (one - tx) + q11.y * tx;
    const T b0 = a00 * (one - ty) + a10 * ty;                             // :417-418
    const T b1 = a01 * (one - ty) + a11 * ty;
    return b0 * (one - tz) + b1 * tz;                                     // :420
}

template <class T, bool TAB>
__global__ __launch_bounds__(256) void advect_velocity_kernel(GridDesc g, SlabCtx sc, T* __restrict__ vx,
                                                               T* __restrict__ vy, T* __restrict__ vz,
                                                               const T* __restrict__ px, const T* __restrict__ py,
                                                               const T* __restrict__ pz,
                                                               const uint8_t* __restrict__ flags, T kx, T ky, T kz,
                                                               long zshift, const T* __restrict__ tab)
{
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y);   // cell_block(): a wave is one row
    const int z = 1 + blockIdx.z;
    if (x > g.W || y > g.H) return;
    const long c = cell(g, x, y, z);
    const unsigned f = flags[c];
    const bool near = (f & F_NEAR) != 0;
    T nx = (T)0, ny = (T)0, nz = (T)0;                   // un-zeroed results (0 inside solids, :375-377)
    T sx = (T)0, sy = (T)0;                              // what setBounds leaves in v_x, v_y
    if (!(f & F_SOLID)) {
        const T oy = vy[c], oz = vz[c];
        if constexpr (TAB) {
            const long plane2 = 2 * (long)(g.H + 2) * (g.D + 2);
            nx = back_trace_tab<T>(g, sc, px, zshift, tab, x, y, z, px[c + zshift], oy, oz, kx, ky, kz);
            sx = near ? (T)0 : nx;
            ny = back_trace_tab<T>(g, sc, py, zshift, tab + plane2, x, y, z, sx, py[c + zshift], oz, kx, ky, kz);
            sy = near ? (T)0 : ny;
            nz = back_trace_tab<T>(g, sc, pz, zshift, tab + 2 * plane2, x, y, z, sx, sy, pz[c + zshift], kx, ky, kz);
        } else {
            nx = back_trace<T>(g, sc, px, zshift, x, y, z, px[c + zshift], oy, oz, kx, ky, kz);
            sx = near ? (T)0 : nx;
            ny = back_trace<T>(g, sc, py, zshift, x, y, z, sx, py[c + zshift], oz, kx, ky, kz);
            sy = near ? (T)0 : ny;
            nz = back_trace<T>(g, sc, pz, zshift, x, y, z, sx, sy, pz[c + zshift], kx, ky, kz);
        }
    }
    vx[c] = near ? (T)0 : nx;
    vy[c] = near ? (T)0 : ny;
    vz[c] = near ? (T)0 : nz;
    write_face_ghosts(g, sc, vx, c, x, y, z, nx, 1);
    write_face_ghosts(g, sc, vy, c, x, y, z, ny, 2);
    write_face_ghosts(g, sc, vz, c, x, y, z, nz, 3);
}

// ---- row forms of the two advection kernels (option advect_kernels=row; NOT the default) ------------
// Measured slower than the per-cell kernels above on MI355X (512^3: 2.8 against 2.4 ms per step fp32, 5.0
// against 3.5 fp64, profiles/r02g_advect_row_vs_cell_*.json): the per-cell form already gets what this form
// was built for -- a wave's 64 clamped traces land on the same two or three cache lines of the inlet column,
// so its gathers coalesce by themselves -- and it keeps four times as many waves in flight to hide the
// latency of the dependent gather.  Kept because it is exercised by the tests (bit-identical) and documents
// the experiment.
// Same arithmetic, restructured around what the memory system sees:
//  * a lane owns four x-consecutive cells: the velocities, the cell's own source values and the result move as
//    one dwordx4 each (a wave = 1 KiB of a row per stream), the solid / near-solid tests come from the kill byte
//    (one byte per four cells) instead of four flag bytes;
//  * the back-trace is data-dependent, but in a wind tunnel most of it is not: dt*W*u_x is hundreds of cells
//    (SURVEY 7.3-3), so the x coordinate of almost every trace clamps to 0.5 -- x0 = 0, weight exactly 0.5
//    (:388, :392-401) -- and the x interpolation (:412-415) only ever combines the ghost column x = 0 with
//    x = 1 (or x = W with W+1 at the other clamp).  advect_columns_kernel does that interpolation once per
//    (y, z) into two small tables (2 x (H+2)(D+2) values per source field, L2-resident); a clamped trace then
//    reads 4 table values as two 8-byte loads instead of 8 scattered values of the big array.  Same
//    expression, same operands, same rounding: bit-identical with the per-cell kernels (tests).
// Traces that do not clamp gather as before.  Wave shuffles do not apply here: which lanes share source rows
// depends on the velocities.
template <class T>
__global__ __launch_bounds__(256) void advect_columns_kernel(GridDesc g, const T* __restrict__ p0, const T* __restrict__ p1,
                                                              const T* __restrict__ p2, int nsrc, T* __restrict__ tab)
{
    // tab[(2*k + side) * (H+2)(D+2) + y + z*(H+2)]: source k, side 0 = columns (0, 1), side 1 = columns (W, W+1)
    const long plane = (long)(g.H + 2) * (g.D + 2);
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const int y = (int)(i % (g.H + 2)), z = (int)(i / (g.H + 2));
    const T one = (T)1, half = (T)0.5;
    for (int k = 0; k < nsrc; ++k) {
        const T* s = (k == 0 ? p0 : k == 1 ? p1 : p2) + cell(g, 0, y, z);
        tab[(2 * k + 0) * plane + i] = s[0] * (one - half) + s[1] * half;                 // :412-415 with tx = 0.5
        tab[(2 * k + 1) * plane + i] = s[g.W] * (one - half) + s[g.W + 1] * half;
    }
}

template <class T>
__device__ __forceinline__ T back_trace_tab(const GridDesc& g, const SlabCtx& sc, const T* __restrict__ src, long zshift,
                                            const T* __restrict__ tab, int x, int y, int z, T ux, T uy, T uz, T kx, T ky, T kz)
{
    const T one = (T)1, half = (T)0.5;
    const T px = clamp_ref<T>((T)x - kx * ux, half, (T)g.W + half);          // :384-390
    const T py = clamp_ref<T>((T)y - ky * uy, half, (T)g.H + half);
    const T pz = clamp_ref<T>((T)(z + sc.zoff) - kz * uz, half, (T)sc.Dglobal + half);
    const int y0 = (int)floor(py), z0 = (int)floor(pz);
    const T ty = py - (T)y0, tz = pz - (T)z0;
    T a00, a01, a10, a11;
    const bool lo = (px == half), hi = (px == (T)g.W + half);
    if (tab != nullptr && (lo || hi)) {
        // x0 = 0 (or W) and tx = 0.5 exactly: the x interpolation was done by advect_columns_kernel
        const long plane = (long)(g.H + 2) * (g.D + 2);
        const T* t = tab + (hi ? plane : 0) + y0 + (long)z0 * (g.H + 2);
        a00 = t[0];
        a10 = t[1];
        a01 = t[g.H + 2];
        a11 = t[g.H + 3];
    } else {
        const int x0 = (int)floor(px);
        const T tx = px - (T)x0;
        const T* s = src + zshift + cell(g, x0, y0, z0 - sc.zoff);
        const auto q00 = ld_pair(s), q01 = ld_pair(s + g.sz), q10 = ld_pair(s + g.sy), q11 = ld_pair(s + g.sy + g.sz);
        a00 = q00.x * (one - tx) + q00.y * tx;                                // :412-415
        a01 = q01.x * (one - tx) + q01.y * tx;
        a10 = q10.x * (one - tx) + q10.y * tx;
        a11 = q11.x * (one - tx) + q11.y * tx;
    }
    const T b0 = a00 * (one - ty) + a10 * ty;                                 // :417-418
    const T b1 = a01 * (one - ty) + a11 * ty;
    return b0 * (one - tz) + b1 * tz;                                         // :420
}

// advect(b, field, prev) + setBounds(b, field): four cells per lane
template <class T>
__global__ __launch_bounds__(256) void advect_row_kernel(GridDesc g, SlabCtx sc, int b, T* __restrict__ field,
                                                          const T* __restrict__ prev, const T* vx, const T* vy, const T* vz,
                                                          const uint8_t* __restrict__ kill, const T* __restrict__ tab, T kx,
                                                          T ky, T kz, long zshift)
{
    const int x0 = 1 + (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = 1 + blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // a wave is one row
    const int z = 1 + blockIdx.z;
    if (x0 > g.W || y > g.H) return;
    const long c = cell(g, x0, y, z);
    const unsigned kb = kill[(c + 3) >> 2];              // bits 0-3 solid, bits 4-7 solid or next to a solid
    T own[4], ax[4], ay[4], az[4], u[4];
    ld_row(prev + c + zshift, b != 0, own);
    ld_row(vx + c, b != 1, ax);
    ld_row(vy + c, b != 2, ay);
    ld_row(vz + c, b != 3, az);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = (T)0;                                     // solid cells: 0 (:375-377)
        if (x0 + e <= g.W && !((kb >> e) & 1u)) {
            const T ux = (b == 1) ? own[e] : ax[e];      // :380-382
            const T uy = (b == 2) ? own[e] : ay[e];
            const T uz = (b == 3) ? own[e] : az[e];
            u[e] = back_trace_tab<T>(g, sc, prev, zshift, tab, x0 + e, y, z, ux, uy, uz, kx, ky, kz);
        }
    }
    store_row_bounds<T>(g, sc, field, c, x0, y, z, u, (b == 0) ? (kb & 15u) : (kb >> 4), b);
}

// the three velocity advections of a step in one pass (see advect_velocity_kernel): four cells per lane
template <class T>
__global__ __launch_bounds__(256) void advect_velocity_row_kernel(GridDesc g, SlabCtx sc, T* __restrict__ vx, T* __restrict__ vy,
                                                                   T* __restrict__ vz, const T* __restrict__ px,
                                                                   const T* __restrict__ py, const T* __restrict__ pz,
                                                                   const uint8_t* __restrict__ kill, const T* __restrict__ tab,
                                                                   T kx, T ky, T kz, long zshift)
{
    const int x0 = 1 + (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = 1 + blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z = 1 + blockIdx.z;
    if (x0 > g.W || y > g.H) return;
    const long c = cell(g, x0, y, z);
    const unsigned kb = kill[(c + 3) >> 2];
    const unsigned near4 = kb >> 4;
    const long plane2 = 2 * (long)(g.H + 2) * (g.D + 2);
    const T* tx_ = tab, *ty_ = tab ? tab + plane2 : nullptr, *tz_ = tab ? tab + 2 * plane2 : nullptr;
    T ox[4], oy[4], oz[4], qy[4], qz[4], nx[4], ny[4], nz[4];
    ld_row(px + c + zshift, true, ox);
    ld_row(py + c + zshift, true, qy);
    ld_row(pz + c + zshift, true, qz);
    ld_row(vy + c, true, oy);
    ld_row(vz + c, true, oz);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        nx[e] = ny[e] = nz[e] = (T)0;                    // un-zeroed results (0 inside solids, :375-377)
        if (x0 + e <= g.W && !((kb >> e) & 1u)) {
            const bool near = ((near4 >> e) & 1u) != 0;
            nx[e] = back_trace_tab<T>(g, sc, px, zshift, tx_, x0 + e, y, z, ox[e], oy[e], oz[e], kx, ky, kz);
            const T sx = near ? (T)0 : nx[e];            // what setBounds leaves in v_x
            ny[e] = back_trace_tab<T>(g, sc, py, zshift, ty_, x0 + e, y, z, sx, qy[e], oz[e], kx, ky, kz);
            const T sy = near ? (T)0 : ny[e];
            nz[e] = back_trace_tab<T>(g, sc, pz, zshift, tz_, x0 + e, y, z, sx, sy, qz[e], kx, ky, kz);
        }
    }
    store_row_bounds<T>(g, sc, vx, c, x0, y, z, nx, near4, 1);
    store_row_bounds<T>(g, sc, vy, c, x0, y, z, ny, near4, 2);
    store_row_bounds<T>(g, sc, vz, c, x0, y, z, nz, near4, 3);
}

// ---- tile form (option advect_kernels=tile; round 3) ---------------------------------------------------------------
// Where the flow is rough (the properly projected flow of solver=mg: |u_y|, |u_z| of order one, i.e. traces that leave
// their cell by tens of rows and planes) the 64 traces of a wave end in up to 64 different cache lines per load and the
// per-cell kernels spend their time in the L2 -> L1 path.  The x coordinate still clamps to the
// inlet for almost every trace, so what is gathered is the x-interpolated inlet column table (advect_columns_kernel):
// a workgroup owns TY x TZ (y, z) rows of cells over the whole row length, stages the window of that table that
// traces of at most R rows / planes can reach in LDS -- coalesced, once per 32 K cells -- and gathers from there; a
// trace that leaves the window, or does not clamp, takes the per-cell path (global table or array).  Same table values,
// same expressions: bit-identical with the other forms.
template <class T, int NF>
__global__ __launch_bounds__(1024) void advect_tile_kernel(GridDesc g, SlabCtx sc, int b, T* __restrict__ f0, T* __restrict__ f1,
                                                            T* __restrict__ f2, const T* __restrict__ p0, const T* __restrict__ p1,
                                                            const T* __restrict__ p2, const T* vx, const T* vy, const T* vz,
                                                            const uint8_t* __restrict__ flags, T kx, T ky, T kz,
                                                            const T* __restrict__ tab, int R)
{
    constexpr int TY = 8, TZ = 8;
    extern __shared__ unsigned char win_raw[];
    T* win = reinterpret_cast<T*>(win_raw);
    const int ty0 = 1 + blockIdx.x * TY, tz0 = 1 + blockIdx.y * TZ;
    // window of table rows (y) and planes (z) this tile's traces can reach, clipped to the table
    const int wy0 = max(0, ty0 - R), wy1 = min(g.H + 1, ty0 + TY + R), wz0 = max(0, tz0 - R), wz1 = min(g.D + 1, tz0 + TZ + R);
    const int WY = wy1 - wy0 + 1, WZ = wz1 - wz0 + 1;
    const long tplane = (long)(g.H + 2) * (g.D + 2);
    for (int k = 0; k < NF; ++k) {
        const T* t = tab + 2 * k * tplane;                   // the inlet side of source k
        for (int i = threadIdx.x; i < WY * WZ; i += 1024) {
            const int yy = i % WY, zz = i / WY;
            win[(long)k * WY * WZ + i] = t[(wy0 + yy) + (long)(wz0 + zz) * (g.H + 2)];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T one = (T)1, half = (T)0.5;
    // one trace of source k (arguments as back_trace_tab)
    auto trace = [&](int k, const T* src, int x, int y, int z, T ux, T uy, T uz) -> T {
        const T px = clamp_ref<T>((T)x - kx * ux, half, (T)g.W + half);          // :384-390
        const T py = clamp_ref<T>((T)y - ky * uy, half, (T)g.H + half);
        const T pz = clamp_ref<T>((T)z - kz * uz, half, (T)g.D + half);
        const int y0 = (int)floor(py), z0 = (int)floor(pz);
        if (px == half && y0 >= wy0 && y0 + 1 <= wy1 && z0 >= wz0 && z0 + 1 <= wz1) {
            const T ty = py - (T)y0, tz = pz - (T)z0;
            const T* w = win + (long)k * WY * WZ + (y0 - wy0) + (long)(z0 - wz0) * WY;
            const T a00 = w[0], a10 = w[1], a01 = w[WY], a11 = w[WY + 1];
            const T b0 = a00 * (one - ty) + a10 * ty;                         // :417-418
            const T b1 = a01 * (one - ty) + a11 * ty;
            return b0 * (one - tz) + b1 * tz;                                 // :420
        }
        return back_trace_tab<T>(g, sc, src, 0, tab + 2 * k * tplane, x, y, z, ux, uy, uz, kx, ky, kz);
    };
    // 16 waves share the TY x TZ rows of the tile; a wave walks its rows 64 cells at a time
    for (int row = wave; row < TY * TZ; row += 16) {
        const int y = ty0 + row % TY, z = tz0 + row / TY;
        if (y > g.H || z > g.D) continue;                    // wave-uniform
        for (int x = 1 + lane; x <= g.W; x += 64) {
            const long c = cell(g, x, y, z);
            const unsigned f = flags[c];
            const bool near = (f & F_NEAR) != 0;
            if constexpr (NF == 1) {
                T u = (T)0;
                if (!(f & F_SOLID)) {
                    T ux, uy, uz;
                    if (b == 0) {                            // uniform
                        ux = vx[c]; uy = vy[c]; uz = vz[c];
                    } else {
                        const T own = p0[c];
                        ux = (b == 1) ? own : vx[c];         // :380-382
                        uy = (b == 2) ? own : vy[c];
                        uz = (b == 3) ? own : vz[c];
                    }
                    u = trace(0, p0, x, y, z, ux, uy, uz);
                }
                f0[c] = ((b != 0) && near) ? (T)0 : u;
                write_face_ghosts(g, sc, f0, c, x, y, z, u, b);
            } else {
                T nx = (T)0, ny = (T)0, nz = (T)0;           // un-zeroed results (0 inside solids, :375-377)
                if (!(f & F_SOLID)) {
                    const T oy = f1[c], oz = f2[c];          // f0, f1, f2 = v_x, v_y, v_z: read at the own cell, then overwritten
                    nx = trace(0, p0, x, y, z, p0[c], oy, oz);
                    const T sx = near ? (T)0 : nx;           // what setBounds leaves in v_x
                    ny = trace(1, p1, x, y, z, sx, p1[c], oz);
                    const T sy = near ? (T)0 : ny;
                    nz = trace(2, p2, x, y, z, sx, sy, p2[c]);
                }
                f0[c] = near ? (T)0 : nx;
                f1[c] = near ? (T)0 : ny;
                f2[c] = near ? (T)0 : nz;
                write_face_ghosts(g, sc, f0, c, x, y, z, nx, 1);
                write_face_ghosts(g, sc, f1, c, x, y, z, ny, 2);
                write_face_ghosts(g, sc, f2, c, x, y, z, nz, 3);
            }
        }
    }
}

// window radius that fits 64 KB of LDS for NF staged tables (the tile is 8 x 8 rows)
template <class T>
static int tile_window(int want, int nf)
{
    int r = want < 1 ? 1 : want;
    while (r > 1 && (long)nf * (8 + 2 * r + 1) * (8 + 2 * r + 1) * (long)sizeof(T) > 64 * 1024) --r;
    return r;
}

static inline dim3 row_grid(const GridDesc& g) { return dim3((g.W + 255) / 256, (g.H + 3) / 4, g.D); }

// the clamp tables only describe a source array that is this GPU's whole domain (a slab traces into the gathered array)
template <class T>
static const T* build_columns(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const T* p0, const T* p1, const T* p2,
                              int nsrc, T* coltab, long zshift)
{
    if (!coltab || zshift != 0 || !(sc.lo_wall && sc.hi_wall)) return nullptr;
    const long plane = (long)(g.H + 2) * (g.D + 2);
    hipLaunchKernelGGL((advect_columns_kernel<T>), dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, g, p0, p1, p2, nsrc,
                       coltab);
    return coltab;
}

template <class T>
void launch_advect_velocity(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, T* vx, T* vy, T* vz,
                            const T* px, const T* py, const T* pz, const uint8_t* flags, const uint8_t* kill, T* coltab, T kx,
                            T ky, T kz, long zshift)
{
    const T* tab = (tune.advect_cell == 1) ? nullptr : build_columns<T>(st, g, sc, px, py, pz, 3, coltab, zshift);
    if (tune.advect_cell == 3 && tab) {
        const int R = tile_window<T>(tune.advect_window, 3);
        const size_t lds = (size_t)3 * (8 + 2 * R + 1) * (8 + 2 * R + 1) * sizeof(T);
        hipLaunchKernelGGL((advect_tile_kernel<T, 3>), dim3((g.H + 7) / 8, (g.D + 7) / 8), dim3(1024), lds, st, g, sc, 0, vx, vy, vz, px,
                           py, pz, vx, vy, vz, flags, kx, ky, kz, tab, R);
        return;
    }
    if (tune.advect_cell) {
        if (tab)
            hipLaunchKernelGGL((advect_velocity_kernel<T, true>), cell_grid(g), cell_block(), 0, st, g, sc, vx, vy, vz, px, py,
                               pz, flags, kx, ky, kz, zshift, tab);
        else
            hipLaunchKernelGGL((advect_velocity_kernel<T, false>), cell_grid(g), cell_block(), 0, st, g, sc, vx, vy, vz, px, py,
                               pz, flags, kx, ky, kz, zshift, tab);
        return;
    }
    hipLaunchKernelGGL((advect_velocity_row_kernel<T>), row_grid(g), dim3(256), 0, st, g, sc, vx, vy, vz, px, py, pz, kill, tab,
                       kx, ky, kz, zshift);
}
template void launch_advect_velocity<float>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, float*, float*,
                                            float*, const float*, const float*, const float*, const uint8_t*, const uint8_t*,
                                            float*, float, float, float, long);
template void launch_advect_velocity<double>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, double*, double*,
                                             double*, const double*, const double*, const double*, const uint8_t*,
                                             const uint8_t*, double*, double, double, double, long);

template <class T>
void launch_advect(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int b, T* field, const T* prev,
                   const T* vx, const T* vy, const T* vz, const uint8_t* flags, const uint8_t* kill, T* coltab, T kx, T ky, T kz,
                   long prev_zshift)
{
    const T* tab = (tune.advect_cell == 1) ? nullptr : build_columns<T>(st, g, sc, prev, prev, prev, 1, coltab, prev_zshift);
    if (tune.advect_cell == 3 && tab) {
        const int R = tile_window<T>(tune.advect_window, 1);
        const size_t lds = (size_t)(8 + 2 * R + 1) * (8 + 2 * R + 1) * sizeof(T);
        hipLaunchKernelGGL((advect_tile_kernel<T, 1>), dim3((g.H + 7) / 8, (g.D + 7) / 8), dim3(1024), lds, st, g, sc, b, field, field,
                           field, prev, prev, prev, vx, vy, vz, flags, kx, ky, kz, tab, R);
        return;
    }
    if (tune.advect_cell) {
        if (tab)
            hipLaunchKernelGGL((advect_kernel<T, true>), cell_grid(g), cell_block(), 0, st, g, sc, b, field, prev, vx, vy, vz,
                               flags, kx, ky, kz, prev_zshift, tab);
        else
            hipLaunchKernelGGL((advect_kernel<T, false>), cell_grid(g), cell_block(), 0, st, g, sc, b, field, prev, vx, vy, vz,
                               flags, kx, ky, kz, prev_zshift, tab);
        return;
    }
    hipLaunchKernelGGL((advect_row_kernel<T>), row_grid(g), dim3(256), 0, st, g, sc, b, field, prev, vx, vy, vz, kill, tab, kx,
                       ky, kz, prev_zshift);
}
template void launch_advect<float>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, int, float*, const float*,
                                   const float*, const float*, const float*, const uint8_t*, const uint8_t*, float*, float, float,
                                   float, long);
template void launch_advect<double>(hipStream_t, const SweepTune&, const GridDesc&, const SlabCtx&, int, double*, const double*,
                                    const double*, const double*, const double*, const uint8_t*, const uint8_t*, double*, double,
                                    double, double, long);

// =====================================================================================
// Flag bytes from the obstacle array.  Tests follow the reference literally:
// solid <=> obs == 1 (simulation.cpp:222), fluid neighbour <=> in range && obs == 0 (:307).
// Under z-slabs "in range" refers to the global depth; the halo planes of `obs` hold the
// neighbouring slabs' cells.
// =====================================================================================
template <class T>
__global__ void build_flags_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ obs, uint8_t* __restrict__ flags,
                                   int zlo)
{
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = zlo + blockIdx.z;
    if (x > g.W || y > g.H) return;
    const long c = cell(g, x, y, z);
    const int zg = z + sc.zoff;
    const bool rxp = x + 1 <= g.W, rxm = x - 1 >= 1, ryp = y + 1 <= g.H, rym = y - 1 >= 1;
    const bool rzp = zg + 1 <= sc.Dglobal, rzm = zg - 1 >= 1;
    const T one = (T)1, zero = (T)0;
    unsigned f = 0;
    if (obs[c] == one) {
        f = F_SOLID;
    } else {
        bool near = (rxp && obs[c + 1] == one) || (rxm && obs[c - 1] == one) || (ryp && obs[c + g.sy] == one) ||
                    (rym && obs[c - g.sy] == one) || (rzp && obs[c + g.sz] == one) || (rzm && obs[c - g.sz] == one);
        if (near) f |= F_NEAR;
    }
    if (rxp && obs[c + 1] == zero) f |= F_XP;
    if (rxm && obs[c - 1] == zero) f |= F_XM;
    if (ryp && obs[c + g.sy] == zero) f |= F_YP;
    if (rym && obs[c - g.sy] == zero) f |= F_YM;
    if (rzp && obs[c + g.sz] == zero) f |= F_ZP;
    if (rzm && obs[c - g.sz] == zero) f |= F_ZM;
    flags[c] = (uint8_t)f;
}

template <class T>
void launch_build_flags(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const T* obs, uint8_t* flags)
{
    // with zh-deep halos the flags of the first zh-1 halo planes on a slab side are needed too (a fused
    // pass recomputes the lower levels of those planes); the outermost halo plane only serves as their z neighbour
    const int zlo = (g.zh >= 2 && !sc.lo_wall) ? 2 - g.zh : 1, zhi = (g.zh >= 2 && !sc.hi_wall) ? g.D + g.zh - 1 : g.D;
    dim3 grid = cell_grid(g);
    grid.z = zhi - zlo + 1;
    hipLaunchKernelGGL((build_flags_kernel<T>), grid, cell_block(), 0, st, g, sc, obs, flags, zlo);
}
template void launch_build_flags<float>(hipStream_t, const GridDesc&, const SlabCtx&, const float*, uint8_t*);
template void launch_build_flags<double>(hipStream_t, const GridDesc&, const SlabCtx&, const double*, uint8_t*);

// Kill bytes for the sweep kernels: one byte per lane group (four x-consecutive cells starting at
// x = 1 mod 4), bits 0-3 = solid, bits 4-7 = solid or next to a solid; byte index = (cell + 3) / 4.
// A quarter of the flag traffic, and the sweeps need nothing else of the flags.
__global__ void build_kill_kernel(GridDesc g, const uint8_t* __restrict__ flags, uint8_t* __restrict__ kill, int zlo)
{
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;            // group index along x
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = zlo + blockIdx.z;
    const int x0 = 1 + 4 * gx;
    if (x0 > g.W || y > g.H) return;
    const long c = cell(g, x0, y, z);
    const unsigned f = *reinterpret_cast<const unsigned*>(flags + c);
    unsigned out = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned fe = (f >> (8 * e)) & 0xffu;
        if (fe & F_SOLID) out |= 1u << e;
        if (fe & (F_SOLID | F_NEAR)) out |= 16u << e;
    }
    kill[(c + 3) >> 2] = (uint8_t)out;
}
void launch_build_kill(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, uint8_t* kill)
{
    const int zlo = (g.zh >= 2 && !sc.lo_wall) ? 2 - g.zh : 1, zhi = (g.zh >= 2 && !sc.hi_wall) ? g.D + g.zh - 1 : g.D;
    const int ng = (g.W + 3) / 4;
    hipLaunchKernelGGL(build_kill_kernel, dim3((ng + 63) / 64, (g.H + 3) / 4, zhi - zlo + 1), dim3(64, 4, 1), 0, st, g,
                       flags, kill, zlo);
}

// =====================================================================================
// Inlet forcing: velocity (speed,0,0) on the x=1 face (simulation.cpp:103-105) and
// +amount density on the same face (simulation.cpp:65-67).
// =====================================================================================
template <class T>
__global__ void inlet_velocity_kernel(GridDesc g, T* vx, T* vy, T* vz, T speed, int zlo)
{
    const int y = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int z = zlo + blockIdx.y;
    if (y > g.H) return;
    const long c = cell(g, 1, y, z);
    vx[c] = speed;
    vy[c] = (T)0;
    vz[c] = (T)0;
}
template <class T>
__global__ void inlet_density_kernel(GridDesc g, T* dens, T amount, int zlo)
{
    const int y = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int z = zlo + blockIdx.y;
    if (y > g.H) return;
    dens[cell(g, 1, y, z)] += amount;
}
template <class T>
void launch_inlet_velocity(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* vx, T* vy, T* vz, T speed)
{
    // a slab also forces the inlet cells of its halo planes (they are interior planes of the
    // neighbouring slab), so the halos stay current without an exchange
    const int zlo = sc.lo_wall ? 1 : 1 - g.zh, zhi = sc.hi_wall ? g.D : g.D + g.zh;
    hipLaunchKernelGGL((inlet_velocity_kernel<T>), dim3((g.H + 63) / 64, zhi - zlo + 1), dim3(64), 0, st, g, vx, vy, vz,
                       speed, zlo);
}
template <class T>
void launch_inlet_density(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* dens, T amount)
{
    const int zlo = sc.lo_wall ? 1 : 1 - g.zh, zhi = sc.hi_wall ? g.D : g.D + g.zh;
    hipLaunchKernelGGL((inlet_density_kernel<T>), dim3((g.H + 63) / 64, zhi - zlo + 1), dim3(64), 0, st, g, dens, amount,
                       zlo);
}
template void launch_inlet_velocity<float>(hipStream_t, const GridDesc&, const SlabCtx&, float*, float*, float*, float);
template void launch_inlet_velocity<double>(hipStream_t, const GridDesc&, const SlabCtx&, double*, double*, double*, double);
template void launch_inlet_density<float>(hipStream_t, const GridDesc&, const SlabCtx&, float*, float);
template void launch_inlet_density<double>(hipStream_t, const GridDesc&, const SlabCtx&, double*, double);

// =====================================================================================
// Layout conversion: pitched device field <-> the reference's dense padded array
// (simulation.h:9, the frame-dump layout of simulation.cpp:143-147).
// =====================================================================================
template <class T, class U>
__global__ void pack_kernel(GridDesc g, const T* __restrict__ f, U* __restrict__ dense, int zlo)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = zlo + blockIdx.z;
    if (x > g.W + 1 || y > g.H + 1) return;
    dense[(long)x + (long)y * (g.W + 2) + (long)(z - zlo) * (g.W + 2) * (g.H + 2)] = (U)f[cell(g, x, y, z)];
}
template <class T, class U>
__global__ void unpack_kernel(GridDesc g, const U* __restrict__ dense, T* __restrict__ f, int zlo)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = zlo + blockIdx.z;
    if (x > g.W + 1 || y > g.H + 1) return;
    f[cell(g, x, y, z)] = (T)dense[(long)x + (long)y * (g.W + 2) + (long)(z - zlo) * (g.W + 2) * (g.H + 2)];
}
template <class T, class U>
void launch_pack(hipStream_t st, const GridDesc& g, const T* field, U* dense, int zlo, int zhi)
{
    if (zhi < zlo) return;
    hipLaunchKernelGGL((pack_kernel<T, U>), dim3((g.W + 2 + 63) / 64, (g.H + 2 + 3) / 4, zhi - zlo + 1), dim3(64, 4, 1),
                       0, st, g, field, dense, zlo);
}
template <class T, class U>
void launch_unpack(hipStream_t st, const GridDesc& g, const U* dense, T* field, int zlo, int zhi)
{
    if (zhi < zlo) return;
    hipLaunchKernelGGL((unpack_kernel<T, U>), dim3((g.W + 2 + 63) / 64, (g.H + 2 + 3) / 4, zhi - zlo + 1),
                       dim3(64, 4, 1), 0, st, g, dense, field, zlo);
}
#define FS_INST_PACK(T, U)                                                                     \
    template void launch_pack<T, U>(hipStream_t, const GridDesc&, const T*, U*, int, int);     \
    template void launch_unpack<T, U>(hipStream_t, const GridDesc&, const U*, T*, int, int);
FS_INST_PACK(float, float)
FS_INST_PACK(float, double)
FS_INST_PACK(double, float)
FS_INST_PACK(double, double)
FS_INST_PACK(float, uint8_t)
FS_INST_PACK(double, uint8_t)

template <class T>
__global__ void copy_kernel(const T* __restrict__ src, T* __restrict__ dst, long n4)
{
    // whole allocation, 16 B per lane (allocation length is a multiple of 4 elements)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<V4<T>*>(dst)[i] = reinterpret_cast<const V4<T>*>(src)[i];
}
template <class T>
void launch_copy(hipStream_t st, const GridDesc& g, const T* src, T* dst)
{
    // src/dst are the shifted pointers; copy the underlying allocations
    long n4 = g.n / 4;
    hipLaunchKernelGGL((copy_kernel<T>), dim3(2048), dim3(256), 0, st, src - g.lead, dst - g.lead, n4);
}
template void launch_copy<float>(hipStream_t, const GridDesc&, const float*, float*);
template void launch_copy<double>(hipStream_t, const GridDesc&, const double*, double*);

// =====================================================================================
// Diagnostics of Simulation::run(): sum / min / max of a whole padded array
// (simulation.cpp:73-90).  Two-stage reduction in double.
// =====================================================================================
template <class T>
__global__ __launch_bounds__(256) void stats_partial_kernel(GridDesc g, const T* __restrict__ f, double* part, int zlo,
                                                             int zhi)
{
    const long rows = (long)(g.H + 2) * (zhi - zlo + 1);
    double s = 0.0, mn = 1e300, mx = -1e300;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        int y = (int)(r % (g.H + 2)), z = zlo + (int)(r / (g.H + 2));
        for (int x = threadIdx.x; x <= g.W + 1; x += blockDim.x) {
            double v = (double)f[cell(g, x, y, z)];
            s += v;
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
    }
    __shared__ double sh[3][256];
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = mn; sh[2][threadIdx.x] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + k];
            sh[1][threadIdx.x] = fmin(sh[1][threadIdx.x], sh[1][threadIdx.x + k]);
            sh[2][threadIdx.x] = fmax(sh[2][threadIdx.x], sh[2][threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x + 0] = sh[0][0];
        part[3 * blockIdx.x + 1] = sh[1][0];
        part[3 * blockIdx.x + 2] = sh[2][0];
    }
}
__global__ void stats_final_kernel(const double* part, int n, double* out3)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0, mn = 1e300, mx = -1e300;
    for (int i = 0; i < n; ++i) {
        s += part[3 * i];
        mn = fmin(mn, part[3 * i + 1]);
        mx = fmax(mx, part[3 * i + 2]);
    }
    out3[0] = s; out3[1] = mn; out3[2] = mx;
}
template <class T>
void launch_stats(hipStream_t st, const GridDesc& g, const T* field, double* out3, double* scratch, int nscratch,
                  int zlo, int zhi)
{
    int nb = nscratch / 3;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL((stats_partial_kernel<T>), dim3(nb), dim3(256), 0, st, g, field, scratch, zlo, zhi);
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(64), 0, st, scratch, nb, out3);
}
template void launch_stats<float>(hipStream_t, const GridDesc&, const float*, double*, double*, int, int, int);
template void launch_stats<double>(hipStream_t, const GridDesc&, const double*, double*, double*, int, int, int);

template <class T>
__global__ void point_kernel(T* p, long idx, T amount, int set_instead)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) p[idx] = set_instead ? amount : p[idx] + amount;
}
template <class T>
void launch_point_add(hipStream_t st, T* p, long idx, T amount, int set_instead)
{
    hipLaunchKernelGGL((point_kernel<T>), dim3(1), dim3(64), 0, st, p, idx, amount, set_instead);
}
template void launch_point_add<float>(hipStream_t, float*, long, float, int);
template void launch_point_add<double>(hipStream_t, double*, long, double, int);

}  // namespace fs

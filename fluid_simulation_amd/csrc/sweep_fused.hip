// sweep_fused.hip -- NL Jacobi sweeps per pass over memory (temporal blocking), register-centred.
//
//   linearSolver  simulation.cpp:251-273 (NL iterations of the k loop, neighbours read from the
//                 previous iterate), each followed by
//   setBounds     simulation.cpp:183-246,
// bit-identical with NL launches of jacobi_sweep_kernel (kernels.hip).
//
// One sweep moves 12 B per cell (fp32) for 8 flops; the only way past the HBM roof is to apply
// several sweeps while the data is on chip.  Level 0 = `src`, level j = the iterate after j
// sweeps + setBounds (levels 1..NL-1 never reach memory), level NL = `dst`.
//
// A workgroup owns the full row width (NXW waves of 256 cells) times a band of BY = NYW*RY rows
// and marches along z.  A wave keeps its own cells of every level for three consecutive planes in
// registers (three slots per level, rotated by unrolling the march three times, so nothing is
// ever copied); LDS only carries what a wave needs from its neighbours -- the rows just above
// and below its patch and the two columns beside it -- in a two-slot ring per intermediate level.
// In iteration zl a wave computes level 1 of plane zl, level 2 of plane zl-1 (and level 3 of plane
// zl-2); level j+1 of plane P reads the level-j tile of plane P that was published one iteration
// earlier, so one barrier per iteration is enough and two slots per ring suffice.  The right-hand
// side of a plane is needed in NL consecutive iterations; each thread parks its own values in a
// private LDS ring instead of holding NL+1 planes of it in registers.
//
// setBounds between the levels, exactly as it would happen in memory: interior cells are zeroed
// (settle4); the ghost columns x = 0, W+1 and ghost rows y = 0, H+1 of a level are written into
// the LDS tile from the unzeroed values (simulation.cpp:186-201), the ghost planes z = 0, D+1
// into the register slot the missing plane would occupy (:208-214).
// Bands overlap by 2(NL-1) rows and z chunks by 2(NL-1) planes.
//
// Instantiations (LDS = (NL-1)*2*BY*TW + NL*(BY-2)*RW elements must stay under 160 KB):
//   fp32 NL=3  rows <= 512 cells   the solver kernel at 256^3 and 512^3: 0.43 ms per pass at 512^3
//                                  = 0.14 ms per sweep, 12 waves of two rows, 152 VGPRs, no scratch
//   fp32 NL=2  rows 513..1024      config 4: bands of 8 or 9 rows (6 or 7 productive) where the older
//                                  jacobi_pair_kernel (kernels.hip) only fits 6 (4 productive)
//   fp64 NL=2  rows <= 512         config 5: bands of 10 rows (8 productive) against 8 (6)
// The two NL=2 families are candidates of the host's one-off timing, which so far always prefers the pair
// kernel to them (0.83 vs 0.71 ms at 1024x512x512, 0.84 vs 0.77 fp64 at 512^3: both sit at the same memory
// ceiling, and per level this structure costs more instructions); three sweeps per pass is where it pays.
// SLAB = true adds what a z-slab of a multi-GPU run needs: an output plane range (the boundary
// regions are computed first), no physical wall on a side that borders another slab -- the levels
// are then computed NL-1 planes into the neighbour's planes from the NL-deep halo -- and an optional
// second range in the same launch.  SLAB = false compiles all of that away (whole domain on one GPU).
//
// What made the difference in speed is written up in DESIGN.md section 4 (uniform wave index through
// readfirstlane, scalar plane pointers + opaque 32-bit lane offsets, unpredicated loads).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "kernels_dev.h"

namespace fs {

template <int N>
struct IC {
    static constexpr int value = N;
};

// Experiment switch (build with -DFS_EXP_NT=1|2|3): nontemporal loads of the right-hand side (1) and / or nontemporal
// stores of the result (2).  Measured at 512^3 on one box (profiles/r02k_nontemporal_ab_c3.json): 0 -> 0.1465, 1 -> 0.150,
// 2 -> 0.146, 3 -> 0.148 ms per sweep; the shipped library is built with 0.
#ifndef FS_EXP_NT
#define FS_EXP_NT 0
#endif
template <class T>
using Vec4 = T __attribute__((ext_vector_type(4)));

template <class T, int NL, int NXW, int NYW, int RY, bool ALIGNED, bool SLAB, int WALLSEL>
__global__ __launch_bounds__(NXW* NYW * 64) void jacobi_fused_kernel(GridDesc g, SlabCtx sc, const T* __restrict__ src,
                                                                      const T* __restrict__ rhs, T* __restrict__ dst,
                                                                      const uint8_t* __restrict__ flags, int b, T a, T inv_c,
                                                                      int z_first, int z_last, int zc_len, int z_stride,
                                                                      int nbands, int nblk, PeerPush pp)
{
    static_assert(NL == 2 || NL == 3, "two or three sweeps per pass");
    constexpr int BY = NYW * RY, TW = NXW * 256 + 8, RW = NXW * 256;
    constexpr int OV = NL - 1;                           // rows / planes a band / chunk loses per side
    constexpr int ES = (int)sizeof(T);
    // WALLSEL = 2, the "lean interior" experiment (round-2 verdict, item 5): no right-hand-side ring -- the later levels re-read
    // the rhs rows from memory (L2 / MALL hits) -- which frees 61 KB of LDS for 16-row bands, and only the wall-free body
    // (workgroups that touch no wall).  See DESIGN.md section 4 for what it measured.
    constexpr bool RS = (WALLSEL != 2);
    static_assert(((NL - 1) * 2 * BY * TW + (RS ? NL * (BY - 2) * RW : 0)) * ES <= 160 * 1024, "LDS budget");
    __shared__ T ring[NL - 1][2][BY][TW];                // [level-1][plane & 1][tile row][x + 3]
    __shared__ T rsave[RS ? NL : 1][RS ? BY - 2 : 1][RW];   // thread-private: rhs of the last NL planes

    const int v = xcd_contiguous(blockIdx.x, nblk);
    const int band = v % nbands, zc = v / nbands;
    // readfirstlane: the wave index is the same in all 64 lanes, but only this tells the compiler so -- rows,
    // row pointers and every row test then live in scalar registers and branch as scalars
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wx = wave % NXW, wy = wave / NXW;
    const int W = g.W, H = g.H, D = g.D;
    // tile row t <-> grid row s + t.  Band k outputs the BY - 2 OV rows k (BY - 2 OV) + 1 ..: every band the same number,
    // also the bands at the y walls (a wall costs no rows, so with s = k (BY - 2 OV) the first band would output NL - 2 more
    // than the others; round 3 shifts the origin instead, which makes the wall bands the lighter ones)
    const int s = band * (BY - 2 * OV) - (OV - 1);
    const int ty0 = wy * RY, y0 = s + ty0;
    const int xl = wx * 256 + lane * 4;
    const int x0 = 1 + xl;
    const bool lane_on = ALIGNED || (x0 <= W);
    const bool full_group = ALIGNED || (x0 + 3 <= W);
    // physical z walls: both on one GPU; on a slab only the first / last rank has one
    const bool lo_wall = SLAB ? (sc.lo_wall != 0) : true, hi_wall = SLAB ? (sc.hi_wall != 0) : true;
    const int zbeg = SLAB ? z_first + zc * z_stride : 1 + zc * zc_len;                  // level-NL output planes
    const int zend = min(SLAB ? z_last : D, zbeg + zc_len - 1);
    if (zbeg > zend) return;                             // block-uniform
    // planes of level j: NL-j beyond the output chunk on each side; beyond a physical wall there is no
    // such plane (its ghost is derived below), beyond a slab boundary it is the neighbour's plane,
    // recomputed here from the NL-deep halo
    const int zlo_lim = lo_wall ? 1 : 1 - OV, zhi_lim = hi_wall ? D : D + OV;
    const int lo1 = max(zlo_lim, zbeg - OV), hi1 = min(zhi_lim, zend + OV);             // level-1 planes
    const int lo2 = max(zlo_lim, zbeg - (OV - 1)), hi2 = min(zhi_lim, zend + (OV - 1)); // level-2 planes
    const int zmax0 = hi_wall ? D + 1 : D + NL, zmax1 = hi_wall ? D : D + OV;           // last level-0 / level-1 plane that exists
    // Everything below can exist twice (WALLSEL = 1): the general body, and one for workgroups that touch no wall in y or z
    // (interior bands x interior z chunks), in which every wall test, ghost row / ghost plane and the registers that carry
    // them are compiled out -- 523 instead of 837 instructions per plane, 126 instead of 152 VGPRs on its own.  The two are
    // separate loops selected once per workgroup (two bodies inside ONE loop spill at their merge, round 1).
    // The march's state lives outside the bodies so that a workgroup can change body between plane iterations (round 3):
    T L0[3][RY][4], L1[3][RY][4] = {}, L2[3][RY][4] = {};
    T hb[4], ht[4], eL[RY], eR[RY], rcur[RY][4];
    T gz1[RY][4] = {}, gz2[RY][4] = {};                 // ghost plane D+1 of levels 1 and 2 while it waits for its register slot
    unsigned flc[RY], kl[3][RY] = {};                    // (not in idle load registers: a select on those would wait for the loads)
    // `ngroups` groups of three plane iterations from z_from on with one of the bodies (the register slots rotate with period
    // three: a group starts and ends with every plane in its home slot, so bodies can alternate between groups), then, with
    // tail_to >= 0, the march's last one or two iterations up to tail_to
    auto run = [&](auto wallc, int z_from, int ngroups, int tail_to) {
    constexpr bool WALLS = decltype(wallc)::value != 0;
    const bool lo_wall_c = WALLS && lo_wall, hi_wall_c = WALLS && hi_wall;      // the ghost-plane code of the physical z walls
    // rows a level can be computed for: one fewer per level at a band edge, none lost at a wall
    const bool top_in_tile = WALLS && (s + BY - 1 >= H + 1), bottom_in_tile = WALLS && (s <= 0);
    const int r2lo = bottom_in_tile ? 1 : s + 1, r2hi = top_in_tile ? H : s + BY - 2;
    const int r3lo = bottom_in_tile ? 1 : s + 2, r3hi = top_in_tile ? H : s + BY - 3;
    const int kill_shift = (b == 0) ? 0 : 4;
    const T zero = (T)0;

    // Addressing: a plane pointer (wave-uniform, advanced by the march) plus a 32-bit byte offset per
    // row that never changes.  Loads are never predicated (a predicated load merges with a default
    // value, which makes the compiler wait for it on the spot instead of one iteration later): rows
    // outside the array are clamped to a row inside it and lanes beyond the row end read the row
    // start; what they fetch is never used for a cell that exists.
    // Per-row byte offsets inside a plane sit in vector registers, the plane pointers in scalar ones --
    // the scalar file is the scarce one here (every spilled scalar costs a v_readlane plus hazard nops).
    const unsigned col0 = lane_on ? (unsigned)x0 : 1u;
    auto clampy = [&](int y) { return (unsigned)min(max(y, 0), H + 1); };
    unsigned oc[RY];                                     // bytes into a plane of T; kill byte index = (element + 3) >> 2
#pragma unroll
    for (int r = 0; r < RY; ++r) oc[r] = (col0 + clampy(y0 + r) * (unsigned)g.sy) * (unsigned)ES;
    const long plane_b = (long)g.sz * ES, plane_f = (long)(g.sz >> 2), row_b = (long)g.sy * ES;
    // rows below / above the patch: the first / last own row's offset with a wave-uniform step (0 where clamped)
    const long step_b = (long)(clampy(y0) - clampy(y0 - 1)) * row_b, step_t = (long)(clampy(y0 + RY) - clampy(y0 + RY - 1)) * row_b;
    auto plane_of = [&](const T* base, int z) { return reinterpret_cast<const char*>(base) + (long)z * plane_b; };   // wave-uniform
    auto ld4 = [&](const char* ptr, T (&out)[4]) {
        V4<T> q = *reinterpret_cast<const V4<T>*>(ptr);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = q.e[e];
    };

    auto load_core = [&](int z, T (&out)[RY][4]) {
        const char* sp = plane_of(src, z);
#pragma unroll
        for (int r = 0; r < RY; ++r) ld4(sp + oc[r], out[r]);
    };
    auto load_side = [&](int z) {                        // what level 1 of plane z needs beside the wave's own cells
        const char* sp = plane_of(src, z);
        const char* rp = plane_of(rhs, z);
        const uint8_t* fp = flags + (long)z * plane_f;
        ld4(sp - step_b + oc[0], hb);
        ld4(sp + step_t + oc[RY - 1], ht);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            eL[r] = *reinterpret_cast<const T*>(sp + oc[r] - ES);        // every lane fetches its own x neighbours:
            eR[r] = *reinterpret_cast<const T*>(sp + oc[r] + 4 * ES);    // no shuffles, no edge lanes
            if (FS_EXP_NT & 1) {
                const Vec4<T> q = __builtin_nontemporal_load(reinterpret_cast<const Vec4<T>*>(rp + oc[r]));
#pragma unroll
                for (int e = 0; e < 4; ++e) rcur[r][e] = q[e];
            } else ld4(rp + oc[r], rcur[r]);
            flc[r] = (unsigned)fp[(oc[r] / (unsigned)ES + 3u) >> 2];
        }
    };

    // one stencil application; simulation.cpp:264-269 order x+1, x-1, y+1, y-1, z+1, z-1
    // (Writing the four cells as two interleaved packed chains removes 55 of the 73 hazard s_nops per plane iteration of
    // the wall-free body and gains nothing -- 0.1301 against 0.1282 ms per sweep, round 3: the other waves of the SIMD fill
    // those slots.)
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
    // what setBounds leaves in memory for the lane's four cells of an interior row.  Most waves
    // have no solid cell anywhere near: they skip the zeroing logic on one wave-uniform test.
    auto settle4 = [&](const T (&u)[4], unsigned fl, T (&st)[4]) {
        const unsigned kb = (fl >> kill_shift) & 15u;
        if (ALIGNED && __builtin_amdgcn_ballot_w64(kb != 0) == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) st[e] = u[e];
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool kill = ((kb >> e) & 1u) != 0;
            if (ALIGNED) st[e] = kill ? zero : u[e];
            else {
                const int x = x0 + e;
                T ghost_src = (e > 0) ? u[e - 1] : zero;
                st[e] = (x <= W) ? (kill ? zero : u[e]) : ((x == W + 1) ? ghost_src : zero);
            }
        }
    };
    auto lds_get = [&](const T* rowp, T (&out)[4]) {
        V4<T> q = *reinterpret_cast<const V4<T>*>(rowp);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = q.e[e];
    };
    auto lds_set = [&](T* rowp, const T (&in)[4]) {
        V4<T> q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q.e[e] = in[e];
        *reinterpret_cast<V4<T>*>(rowp) = q;
    };
    auto face4 = [&](const T (&u)[4], bool negate, T (&out)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = (ALIGNED || x0 + e <= W) ? (negate ? -u[e] : u[e]) : zero;
    };
    // publish a freshly computed row of level RG+1: the settled cells plus the ghosts its setBounds writes
    auto publish = [&](auto rgc, int P, int r, const T (&u)[4], const T (&st)[4]) {
        constexpr int RG = decltype(rgc)::value;
        const int y = y0 + r, t = ty0 + r;
        T(*tl)[TW] = ring[RG][P & 1];
        lds_set(&tl[t][x0 + 3], st);
        if (x0 == 1) tl[t][3] = (b == 1) ? -u[0] : u[0];                                          // :189-190
        if (full_group && x0 + 3 == W) tl[t][W + 4] = u[3];                                       // :191
        if (WALLS && (y == 1 || y == H)) {
            T f[4];
            face4(u, b == 2, f);
            if (y == 1 && t >= 1) lds_set(&tl[t - 1][x0 + 3], f);                                  // :198-199
            if (y == H && t + 1 < BY) lds_set(&tl[t + 1][x0 + 3], f);                              // :200-201
        }
    };

    // level j+1 of plane P, row r, from the wave's level-j registers (planes P-1, P, P+1) and the
    // level-j tile of plane P in LDS (x neighbours, and the y neighbours the wave does not own)
    auto next_row = [&](auto rgc, int P, int r, const T (&s0)[RY][4], const T (&s1)[RY][4], const T (&s2)[RY][4],
                        const T (&rh)[4], T (&u)[4]) {
        constexpr int RG = decltype(rgc)::value;
        const int y = y0 + r, t = ty0 + r;
        T(*tl)[TW] = ring[RG][P & 1];
        const T left = tl[t][x0 + 2];                    // neighbour lane's / wave's cell, or the ghost column x = 0
        const T right = tl[t][x0 + 7];                   // ... or the ghost column x = W+1
        T ym[4], yp[4];
        if (r > 0 && (!WALLS || y != 1)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ym[e] = s1[r > 0 ? r - 1 : 0][e];
        } else lds_get(&tl[t - 1][x0 + 3], ym);          // another wave's row, or the ghost row y = 0
        if (r < RY - 1 && (!WALLS || y != H)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) yp[e] = s1[r < RY - 1 ? r + 1 : r][e];
        } else lds_get(&tl[t + 1][x0 + 3], yp);          // another wave's row, or the ghost row y = H+1
        relax4(s1[r], left, right, ym, yp, s0[r], s2[r], rh, u);
    };

    // final level: the stores of the sweep plus those of its setBounds
    auto store_final = [&](int zo, int r, const T (&u)[4], unsigned fl) {
        const int y = y0 + r;
        T st[4];
        settle4(u, fl, st);
        char* base = reinterpret_cast<char*>(dst) + (long)zo * plane_b + oc[r];
        // z-slab push exchange: the planes the neighbours need next also go straight into their halo planes (wave-uniform)
        const long dl = (SLAB && zo <= pp.planes) ? pp.lo : 0, dh = (SLAB && zo > D - pp.planes) ? pp.hi : 0;
        if (FS_EXP_NT & 2) {
            Vec4<T> q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = st[e];
            __builtin_nontemporal_store(q, reinterpret_cast<Vec4<T>*>(base));
        } else {
            V4<T> q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q.e[e] = st[e];
            put<SLAB>(reinterpret_cast<V4<T>*>(base), q, dl, dh);
        }
        if (x0 == 1) put<SLAB>(reinterpret_cast<T*>(base - ES), (b == 1) ? -u[0] : u[0], dl, dh);   // :189-190
        if (full_group && x0 + 3 == W) put<SLAB>(reinterpret_cast<T*>(base + 4 * ES), u[3], dl, dh);   // :191
        if (WALLS && (y == 1 || y == H)) {
            T f[4];
            V4<T> qq;
            face4(u, b == 2, f);
#pragma unroll
            for (int e = 0; e < 4; ++e) qq.e[e] = f[e];
            if (y == 1) put<SLAB>(reinterpret_cast<V4<T>*>(base - row_b), qq, dl, dh);               // :198-201
            if (y == H) put<SLAB>(reinterpret_cast<V4<T>*>(base + row_b), qq, dl, dh);
        }
        const bool zlo_face = (zo == 1) && lo_wall_c, zhi_face = (zo == D) && hi_wall_c;
        if (zlo_face || zhi_face) {
            T f[4];
            V4<T> qq;
            face4(u, b == 3, f);
#pragma unroll
            for (int e = 0; e < 4; ++e) qq.e[e] = f[e];
            if (zlo_face) *reinterpret_cast<V4<T>*>(base - plane_b) = qq;                           // :208-214
            if (zhi_face) *reinterpret_cast<V4<T>*>(base + plane_b) = qq;
        }
    };

    // One step of the march.  PH: which register slot is which plane (rotates with period 3).
    auto iter = [&](auto phc, int zl) {
        constexpr int PH = decltype(phc)::value;
        constexpr int I0 = PH, I1 = (PH + 1) % 3, I2 = (PH + 2) % 3;
        // level-0 planes zl-1, zl, zl+1 sit in L0[I0], L0[I1], L0[I2]; level j planes
        // p-1, p (and the one computed now, p+1) in Lj[I0], Lj[I1], Lj[I2]
        // The row offsets are made opaque once per iteration: otherwise loop strength reduction folds
        // each (plane pointer + row offset) into its own 64-bit vector induction variable -- two VGPRs
        // per access kept across the loop -- instead of a scalar base plus this 32-bit offset.
#pragma unroll
        for (int r = 0; r < RY; ++r) asm volatile("" : "+v"(oc[r]));
        // the rhs ring: NL = 3 rotates with the register slots (compile-time slot), NL = 2 by plane parity
        T(*rs_put)[RW] = rsave[(NL == 3) ? PH : (zl & 1)];
        T(*rs_get2)[RW] = rsave[(NL == 3) ? I2 : ((zl - 1) & 1)];
        const bool wall_lo1 = (zl == 1) && lo_wall_c, wall_hi1 = (zl == D) && hi_wall_c;
        if (zl <= hi1) {                                 // ---- level 1 of plane zl
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r, t = ty0 + r;
                const bool row_on = (y >= 1) && (y <= H);     // wave-uniform
                if (row_on && lane_on) {
                    T ym[4], yp[4], u[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ym[e] = (r > 0) ? L0[I1][r > 0 ? r - 1 : 0][e] : hb[e];
                        yp[e] = (r < RY - 1) ? L0[I1][r < RY - 1 ? r + 1 : r][e] : ht[e];
                    }
                    relax4(L0[I1][r], eL[r], eR[r], ym, yp, L0[I0][r], L0[I2][r], rcur[r], u);
                    settle4(u, flc[r], L1[I2][r]);
                    publish(IC<0>{}, zl, r, u, L1[I2][r]);
                    if (RS && t >= 1 && t <= BY - 2) lds_set(&rs_put[t - 1][xl], rcur[r]);
                    if (wall_lo1) face4(u, b == 3, L1[I1][r]);      // ghost plane z = 0 takes plane 0's slot, :208-210
                    if (wall_hi1) face4(u, b == 3, gz1[r]);         // ghost plane z = D+1 waits for its slot, :212-214
                }
                kl[PH][r] = flc[r];
            }
        }
        // next plane's level-0 data, one iteration ahead.  Unconditional (a conditional load merges
        // with the old register contents, and the merge waits for the load); past the last plane the
        // same plane is fetched again and never used.
        // lean variant: the right-hand sides of the two older planes, issued BEFORE the prefetch (memory returns in order:
        // a wait for these must not wait for the next plane's loads as well)
        T rh2[RS ? 1 : RY][4], rh3[RS ? 1 : RY][4];
        if constexpr (!RS) {
            const char* rp2 = plane_of(rhs, max(zl - 1, lo1));
            const char* rp3 = plane_of(rhs, max(zl - 2, lo1));
#pragma unroll
            for (int r = 0; r < RY; ++r) { ld4(rp2 + oc[r], rh2[r]); ld4(rp3 + oc[r], rh3[r]); }
        }
        __builtin_amdgcn_sched_barrier(0);               // pin the loads here: left alone, the scheduler sinks them to the
        load_core(min(zl + 2, zmax0), L0[I0]);           // end of the iteration to shorten live ranges, and the next
        load_side(min(zl + 1, zmax1));                   // iteration then starts by waiting a full memory latency
        asm volatile("" ::: "memory");                   // (the IR-level sink pass does the same across blocks)
        __builtin_amdgcn_sched_barrier(0);
        const int P2 = zl - 1;
        const bool wall_lo2 = (P2 == 1) && lo_wall_c, wall_hi2 = (P2 == D) && hi_wall_c;
        if constexpr (NL == 3) {
            if (P2 >= lo2 && P2 <= hi2) {                // ---- level 2 of plane zl-1 (intermediate)
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const int y = y0 + r, t = ty0 + r;
                    if (y >= r2lo && y <= r2hi) {        // wave-uniform
                        T rh[4], u[4];
                        if constexpr (RS) lds_get(&rs_get2[t - 1][xl], rh);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) rh[e] = rh2[r][e];
                        }
                        next_row(IC<0>{}, P2, r, L1[I0], L1[I1], L1[I2], rh, u);
                        if (lane_on) {
                            settle4(u, kl[I2][r], L2[I2][r]);
                            publish(IC<NL - 2>{}, P2, r, u, L2[I2][r]);
                            if (wall_lo2) face4(u, b == 3, L2[I1][r]);
                            if (wall_hi2) face4(u, b == 3, gz2[r]);
                        }
                    }
                }
            }
        } else {
            if (P2 >= zbeg && P2 <= zend) {              // ---- level 2 of plane zl-1 (final)
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const int y = y0 + r, t = ty0 + r;
                    if (y >= r2lo && y <= r2hi) {
                        T rh[4], u[4];
                        lds_get(&rs_get2[t - 1][xl], rh);
                        next_row(IC<0>{}, P2, r, L1[I0], L1[I1], L1[I2], rh, u);
                        if (lane_on) store_final(P2, r, u, kl[I2][r]);
                    }
                }
            }
        }
        if (wall_hi1) {                                  // level 1's oldest slot is free now: it becomes plane D+1
#pragma unroll
            for (int r = 0; r < RY; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) L1[I0][r][e] = gz1[r][e];
        }
        if constexpr (NL == 3) {
            const int P3 = zl - 2;
            if (P3 >= zbeg && P3 <= zend) {              // ---- level 3 of plane zl-2
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const int y = y0 + r, t = ty0 + r;
                    if (y >= r3lo && y <= r3hi) {
                        T rh[4], u[4];
                        if constexpr (RS) lds_get(&rsave[I1][t - 1][xl], rh);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) rh[e] = rh3[r][e];
                        }
                        next_row(IC<NL - 2>{}, P3, r, L2[I0], L2[I1], L2[I2], rh, u);
                        if (lane_on) store_final(P3, r, u, kl[I1][r]);
                    }
                }
            }
            if (wall_hi2) {
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) L2[I0][r][e] = gz2[r][e];
            }
        }
        __syncthreads();
    };

    if (z_from == lo1) {                                 // the march starts here: first loads
        load_core(lo1 - 1, L0[0]);
        load_core(lo1, L0[1]);
        load_core(min(lo1 + 1, zmax0), L0[2]);
        load_side(lo1);
    }
    int zl = z_from;
    for (int k = 0; k < ngroups; ++k) {
        iter(IC<0>{}, zl);
        iter(IC<1>{}, zl + 1);
        iter(IC<2>{}, zl + 2);
        zl += 3;
    }
    if (zl <= tail_to) {
        iter(IC<0>{}, zl);
        if (zl + 1 <= tail_to) iter(IC<1>{}, zl + 1);
    }
    };
    const int zl_end = zend + OV;                        // the march: iterations lo1 .. zl_end
    const int total = zl_end - lo1 + 1, ngroups = total / 3;
    if constexpr (WALLSEL == 1) {
        // Two bodies.  A band at a y wall runs the general one throughout (and is the lighter band, see `s`).  Any other band
        // needs it only where a level touches a physical z wall: the iterations up to NL (plane 1 as level 1 .. NL) and those
        // from D on; the groups of three iterations that contain none of those run the wall-free body.
        const bool ywall = (s <= 0) || (s + BY - 1 >= H + 1);
        // groups [0, g0) general, [g0, g1) wall-free, [g1, ngroups) + tail general
        int g0 = ngroups, g1 = ngroups;
        if (!ywall) {
            g0 = (lo_wall && lo1 <= NL) ? min(ngroups, (NL - lo1) / 3 + 1) : 0;
            g1 = hi_wall ? max(g0, min(ngroups, (D - lo1) / 3)) : ngroups;     // group k covers lo1 + 3k .. lo1 + 3k + 2 < D
        }
        run(IC<1>{}, lo1, g0, -1);
        run(IC<0>{}, lo1 + 3 * g0, g1 - g0, -1);
        run(IC<1>{}, lo1 + 3 * g1, ngroups - g1, zl_end);
    } else {
        run(IC<WALLSEL == 2 ? 0 : 1>{}, lo1, ngroups, zl_end);
    }
}

// ---------------------------------------------------------------------------------------------
// launch plans
// ---------------------------------------------------------------------------------------------
template <int NL>
static int fused_bands(int H, int BY)
{
    // band k outputs rows k (BY - 2 (NL-1)) + 1 .. (k + 1)(BY - 2 (NL-1)), the last band up to row H
    const int step = BY - 2 * (NL - 1);
    return (H + step - 1) / step;
}

template <class T, int NL, int NXW, int NYW, int RY>
static void launch_fused_v(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, const T* src,
                           const T* rhs, T* dst, const uint8_t* flags, int b, T a, T inv_c, int z_first, int z_last, int alt,
                           int second_first, const PeerPush* push)
{
    const PeerPush pp = (push && second_first < 0) ? *push : PeerPush();
    constexpr int BY = NYW * RY, THREADS = NXW * NYW * 64;
    const int planes = z_last - z_first + 1;
    if (planes <= 0) return;
    const int nbands = fused_bands<NL>(g.H, BY);
    const bool aligned = (g.W == NXW * 256);
    const bool whole = sc.lo_wall && sc.hi_wall && z_first == 1 && z_last == g.D && second_first < 0;
    int zc_len, z_stride, nblk;
    if (second_first >= 0) {
        // two equally long ranges (the slab's two boundary regions) as two chunks of one launch
        zc_len = planes;
        z_stride = second_first - z_first;
        z_last = second_first + planes - 1;
        nblk = nbands * 2;
    } else {
        // z chunks re-read 2 NL level-0 planes and recompute the levels below the last on 2(NL-1) + ... planes:
        // keep them long; pick the count that fills the CUs most evenly.  `alt` picks the alt-th best chunk
        // count by this model (the host driver times alt = 0, 1, 2 once per grid).
        int cand_nzc[3] = {1, 1, 1};
        double cand_eff[3] = {-1.0, -1.0, -1.0};
        const int slots = tune.cu_slots > 0 ? tune.cu_slots : 256;
        for (int nzc = 1; nzc <= 64 && (nzc == 1 || planes / nzc >= 16); ++nzc) {
            const long blocks = (long)nbands * nzc;
            const long rounds = (blocks + slots - 1) / slots;
            const int len = (planes + nzc - 1) / nzc;
            const double eff = (double)blocks / (double)(rounds * slots) * (double)len / (double)(len + 2 * NL - 1);
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
        zc_len = (planes + cand_nzc[pick] - 1) / cand_nzc[pick];
        if (tune.pair_zc > 0) zc_len = tune.pair_zc < planes ? tune.pair_zc : planes;
        const int nzc = (planes + zc_len - 1) / zc_len;
        z_stride = zc_len;
        nblk = nbands * nzc;
    }
    // whole-domain, lane-aligned rows (the benchmark grids) get the build without any slab logic; everything
    // else the general one
#define FS_LAUNCH(AL, SL, WS)                                                                                            \
    hipLaunchKernelGGL((jacobi_fused_kernel<T, NL, NXW, NYW, RY, AL, SL, WS>), dim3(nblk), dim3(THREADS), 0, st, g, sc, src, \
                       rhs, dst, flags, b, a, inv_c, z_first, z_last, zc_len, z_stride, nbands, nblk, pp)
    // The wall-free second body exists for the three-sweep kernel on lane-aligned whole-domain grids (the benchmark grids).
    // Rounds 1-2 selected a body per WORKGROUP: 9 % (512^3) to 14 % (256^3) faster for the interior ones, but a pass ends with its
    // slowest workgroup, and at 512^3 (256 workgroups, one per CU) half of them touch a z wall.  Round 3 selects per GROUP OF
    // THREE PLANE ITERATIONS instead (only the first and last groups of a z-wall chunk run the general body) and shifts the band
    // origin so that the two y-wall bands, which run it throughout, carry one row less per level: 0.1448 -> 0.1371 ms per sweep
    // at 512^3 on one box, 0.1380 -> 0.1282 on another, 0.0172 -> 0.0166 at 256^3 (profiles/r3y_*); letting the y-wall bands
    // run the wall-free body too (timing only) changes nothing any more, so "auto" now means always.
    const bool two_bodies = tune.wall_free >= 1;
    if constexpr (NL == 3) {                             // (the two-body builds exist for three sweeps only)
        if (aligned && two_bodies) {
            if (whole) FS_LAUNCH(true, false, 1);
            else FS_LAUNCH(true, true, 1);               // z-slabs: an inner rank has no z wall at all
            return;
        }
    }
    if (aligned && whole) FS_LAUNCH(true, false, 0);
    else if (aligned) FS_LAUNCH(true, true, 0);
    else FS_LAUNCH(false, true, 0);
#undef FS_LAUNCH
}

// Which (T, NL) this file has a kernel for on this grid.  On a z-slab the halo must be NL planes deep.
template <>
bool fused_supported<float>(const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int levels)
{
    const bool whole = sc.lo_wall && sc.hi_wall;
    if (!whole && g.zh < levels) return false;
    if (levels == 3) return g.W <= 512 && tune.fuse >= 3;
    if (levels == 2) return g.W > 512 && g.W <= 1024 && tune.fuse >= 2;
    return false;
}
template <>
bool fused_supported<double>(const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int levels)
{
    const bool whole = sc.lo_wall && sc.hi_wall;
    if (!whole && g.zh < levels) return false;
    return levels == 2 && g.W <= 512 && tune.fuse >= 2;
}

template <>
int fused_shape_count<float>(const GridDesc& g, int levels)
{
    if (levels == 3) return (g.W <= 256) ? 3 : 2;
    return (g.W > 768) ? 2 : 1;
}
template <>
int fused_shape_count<double>(const GridDesc& g, int) { return (g.W <= 256) ? 1 : 2; }

// plan = workgroup shape + 8 * (which of the launcher's three best z-chunk counts); all plans give the same bits,
// the host driver times them once per grid.
template <>
void launch_jacobi_fused<float>(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int levels,
                                const float* src, const float* rhs, float* dst, const uint8_t* flags, int b, float a,
                                float inv_c, int z_first, int z_last, int plan, int second_first, const PeerPush* push)
{
    if (plan < 0) plan = 0;
    const int alt = plan >> 3, shape = plan & 7;
#define FS_F(NL, NX, NY, RY) launch_fused_v<float, NL, NX, NY, RY>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, alt, second_first, push)
    if (levels == 3 && tune.abl == 16 && g.W == 512 && sc.lo_wall && sc.hi_wall && z_first == 1 && z_last == g.D && second_first < 0) {
        // EXPERIMENT, timing only (wrong at the walls): every workgroup runs the lean interior body in 16-row bands
        constexpr int BY = 16;
        const int nbands = fused_bands<3>(g.H, BY);
        const int nzc = alt == 0 ? 5 : alt == 1 ? 6 : 4;
        const int zc_len = (g.D + nzc - 1) / nzc, nblk = nbands * nzc;
        hipLaunchKernelGGL((jacobi_fused_kernel<float, 3, 2, 8, 2, true, false, 2>), dim3(nblk), dim3(1024), 0, st, g, sc, src, rhs, dst, flags,
                           b, a, inv_c, z_first, z_last, zc_len, zc_len, nbands, nblk, PeerPush());
        return;
    }
    if (levels == 3) {
        // Two rows per wave throughout (three rows and 8 waves were slower: the instruction stream of a wave
        // is what limits this kernel).  Rows up to 256 cells: bands of 20, 16 or 12 rows (the smaller ones trade
        // recomputed rows for longer z chunks and, at 12 rows, two workgroups per CU); up to 512 cells: 12 or 10 rows.
        if (g.W <= 256) {
            if (shape == 1) FS_F(3, 1, 8, 2);
            else if (shape == 2) FS_F(3, 1, 6, 2);
            else FS_F(3, 1, 10, 2);
        } else {
            if (shape == 1) FS_F(3, 2, 5, 2);
            else FS_F(3, 2, 6, 2);
        }
    } else {
        // rows of 513..1024 cells: 16 waves x two rows (8-row bands, <= 128 VGPRs) or 12 waves x three rows (9-row bands)
        if (g.W <= 768) FS_F(2, 3, 4, 2);
        else if (shape == 1) FS_F(2, 4, 3, 3);
        else FS_F(2, 4, 4, 2);
    }
#undef FS_F
}
template <>
void launch_jacobi_fused<double>(hipStream_t st, const SweepTune& tune, const GridDesc& g, const SlabCtx& sc, int,
                                 const double* src, const double* rhs, double* dst, const uint8_t* flags, int b, double a,
                                 double inv_c, int z_first, int z_last, int plan, int second_first, const PeerPush* push)
{
    if (plan < 0) plan = 0;
    const int alt = plan >> 3, shape = plan & 7;
#define FS_F(NL, NX, NY, RY) launch_fused_v<double, NL, NX, NY, RY>(st, tune, g, sc, src, rhs, dst, flags, b, a, inv_c, z_first, z_last, alt, second_first, push)
    // fp64: two sweeps per pass; rows up to 256 cells: 20-row bands; up to 512: 10-row bands (10 waves) or 8 (8 waves, 256 VGPRs)
    if (g.W <= 256) FS_F(2, 1, 10, 2);
    else if (shape == 1) FS_F(2, 2, 4, 2);
    else FS_F(2, 2, 5, 2);
#undef FS_F
}

}  // namespace fs

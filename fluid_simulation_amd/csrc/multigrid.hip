// multigrid.hip -- coarse levels of the optional multigrid pressure solve (see multigrid.h).
//
// Every expression keeps the association order written in oracle/cpu_ref_mg.h; the build is
// -ffp-contract=off, so the two agree bit for bit (tests/test_gpu_multigrid.py).  The kernels are
// plain one-thread-per-cell passes: all coarse levels together hold 1/7 of the cells of level 0,
// whose smoothing (jacobi_pair_kernel<.., RB>) and whose two transfer passes dominate a cycle.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "kernels_dev.h"
#include "multigrid.h"

namespace fs {

namespace {

constexpr int MG_MIN_DIM = 4;

template <class T>
__device__ __forceinline__ long lat(const MgLevel<T>& l, int x, int y, int z)
{
    return (long)x + (long)y * l.sy + (long)z * l.sz;
}

// ---- level-0 coefficients from the flag bytes (never stored) --------------------------------
// z-slab runs: "in range" follows the GLOBAL depth -- plane z of a slab is global plane sc.zoff + z, and the halo planes of
// the flag bytes hold the neighbouring slabs' cells (the flag build covers them, kernels.hip)
__device__ __forceinline__ bool in_range(const GridDesc& g, const SlabCtx& sc, int x, int y, int z)
{
    const int zg = z + sc.zoff;
    return x >= 1 && x <= g.W && y >= 1 && y <= g.H && zg >= 1 && zg <= sc.Dglobal;
}
__device__ __forceinline__ bool fluid0(const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, int x, int y, int z)
{
    return in_range(g, sc, x, y, z) && !(flags[cell(g, x, y, z)] & F_SOLID);
}
__device__ __forceinline__ int solid0(const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, int x, int y, int z)
{
    return (in_range(g, sc, x, y, z) && (flags[cell(g, x, y, z)] & F_SOLID)) ? 1 : 0;
}
template <class T>
__device__ __forceinline__ T w0(const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, int x, int y, int z, int axis)
{
    const int xm = x - (axis == 0), ym = y - (axis == 1), zm = z - (axis == 2);
    return (fluid0(g, sc, flags, x, y, z) && fluid0(g, sc, flags, xm, ym, zm)) ? (T)1 : (T)0;
}
template <class T>
__device__ __forceinline__ T d0(const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, int x, int y, int z)
{
    if (!fluid0(g, sc, flags, x, y, z)) return (T)0;
    return (T)(solid0(g, sc, flags, x + 1, y, z) + solid0(g, sc, flags, x - 1, y, z) + solid0(g, sc, flags, x, y + 1, z) +
               solid0(g, sc, flags, x, y - 1, z) + solid0(g, sc, flags, x, y, z + 1) + solid0(g, sc, flags, x, y, z - 1));
}

// Coefficients of level c from the level below: stored level f, or (FROM0) level 0's flag bytes.
template <class T, bool FROM0>
__global__ __launch_bounds__(256) void mg_coarsen_kernel(GridDesc g, SlabCtx sc, const uint8_t* __restrict__ flags, MgLevel<T> f, MgLevel<T> c)
{
    const int X = 1 + blockIdx.x * 64 + threadIdx.x, Y = 1 + blockIdx.y * 4 + threadIdx.y, Z = 1 + blockIdx.z;
    if (X > c.W + 1 || Y > c.H + 1 || Z > c.D + 1) return;
    const long C = lat(c, X, Y, Z);
    const int x = 2 * X - 1, y = 2 * Y - 1, z = 2 * Z - 1;      // first child
    const T q = (T)0.25, hf = (T)0.5;
    auto FW = [&](int axis, int xx, int yy, int zz) -> T {
        if constexpr (FROM0) return w0<T>(g, sc, flags, xx, yy, zz, axis);
        else return (axis == 0 ? f.wx : axis == 1 ? f.wy : f.wz)[lat(f, xx, yy, zz)];
    };
    auto FD = [&](int xx, int yy, int zz) -> T {
        if constexpr (FROM0) return d0<T>(g, sc, flags, xx, yy, zz);
        else return f.d[lat(f, xx, yy, zz)];
    };
    if (Y <= c.H && Z <= c.D) c.wx[C] = q * (((FW(0, x, y, z) + FW(0, x, y + 1, z)) + FW(0, x, y, z + 1)) + FW(0, x, y + 1, z + 1));
    if (X <= c.W && Z <= c.D) c.wy[C] = q * (((FW(1, x, y, z) + FW(1, x + 1, y, z)) + FW(1, x, y, z + 1)) + FW(1, x + 1, y, z + 1));
    if (X <= c.W && Y <= c.H) c.wz[C] = q * (((FW(2, x, y, z) + FW(2, x + 1, y, z)) + FW(2, x, y + 1, z)) + FW(2, x + 1, y + 1, z));
    if (X <= c.W && Y <= c.H && Z <= c.D)
        c.d[C] = hf * (((((((FD(x, y, z) + FD(x + 1, y, z)) + FD(x, y + 1, z)) + FD(x + 1, y + 1, z)) + FD(x, y, z + 1)) +
                         FD(x + 1, y, z + 1)) + FD(x, y + 1, z + 1)) + FD(x + 1, y + 1, z + 1));
}

template <class T>
__global__ __launch_bounds__(256) void mg_diag_kernel(MgLevel<T> c)
{
    const int X = 1 + blockIdx.x * 64 + threadIdx.x, Y = 1 + blockIdx.y * 4 + threadIdx.y, Z = 1 + blockIdx.z;
    if (X > c.W || Y > c.H) return;
    const long C = lat(c, X, Y, Z);
    const T one = (T)1;
    const T xm = c.wx[C], xp = c.wx[C + 1], ym = c.wy[C], yp = c.wy[C + c.sy], zm = c.wz[C], zp = c.wz[C + c.sz], d = c.d[C];
    c.dg[C] = (((((xm + xp) + ym) + yp) + zm) + zp) + d;
    // regular: the general expressions reduce to the plain 7-point ones bit for bit (1 * e = e, diagonal exactly 6)
    c.reg[C] = (xm == one && xp == one && ym == one && yp == one && zm == one && zp == one && d == (T)0) ? 1 : 0;
}

template <class T>
__device__ __forceinline__ T plain_neighbours(const MgLevel<T>& l, long c)
{
    const T* e = l.e;
    return ((((e[c - 1] + e[c + 1]) + e[c - l.sy]) + e[c + l.sy]) + e[c - l.sz]) + e[c + l.sz];
}

template <class T>
__device__ __forceinline__ T weighted_neighbours(const MgLevel<T>& l, long c)
{
    const T* e = l.e;
    return ((((l.wx[c] * e[c - 1] + l.wx[c + 1] * e[c + 1]) + l.wy[c] * e[c - l.sy]) + l.wy[c + l.sy] * e[c + l.sy]) +
            l.wz[c] * e[c - l.sz]) + l.wz[c + l.sz] * e[c + l.sz];
}

// one colour of a red-black Gauss-Seidel iteration: the cells with (x + y + z) & 1 == colour
template <class T>
__global__ __launch_bounds__(256) void mg_smooth_kernel(MgLevel<T> l, int colour)
{
    const int y = 1 + blockIdx.y * 4 + threadIdx.y, z = 1 + blockIdx.z;
    const int x = 1 + 2 * (blockIdx.x * 64 + threadIdx.x) + (((y + z + l.zoff + colour) & 1) ? 0 : 1);   // colour by global z
    if (x > l.W || y > l.H) return;
    const long c = lat(l, x, y, z);
    if (l.reg[c]) {
        l.e[c] = (l.b[c] + plain_neighbours(l, c)) / (T)6;
        return;
    }
    const T dg = l.dg[c];
    if (!(dg > (T)0)) return;
    l.e[c] = (l.b[c] + weighted_neighbours(l, c)) / dg;
}

template <class T>
__device__ __forceinline__ T residual_at(const MgLevel<T>& l, long c)
{
    if (l.reg[c]) return (l.b[c] + plain_neighbours(l, c)) - (T)6 * l.e[c];
    const T dg = l.dg[c];
    if (!(dg > (T)0)) return (T)0;
    return (l.b[c] + weighted_neighbours(l, c)) - dg * l.e[c];
}

// right-hand side of level c = 1/2 * sum of the eight children's residuals on level f; e_c = 0
template <class T>
__global__ __launch_bounds__(256) void mg_restrict_kernel(MgLevel<T> f, MgLevel<T> c)
{
    const int X = 1 + blockIdx.x * 64 + threadIdx.x, Y = 1 + blockIdx.y * 4 + threadIdx.y, Z = 1 + blockIdx.z;
    if (X > c.W || Y > c.H) return;
    const long C = lat(c, X, Y, Z);
    const int x = 2 * X - 1, y = 2 * Y - 1, z = 2 * Z - 1;
    T r = residual_at(f, lat(f, x, y, z));
    r = r + residual_at(f, lat(f, x + 1, y, z));
    r = r + residual_at(f, lat(f, x, y + 1, z));
    r = r + residual_at(f, lat(f, x + 1, y + 1, z));
    r = r + residual_at(f, lat(f, x, y, z + 1));
    r = r + residual_at(f, lat(f, x + 1, y, z + 1));
    r = r + residual_at(f, lat(f, x, y + 1, z + 1));
    r = r + residual_at(f, lat(f, x + 1, y + 1, z + 1));
    c.b[C] = (c.dg[C] > (T)0) ? (T)0.5 * r : (T)0;
    c.e[C] = (T)0;
}

// level 0: residual of the reference's fixed point, sum order of simulation.cpp:264-268
template <class T>
__device__ __forceinline__ T residual0_at(const GridDesc& g, const uint8_t* flags, const T* p, const T* rhs, int x, int y, int z)
{
    const long c = cell(g, x, y, z);
    if (flags[c] & F_SOLID) return (T)0;
    const T nb = p[c + 1] + p[c - 1] + p[c + g.sy] + p[c - g.sy] + p[c + g.sz] + p[c - g.sz];
    return (rhs[c] + nb) - (T)6 * p[c];
}

// A lane owns four x-consecutive fine cells of the two rows and two planes below coarse row (Y, Z): 16 residuals from
// twelve 16-byte row loads of p (plus the columns beside them), four of rhs and four flag words, all in flight at once;
// they make two coarse cells, summed in the oracle's order (x fastest, then y, then z).
template <class T>
__global__ __launch_bounds__(256) void mg_restrict0_kernel(GridDesc g, const uint8_t* __restrict__ flags, const T* __restrict__ p,
                                                            const T* __restrict__ rhs, MgLevel<T> c)
{
    const int x0 = 1 + 4 * (blockIdx.x * 64 + threadIdx.x);
    const int Y = 1 + blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y), Z = 1 + blockIdx.z;
    if (x0 > g.W || Y > c.H) return;
    const int y0 = 2 * Y - 1, z0 = 2 * Z - 1;
    const long i00 = cell(g, x0, y0, z0);
    auto ld = [&](long i) { return *reinterpret_cast<const V4<T>*>(p + i); };
    // centre rows [dz][dy], the rows beside them in y ([dz][0] = y0-1, [dz][1] = y0+2) and in z ([0][dy] = z0-1, [1][dy] = z0+2)
    V4<T> cen[2][2], ynb[2][2], znb[2][2], rh[2][2];
    T xl[2][2], xr[2][2];
    unsigned fw[2][2];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const long i = i00 + dy * g.sy + dz * g.sz;
            cen[dz][dy] = ld(i);
            rh[dz][dy] = *reinterpret_cast<const V4<T>*>(rhs + i);
            fw[dz][dy] = *reinterpret_cast<const unsigned*>(flags + i);
            xl[dz][dy] = p[i - 1];
            xr[dz][dy] = p[i + 4];
            znb[dz][dy] = ld(i00 + dy * g.sy + (dz ? 2 : -1) * g.sz);
        }
        ynb[dz][0] = ld(i00 - g.sy + dz * g.sz);
        ynb[dz][1] = ld(i00 + 2 * g.sy + dz * g.sz);
    }
    T sum[2] = {(T)0, (T)0};                              // coarse cells X, X+1
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const V4<T>& v = cen[dz][dy];
                const T xp = (e < 3) ? v.e[e < 3 ? e + 1 : 3] : xr[dz][dy];
                const T xm = (e > 0) ? v.e[e > 0 ? e - 1 : 0] : xl[dz][dy];
                const T yp = dy ? ynb[dz][1].e[e] : cen[dz][1].e[e];
                const T ym = dy ? cen[dz][0].e[e] : ynb[dz][0].e[e];
                const T zp = dz ? znb[1][dy].e[e] : cen[1][dy].e[e];
                const T zm = dz ? cen[0][dy].e[e] : znb[0][dy].e[e];
                const T nb = xp + xm + yp + ym + zp + zm;     // order of simulation.cpp:264-268
                const bool solid = ((fw[dz][dy] >> (8 * e)) & F_SOLID) != 0;
                const T r = (solid || x0 + e > g.W) ? (T)0 : (rh[dz][dy].e[e] + nb) - (T)6 * v.e[e];
                // the oracle's running sum visits the children in exactly this loop order (dz outer, dy, then x)
                if (dz == 0 && dy == 0 && (e & 1) == 0) sum[e >> 1] = r;
                else sum[e >> 1] = sum[e >> 1] + r;
            }
    const int X = (x0 + 1) >> 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (X + k > c.W) continue;
        const long C = lat(c, X + k, Y, Z);
        c.b[C] = (c.dg[C] > (T)0) ? (T)0.5 * sum[k] : (T)0;
        c.e[C] = (T)0;
    }
}

// trilinear interpolation of level c's correction at fine cell (x, y, z); neighbour index clamped at the walls
template <class T>
__device__ __forceinline__ T interp(const MgLevel<T>& c, int x, int y, int z)
{
    const int X = (x + 1) >> 1, Y = (y + 1) >> 1, Z = (z + 1) >> 1;
    const int Xn = min(max((x & 1) ? X - 1 : X + 1, 1), c.W);
    const int Yn = min(max((y & 1) ? Y - 1 : Y + 1, 1), c.H);
    // (z-slabs: the fine level's plane offset is even, so local parity is global parity; beyond a slab side without a
    // physical wall lies the neighbour's plane, held in the halo plane of e -- no clamp there)
    const int Zn = min(max((z & 1) ? Z - 1 : Z + 1, c.lo_wall ? 1 : 0), c.hi_wall ? c.D : c.D + 1);
    const T a = (T)0.75, q = (T)0.25;
    const T* e = c.e;
    const T x00 = a * e[lat(c, X, Y, Z)] + q * e[lat(c, Xn, Y, Z)];
    const T x10 = a * e[lat(c, X, Yn, Z)] + q * e[lat(c, Xn, Yn, Z)];
    const T x01 = a * e[lat(c, X, Y, Zn)] + q * e[lat(c, Xn, Y, Zn)];
    const T x11 = a * e[lat(c, X, Yn, Zn)] + q * e[lat(c, Xn, Yn, Zn)];
    const T y0 = a * x00 + q * x10;
    const T y1 = a * x01 + q * x11;
    return a * y0 + q * y1;
}

template <class T>
__global__ __launch_bounds__(256) void mg_prolong_kernel(MgLevel<T> c, MgLevel<T> f)
{
    const int x = 1 + blockIdx.x * 64 + threadIdx.x, y = 1 + blockIdx.y * 4 + threadIdx.y, z = 1 + blockIdx.z;
    if (x > f.W || y > f.H) return;
    const long i = lat(f, x, y, z);
    if (f.dg[i] > (T)0) f.e[i] = f.e[i] + interp(c, x, y, z);
}

// level 0: p += correction at the cells that are not solid, then setBounds(0, p) (solids already hold 0).
// A lane owns four x-consecutive cells of the two rows and two planes below coarse row (Y, Z) (16-byte accesses, four
// rows in flight); the 16 cells share the coarse columns X-1 .. X+2 of the 3 x 3 coarse rows around (Y, Z).
template <class T>
__global__ __launch_bounds__(256) void mg_prolong0_kernel(MgLevel<T> c, GridDesc g, SlabCtx sc, const uint8_t* __restrict__ flags,
                                                           T* __restrict__ p)
{
    const int x0 = 1 + 4 * (blockIdx.x * 64 + threadIdx.x);
    const int Y = 1 + blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y), Z = 1 + blockIdx.z;
    if (x0 > g.W || Y > c.H) return;
    const int X = (x0 + 1) >> 1;
    // coarse columns of the four cells: own (X, X, X+1, X+1), neighbour (X-1, X+1, X, X+2), clamped at the walls
    const int col[4] = { max(X - 1, 1), X, min(X + 1, c.W), min(X + 2, c.W) };
    const int Yc[3] = { max(Y - 1, 1), Y, min(Y + 1, c.H) };
    const int Zc[3] = { max(Z - 1, c.lo_wall ? 1 : 0), Z, min(Z + 1, c.hi_wall ? c.D : c.D + 1) };   // halo plane beyond a slab side
    T R[3][3][4];                                         // [z][y][column]
#pragma unroll
    for (int zi = 0; zi < 3; ++zi)
#pragma unroll
        for (int yi = 0; yi < 3; ++yi) {
            const long row = lat(c, 0, Yc[yi], Zc[zi]);
#pragma unroll
            for (int k = 0; k < 4; ++k) R[zi][yi][k] = c.e[row + col[k]];
        }
    const T a = (T)0.75, q = (T)0.25;
    const long i00 = cell(g, x0, 2 * Y - 1, 2 * Z - 1);
    V4<T> v[2][2];
    unsigned fw[2][2];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            v[dz][dy] = *reinterpret_cast<const V4<T>*>(p + i00 + dy * g.sy + dz * g.sz);
            fw[dz][dy] = *reinterpret_cast<const unsigned*>(flags + i00 + dy * g.sy + dz * g.sz);
        }
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = 2 * Y - 1 + dy, z = 2 * Z - 1 + dz;
            const long i = i00 + dy * g.sy + dz * g.sz;
            const int yn = dy ? 2 : 0, zn = dz ? 2 : 0;      // the neighbour row of an odd y (dy = 0) is the one below, ...
            T u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int own = 1 + (e >> 1), nbr = (e & 1) ? own + 1 : own - 1;
                const T x00 = a * R[1][1][own] + q * R[1][1][nbr];
                const T x10 = a * R[1][yn][own] + q * R[1][yn][nbr];
                const T x01 = a * R[zn][1][own] + q * R[zn][1][nbr];
                const T x11 = a * R[zn][yn][own] + q * R[zn][yn][nbr];
                const T y0 = a * x00 + q * x10;
                const T y1 = a * x01 + q * x11;
                const T corr = a * y0 + q * y1;
                const bool solid = ((fw[dz][dy] >> (8 * e)) & F_SOLID) != 0;
                u[e] = (x0 + e <= g.W) ? (solid ? v[dz][dy].e[e] : v[dz][dy].e[e] + corr) : (T)0;
            }
            // setBounds(0, p): ghost faces mirror the (un-zeroed) interior values; ghost edges and row padding stay 0
            V4<T> o, face;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = x0 + e;
                o.e[e] = (x <= g.W) ? u[e] : ((x == g.W + 1) ? u[e > 0 ? e - 1 : 0] : (T)0);
                face.e[e] = (x <= g.W) ? u[e] : (T)0;
            }
            *reinterpret_cast<V4<T>*>(p + i) = o;
            if (x0 == 1) p[i - 1] = u[0];
            if (x0 + 3 == g.W) p[i + 4] = u[3];           // the ghost x = W+1 when W is a multiple of 4
            if (y == 1) *reinterpret_cast<V4<T>*>(p + i - g.sy) = face;
            if (y == g.H) *reinterpret_cast<V4<T>*>(p + i + g.sy) = face;
            if (z == 1 && sc.lo_wall) *reinterpret_cast<V4<T>*>(p + i - g.sz) = face;
            if (z == g.D && sc.hi_wall) *reinterpret_cast<V4<T>*>(p + i + g.sz) = face;
        }
}

// The small levels (at most BOTTOM_CELLS cells each) as ONE launch of one workgroup: their launches would cost more than
// their work (a 512^3 cycle has three such levels and 60 coarsest-level half-sweeps; 32^3 cells on one CU is already slower than its launches).  Same per-cell expressions as the
// kernels above; __syncthreads() orders the phases (one workgroup: one CU, one L1).
constexpr long BOTTOM_CELLS = 4096;
constexpr int BOTTOM_MAX = 10;

template <class T>
struct MgBottom {
    MgLevel<T> lv[BOTTOM_MAX];
    int n, pre, post, coarse;
};

template <class T>
__global__ __launch_bounds__(1024) void mg_bottom_kernel(MgBottom<T> B)
{
    const int tid = threadIdx.x;
    auto smooth = [&](const MgLevel<T>& l, int n) {
        const int hw = (l.W + 1) >> 1, cnt = hw * l.H * l.D;
        for (int it = 0; it < n; ++it)
            for (int colour = 0; colour < 2; ++colour) {
                for (int j = tid; j < cnt; j += 1024) {
                    const int y = 1 + (j / hw) % l.H, z = 1 + j / (hw * l.H);
                    const int x = 1 + 2 * (j % hw) + (((y + z + colour) & 1) ? 0 : 1);
                    if (x > l.W) continue;
                    const long c = lat(l, x, y, z);
                    const T dg = l.dg[c];
                    if (dg > (T)0) l.e[c] = (l.b[c] + weighted_neighbours(l, c)) / dg;
                }
                __syncthreads();
            }
    };
    auto restrict_to = [&](const MgLevel<T>& f, const MgLevel<T>& c) {
        const int cnt = c.W * c.H * c.D;
        for (int j = tid; j < cnt; j += 1024) {
            const int X = 1 + j % c.W, Y = 1 + (j / c.W) % c.H, Z = 1 + j / (c.W * c.H);
            const long C = lat(c, X, Y, Z);
            const int x = 2 * X - 1, y = 2 * Y - 1, z = 2 * Z - 1;
            T r = residual_at(f, lat(f, x, y, z));
            r = r + residual_at(f, lat(f, x + 1, y, z));
            r = r + residual_at(f, lat(f, x, y + 1, z));
            r = r + residual_at(f, lat(f, x + 1, y + 1, z));
            r = r + residual_at(f, lat(f, x, y, z + 1));
            r = r + residual_at(f, lat(f, x + 1, y, z + 1));
            r = r + residual_at(f, lat(f, x, y + 1, z + 1));
            r = r + residual_at(f, lat(f, x + 1, y + 1, z + 1));
            c.b[C] = (c.dg[C] > (T)0) ? (T)0.5 * r : (T)0;
            c.e[C] = (T)0;
        }
        __syncthreads();
    };
    auto prolong_to = [&](const MgLevel<T>& c, const MgLevel<T>& f) {
        const int cnt = f.W * f.H * f.D;
        for (int j = tid; j < cnt; j += 1024) {
            const int x = 1 + j % f.W, y = 1 + (j / f.W) % f.H, z = 1 + j / (f.W * f.H);
            const long i = lat(f, x, y, z);
            if (f.dg[i] > (T)0) f.e[i] = f.e[i] + interp(c, x, y, z);
        }
        __syncthreads();
    };
    for (int k = 0; k < B.n - 1; ++k) {
        smooth(B.lv[k], B.pre);
        restrict_to(B.lv[k], B.lv[k + 1]);
    }
    smooth(B.lv[B.n - 1], B.coarse);
    for (int k = B.n - 2; k >= 0; --k) {
        prolong_to(B.lv[k + 1], B.lv[k]);
        smooth(B.lv[k], B.post);
    }
}

inline dim3 blk() { return dim3(64, 4, 1); }
inline dim3 grd(int nx, int ny, int nz) { return dim3((nx + 63) / 64, (ny + 3) / 4, nz); }

}  // namespace

template <class T>
void Multigrid<T>::release()
{
    if (pool) hipFree(pool);
    if (reg_pool) hipFree(reg_pool);
    pool = nullptr;
    reg_pool = nullptr;
    pool_elems = 0;
    lv.clear();
    W0 = H0 = D0 = 0;
}

// This rank's planes of replicated level l as a slab: the pointers are shifted so that plane z of the view is global plane
// zoff + z; the planes beside it are the neighbours' (the array holds the whole level), so the view needs no halo exchange.
template <class T>
MgLevel<T> Multigrid<T>::slab_view(int l, int fine_zoff, int fine_D, const SlabCtx& sc) const
{
    MgLevel<T> v = lv[l];
    v.zoff = fine_zoff / 2;
    v.D = fine_D / 2;
    v.lo_wall = sc.lo_wall;
    v.hi_wall = sc.hi_wall;
    const long shift = (long)v.zoff * v.sz;
    for (T** a : { &v.wx, &v.wy, &v.wz, &v.d, &v.dg, &v.e, &v.b }) *a += shift;
    v.reg += shift;
    return v;
}

template <class T>
int Multigrid<T>::build(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, int nranks_, int rank_,
                        int min_planes, const MgHooks<T>* hooks)
{
    const int Dg = sc.Dglobal;
    if (g.W != W0 || g.H != H0 || g.D != D0 || nranks_ != nranks || rank_ != rank || lv.empty()) {
        release();
        W0 = g.W; H0 = g.H; D0 = g.D;
        nranks = nranks_;
        rank = rank_;
        MgLevel<T> l0{};
        l0.W = g.W; l0.H = g.H; l0.D = g.D;
        lv.push_back(l0);
        // the hierarchy is that of the GLOBAL grid (a slab run must coarsen exactly like the same run on one GPU)
        int W = g.W, H = g.H, D = Dg;
        first_repl = 1;
        bool dist = nranks > 1;                           // one GPU: every level is held whole
        size_t total = 0;
        while (W % 2 == 0 && H % 2 == 0 && D % 2 == 0 && W / 2 >= MG_MIN_DIM && H / 2 >= MG_MIN_DIM && D / 2 >= MG_MIN_DIM) {
            W /= 2; H /= 2; D /= 2;
            MgLevel<T> l{};
            l.W = W; l.H = H;
            // a level stays distributed while every rank keeps at least min_planes planes of it
            dist = dist && D % nranks == 0 && D / nranks >= (min_planes > 1 ? min_planes : 1);
            if (dist) {
                l.D = D / nranks;
                l.zoff = rank * l.D;
                l.lo_wall = sc.lo_wall;
                l.hi_wall = sc.hi_wall;
                first_repl = (int)lv.size() + 1;
            } else {
                l.D = D;
            }
            l.sy = W + 2;
            l.sz = l.sy * (H + 2);
            l.n = l.sz * (l.D + 2);
            total += 7 * (size_t)l.n;
            lv.push_back(l);
        }
        if (nranks > 1 && lv.size() > 1 && (g.D % 2) != 0) { release(); return 2; }   // children of a coarse cell must be one rank's
        if (total) {
            hipError_t e = hipMalloc((void**)&pool, total * sizeof(T));
            if (e == hipSuccess) e = hipMalloc((void**)&reg_pool, total / 7);
            if (e != hipSuccess) { release(); return 1; }
            pool_elems = total;
            T* q = pool;
            uint8_t* rq = reg_pool;
            for (size_t i = 1; i < lv.size(); ++i) {
                MgLevel<T>& l = lv[i];
                T** arrs[] = { &l.wx, &l.wy, &l.wz, &l.d, &l.dg, &l.e, &l.b };
                for (T** a : arrs) { *a = q; q += l.n; }
                l.reg = rq;
                rq += l.n;
            }
        }
        if (nranks > 1) return -1;                       // fresh allocation: the caller exports it to its peers, then calls again
    }
    if (pool) {
        hipError_t e = hipMemsetAsync(pool, 0, pool_elems * sizeof(T), st);     // ghosts, dead cells and wall faces stay 0
        if (e == hipSuccess) e = hipMemsetAsync(reg_pool, 0, pool_elems / 7, st);
        if (e != hipSuccess) return 1;
    }
    const int nl = levels();
    for (int i = 1; i < nl; ++i) {
        // coefficients of level i from level i-1.  The first replicated level below a distributed one: every rank computes
        // its planes (as a slab view), then the four coefficient arrays are gathered.
        const bool seam = nranks > 1 && i == first_repl;
        const int fz = (i == 1) ? sc.zoff : lv[i - 1].zoff, fD = (i == 1) ? g.D : lv[i - 1].D;
        MgLevel<T> c = seam ? slab_view(i, fz, fD, sc) : lv[i];
        if (i == 1)
            hipLaunchKernelGGL((mg_coarsen_kernel<T, true>), grd(c.W + 1, c.H + 1, c.D + 1), blk(), 0, st, g, sc, flags, lv[0], c);
        else
            hipLaunchKernelGGL((mg_coarsen_kernel<T, false>), grd(c.W + 1, c.H + 1, c.D + 1), blk(), 0, st, g, sc, flags, lv[i - 1], c);
        if (seam) {
            if (!hooks) return 1;
            for (T* a : { lv[i].wx, lv[i].wy, lv[i].wz, lv[i].d })
                if (hooks->gather(lv[i], a, c.D, c.zoff)) return 3;
        }
        hipLaunchKernelGGL((mg_diag_kernel<T>), grd(lv[i].W, lv[i].H, lv[i].D), blk(), 0, st, lv[i]);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Levels first .. coarsest, all held whole by this rank: pre-smooth and restrict on the way down, coarse_iters on the
// coarsest, prolong and post-smooth on the way up; the levels of at most BOTTOM_CELLS cells as one single-workgroup launch.
template <class T>
void Multigrid<T>::vcycle_replicated(hipStream_t st, int first, int pre, int post, int coarse_iters)
{
    const int nl = levels();
    if (first >= nl) return;
    auto smooth = [&](MgLevel<T>& l, int n) {
        for (int it = 0; it < n; ++it)
            for (int colour = 0; colour < 2; ++colour)
                hipLaunchKernelGGL((mg_smooth_kernel<T>), grd((l.W + 1) / 2, l.H, l.D), blk(), 0, st, l, colour);
    };
    int lb = nl;                                          // first level of the single-workgroup bottom
    for (int l = first; l < nl; ++l)
        if ((long)lv[l].W * lv[l].H * lv[l].D <= BOTTOM_CELLS && nl - l <= BOTTOM_MAX) { lb = l; break; }
    const int top = lb < nl - 1 ? lb : nl - 1;            // the level the launches below stop at
    for (int l = first; l < top; ++l) {                   // down
        smooth(lv[l], pre);
        hipLaunchKernelGGL((mg_restrict_kernel<T>), grd(lv[l + 1].W, lv[l + 1].H, lv[l + 1].D), blk(), 0, st, lv[l], lv[l + 1]);
    }
    if (lb < nl) {
        MgBottom<T> B;
        B.n = nl - lb;
        for (int k = 0; k < B.n; ++k) B.lv[k] = lv[lb + k];
        B.pre = pre; B.post = post; B.coarse = coarse_iters;
        hipLaunchKernelGGL((mg_bottom_kernel<T>), dim3(1), dim3(1024), 0, st, B);
    } else {
        smooth(lv[nl - 1], coarse_iters);
    }
    for (int l = top - 1; l >= first; --l) {              // up
        hipLaunchKernelGGL((mg_prolong_kernel<T>), grd(lv[l].W, lv[l].H, lv[l].D), blk(), 0, st, lv[l + 1], lv[l]);
        smooth(lv[l], post);
    }
}

template <class T>
int Multigrid<T>::coarse_correction(hipStream_t st, const GridDesc& g, const SlabCtx& sc, const uint8_t* flags, T* p, const T* rhs,
                                    int pre, int post, int coarse_iters, const MgHooks<T>* hooks)
{
    const int nl = levels();
    if (nl < 2) return 0;
    if (nranks == 1) {
        hipLaunchKernelGGL((mg_restrict0_kernel<T>), grd((g.W + 3) / 4, lv[1].H, lv[1].D), blk(), 0, st, g, flags, p, rhs, lv[1]);
        vcycle_replicated(st, 1, pre, post, coarse_iters);
        hipLaunchKernelGGL((mg_prolong0_kernel<T>), grd((g.W + 3) / 4, lv[1].H, lv[1].D), blk(), 0, st, lv[1], g, sc, flags, p);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    // ---- z-slabs: the same operations in the same order; what differs is where a level lives
    if (!hooks) return 1;
    const int fr = first_repl;                             // levels 1 .. fr-1 distributed, fr .. nl-1 held whole by every rank
    auto fine_zoff = [&](int l) { return l == 0 ? sc.zoff : lv[l].zoff; };
    auto fine_D = [&](int l) { return l == 0 ? g.D : lv[l].D; };
    // level l as the level above it sees it: itself if distributed, else this rank's planes of the replicated array
    auto as_child_of = [&](int l) { return (l == fr) ? slab_view(l, fine_zoff(l - 1), fine_D(l - 1), sc) : lv[l]; };
    // after a restriction into level l: a distributed level's halo planes of e must be the neighbours' zeros; a replicated
    // level needs everybody's planes of b and a zero e everywhere
    auto after_restrict = [&](int l, const MgLevel<T>& written) -> int {
        if (l == fr) {
            if (hooks->gather(lv[l], lv[l].b, written.D, written.zoff)) return 3;
            if (hipMemsetAsync(lv[l].e, 0, (size_t)lv[l].n * sizeof(T), st) != hipSuccess) return 1;
        } else {
            if (hipMemsetAsync(lv[l].e, 0, (size_t)lv[l].sz * sizeof(T), st) != hipSuccess) return 1;
            if (hipMemsetAsync(lv[l].e + (long)(lv[l].D + 1) * lv[l].sz, 0, (size_t)lv[l].sz * sizeof(T), st) != hipSuccess) return 1;
        }
        return 0;
    };
    // red-black smoothing of a distributed level: each colour reads the other colour's cells of the neighbours' boundary planes
    auto smooth_dist = [&](MgLevel<T>& l, int n) -> int {
        for (int it = 0; it < n; ++it)
            for (int colour = 0; colour < 2; ++colour) {
                hipLaunchKernelGGL((mg_smooth_kernel<T>), grd((l.W + 1) / 2, l.H, l.D), blk(), 0, st, l, colour);
                if (hooks->halo(l, l.e)) return 3;
            }
        return 0;
    };
    int rc;
    {
        MgLevel<T> c1 = as_child_of(1);
        hipLaunchKernelGGL((mg_restrict0_kernel<T>), grd((g.W + 3) / 4, c1.H, c1.D), blk(), 0, st, g, flags, p, rhs, c1);
        if ((rc = after_restrict(1, c1))) return rc;
    }
    for (int l = 1; l < fr; ++l) {                         // down the distributed levels
        if (l == nl - 1) {                                 // the coarsest level of all is still distributed
            if ((rc = smooth_dist(lv[l], coarse_iters))) return rc;
            break;
        }
        if ((rc = smooth_dist(lv[l], pre))) return rc;
        MgLevel<T> c = as_child_of(l + 1);
        hipLaunchKernelGGL((mg_restrict_kernel<T>), grd(c.W, c.H, c.D), blk(), 0, st, lv[l], c);
        if ((rc = after_restrict(l + 1, c))) return rc;
    }
    vcycle_replicated(st, fr, pre, post, coarse_iters);   // nothing to communicate below
    for (int l = (fr < nl ? fr : nl - 1) - 1; l >= 1; --l) {   // up the distributed levels
        MgLevel<T> c = as_child_of(l + 1);
        hipLaunchKernelGGL((mg_prolong_kernel<T>), grd(lv[l].W, lv[l].H, lv[l].D), blk(), 0, st, c, lv[l]);
        if (hooks->halo(lv[l], lv[l].e)) return 3;
        if ((rc = smooth_dist(lv[l], post))) return rc;
    }
    {
        MgLevel<T> c1 = as_child_of(1);
        hipLaunchKernelGGL((mg_prolong0_kernel<T>), grd((g.W + 3) / 4, c1.H, c1.D), blk(), 0, st, c1, g, sc, flags, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

template struct Multigrid<float>;
template struct Multigrid<double>;

}  // namespace fs

// surface.hip -- see surface.h.  Marching cubes over the padded obstacle array at level 0.5:
// one vertex per grid edge whose end points lie on different sides of 0.5 (placed by linear
// interpolation; obs is 0/1, so at the edge's midpoint), triangles from a 256-entry case table.
//
// The table is not copied from anywhere: it is constructed at first use (build_table) from the
// geometry of the cube -- on every cube face the crossing points are joined by segments, the
// segments of the six faces close into loops, every loop is oriented solid -> fluid and cut into a
// triangle fan.  A face with four crossings (diagonal corners inside) always cuts each inside corner
// off separately; that choice depends only on the four values on the face, so the two cubes sharing
// the face agree and the mesh has no cracks (tests/test_surface.py checks all 3 x 4096 pairs of
// neighbouring cubes on the CPU, and closedness of whole meshes on the GPU).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/fluidsim.h"
#include "kernels_dev.h"
#include "surface.h"

namespace fs {

namespace {

struct Table {
    unsigned char ntri[256];
    unsigned char edges[256][3 * SURF_MAX_TRIS];
    bool ok;
};

// other axes of axis a, lower first
inline void others(int a, int& u, int& v)
{
    u = (a == 0) ? 1 : 0;
    v = (a == 2) ? 1 : 2;
}

Table build_table()
{
    Table t;
    memset(&t, 0, sizeof t);
    t.ok = true;
    int e_c0[12], e_c1[12];                                   // end corners of every cube edge (c0 has bit `axis` clear)
    for (int a = 0; a < 3; ++a) {
        int u, v;
        others(a, u, v);
        for (int ov = 0; ov < 2; ++ov)
            for (int ou = 0; ou < 2; ++ou) {
                const int id = 4 * a + 2 * ov + ou;
                e_c0[id] = (ou << u) | (ov << v);
                e_c1[id] = e_c0[id] | (1 << a);
            }
    }
    auto edge_between = [&](int ca, int cb) {
        for (int e = 0; e < 12; ++e)
            if ((e_c0[e] == ca && e_c1[e] == cb) || (e_c0[e] == cb && e_c1[e] == ca)) return e;
        return -1;
    };
    for (int cfg = 0; cfg < 256; ++cfg) {
        auto in = [&](int c) { return ((cfg >> c) & 1) != 0; };
        int nbr[12][2], cnt[12];
        for (int e = 0; e < 12; ++e) cnt[e] = 0;
        auto join = [&](int ea, int eb) {
            if (ea < 0 || eb < 0 || cnt[ea] >= 2 || cnt[eb] >= 2) { t.ok = false; return; }
            nbr[ea][cnt[ea]++] = eb;
            nbr[eb][cnt[eb]++] = ea;
        };
        for (int d = 0; d < 3; ++d)
            for (int sd = 0; sd < 2; ++sd) {
                int u, v;
                others(d, u, v);
                static const int pu[4] = {0, 1, 1, 0}, pv[4] = {0, 0, 1, 1};
                int cyc[4];
                for (int k = 0; k < 4; ++k) cyc[k] = (sd << d) | (pu[k] << u) | (pv[k] << v);
                int cross[4], n = 0;
                for (int k = 0; k < 4; ++k) {
                    cross[k] = in(cyc[k]) != in(cyc[(k + 1) & 3]);
                    n += cross[k];
                }
                if (n == 2) {
                    int ks[2], m = 0;
                    for (int k = 0; k < 4; ++k)
                        if (cross[k]) ks[m++] = k;
                    join(edge_between(cyc[ks[0]], cyc[(ks[0] + 1) & 3]), edge_between(cyc[ks[1]], cyc[(ks[1] + 1) & 3]));
                } else if (n == 4) {
                    for (int k = 0; k < 4; ++k)
                        if (in(cyc[k])) join(edge_between(cyc[(k + 3) & 3], cyc[k]), edge_between(cyc[k], cyc[(k + 1) & 3]));
                }
            }
        bool seen[12] = {false};
        int nt = 0;
        for (int e0 = 0; e0 < 12; ++e0) {
            if (cnt[e0] == 0 || seen[e0]) continue;
            if (cnt[e0] != 2) { t.ok = false; continue; }
            int loop[12], n = 0, prev = -1, cur = e0;
            do {
                if (n >= 12 || cnt[cur] != 2) { t.ok = false; break; }
                loop[n++] = cur;
                seen[cur] = true;
                const int nxt = (nbr[cur][0] != prev) ? nbr[cur][0] : nbr[cur][1];
                prev = cur;
                cur = nxt;
            } while (cur != e0);
            if (n < 3) { t.ok = false; continue; }
            // orientation: Newell normal of the loop (vertices at the edge midpoints) against the summed
            // solid -> fluid directions of the edges it crosses
            double P[12][3], N[3] = {0, 0, 0}, R[3] = {0, 0, 0};
            for (int i = 0; i < n; ++i) {
                const int e = loop[i], ci = in(e_c0[e]) ? e_c0[e] : e_c1[e], co = in(e_c0[e]) ? e_c1[e] : e_c0[e];
                for (int k = 0; k < 3; ++k) {
                    P[i][k] = 0.5 * (((e_c0[e] >> k) & 1) + ((e_c1[e] >> k) & 1));
                    R[k] += ((co >> k) & 1) - ((ci >> k) & 1);
                }
            }
            for (int i = 0; i < n; ++i) {
                const double* p = P[i];
                const double* q = P[(i + 1) % n];
                N[0] += p[1] * q[2] - p[2] * q[1];
                N[1] += p[2] * q[0] - p[0] * q[2];
                N[2] += p[0] * q[1] - p[1] * q[0];
            }
            const double dot = N[0] * R[0] + N[1] * R[1] + N[2] * R[2];
            if (dot == 0.0) t.ok = false;
            if (dot < 0.0)
                for (int i = 0; i < n / 2; ++i) { int tmp = loop[i]; loop[i] = loop[n - 1 - i]; loop[n - 1 - i] = tmp; }
            // triangle fan; the apex is chosen so that no triangle lies flat in a face of the cube (three
            // vertices on one face would put a sliver of surface between two cubes)
            auto flat = [&](int ea, int eb, int ec) {
                for (int k = 0; k < 3; ++k)
                    for (int side = 0; side < 2; ++side) {
                        auto on = [&](int e) { return ((e_c0[e] >> k) & 1) == side && ((e_c1[e] >> k) & 1) == side; };
                        if (on(ea) && on(eb) && on(ec)) return true;
                    }
                return false;
            };
            int apex = -1;
            for (int k = 0; k < n && apex < 0; ++k) {
                bool good = true;
                for (int i = 1; i + 1 < n && good; ++i) good = !flat(loop[k], loop[(k + i) % n], loop[(k + i + 1) % n]);
                if (good) apex = k;
            }
            if (apex < 0) { t.ok = false; apex = 0; }
            for (int i = 1; i + 1 < n; ++i) {
                if (nt >= SURF_MAX_TRIS) { t.ok = false; break; }
                t.edges[cfg][3 * nt + 0] = (unsigned char)loop[apex];
                t.edges[cfg][3 * nt + 1] = (unsigned char)loop[(apex + i) % n];
                t.edges[cfg][3 * nt + 2] = (unsigned char)loop[(apex + i + 1) % n];
                ++nt;
            }
        }
        t.ntri[cfg] = (unsigned char)nt;
    }
    return t;
}

const Table& table()
{
    static const Table t = build_table();
    return t;
}

__constant__ unsigned char c_ntri[256];
__constant__ unsigned char c_edges[256][3 * SURF_MAX_TRIS];

template <class T>
__device__ __forceinline__ bool solid(T v) { return v > (T)0.5; }

constexpr int VB_SHIFT = 28;                             // vbase word: first vertex index | (owned-edge mask << 28)
constexpr int VB_MASK = (1 << VB_SHIFT) - 1;

// crossing edges owned by grid point (x, y, z): bit a set <=> the edge towards +axis a crosses 0.5
template <class T>
__device__ __forceinline__ unsigned owned_edges(const GridDesc& g, const T* __restrict__ obs, int x, int y, int z, long c)
{
    const bool in0 = solid(obs[c]);
    unsigned m = 0;
    if (x <= g.W && solid(obs[c + 1]) != in0) m |= 1u;
    if (y <= g.H && solid(obs[c + g.sy]) != in0) m |= 2u;
    if (z <= g.D && solid(obs[c + g.sz]) != in0) m |= 4u;
    return m;
}
template <class T>
__device__ __forceinline__ unsigned cube_config(const GridDesc& g, const T* __restrict__ obs, long c)
{
    unsigned cfg = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (solid(obs[c + (i & 1) + ((i >> 1) & 1) * g.sy + ((i >> 2) & 1) * g.sz])) cfg |= 1u << i;
    return cfg;
}

// exclusive prefix sum of one int per thread over a 256-thread workgroup; *total = the workgroup's sum
__device__ __forceinline__ int block_scan256(int v, int* total)
{
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();                                     // wsum may still be read from the previous call
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += wsum[w];
    }
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - v;
}

// pass 1: vertices and triangles per grid row (y, z)
template <class T>
__global__ __launch_bounds__(256) void surf_count_kernel(GridDesc g, const T* __restrict__ obs, int* __restrict__ rowV,
                                                          int* __restrict__ rowT)
{
    const int row = blockIdx.x, y = row % (g.H + 2), z = row / (g.H + 2);
    int nv = 0, nt = 0;
    for (int x = threadIdx.x; x <= g.W + 1; x += 256) {
        const long c = cell(g, x, y, z);
        nv += __popc(owned_edges(g, obs, x, y, z, c));
        if (x <= g.W && y <= g.H && z <= g.D) nt += c_ntri[cube_config(g, obs, c)];
    }
    int tv, tt;
    block_scan256(nv, &tv);
    block_scan256(nt, &tt);
    if (threadIdx.x == 0) {
        rowV[row] = tv;
        rowT[row] = tt;
    }
}

// pass 2: exclusive scan over the rows (one workgroup); offs[nrows] = total; overflow[0] set when a total leaves int range
__global__ __launch_bounds__(256) void surf_scan_kernel(const int* __restrict__ cnt, int* __restrict__ offs, int nrows, int limit,
                                                         int* overflow)
{
    long carry = 0;
    for (int i0 = 0; i0 < nrows; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const int v = (i < nrows) ? cnt[i] : 0;
        int tot;
        const int ex = block_scan256(v, &tot);
        if (i < nrows) offs[i] = (int)(carry + ex);
        carry += tot;
        if (carry > (long)limit) { if (threadIdx.x == 0) overflow[0] = 1; carry = limit; }
    }
    if (threadIdx.x == 0) offs[nrows] = (int)carry;
}

// pass 3: vertices, and per grid point the index of its first vertex + which of its three edges cross
template <class T>
__global__ __launch_bounds__(256) void surf_vertex_kernel(GridDesc g, const T* __restrict__ obs, const int* __restrict__ rowVoff,
                                                           int* __restrict__ vbase, float* __restrict__ verts)
{
    const int row = blockIdx.x, y = row % (g.H + 2), z = row / (g.H + 2);
    int running = rowVoff[row];
    const long dense_row = (long)row * (g.W + 2);
    for (int x0 = 0; x0 <= g.W + 1; x0 += 256) {
        const int x = x0 + threadIdx.x;
        const bool on = x <= g.W + 1;
        const long c = cell(g, on ? x : 0, y, z);
        const unsigned m = on ? owned_edges(g, obs, x, y, z, c) : 0u;
        int tot;
        const int first = running + block_scan256(__popc(m), &tot);
        running += tot;
        if (!on) continue;
        vbase[dense_row + x] = first | (int)(m << VB_SHIFT);
        const float v0 = (float)obs[c];
        int k = first;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!(m & (1u << a))) continue;
            const float v1 = (float)obs[c + (a == 0 ? 1 : a == 1 ? g.sy : g.sz)];
            const float t = (0.5f - v0) / (v1 - v0);         // linear interpolation to the 0.5 level
            verts[3 * (long)k + 0] = (float)x + (a == 0 ? t : 0.0f);
            verts[3 * (long)k + 1] = (float)y + (a == 1 ? t : 0.0f);
            verts[3 * (long)k + 2] = (float)z + (a == 2 ? t : 0.0f);
            ++k;
        }
    }
}

// pass 4: triangles of every cube, vertex indices through vbase of the grid point that owns the edge
template <class T>
__global__ __launch_bounds__(256) void surf_triangle_kernel(GridDesc g, const T* __restrict__ obs, const int* __restrict__ rowToff,
                                                             const int* __restrict__ vbase, int* __restrict__ tris)
{
    const int row = blockIdx.x, y = row % (g.H + 2), z = row / (g.H + 2);
    if (y > g.H || z > g.D) return;                      // block-uniform: no cube starts on the last row / plane
    int running = rowToff[row];
    const long px = 1, py = g.W + 2, pz = (long)(g.W + 2) * (g.H + 2);   // dense point strides
    const long p0 = (long)row * (g.W + 2);
    for (int x0 = 0; x0 <= g.W; x0 += 256) {
        const int x = x0 + threadIdx.x;
        const bool on = x <= g.W;
        const unsigned cfg = on ? cube_config(g, obs, cell(g, x, y, z)) : 0u;
        const int nt = c_ntri[cfg];
        int tot;
        int k = running + block_scan256(nt, &tot);
        running += tot;
        for (int i = 0; i < nt; ++i, ++k) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = c_edges[cfg][3 * i + j];
                const int a = e >> 2, ov = (e >> 1) & 1, ou = e & 1;
                const long su = (a == 0) ? py : px, sv = (a == 2) ? py : pz;   // strides of the lower / higher other axis
                const int vb = vbase[p0 + x + ou * su + ov * sv];
                const unsigned m = (unsigned)vb >> VB_SHIFT;
                tris[3 * (long)k + j] = (vb & VB_MASK) + __popc(m & ((1u << a) - 1u));
            }
        }
    }
}

hipError_t upload_table()
{
    // __constant__ symbols exist once per device: upload for every device a handle lives on
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
    const Table& t = table();
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_ntri), t.ntri, sizeof t.ntri);
    if (e != hipSuccess) return e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_edges), t.edges, sizeof t.edges);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) done[dev] = true;
    return hipSuccess;
}

}  // namespace

int surface_case(int config, int* edges)
{
    const Table& t = table();
    if (!t.ok || config < 0 || config > 255) return -1;
    for (int i = 0; i < 3 * t.ntri[config]; ++i) edges[i] = t.edges[config][i];
    return t.ntri[config];
}

void surface_free(SurfaceResult* r)
{
    if (r->d_verts) hipFree(r->d_verts);
    if (r->d_tris) hipFree(r->d_tris);
    r->d_verts = nullptr;
    r->d_tris = nullptr;
    r->nverts = r->ntris = 0;
}

template <class T>
int extract_surface(hipStream_t st, const GridDesc& g, const T* obs, SurfaceResult* out, const char** err)
{
    static const char* msg_table = "the marching-cubes case table failed its construction checks";
    static const char* msg_hip = "HIP call failed during surface extraction";
    static const char* msg_big = "obstacle surface too large (more than 2^28 vertices or 2^30 triangles)";
    *out = SurfaceResult();
    if (!table().ok) { *err = msg_table; return FS_EINVAL; }
    if (upload_table() != hipSuccess) { *err = msg_hip; return FS_EHIP; }
    const int nrows = (g.H + 2) * (g.D + 2);
    const long npoints = (long)nrows * (g.W + 2);
    int *rows = nullptr, *vbase = nullptr;                // rows: rowV, rowT, rowVoff (+1), rowToff (+1), overflow flag
    int rc = FS_OK;
    int totals[2] = {0, 0}, overflow = 0;
    do {
        if (hipMalloc((void**)&rows, sizeof(int) * (4 * (size_t)nrows + 3)) != hipSuccess ||
            hipMalloc((void**)&vbase, sizeof(int) * (size_t)npoints) != hipSuccess) { *err = msg_hip; rc = FS_ENOMEM; break; }
        int *rowV = rows, *rowT = rows + nrows, *offV = rows + 2 * nrows, *offT = offV + nrows + 1, *ovf = offT + nrows + 1;
        if (hipMemsetAsync(ovf, 0, sizeof(int), st) != hipSuccess) { *err = msg_hip; rc = FS_EHIP; break; }
        hipLaunchKernelGGL((surf_count_kernel<T>), dim3(nrows), dim3(256), 0, st, g, obs, rowV, rowT);
        hipLaunchKernelGGL(surf_scan_kernel, dim3(1), dim3(256), 0, st, rowV, offV, nrows, VB_MASK, ovf);
        hipLaunchKernelGGL(surf_scan_kernel, dim3(1), dim3(256), 0, st, rowT, offT, nrows, 1 << 30, ovf);
        if (hipMemcpyAsync(&totals[0], offV + nrows, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(&totals[1], offT + nrows, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(&overflow, ovf, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { *err = msg_hip; rc = FS_EHIP; break; }
        if (overflow) { *err = msg_big; rc = FS_ENOMEM; break; }
        out->nverts = totals[0];
        out->ntris = totals[1];
        if (out->nverts == 0 || out->ntris == 0) { out->nverts = out->ntris = 0; break; }
        if (hipMalloc((void**)&out->d_verts, sizeof(float) * 3 * (size_t)out->nverts) != hipSuccess ||
            hipMalloc((void**)&out->d_tris, sizeof(int) * 3 * (size_t)out->ntris) != hipSuccess) { *err = msg_hip; rc = FS_ENOMEM; break; }
        hipLaunchKernelGGL((surf_vertex_kernel<T>), dim3(nrows), dim3(256), 0, st, g, obs, offV, vbase, out->d_verts);
        hipLaunchKernelGGL((surf_triangle_kernel<T>), dim3(nrows), dim3(256), 0, st, g, obs, offT, vbase, out->d_tris);
        if (hipStreamSynchronize(st) != hipSuccess) { *err = msg_hip; rc = FS_EHIP; break; }
    } while (0);
    if (rows) hipFree(rows);
    if (vbase) hipFree(vbase);
    if (rc) surface_free(out);
    return rc;
}
template int extract_surface<float>(hipStream_t, const GridDesc&, const float*, SurfaceResult*, const char**);
template int extract_surface<double>(hipStream_t, const GridDesc&, const double*, SurfaceResult*, const char**);

}  // namespace fs

// voxelize.hip -- STL -> solid voxels on the GPU.
//
// Restates loadSTLIntoObstacles (object_loader.cpp:270-452) at its one-thread semantics:
// the reference draws six numbers from one std::minstd_rand stream for every sample point
// that survives the coarse-grid rejection, in i,j,k order.  Here the rejection test runs
// for all samples in parallel, an exclusive scan gives every surviving sample its rank r
// in that order, and the sample jumps the generator ahead by 6r draws (x_n = x_0 * a^n mod
// m), so each sample sees exactly the numbers the sequential loop would have given it and
// the mask is reproducible bit for bit from (STL, arguments, seed).
//
// Mesh parsing, the Euler rotation (host cosf/sinf, like the reference) and the 64^3
// coarse occupancy grid are one-off host work on O(triangles) data; the
// O(samples x triangles) ray-parity test is the kernel.  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "kernels.h"
#include "voxelize.h"

namespace fs {

namespace {

struct P3 { float x, y, z; };
struct Tri { P3 a, b, c; };

struct VoxParams {
    int ns;
    float lo, res, cell;
    float gscale, cx, cy, cz, tx, ty, tz;
    int W, H, D;
    unsigned seed;
    int ntri;
};

constexpr int GRID = 64;                 // object_loader.cpp:382
constexpr unsigned LCG_A = 48271u;       // std::minstd_rand
constexpr unsigned LCG_M = 2147483647u;

std::string strip(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\n\r");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\n\r");
    return s.substr(a, b - a + 1);
}

// object_loader.cpp:98-174
bool read_stl(const char* path, std::vector<Tri>& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line;
    std::getline(f, line);
    const bool binary = (strip(line).find("solid") != 0);      // :107
    f.close();
    if (binary) {
        f.open(path, std::ios::binary);
        if (!f) return false;
        f.seekg(80);
        uint32_t n = 0;
        f.read(reinterpret_cast<char*>(&n), 4);
        if (!f) return true;
        out.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            float rec[12];
            uint16_t attr;
            f.read(reinterpret_cast<char*>(rec), sizeof rec);
            f.read(reinterpret_cast<char*>(&attr), 2);
            if (!f) break;                                       // truncated file: keep what is complete
            Tri t;
            t.a = { rec[3], rec[4], rec[5] };
            t.b = { rec[6], rec[7], rec[8] };
            t.c = { rec[9], rec[10], rec[11] };
            out.push_back(t);
        }
    } else {
        f.open(path);
        if (!f) return false;
        Tri cur{};
        int vi = 0;
        while (std::getline(f, line)) {
            line = strip(line);
            if (line == "outer loop") { vi = 0; continue; }
            if (line == "endloop") continue;
            if (line == "endfacet") { if (vi == 3) out.push_back(cur); continue; }
            if (line.compare(0, 6, "vertex") == 0) {
                std::istringstream iss(line.substr(6));
                float x, y, z;
                if (iss >> x >> y >> z) {
                    P3 v = { x, y, z };
                    if (vi == 0) cur.a = v; else if (vi == 1) cur.b = v; else if (vi == 2) cur.c = v;
                    vi = (vi + 1) % 4;                           // :166
                }
            }
        }
    }
    return true;
}

// object_loader.cpp:177-202
P3 rotate(const P3& p, float dx, float dy, float dz)
{
    const float rx = dx * M_PI / 180.0f, ry = dy * M_PI / 180.0f, rz = dz * M_PI / 180.0f;
    const float cx = cosf(rx), sx = sinf(rx), cy = cosf(ry), sy = sinf(ry), cz = cosf(rz), sz = sinf(rz);
    P3 o;
    o.x = (cy * cz) * p.x + (-cy * sz) * p.y + (sy) * p.z;
    o.y = (sx * sy * cz + cx * sz) * p.x + (-sx * sy * sz + cx * cz) * p.y + (-sx * cy) * p.z;
    o.z = (-cx * sy * cz + sx * sz) * p.x + (cx * sy * sz + sx * cz) * p.y + (cx * cy) * p.z;
    return o;
}

// ---------------------------------------------------------------------------- device

__device__ __forceinline__ unsigned mulmod(unsigned a, unsigned b)
{
    return (unsigned)(((unsigned long long)a * b) % LCG_M);
}
__device__ unsigned powmod(unsigned base, unsigned long long e)
{
    unsigned r = 1u;
    while (e) {
        if (e & 1ull) r = mulmod(r, base);
        base = mulmod(base, base);
        e >>= 1;
    }
    return r;
}

// sample n = (i*ns + j)*ns + k, the reference's loop order (object_loader.cpp:403-405)
__device__ __forceinline__ void sample_point(const VoxParams& q, long n, float& x, float& y, float& z)
{
    const int k = (int)(n % q.ns), j = (int)((n / q.ns) % q.ns), i = (int)(n / ((long)q.ns * q.ns));
    x = q.lo + i * q.res;                                        // :407-409
    y = q.lo + j * q.res;
    z = q.lo + k * q.res;
}

// VoxelGrid::contains, object_loader.cpp:79-87
__global__ void keep_kernel(VoxParams q, const uint8_t* __restrict__ occ, uint8_t* __restrict__ keep, long nsamp)
{
    long n = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (n >= nsamp) return;
    float x, y, z;
    sample_point(q, n, x, y, z);
    uint8_t k = 0;
    if (!(x < q.lo || y < q.lo || z < q.lo)) {
        int ix = (int)((x - q.lo) / q.cell), iy = (int)((y - q.lo) / q.cell), iz = (int)((z - q.lo) / q.cell);
        if (ix >= 0 && ix < GRID && iy >= 0 && iy < GRID && iz >= 0 && iz < GRID)
            k = occ[ix + iy * GRID + iz * GRID * GRID];
    }
    keep[n] = k;
}

__global__ void compact_kernel(const uint8_t* __restrict__ keep, const unsigned* __restrict__ rank,
                               unsigned* __restrict__ list, long nsamp)
{
    long n = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (n >= nsamp) return;
    if (keep[n]) list[rank[n]] = (unsigned)n;
}

// Moeller-Trumbore exactly as object_loader.cpp:205-233
__device__ __forceinline__ bool ray_hits(float ox, float oy, float oz, float dx, float dy, float dz, const float* t)
{
    const float EPS = 1e-6f;
    const float e1x = t[3] - t[0], e1y = t[4] - t[1], e1z = t[5] - t[2];
    const float e2x = t[6] - t[0], e2y = t[7] - t[1], e2z = t[8] - t[2];
    const float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
    const float det = e1x * hx + e1y * hy + e1z * hz;
    if (fabsf(det) < EPS) return false;
    const float f = 1.0f / det;
    const float sx = ox - t[0], sy = oy - t[1], sz = oz - t[2];
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if (u < 0.0f || u > 1.0f) return false;
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    const float v = f * (dx * qx + dy * qy + dz * qz);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float tt = f * (e2x * qx + e2y * qy + e2z * qz);
    return tt > 1e-3f;
}

__device__ __forceinline__ unsigned lcg_step(unsigned& st)
{
    st = mulmod(st, LCG_A);
    return st;
}
// std::uniform_real_distribution<float>(0.1f, 1.0f) on minstd_rand (libstdc++): one draw,
// generate_canonical = float(x - min) / float(range) with range 2^31-2 rounding to 2^31.
__device__ __forceinline__ float lcg_unit(unsigned& st)
{
    float r = (float)(unsigned long long)(lcg_step(st) - 1u) / 2147483648.0f;
    if (r >= 1.0f) r = 0.99999994f;                              // nextafter(1.0f, 0.0f)
    return r * (1.0f - 0.1f) + 0.1f;
}

constexpr int TILE = 256;   // triangles staged in LDS per pass (9 KB)

// One lane per surviving sample; all triangles stream through LDS in tiles that the whole
// block tests against (object_loader.cpp:417-444).
__global__ __launch_bounds__(256) void parity_kernel(VoxParams q, const float* __restrict__ tri,
                                                      const unsigned* __restrict__ list, long nkept,
                                                      int* __restrict__ cells, unsigned long long* __restrict__ count)
{
    __shared__ float sh[TILE * 9];
    const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const bool on = r < nkept;
    float px = 0, py = 0, pz = 0, dx = 1, dy = 1, dz = 1;
    if (on) {
        sample_point(q, (long)list[r], px, py, pz);
        unsigned s0 = q.seed % LCG_M;                            // minstd_rand(seed): 0 maps to 1
        if (s0 == 0) s0 = 1;
        unsigned st = mulmod(s0, powmod(LCG_A, 6ull * (unsigned long long)r));
        px += (float)(unsigned long long)(lcg_step(st) % 1000u) * 1e-6f - 5e-4f;   // :417-419
        py += (float)(unsigned long long)(lcg_step(st) % 1000u) * 1e-6f - 5e-4f;
        pz += (float)(unsigned long long)(lcg_step(st) % 1000u) * 1e-6f - 5e-4f;
        dx = lcg_unit(st);                                       // :422
        dy = lcg_unit(st);
        dz = lcg_unit(st);
    }
    int crossings = 0;
    for (int base = 0; base < q.ntri; base += TILE) {
        const int cnt = min(TILE, q.ntri - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 9; i += blockDim.x) sh[i] = tri[(long)base * 9 + i];
        __syncthreads();
        if (on)
            for (int t = 0; t < cnt; ++t) crossings += ray_hits(px, py, pz, dx, dy, dz, &sh[t * 9]) ? 1 : 0;
    }
    if (on && (crossings & 1)) {
        const int gx = (int)((px - 0.0f) * q.gscale + q.cx + q.tx);   // :432-434, truncation toward zero
        const int gy = (int)((py - 0.0f) * q.gscale + q.cy + q.ty);
        const int gz = (int)((pz - 0.0f) * q.gscale + q.cz + q.tz);
        if (gx >= 1 && gx <= q.W && gy >= 1 && gy <= q.H && gz >= 1 && gz <= q.D) {
            unsigned long long at = atomicAdd(count, 1ull);
            cells[at] = gx + gy * (q.W + 2) + gz * (q.W + 2) * (q.H + 2);
        }
    }
}

template <class T>
__global__ void mark_cells_kernel(GridDesc g, SlabCtx sc, T* obs, const int* __restrict__ cells, long n)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = cells[i];
    const int x = id % (g.W + 2), y = (id / (g.W + 2)) % (g.H + 2), z = id / ((g.W + 2) * (g.H + 2));
    const int zl = z - sc.zoff;
    // a slab also records the solids of its halo planes (the flag build reads them)
    if (zl < 1 - g.zh || zl > g.D + g.zh) return;
    obs[(long)x + (long)y * g.sy + (long)zl * g.sz] = (T)1;     // Simulation::addObstacle, simulation.cpp:157
}

#define VX_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) { out->error = std::string(#expr ": ") + hipGetErrorString(e_); return -3; } \
    } while (0)

}  // namespace

template <class T>
void launch_mark_cells(hipStream_t st, const GridDesc& g, const SlabCtx& sc, T* obs, const int* cells, long n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL((mark_cells_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, sc, obs, cells, n);
}
template void launch_mark_cells<float>(hipStream_t, const GridDesc&, const SlabCtx&, float*, const int*, long);
template void launch_mark_cells<double>(hipStream_t, const GridDesc&, const SlabCtx&, double*, const int*, long);

int voxelize_stl(hipStream_t st, const char* path, int W, int H, int D, float scale, float rot_x, float rot_y,
                 float rot_z, float tr_x, float tr_y, float tr_z, unsigned seed, bool quiet, VoxelResult* out)
{
    std::vector<Tri> raw;
    if (!read_stl(path, raw)) {
        fprintf(stderr, "Error: Cannot open STL file: %s\n", path);          // :101
        fprintf(stderr, "Failed to load STL: %s\n", path);                   // :283
        out->error = std::string("cannot open STL file: ") + path;
        return -2;
    }
    if (!quiet) printf("Loaded %zu triangles.\n", raw.size());               // :172
    if (raw.empty()) {
        fprintf(stderr, "Failed to load STL: %s\n", path);
        out->error = std::string("no triangles in STL file: ") + path;
        return -2;
    }

    // objCenter is the origin: orig_min/orig_max are never updated (:288-296)
    std::vector<float> rt(raw.size() * 9);
    float r2 = 0.0f;
    for (size_t i = 0; i < raw.size(); ++i) {
        const P3 v[3] = { raw[i].a, raw[i].b, raw[i].c };
        for (int k = 0; k < 3; ++k) {
            P3 r = rotate(v[k], rot_x, rot_y, rot_z);                        // :302-316
            rt[i * 9 + k * 3 + 0] = r.x; rt[i * 9 + k * 3 + 1] = r.y; rt[i * 9 + k * 3 + 2] = r.z;
            float d2 = v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z;  // unrotated, :328-333
            if (d2 > r2) r2 = d2;
        }
    }
    const float radius = std::sqrt(r2);
    const float pad = radius * 0.05f;                                        // :349
    const float lo = (0.0f - radius) - pad, hi = (0.0f + radius) + pad;
    const float span = hi - lo;                                              // objSize
    float res = span / 200.0f;                                               // :368
    if (res < 0.02f) res = 0.02f;
    const int ns = (int)(span / res);                                        // :370-372
    if (!quiet) {
        printf("Rotated object voxelization:\n  Grid: %d x %d x %d\n  Resolution: %g\n", ns, ns, ns, res);
        printf("  Rotations: X=%g\xC2\xB0 Y=%g\xC2\xB0 Z=%g\xC2\xB0\n", rot_x, rot_y, rot_z);
    }

    // coarse occupancy grid: triangle AABBs rasterised with (int) truncation (:54-77, :380-389)
    const float cell = res * 5.0f;
    std::vector<uint8_t> occ((size_t)GRID * GRID * GRID, 0);
    for (size_t i = 0; i < raw.size(); ++i) {
        const float* t = &rt[i * 9];
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(std::min(t[a], t[3 + a]), t[6 + a]);
            mx[a] = std::max(std::max(t[a], t[3 + a]), t[6 + a]);
        }
        int c0[3], c1[3];
        for (int a = 0; a < 3; ++a) {
            c0[a] = std::max(0, (int)((mn[a] - lo) / cell));
            c1[a] = std::min(GRID - 1, (int)((mx[a] - lo) / cell));
        }
        for (int z = c0[2]; z <= c1[2]; ++z)
            for (int y = c0[1]; y <= c1[1]; ++y)
                for (int x = c0[0]; x <= c1[0]; ++x) occ[x + y * GRID + z * GRID * GRID] = 1;
    }
    if (!quiet) printf("Built spatial grid for fast rejection.\n");          // :391

    VoxParams q;
    q.ns = ns; q.lo = lo; q.res = res; q.cell = cell;
    q.gscale = scale * std::min(std::min((float)W, (float)H), (float)D) / span;   // :429
    q.cx = (float)W / 2; q.cy = (float)H / 2; q.cz = (float)D / 2;                // :430
    q.tx = tr_x; q.ty = tr_y; q.tz = tr_z;
    q.W = W; q.H = H; q.D = D; q.seed = seed; q.ntri = (int)raw.size();

    out->ntri = (long)raw.size();
    out->ns = ns;
    out->resolution = res;
    out->added = 0;
    const long nsamp = (long)ns * ns * ns;
    if (nsamp <= 0) {
        if (!quiet) printf("Added 0 obstacle points.\n");
        return 0;
    }

    // one device slab for everything: tri | occ | keep | rank | list | count
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_tri = 0;
    const size_t o_occ = o_tri + up(rt.size() * sizeof(float));
    const size_t o_keep = o_occ + up(occ.size());
    const size_t o_rank = o_keep + up((size_t)nsamp);
    const size_t o_list = o_rank + up((size_t)nsamp * 4);
    const size_t o_cnt = o_list + up((size_t)nsamp * 4);
    size_t scan_bytes = 0;
    rocprim::exclusive_scan((void*)nullptr, scan_bytes, (uint8_t*)nullptr, (unsigned*)nullptr, 0u, (size_t)nsamp,
                            rocprim::plus<unsigned>(), st);
    const size_t o_scan = o_cnt + 256;
    const size_t total = o_scan + up(scan_bytes);
    char* w = nullptr;
    VX_HIP(hipMalloc((void**)&w, total));
    out->d_work = w;
    VX_HIP(hipMemcpyAsync(w + o_tri, rt.data(), rt.size() * sizeof(float), hipMemcpyHostToDevice, st));
    VX_HIP(hipMemcpyAsync(w + o_occ, occ.data(), occ.size(), hipMemcpyHostToDevice, st));
    VX_HIP(hipMemsetAsync(w + o_cnt, 0, 256, st));

    uint8_t* d_keep = (uint8_t*)(w + o_keep);
    unsigned* d_rank = (unsigned*)(w + o_rank);
    unsigned* d_list = (unsigned*)(w + o_list);
    unsigned long long* d_cnt = (unsigned long long*)(w + o_cnt);
    const unsigned nb = (unsigned)((nsamp + 255) / 256);
    hipLaunchKernelGGL(keep_kernel, dim3(nb), dim3(256), 0, st, q, (const uint8_t*)(w + o_occ), d_keep, nsamp);
    VX_HIP(rocprim::exclusive_scan((void*)(w + o_scan), scan_bytes, d_keep, d_rank, 0u, (size_t)nsamp,
                                   rocprim::plus<unsigned>(), st));
    hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(256), 0, st, (const uint8_t*)d_keep, (const unsigned*)d_rank,
                       d_list, nsamp);
    unsigned last_rank = 0;
    uint8_t last_keep = 0;
    VX_HIP(hipMemcpyAsync(&last_rank, d_rank + (nsamp - 1), 4, hipMemcpyDeviceToHost, st));
    VX_HIP(hipMemcpyAsync(&last_keep, d_keep + (nsamp - 1), 1, hipMemcpyDeviceToHost, st));
    VX_HIP(hipStreamSynchronize(st));
    const long nkept = (long)last_rank + (last_keep ? 1 : 0);

    if (nkept > 0) {
        VX_HIP(hipMalloc((void**)&out->d_cells, (size_t)nkept * sizeof(int)));
        hipLaunchKernelGGL(parity_kernel, dim3((unsigned)((nkept + 255) / 256)), dim3(256), 0, st, q,
                           (const float*)(w + o_tri), (const unsigned*)d_list, nkept, out->d_cells, d_cnt);
        unsigned long long cnt = 0;
        VX_HIP(hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
        VX_HIP(hipStreamSynchronize(st));
        out->added = (long)cnt;
    }
    if (!quiet) printf("Added %ld obstacle points.\n", out->added);          // :451
    return 0;
}

void voxelize_free(VoxelResult* r)
{
    if (r->d_cells) hipFree(r->d_cells);
    if (r->d_work) hipFree(r->d_work);
    r->d_cells = nullptr;
    r->d_work = nullptr;
}

}  // namespace fs

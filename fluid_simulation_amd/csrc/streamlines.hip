// streamlines.hip -- see streamlines.h.  Arithmetic follows GUI/utils.py expression by expression in
// float64 (numpy promotes the float32 grid values when they meet the float64 coordinates); the file
// is compiled with -ffp-contract=off like the rest.
#include <hip/hip_runtime.h>

#include <climits>

#include "streamlines.h"

namespace fs {

template <class T>
__global__ void obs_bbox_kernel(GridDesc g, const T* __restrict__ obs, int* box)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x > g.W + 1 || y > g.H + 1) return;
    if (obs[(long)x + (long)y * g.sy + (long)z * g.sz] > (T)0.5) {           // utils.py:127
        atomicMin(&box[0], x);
        atomicMin(&box[1], y);
        atomicMin(&box[2], z);
        atomicMax(&box[3], x);
        atomicMax(&box[4], y);
        atomicMax(&box[5], z);
    }
}

__global__ void bbox_init_kernel(int* box)
{
    if (threadIdx.x < 3) box[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) box[threadIdx.x] = INT_MIN;
}

template <class T>
void launch_obs_bbox(hipStream_t st, const GridDesc& g, const T* obs, int* box)
{
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, st, box);
    dim3 block(64, 4), grid((g.W + 2 + 63) / 64, (g.H + 2 + 3) / 4, g.D + 2);
    hipLaunchKernelGGL((obs_bbox_kernel<T>), grid, block, 0, st, g, obs, box);
}
template void launch_obs_bbox<float>(hipStream_t, const GridDesc&, const float*, int*);
template void launch_obs_bbox<double>(hipStream_t, const GridDesc&, const double*, int*);

// _interpolate_scalar, utils.py:40-76
template <class T>
__device__ double interp(const GridDesc& g, const T* __restrict__ f, const StreamParams& p, double x, double y, double z)
{
    x = fmin(fmax(x, 0.0), p.clip_hi[0]);                // np.clip
    y = fmin(fmax(y, 0.0), p.clip_hi[1]);
    z = fmin(fmax(z, 0.0), p.clip_hi[2]);
    const int x0 = (int)x, y0 = (int)y, z0 = (int)z;
    const double xd = x - x0, yd = y - y0, zd = z - z0;
    const T* q = f + (long)x0 + (long)y0 * g.sy + (long)z0 * g.sz;
    const double c000 = (double)q[0], c100 = (double)q[1];
    const double c010 = (double)q[g.sy], c110 = (double)q[g.sy + 1];
    const double c001 = (double)q[g.sz], c101 = (double)q[g.sz + 1];
    const double c011 = (double)q[g.sz + g.sy], c111 = (double)q[g.sz + g.sy + 1];
    const double c00 = c000 * (1 - xd) + c100 * xd;
    const double c01 = c001 * (1 - xd) + c101 * xd;
    const double c10 = c010 * (1 - xd) + c110 * xd;
    const double c11 = c011 * (1 - xd) + c111 * xd;
    const double c0 = c00 * (1 - yd) + c10 * yd;
    const double c1 = c01 * (1 - yd) + c11 * yd;
    return c0 * (1 - zd) + c1 * zd;
}

template <class T>
__global__ void streamline_kernel(GridDesc g, const T* __restrict__ vx, const T* __restrict__ vy,
                                  const T* __restrict__ vz, const T* __restrict__ obs, StreamParams p,
                                  const double* __restrict__ seeds, const int* __restrict__ cand, int ncand,
                                  int* __restrict__ count, double* __restrict__ pts, double* __restrict__ vel)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;      // index into the candidate list
    if (s >= ncand) return;
    const int id = cand[s];
    const int ix = id % p.nx, iy = (id / p.nx) % p.ny, iz = id / (p.nx * p.ny);
    const double sx = seeds[ix], sy = seeds[p.nx + iy], sz = seeds[p.nx + p.ny + iz];
    count[2 * s] = 0;
    count[2 * s + 1] = 0;
    // the bounding-box cull (utils.py:147-150) already happened on the host; :153-154 seeds inside obstacles
    if (obs[(long)(int)sx + (long)(int)sy * g.sy + (long)(int)sz * g.sz] > (T)0.5) return;

    const double v0x = interp(g, vx, p, sx, sy, sz), v0y = interp(g, vy, p, sx, sy, sz), v0z = interp(g, vz, p, sx, sy, sz);
    for (int part = 0; part < 2; ++part) {               // backward, then forward (:157-164)
        const double direction = part == 0 ? -1.0 : 1.0;
        double* P = pts + ((size_t)(2 * s + part) * (p.half + 1)) * 3;
        double* V = vel + ((size_t)(2 * s + part) * (p.half + 1)) * 3;
        P[0] = sx; P[1] = sy; P[2] = sz;                 // :81-82
        V[0] = v0x; V[1] = v0y; V[2] = v0z;
        int n = 1;
        double px = sx, py = sy, pz = sz;
        for (int it = 0; it < p.half; ++it) {            // _integrate_streamline_part, :84-113
            const double u = interp(g, vx, p, px, py, pz), v = interp(g, vy, p, px, py, pz), w = interp(g, vz, p, px, py, pz);
            const double speed = sqrt((u * u + v * v) + w * w);
            if (speed < 1e-6) break;
            px = px + direction * (u / speed) * p.step_size;
            py = py + direction * (v / speed) * p.step_size;
            pz = pz + direction * (w / speed) * p.step_size;
            if (isnan(px) || isnan(py) || isnan(pz) || isinf(px) || isinf(py) || isinf(pz)) break;
            if (!(1 <= px && px < p.bound_hi[0] && 1 <= py && py < p.bound_hi[1] && 1 <= pz && pz < p.bound_hi[2])) break;
            if (interp(g, obs, p, px, py, pz) > 0.5) break;
            P[3 * n] = px; P[3 * n + 1] = py; P[3 * n + 2] = pz;
            V[3 * n] = u; V[3 * n + 1] = v; V[3 * n + 2] = w;
            ++n;
        }
        count[2 * s + part] = n;
    }
}

template <class T>
void launch_streamlines(hipStream_t st, const GridDesc& g, const T* vx, const T* vy, const T* vz, const T* obs,
                        const StreamParams& p, const double* seeds, const int* cand, int ncand, int* count, double* pts,
                        double* vel)
{
    if (ncand <= 0) return;
    hipLaunchKernelGGL((streamline_kernel<T>), dim3((ncand + 63) / 64), dim3(64), 0, st, g, vx, vy, vz, obs, p, seeds, cand,
                       ncand, count, pts, vel);
}
template void launch_streamlines<float>(hipStream_t, const GridDesc&, const float*, const float*, const float*, const float*,
                                        const StreamParams&, const double*, const int*, int, int*, double*, double*);
template void launch_streamlines<double>(hipStream_t, const GridDesc&, const double*, const double*, const double*,
                                         const double*, const StreamParams&, const double*, const int*, int, int*, double*,
                                         double*);

}  // namespace fs

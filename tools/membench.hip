// tools/membench.hip -- calibration of the achievable HBM / Infinity-Cache streaming rates for
// the access mixes of the solver (development tool).  hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_read(const float4* __restrict__ a, float* out, long n4)
{
    float s = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 1.2345e-30f) out[0] = s;
}
__global__ void k_write(float4* __restrict__ a, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
// contiguous chunk per block instead of grid-stride
__global__ void k_triad_chunk(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, long n4)
{
    long per = (n4 + gridDim.x - 1) / gridDim.x;
    long beg = blockIdx.x * per, end = beg + per < n4 ? beg + per : n4;
    for (long i = beg + threadIdx.x; i < end; i += blockDim.x) {
        float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

int main(int argc, char** argv)
{
    long mb = argc > 1 ? atol(argv[1]) : 543;
    int blocks = argc > 2 ? atoi(argv[2]) : 2048;
    long n4 = mb * 1000000L / 16;
    float4 *a, *b, *c; float* out;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&c, n4 * 16)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16)); CK(hipMemset(c, 0, n4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    auto timeit = [&](const char* name, double bytes, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-16s %4ld MB/array  blocks %5d  %8.1f us  %7.0f GB/s\n", name, mb, blocks, ms * 1e3, bytes / ms / 1e6);
    };
    double B = (double)n4 * 16;
    timeit("read", B, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, out, n4); });
    timeit("write", B, [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, c, n4); });
    timeit("copy", 2 * B, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, c, n4); });
    timeit("triad 2R+1W", 3 * B, [&] { hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, 0, a, b, c, n4); });
    timeit("triad chunked", 3 * B, [&] { hipLaunchKernelGGL(k_triad_chunk, dim3(blocks), dim3(256), 0, 0, a, b, c, n4); });
    return 0;
}

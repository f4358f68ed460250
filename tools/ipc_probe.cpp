// ipc_probe.cpp -- stand-alone check of the FSIPC transport (fluid_simulation_amd/csrc/ipc.h) on ONE GPU:
// forks N rank processes (before any HIP call) that share the device, exports/maps two buffers per rank, runs
// `iters` neighbour exchanges (the halo pattern: my first/last block -> the neighbours' ghost blocks) plus
// reductions, all stream-ordered with no host synchronisation inside the loop, and checks every received block.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/ipc_probe.cpp -o build/ipc_probe -lrt -lpthread
//   ./tools/ipc_probe [ranks=3] [iters=200] [MiB per block=2] [cycles=1]
#include "../fluid_simulation_amd/csrc/ipc.h"

#include <sys/wait.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void fill_kernel(unsigned* p, size_t n, unsigned v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i;
}
__global__ void check_kernel(const unsigned* p, size_t n, unsigned v, unsigned* bad)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != v + (unsigned)i) atomicAdd(bad, 1u);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 2; } } while (0)

static int run_rank(int rank, int nranks, const char* name, int iters, size_t block_words)
{
    fs::IpcTransport t;
    t.name = name;
    t.rank = rank;
    t.nranks = nranks;
    std::string err;
    // layout of a buffer: [ghost lo][first][... ][last][ghost hi], blocks of block_words
    const size_t words = block_words * 6;
    unsigned* buf[2];
    for (int k = 0; k < 2; ++k) {
        CK(hipMalloc((void**)&buf[k], words * 4));
        CK(hipMemset(buf[k], 0, words * 4));
    }
    unsigned* bad = nullptr;
    double* d3 = nullptr;
    CK(hipMalloc((void**)&bad, 4));
    CK(hipMemset(bad, 0, 4));
    CK(hipMalloc((void**)&d3, 24));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k)
        if (t.register_buffer(k, buf[k], words * 4, k == 1, &err)) { fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); return 3; }
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) {
        unsigned* b = buf[it & 1];
        const unsigned tag = 1000003u * (unsigned)(it + 1) + 7919u * (unsigned)rank;
        // "compute": my first and last interior blocks of this pass
        fill_kernel<<<64, 256, 0, st>>>(b + block_words, block_words, tag);
        fill_kernel<<<64, 256, 0, st>>>(b + 4 * block_words, block_words, tag ^ 0x55555555u);
        fs::IpcTransport::Xfer x[2];
        int peers[2], n = 0;
        if (rank > 0) { x[n] = { rank - 1, b + block_words, b + 5 * block_words, block_words * 4 }; peers[n++] = rank - 1; }
        if (rank < nranks - 1) { x[n] = { rank + 1, b + 4 * block_words, b, block_words * 4 }; peers[n++] = rank + 1; }
        if (t.exchange(st, x, n, peers, n, &err)) { fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); return 4; }
        // what must have landed: the lower neighbour's last block in my ghost lo, the upper neighbour's first in ghost hi
        if (rank > 0) check_kernel<<<64, 256, 0, st>>>(b, block_words, (1000003u * (unsigned)(it + 1) + 7919u * (unsigned)(rank - 1)) ^ 0x55555555u, bad);
        if (rank < nranks - 1) check_kernel<<<64, 256, 0, st>>>(b + 5 * block_words, block_words, 1000003u * (unsigned)(it + 1) + 7919u * (unsigned)(rank + 1), bad);
        if (it % 16 == 0) {
            const double v[3] = { (double)(rank + 1) * (it + 1), (double)(rank - it), (double)(rank + it) };
            CK(hipMemcpyAsync(d3, v, 24, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));                // `v` is a stack array; the loop is otherwise free of host syncs
            if (t.reduce3(st, d3, &err)) { fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); return 5; }
            double r[3];
            CK(hipMemcpyAsync(r, d3, 24, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            const double want0 = (double)(it + 1) * nranks * (nranks + 1) / 2.0, want1 = (double)(0 - it), want2 = (double)(nranks - 1 + it);
            if (r[0] != want0 || r[1] != want1 || r[2] != want2) {
                fprintf(stderr, "rank %d: reduction %d gave %g %g %g, want %g %g %g\n", rank, it, r[0], r[1], r[2], want0, want1, want2);
                return 6;
            }
        }
    }
    CK(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned hbad = 0;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    const unsigned derr = t.device_error();
    printf("rank %d/%d: %d exchanges of %zu KiB per direction in %.1f ms (%.1f us each), %u wrong words, device error word %u, %ld copies\n",
           rank, nranks, iters, block_words * 4 / 1024, ms, ms * 1e3 / iters, hbad, derr, t.copies);
    t.destroy();
    for (int k = 0; k < 2; ++k) hipFree(buf[k]);
    hipFree(bad);
    hipFree(d3);
    hipStreamDestroy(st);
    return (hbad || derr) ? 7 : 0;
}

int main(int argc, char** argv)
{
    const int nranks = argc > 1 ? atoi(argv[1]) : 3;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const size_t block_words = (size_t)((argc > 3 ? atof(argv[3]) : 2.0) * 262144);
    const int cycles = argc > 4 ? atoi(argv[4]) : 1;
    char name[64];
    snprintf(name, sizeof name, "/fs_ipc_probe_%d", (int)getpid());
    std::vector<pid_t> kids;
    for (int r = 0; r < nranks; ++r) {
        pid_t p = fork();                                // before any HIP call: every rank initialises the GPU itself
        if (p == 0) {
            // `cycles` transports one after the other in the same process (allocation, export, mapping and teardown repeat)
            int rc = 0;
            for (int c = 0; c < cycles && !rc; ++c) {
                char nm[96];
                snprintf(nm, sizeof nm, "%s_%d", name, c);
                rc = run_rank(r, nranks, nm, iters, block_words);
            }
            fflush(stdout);
            _exit(rc);
        }
        kids.push_back(p);
    }
    int rc = 0;
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1;
    }
    printf(rc ? "ipc_probe: FAILED\n" : "ipc_probe: ok\n");
    return rc;
}

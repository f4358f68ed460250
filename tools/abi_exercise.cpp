// tools/abi_exercise.cpp -- drives most of the C ABI from plain C++ (no Python), meant to be built
// with a host sanitizer:  see the recipe in DESIGN.md ("host sanitizers").  Exit code 0 = ran clean.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "../include/fluidsim.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, fs_last_error()); return 1; } } while (0)

// One rank of a two-rank z-slab run over the FSIPC transport (both ranks share the GPU): the host side of the slab path --
// export / mapping of the arrays, the timed choice of the communication schedule (all four incl. the push schedule), the
// asynchronous reach, the split density solve, windowed gathers, per-rank dumps, reduced statistics, solver=mg on slabs.
static int slab_rank(int rank, const char* id, const char* ascii, const char* dumpdir)
{
    fs_sim* s = fs_create(32, 16, 32, 2, 30, 0.05f, 2.0e-5f, 1.5e-5f, 7);
    if (!s) { fprintf(stderr, "fs_create: %s\n", fs_last_error()); return 1; }
    CHECK(fs_set_option(s, "quiet", "1"));
    CHECK(fs_set_option(s, "dump_dir", dumpdir));
    CHECK(fs_set_option(s, "overlap", "auto"));
    CHECK(fs_set_option(s, "comm_cus", "auto"));
    char idbuf[FS_COMM_ID_BYTES] = {0};
    snprintf(idbuf, sizeof idbuf, "%s", id);
    CHECK(fs_comm_init(s, rank, 2, idbuf));
    long added = 0;
    CHECK(fs_load_stl(s, ascii, 0.6f, 0.f, 0.f, 0.f, 5.f, 0.f, 0.f, &added));
    CHECK(fs_add_obstacle(s, 9, 7, 16));
    CHECK(fs_add_obstacle(s, 9, 7, 17));
    CHECK(fs_run(s));                                            // two steps with dumps
    CHECK(fs_set_velocity(s, 6, 5, 16, 1.5f, -0.5f, 2.0f));      // v_z jumps between steps
    CHECK(fs_run_one(s));
    int plan = -1, syncs = -1;
    CHECK(fs_get_int(s, "overlap_plan", &plan));
    CHECK(fs_get_int(s, "stream_syncs", &syncs));
    if (plan < 0 || plan > 3 || syncs != 0) { fprintf(stderr, "rank %d: plan %d, %d stream syncs\n", rank, plan, syncs); return 11; }
    CHECK(fs_advect(s, 0, FS_DENS, FS_BUFFER));                  // from outside step(): the synchronous reach
    double sum, mn, mx;
    CHECK(fs_field_stats(s, FS_VX, &sum, &mn, &mx));
    CHECK(fs_set_option(s, "solver", "mg"));
    CHECK(fs_set_option(s, "mg_min_planes", "8"));              // level 1 distributed, level 2 held whole
    CHECK(fs_run_one(s));
    CHECK(fs_add_obstacle(s, 11, 5, 15));
    CHECK(fs_run_one(s));
    const size_t n = fs_padded_size(s);
    std::vector<float> f(n);
    CHECK(fs_get_field(s, FS_PRESSURE, f.data(), n, 4));
    CHECK(fs_sync(s));
    CHECK(fs_destroy(s));
    printf("slab rank %d ok: schedule %d, velocity x in [%g, %g]\n", rank, plan, mn, mx);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && strcmp(argv[1], "--slabs") == 0) {
        // before any HIP call: two rank processes
        const char* ascii = argc > 2 ? argv[2] : "tests/golden/plate_ascii.stl";
        const char* dumpdir = argc > 3 ? argv[3] : "/tmp";
        char id[64];
        snprintf(id, sizeof id, "FSIPC:/fs_abi_exercise_%d", (int)getpid());
        pid_t kids[2];
        for (int r = 0; r < 2; ++r) {
            kids[r] = fork();
            if (kids[r] == 0) {
                // _exit, not return: in a forked child the HIP runtime's own exit handler trips a CHECK inside the ASan runtime
                // (sanitizer_allocator_device.h, "dev_runtime_unloaded_") after everything of ours has run clean; address
                // and UB findings are reported where they happen, not at exit (leak checking is off under HIP anyway)
                const int rc = slab_rank(r, id, ascii, dumpdir);
                fflush(stdout);
                fflush(stderr);
                _exit(rc);
            }
        }
        int bad = 0;
        for (int r = 0; r < 2; ++r) {
            int st = 0;
            waitpid(kids[r], &st, 0);
            if (!WIFEXITED(st) || WEXITSTATUS(st)) bad = 1;
        }
        printf(bad ? "slab leg FAILED\n" : "slab leg ok\n");
        return bad;
    }
    const char* stl = argc > 1 ? argv[1] : "tests/golden/sphere_24x12.stl";
    const char* ascii = argc > 2 ? argv[2] : "tests/golden/plate_ascii.stl";
    const char* dumpdir = argc > 3 ? argv[3] : "/tmp";
    for (int fp64 = 0; fp64 < 2; ++fp64) {
        fs_sim* s = fs_create(40, 24, 20, 3, 30, 0.05f, 2.0e-5f, 1.5e-5f, 5);
        if (!s) { fprintf(stderr, "fs_create: %s\n", fs_last_error()); return 1; }
        CHECK(fs_set_option(s, "precision", fp64 ? "fp64" : "fp32"));
        CHECK(fs_set_option(s, "quiet", "1"));
        CHECK(fs_set_option(s, "dump_dir", dumpdir));
        CHECK(fs_set_option(s, "profile", "1"));
        if (fs_set_option(s, "nonsense", "1") != FS_EINVAL) return 2;
        long added = 0;
        CHECK(fs_load_stl(s, stl, 0.5f, 10.f, 20.f, 30.f, -4.f, 0.f, 0.f, &added));
        CHECK(fs_load_stl(s, ascii, 0.6f, 0.f, 0.f, 0.f, 5.f, 0.f, 0.f, &added));
        if (fs_load_stl(s, "/nonexistent.stl", 1.f, 0, 0, 0, 0, 0, 0, nullptr) != FS_EIO) return 3;
        CHECK(fs_add_obstacle(s, 3, 3, 3));
        CHECK(fs_add_density(s, 4, 4, 4, 0.5f));
        CHECK(fs_set_velocity(s, 5, 5, 5, 1.f, 2.f, 3.f));
        if (fs_add_obstacle(s, 0, 1, 1) != FS_EINVAL) return 4;
        const size_t n = fs_padded_size(s);
        std::vector<float> f32(n);
        std::vector<double> f64(n);
        std::vector<uint8_t> mask(n, 0);
        CHECK(fs_get_field(s, FS_OBS, f32.data(), n, 4));
        for (size_t i = 0; i < n; ++i) mask[i] = f32[i] > 0.5f;
        CHECK(fs_set_obstacle_mask(s, mask.data(), n));
        CHECK(fs_run(s));
        CHECK(fs_step(s));
        CHECK(fs_run_one(s));
        CHECK(fs_diffuse(s, 1, FS_VX, FS_VX_PREV));
        CHECK(fs_project(s));
        CHECK(fs_advect(s, 2, FS_VY, FS_VY_PREV));
        CHECK(fs_set_bounds(s, 3, FS_VZ));
        CHECK(fs_linear_solver(s, 0, FS_PRESSURE, FS_DIVERGENCE, 1.0f, 6.0f));
        CHECK(fs_get_field(s, FS_VX, f64.data(), n, 8));
        CHECK(fs_set_field(s, FS_VX, f64.data(), n, 8));
        if (fs_get_field(s, FS_VX, f32.data(), n - 1, 4) != FS_EINVAL) return 5;
        double sum, mn, mx, ms;
        long launches;
        CHECK(fs_field_stats(s, FS_DENS, &sum, &mn, &mx));
        CHECK(fs_get_timing(s, "sweep_pair", &ms, &launches));
        CHECK(fs_time_sweeps(s, 2, FS_VY, FS_VY_PREV, 0.3f, 2.8f, 4, &ms));
        CHECK(fs_dump_frame(s));
        CHECK(fs_sync(s));
        int w;
        CHECK(fs_get_int(s, "width", &w));
        {   // the viewer's streamlines: compute, then fetch into caller memory
            long nl = 0, np = 0;
            CHECK(fs_streamlines(s, 30, 2.0, 100, 0.2, 0.0, &nl, &np));
            std::vector<long> off((size_t)nl + 1);
            std::vector<double> pts((size_t)np * 3 + 1), norm((size_t)nl + 1);
            CHECK(fs_streamlines_fetch(s, off.data(), pts.data(), norm.data()));
            if (off[(size_t)nl] != np) return 6;
        }
        {   // the viewer's obstacle mesh: extract, then fetch into caller memory
            long nv = 0, nt = 0;
            CHECK(fs_obstacle_surface(s, &nv, &nt));
            std::vector<float> verts((size_t)nv * 3 + 1);
            std::vector<int> tris((size_t)nt * 3 + 1);
            CHECK(fs_obstacle_surface_fetch(s, verts.data(), tris.data()));
            for (long i = 0; i < 3 * nt; ++i)
                if (tris[(size_t)i] < 0 || tris[(size_t)i] >= nv) return 7;
            int edges[24];
            if (fs_surface_case_table(1, edges) != 1 || fs_surface_case_table(256, edges) != FS_EINVAL) return 8;
        }
        CHECK(fs_set_option(s, "sweep_fuse", "4"));     // force the three-sweep kernel (fp32 only; fp64 keeps pairs)
        CHECK(fs_run_one(s));
        CHECK(fs_set_option(s, "sweep_fuse", "3"));
        CHECK(fs_set_option(s, "two_sweep_kernel", "fused"));   // jacobi_fused_kernel<NL=2> (fp64 here; fp32 needs rows > 512 cells)
        CHECK(fs_run_one(s));
        CHECK(fs_set_option(s, "two_sweep_kernel", "pair"));
        CHECK(fs_run_one(s));
        CHECK(fs_set_option(s, "two_sweep_kernel", "auto"));
        CHECK(fs_set_option(s, "advect_kernels", "row"));
        CHECK(fs_run_one(s));
        CHECK(fs_set_option(s, "advect_kernels", "cell"));
        CHECK(fs_set_option(s, "solver", "mg"));                // multigrid pressure solve: 40x24x20 halves twice
        CHECK(fs_set_option(s, "mg_cycles", "2"));
        CHECK(fs_run_one(s));
        CHECK(fs_add_obstacle(s, 7, 7, 7));                      // rebuilds the coarse operators
        CHECK(fs_run_one(s));
        CHECK(fs_linear_solver(s, 0, FS_PRESSURE, FS_DIVERGENCE, 1.0f, 6.0f));
        int mgl = 0;
        CHECK(fs_get_int(s, "mg_levels", &mgl));
        if (mgl != 3) return 9;
        if (fs_set_option(s, "mg_pre", "0") != FS_EINVAL) return 10;
        CHECK(fs_set_option(s, "solver", "gs_lex"));
        CHECK(fs_run_one(s));
        CHECK(fs_destroy(s));
        printf("%s ok: %zu cells, density sum %.6g, %ld sample points\n", fp64 ? "fp64" : "fp32", n, sum, added);
    }
    return 0;
}

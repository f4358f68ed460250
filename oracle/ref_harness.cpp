// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never shipped in the product path).
//
// A thin extern "C" shim over the *unmodified* reference sources, which are compiled
// where they lie under /root/reference by oracle/Makefile (outputs only in oracle/_ref/).
// It exists so that (1) oracle/make_golden.py can drive the real reference at
// OMP_NUM_THREADS=1 and record golden vectors, (2) the C restatement in oracle/cpu_ref.c
// can be validated bit-for-bit against the real thing, and (3) bench.py's cpu_baseline
// leg can time the real reference ("kind": "reference") on the GPU box's host cores.
//
// Nothing in here restates solver arithmetic: every number comes out of
// Simulation::step() / loadSTLIntoObstacles() of the reference (simulation.cpp:96-150,
// object_loader.cpp:270-452).  The only logic repeated from the reference is the three
// statement per-step prologue of Simulation::run() (simulation.cpp:65-71: inlet density,
// buffer = dens, step()), because run() itself loops `iter` times and opens files.
#include <vector>
#include <fstream>
#include <thread>
#include <functional>
#include <cstring>
#include <cstdint>

// The field arrays are private members (simulation.h:12-27).  The standard headers
// above are already included (their guards make the re-include a no-op), so the
// keyword swap only touches the reference's own class declaration.
#define private public
#include "simulation.h"
#undef private

extern "C" {

void* ref_create(int w, int h, int d, int iter, int speed, float dt, float diff, float visc, int acc)
{
    return new Simulation(w, h, d, iter, speed, dt, diff, visc, acc);
}

void ref_destroy(void* p) { delete static_cast<Simulation*>(p); }

long ref_size(void* p) { return static_cast<Simulation*>(p)->size; }

void ref_add_obstacle(void* p, int x, int y, int z) { static_cast<Simulation*>(p)->addObstacle(x, y, z); }
void ref_add_density(void* p, int x, int y, int z, float a) { static_cast<Simulation*>(p)->addDensity(x, y, z, a); }
void ref_set_velocity(void* p, int x, int y, int z, float ax, float ay, float az)
{
    static_cast<Simulation*>(p)->setVelocity(x, y, z, ax, ay, az);
}

static std::vector<float>* pick(Simulation* s, int which)
{
    switch (which) {
        case 0: return &s->dens;
        case 1: return &s->v_x;
        case 2: return &s->v_y;
        case 3: return &s->v_z;
        case 4: return &s->obs;
        case 5: return &s->pressure;
        case 6: return &s->divergence;
        case 7: return &s->v_x_prev;
        case 8: return &s->v_y_prev;
        case 9: return &s->v_z_prev;
        case 10: return &s->buffer;
        default: return nullptr;
    }
}

int ref_get_field(void* p, int which, float* dst)
{
    auto* v = pick(static_cast<Simulation*>(p), which);
    if (!v) return -1;
    std::memcpy(dst, v->data(), v->size() * sizeof(float));
    return 0;
}

int ref_set_field(void* p, int which, const float* src)
{
    auto* v = pick(static_cast<Simulation*>(p), which);
    if (!v) return -1;
    std::memcpy(v->data(), src, v->size() * sizeof(float));
    return 0;
}

// Simulation::step() alone (no inlet density, no buffer copy).
void ref_step_only(void* p) { static_cast<Simulation*>(p)->step(); }

// One iteration of the time loop of Simulation::run() (simulation.cpp:63-71).
void ref_run_one(void* p)
{
    Simulation* s = static_cast<Simulation*>(p);
    for (int j = 1; j <= s->height; ++j)
        for (int k = 1; k <= s->depth; ++k)
            s->addDensity(1, j, k, 0.001f);
    s->buffer = s->dens;
    s->step();
}

// The real run(): opens data/*.bin relative to the cwd and dumps every step.
void ref_run(void* p) { static_cast<Simulation*>(p)->run(); }

// The individual private passes, for per-kernel golden vectors.  They contain orphaned
// `omp for` constructs; called outside a parallel region they run on the calling thread.
void ref_diffuse(void* p, int b, int field, int prev)
{
    Simulation* s = static_cast<Simulation*>(p);
    s->diffuse(b, *pick(s, field), *pick(s, prev));
}
void ref_project(void* p)
{
    Simulation* s = static_cast<Simulation*>(p);
    s->project(s->v_x, s->v_y, s->v_z, s->pressure, s->divergence);
}
void ref_advect(void* p, int b, int field, int prev)
{
    Simulation* s = static_cast<Simulation*>(p);
    s->advect(b, *pick(s, field), *pick(s, prev));
}
void ref_set_bounds(void* p, int b, int field)
{
    Simulation* s = static_cast<Simulation*>(p);
    s->setBounds(b, *pick(s, field));
}
void ref_linear_solver(void* p, int b, int field, int prev, float a, float c)
{
    Simulation* s = static_cast<Simulation*>(p);
    s->linearSolver(b, *pick(s, field), *pick(s, prev), a, c);
}

void ref_load_stl(void* p, const char* path, float scale, float rx, float ry, float rz,
                  float tx, float ty, float tz)
{
    loadSTLIntoObstacles(path, *static_cast<Simulation*>(p), scale, rx, ry, rz, tx, ty, tz);
}

// The value object_loader.cpp:399 seeds its per-thread minstd_rand with, evaluated on the
// calling thread (which is OpenMP thread 0 of the loader's parallel region when
// OMP_NUM_THREADS=1).  Recorded next to every golden mask.
unsigned ref_thread_seed(void)
{
    return static_cast<unsigned int>(std::hash<std::thread::id>{}(std::this_thread::get_id()));
}

}  // extern "C"

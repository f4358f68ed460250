/* oracle/cpu_ref.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the reference wind-tunnel solver and STL voxelizer
 * (Ghundi/fluid_simulation: simulation.cpp, object_loader.cpp).  It is the *checker*
 * the HIP path is compared against; it is never linked into, called from, or used as a
 * fallback by the product library (fluid_simulation_amd/csrc -> libfluidsim.so).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors of its own
 * (SURVEY.md section 4), so the pin is the reference itself: oracle/Makefile compiles the
 * unmodified reference sources into oracle/_ref/libref.so, oracle/make_golden.py runs it
 * at OMP_NUM_THREADS=1 and commits its outputs under tests/golden/, and
 * tests/test_oracle_golden.py requires this file (solver mode CR_GS_LEX) to reproduce
 * them bit for bit.  tests/test_oracle_vs_ref.py repeats that live against libref.so
 * whenever it is present.
 *
 * Solver modes (two more, CR_RBSOR and CR_MG, are the build's own optional solvers and have no
 * counterpart in the reference; see relax() and cpu_ref_mg.h; under CR_MG only the projection's
 * pressure solve changes, every other solve is CR_JACOBI):
 *   CR_GS_LEX  the reference's in-place sweep in its own loop order (x outer, y, z inner;
 *              simulation.cpp:258-270), i.e. the reference at one thread.
 *   CR_JACOBI  the same update reading all six neighbours from the previous iterate
 *              (north-star solver; the only change is which buffer neighbours come from).
 *
 * Build with -ffp-contract=off; every expression keeps the reference's association
 * order so results are bit-identical with `c++ -O2` on baseline x86-64 (no FMA).
 * Compile with -DCR_REAL=double for the fp64 field variant (BASELINE config 5): field
 * storage and all arithmetic (derived scalars included) widen to double, starting from
 * the same float-valued parameters.  The reference has no fp64 mode; that variant is
 * defined here and is compared GPU-vs-oracle only.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef CR_REAL
#define CR_REAL float
#endif
typedef CR_REAL real;
/* cube root in the field precision: cbrtf for float (std::cbrt(float), simulation.cpp:295) */
#define CR_CBRT(v) (sizeof(real) == sizeof(float) ? (real)cbrtf((float)(v)) : (real)cbrt((double)(v)))

enum { CR_GS_LEX = 0, CR_JACOBI = 1, CR_RBSOR = 2, CR_MG = 3 };
enum { CR_DENS = 0, CR_VX, CR_VY, CR_VZ, CR_OBS, CR_P, CR_DIV, CR_VX0, CR_VY0, CR_VZ0, CR_BUF, CR_NFIELDS };

typedef struct cr_sim {
    int W, H, D;              /* interior extents (simulation.h:44) */
    int steps, speed, sweeps; /* iter, speed, acc */
    float dt, diff, visc;
    int solver;
    float omega;              /* CR_RBSOR only */
    int mg_cycles, mg_pre, mg_post, mg_coarse;   /* CR_MG only: V-cycles per pressure solve, smoothing steps, coarsest-level iterations */
    size_t n;                 /* padded cell count (simulation.cpp:35) */
    size_t sy, sz;            /* strides: idx = x + y*sy + z*sz (simulation.h:9) */
    real* f[CR_NFIELDS];
    real* scratch;            /* Jacobi next-iterate */
} cr_sim;

#define AT(s, x, y, z) ((size_t)(x) + (size_t)(y) * (s)->sy + (size_t)(z) * (s)->sz)

/* ------------------------------------------------------------------ lifecycle */

cr_sim* cr_create(int w, int h, int d, int iter, int speed, float dt, float diff, float visc, int acc)
{
    cr_sim* s = (cr_sim*)calloc(1, sizeof(cr_sim));
    if (!s) return NULL;
    s->W = w; s->H = h; s->D = d;
    s->steps = iter; s->speed = speed; s->sweeps = acc;
    s->dt = dt; s->diff = diff; s->visc = visc;
    s->solver = CR_GS_LEX;
    s->omega = 1.0f;
    s->mg_cycles = 4; s->mg_pre = 1; s->mg_post = 1; s->mg_coarse = 30;
    s->sy = (size_t)w + 2;
    s->sz = s->sy * ((size_t)h + 2);
    s->n = s->sz * ((size_t)d + 2);
    for (int k = 0; k < CR_NFIELDS; ++k) {
        s->f[k] = (real*)calloc(s->n, sizeof(real));   /* zero-filled: simulation.cpp:38-43 */
        if (!s->f[k]) return NULL;
    }
    s->scratch = (real*)calloc(s->n, sizeof(real));
    return s;
}

void cr_destroy(cr_sim* s)
{
    if (!s) return;
    for (int k = 0; k < CR_NFIELDS; ++k) free(s->f[k]);
    free(s->scratch);
    free(s);
}

void cr_set_solver(cr_sim* s, int mode) { s->solver = mode; }
void cr_set_omega(cr_sim* s, float omega) { s->omega = omega; }
void cr_set_mg(cr_sim* s, int cycles, int pre, int post, int coarse) { s->mg_cycles = cycles; s->mg_pre = pre; s->mg_post = post; s->mg_coarse = coarse; }
int cr_real_bytes(void) { return (int)sizeof(real); }
long cr_size(cr_sim* s) { return (long)s->n; }

void cr_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int cr_get_field(cr_sim* s, int which, real* dst)
{
    if (which < 0 || which >= CR_NFIELDS) return -1;
    memcpy(dst, s->f[which], s->n * sizeof(real));
    return 0;
}

int cr_set_field(cr_sim* s, int which, const real* src)
{
    if (which < 0 || which >= CR_NFIELDS) return -1;
    memcpy(s->f[which], src, s->n * sizeof(real));
    return 0;
}

/* simulation.cpp:155-178 -- 1-based interior coordinates, unchecked in the reference */
void cr_add_obstacle(cr_sim* s, int x, int y, int z) { s->f[CR_OBS][AT(s, x, y, z)] = (real)1; }
void cr_add_density(cr_sim* s, int x, int y, int z, float amount) { s->f[CR_DENS][AT(s, x, y, z)] += (real)amount; }
void cr_set_velocity(cr_sim* s, int x, int y, int z, float ax, float ay, float az)
{
    size_t c = AT(s, x, y, z);
    s->f[CR_VX][c] = (real)ax; s->f[CR_VY][c] = (real)ay; s->f[CR_VZ][c] = (real)az;
}

/* ------------------------------------------------------------------ boundaries
 * simulation.cpp:183-246.  Order matters: the three face passes read the interior
 * *before* the two zeroing passes touch it.  Ghost edges and corners are never written. */
static void enforce_bounds(cr_sim* s, int b, real* q)
{
    const int W = s->W, H = s->H, D = s->D;
    const real* solid = s->f[CR_OBS];

    for (int y = 1; y <= H; ++y)                       /* :187-192 */
        for (int z = 1; z <= D; ++z) {
            real inner = q[AT(s, 1, y, z)];
            q[AT(s, 0, y, z)] = (b == 1) ? -inner : inner;
            q[AT(s, W + 1, y, z)] = q[AT(s, W, y, z)];
        }
    for (int x = 1; x <= W; ++x)                       /* :196-202 */
        for (int z = 1; z <= D; ++z) {
            real lo = q[AT(s, x, 1, z)], hi = q[AT(s, x, H, z)];
            q[AT(s, x, 0, z)] = (b == 2) ? -lo : lo;
            q[AT(s, x, H + 1, z)] = (b == 2) ? -hi : hi;
        }
    for (int x = 1; x <= W; ++x)                       /* :206-215 */
        for (int y = 1; y <= H; ++y) {
            real lo = q[AT(s, x, y, 1)], hi = q[AT(s, x, y, D)];
            q[AT(s, x, y, 0)] = (b == 3) ? -lo : lo;
            q[AT(s, x, y, D + 1)] = (b == 3) ? -hi : hi;
        }

#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 1; z <= D; ++z)                       /* :219-223 */
        for (int y = 1; y <= H; ++y)
            for (int x = 1; x <= W; ++x)
                if (solid[AT(s, x, y, z)] == (real)1) q[AT(s, x, y, z)] = (real)0;

    if (b < 1 || b > 3) return;                        /* :240 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 1; z <= D; ++z)                       /* :227-245 */
        for (int y = 1; y <= H; ++y)
            for (int x = 1; x <= W; ++x) {
                size_t c = AT(s, x, y, z);
                if (solid[c] == (real)1) continue;
                int touch = (x + 1 <= W && solid[c + 1] == (real)1) || (x - 1 >= 1 && solid[c - 1] == (real)1) ||
                            (y + 1 <= H && solid[c + s->sy] == (real)1) || (y - 1 >= 1 && solid[c - s->sy] == (real)1) ||
                            (z + 1 <= D && solid[c + s->sz] == (real)1) || (z - 1 >= 1 && solid[c - s->sz] == (real)1);
                if (touch) q[c] = (real)0;
            }
}

/* NOT in the reference: the build's optional red-black iteration (solver mode CR_RBSOR, SURVEY.md 8f
 * rank 4; also the level-0 smoother of CR_MG), defined here.  One iteration = the cells with even
 * x+y+z, then those with odd x+y+z, each half followed by the reference's setBounds; a cell moves
 * from its value q towards the reference's update r by q + omega*(r - q).  Cells of one colour do not
 * neighbour each other, so the in-place loop has no ordering freedom. */
static void rb_iterations(cr_sim* s, int b, real* q, const real* rhs, real a, real inv_c, real om, int n)
{
    const int W = s->W, H = s->H, D = s->D;
    const size_t sy = s->sy, sz = s->sz;
    for (int it = 0; it < n; ++it)
        for (int colour = 0; colour < 2; ++colour) {
#pragma omp parallel for collapse(2) schedule(static)
            for (int z = 1; z <= D; ++z)
                for (int y = 1; y <= H; ++y)
                    for (int x = 1; x <= W; ++x) {
                        if (((x + y + z) & 1) != colour) continue;
                        size_t c = AT(s, x, y, z);
                        real nb = q[c + 1] + q[c - 1] + q[c + sy] + q[c - sy] + q[c + sz] + q[c - sz];
                        real r = (rhs[c] + a * nb) * inv_c;
                        q[c] = q[c] + om * (r - q[c]);
                    }
            enforce_bounds(s, b, q);
        }
}

/* ------------------------------------------------------------------ linear solver
 * simulation.cpp:251-273.  Every interior cell is updated, solids included; the sum is
 * taken in the order x+1, x-1, y+1, y-1, z+1, z-1 and scaled by a reciprocal. */
static void relax(cr_sim* s, int b, real* q, const real* rhs, real a, real c)
{
    const int W = s->W, H = s->H, D = s->D;
    const size_t sy = s->sy, sz = s->sz;
    const real inv_c = (real)1 / c;                    /* :257 */

    if (s->solver == CR_RBSOR) {
        rb_iterations(s, b, q, rhs, a, inv_c, (real)s->omega, s->sweeps);
        return;
    }
    for (int it = 0; it < s->sweeps; ++it) {
        if (s->solver == CR_GS_LEX) {
            /* reference traversal: x outermost, z innermost, updated in place (:260-262).
             * With >1 thread this is the same chunked race the reference has. */
#pragma omp parallel for collapse(3) schedule(static)
            for (int x = 1; x <= W; ++x)
                for (int y = 1; y <= H; ++y)
                    for (int z = 1; z <= D; ++z) {
                        size_t c = AT(s, x, y, z);
                        real nb = q[c + 1] + q[c - 1] + q[c + sy] + q[c - sy] + q[c + sz] + q[c - sz];
                        q[c] = (rhs[c] + a * nb) * inv_c;
                    }
        } else {
            real* nxt = s->scratch;
#pragma omp parallel for collapse(2) schedule(static)
            for (int z = 1; z <= D; ++z)
                for (int y = 1; y <= H; ++y)
                    for (int x = 1; x <= W; ++x) {
                        size_t c = AT(s, x, y, z);
                        real nb = q[c + 1] + q[c - 1] + q[c + sy] + q[c - sy] + q[c + sz] + q[c - sz];
                        nxt[c] = (rhs[c] + a * nb) * inv_c;
                    }
#pragma omp parallel for collapse(2) schedule(static)
            for (int z = 1; z <= D; ++z)
                for (int y = 1; y <= H; ++y)
                    memcpy(&q[AT(s, 1, y, z)], &nxt[AT(s, 1, y, z)], (size_t)W * sizeof(real));
        }
        enforce_bounds(s, b, q);                       /* :271 */
    }
}

#include "cpu_ref_mg.h"

/* simulation.cpp:278-284: a = dt*diff*W*H*D evaluated left to right in float */
static void spread(cr_sim* s, int b, real* q, const real* rhs)
{
    real a = (real)s->dt * (real)s->diff * s->W * s->H * s->D;
    relax(s, b, q, rhs, a, (real)1 + (real)6 * a);
}

/* ------------------------------------------------------------------ projection
 * simulation.cpp:289-362 */
static void make_solenoidal(cr_sim* s)
{
    const int W = s->W, H = s->H, D = s->D;
    const size_t sy = s->sy, sz = s->sz;
    real *vx = s->f[CR_VX], *vy = s->f[CR_VY], *vz = s->f[CR_VZ];
    real *p = s->f[CR_P], *dv = s->f[CR_DIV];
    const real* solid = s->f[CR_OBS];
    const real h = (real)1 / CR_CBRT((real)(W * H * D));  /* :295 */
    const real mhalf_h = (real)(-0.5f) * h;               /* -0.5f*h*div -> (-0.5f*h)*div */

#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 1; z <= D; ++z)
        for (int y = 1; y <= H; ++y)
            for (int x = 1; x <= W; ++x) {
                size_t c = AT(s, x, y, z);
                p[c] = (real)0;
                if (solid[c] == (real)1) { dv[c] = (real)0; continue; }
                real acc = (real)0;                        /* :306-312, one-sided near solids/walls */
                if (x + 1 <= W && solid[c + 1] == (real)0) acc += vx[c + 1];
                if (x - 1 >= 1 && solid[c - 1] == (real)0) acc -= vx[c - 1];
                if (y + 1 <= H && solid[c + sy] == (real)0) acc += vy[c + sy];
                if (y - 1 >= 1 && solid[c - sy] == (real)0) acc -= vy[c - sy];
                if (z + 1 <= D && solid[c + sz] == (real)0) acc += vz[c + sz];
                if (z - 1 >= 1 && solid[c - sz] == (real)0) acc -= vz[c - sz];
                dv[c] = mhalf_h * acc;
            }

    enforce_bounds(s, 0, dv);
    enforce_bounds(s, 0, p);
    if (s->solver == CR_MG) mg_solve(s, p, dv);        /* the build's own mode, cpu_ref_mg.h */
    else relax(s, 0, p, dv, (real)1, (real)6);         /* :320 */

    const real two_h = (real)2 * h;
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 1; z <= D; ++z)
        for (int y = 1; y <= H; ++y)
            for (int x = 1; x <= W; ++x) {
                size_t c = AT(s, x, y, z);
                if (solid[c] == (real)1) continue;
                int xp = (x + 1 <= W && solid[c + 1] == (real)0), xm = (x - 1 >= 1 && solid[c - 1] == (real)0);
                int yp = (y + 1 <= H && solid[c + sy] == (real)0), ym = (y - 1 >= 1 && solid[c - sy] == (real)0);
                int zp = (z + 1 <= D && solid[c + sz] == (real)0), zm = (z - 1 >= 1 && solid[c - sz] == (real)0);
                real g;
                g = (real)0;                               /* :328-336 */
                if (xp && xm) g = (p[c + 1] - p[c - 1]) / two_h;
                else if (xp)  g = (p[c + 1] - p[c]) / h;
                else if (xm)  g = (p[c] - p[c - 1]) / h;
                vx[c] -= g;
                g = (real)0;                               /* :338-346 */
                if (yp && ym) g = (p[c + sy] - p[c - sy]) / two_h;
                else if (yp)  g = (p[c + sy] - p[c]) / h;
                else if (ym)  g = (p[c] - p[c - sy]) / h;
                vy[c] -= g;
                g = (real)0;                               /* :348-356 */
                if (zp && zm) g = (p[c + sz] - p[c - sz]) / two_h;
                else if (zp)  g = (p[c + sz] - p[c]) / h;
                else if (zm)  g = (p[c] - p[c - sz]) / h;
                vz[c] -= g;
            }

    enforce_bounds(s, 1, vx);
    enforce_bounds(s, 2, vy);
    enforce_bounds(s, 3, vz);
}

/* ------------------------------------------------------------------ advection
 * simulation.cpp:367-424.  The carrying velocity is read at the destination cell:
 * component b from `src`, the others from the *current* velocity arrays (:380-382). */
static real clamp_to(real v, real lo, real hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

static void transport(cr_sim* s, int b, real* q, const real* src)
{
    const int W = s->W, H = s->H, D = s->D;
    const size_t sy = s->sy, sz = s->sz;
    const real *vx = s->f[CR_VX], *vy = s->f[CR_VY], *vz = s->f[CR_VZ];
    const real* solid = s->f[CR_OBS];
    const real kx = (real)s->dt * (real)W, ky = (real)s->dt * (real)H, kz = (real)s->dt * (real)D;
    const real one = (real)1, half = (real)0.5;

#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 1; z <= D; ++z)
        for (int y = 1; y <= H; ++y)
            for (int x = 1; x <= W; ++x) {
                size_t c = AT(s, x, y, z);
                if (solid[c] == one) { q[c] = (real)0; continue; }
                real ux = (b == 1) ? src[c] : vx[c];
                real uy = (b == 2) ? src[c] : vy[c];
                real uz = (b == 3) ? src[c] : vz[c];
                real px = clamp_to((real)x - kx * ux, half, (real)W + half);   /* :384-390 */
                real py = clamp_to((real)y - ky * uy, half, (real)H + half);
                real pz = clamp_to((real)z - kz * uz, half, (real)D + half);
                int x0 = (int)floor((double)px), y0 = (int)floor((double)py), z0 = (int)floor((double)pz);
                real tx = px - (real)x0, ty = py - (real)y0, tz = pz - (real)z0;
                size_t o = AT(s, x0, y0, z0);
                real a00 = src[o] * (one - tx) + src[o + 1] * tx;               /* :412-415 */
                real a01 = src[o + sz] * (one - tx) + src[o + sz + 1] * tx;
                real a10 = src[o + sy] * (one - tx) + src[o + sy + 1] * tx;
                real a11 = src[o + sy + sz] * (one - tx) + src[o + sy + sz + 1] * tx;
                real b0 = a00 * (one - ty) + a10 * ty;                          /* :417-418 */
                real b1 = a01 * (one - ty) + a11 * ty;
                q[c] = b0 * (one - tz) + b1 * tz;                               /* :420 */
            }
    enforce_bounds(s, b, q);
}

/* ------------------------------------------------------------------ time step
 * simulation.cpp:96-150 (minus the frame dump) */
void cr_step_only(cr_sim* s)
{
    for (int y = 1; y <= s->H; ++y)                    /* inlet, :103-105 */
        for (int z = 1; z <= s->D; ++z)
            cr_set_velocity(s, 1, y, z, (float)s->speed, 0.0f, 0.0f);

    memcpy(s->f[CR_VX0], s->f[CR_VX], s->n * sizeof(real));   /* pre-diffusion snapshots, :108-110 */
    memcpy(s->f[CR_VY0], s->f[CR_VY], s->n * sizeof(real));
    memcpy(s->f[CR_VZ0], s->f[CR_VZ], s->n * sizeof(real));

    spread(s, 1, s->f[CR_VX], s->f[CR_VX0]);           /* velocity diffuses with `diff`; visc is dead */
    spread(s, 2, s->f[CR_VY], s->f[CR_VY0]);
    spread(s, 3, s->f[CR_VZ], s->f[CR_VZ0]);
    make_solenoidal(s);
    transport(s, 1, s->f[CR_VX], s->f[CR_VX0]);
    transport(s, 2, s->f[CR_VY], s->f[CR_VY0]);
    transport(s, 3, s->f[CR_VZ], s->f[CR_VZ0]);
    make_solenoidal(s);
    spread(s, 0, s->f[CR_DENS], s->f[CR_BUF]);         /* overwritten by the next line (:135-136) */
    transport(s, 0, s->f[CR_DENS], s->f[CR_BUF]);
}

/* one iteration of the loop in Simulation::run(), simulation.cpp:63-71 */
void cr_run_one(cr_sim* s)
{
    for (int y = 1; y <= s->H; ++y)
        for (int z = 1; z <= s->D; ++z)
            cr_add_density(s, 1, y, z, 0.001f);
    memcpy(s->f[CR_BUF], s->f[CR_DENS], s->n * sizeof(real));
    cr_step_only(s);
}

/* individual passes, for per-kernel parity tests */
void cr_set_bounds(cr_sim* s, int b, int field) { enforce_bounds(s, b, s->f[field]); }
void cr_linear_solver(cr_sim* s, int b, int field, int prev, float a, float c)
{
    if (s->solver == CR_MG && b == 0 && a == 1.0f && c == 6.0f) mg_solve(s, s->f[field], s->f[prev]);   /* the pressure equation's coefficients */
    else relax(s, b, s->f[field], s->f[prev], (real)a, (real)c);
}
void cr_diffuse(cr_sim* s, int b, int field, int prev) { spread(s, b, s->f[field], s->f[prev]); }
void cr_project(cr_sim* s) { make_solenoidal(s); }
void cr_advect(cr_sim* s, int b, int field, int prev) { transport(s, b, s->f[field], s->f[prev]); }

/* Append one frame in the reference's dump layout (simulation.cpp:140-148): five raw
 * padded arrays, always float32 on disk.  `dir` must exist. */
int cr_dump_frame(cr_sim* s, const char* dir, int append)
{
    static const char* names[5] = { "data.bin", "obs.bin", "v_x.bin", "v_y.bin", "v_z.bin" };
    static const int which[5] = { CR_DENS, CR_OBS, CR_VX, CR_VY, CR_VZ };
    char path[4096];
    float* tmp = (float*)malloc(s->n * sizeof(float));
    if (!tmp) return -1;
    for (int k = 0; k < 5; ++k) {
        snprintf(path, sizeof path, "%s/%s", dir, names[k]);
        FILE* fp = fopen(path, append ? "ab" : "wb");
        if (!fp) { free(tmp); return -2; }
        for (size_t i = 0; i < s->n; ++i) tmp[i] = (float)s->f[which[k]][i];
        fwrite(tmp, sizeof(float), s->n, fp);
        fclose(fp);
    }
    free(tmp);
    return 0;
}

/* ================================================================== voxelizer
 * object_loader.cpp:98-452, single-thread semantics.  The reference seeds its RNG from a
 * hash of the thread id (:399), which varies run to run; here the seed is an argument,
 * and goldens record the seed the reference actually used. */

typedef struct { float x, y, z; } vec3;
typedef struct { vec3 a, b, c; } tri_t;

static char* trimmed(char* s)
{
    while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') ++s;
    size_t n = strlen(s);
    while (n && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\n' || s[n - 1] == '\r')) s[--n] = 0;
    return s;
}

/* object_loader.cpp:98-174.  Returns triangle count (0 on failure), array in *out. */
static size_t stl_read(const char* path, tri_t** out)
{
    *out = NULL;
    FILE* fp = fopen(path, "rb");
    if (!fp) return 0;
    char line[1024];
    if (!fgets(line, sizeof line, fp)) line[0] = 0;
    int ascii = (strncmp(trimmed(line), "solid", 5) == 0);     /* :105-107 */
    size_t n = 0, cap = 0;
    tri_t* t = NULL;
    if (!ascii) {
        uint32_t cnt = 0;
        fseek(fp, 80, SEEK_SET);
        if (fread(&cnt, 4, 1, fp) != 1) { fclose(fp); return 0; }
        t = (tri_t*)malloc((size_t)cnt * sizeof(tri_t) + 1);
        for (uint32_t i = 0; i < cnt; ++i) {
            float rec[12]; uint16_t attr;
            size_t got = fread(rec, 4, 12, fp);
            got += fread(&attr, 2, 1, fp);
            /* a short read leaves the reference pushing whatever it has; clean files only */
            (void)got;
            t[n].a = (vec3){ rec[3], rec[4], rec[5] };
            t[n].b = (vec3){ rec[6], rec[7], rec[8] };
            t[n].c = (vec3){ rec[9], rec[10], rec[11] };
            ++n;
        }
    } else {
        rewind(fp);
        tri_t cur; memset(&cur, 0, sizeof cur);
        int vi = 0;
        while (fgets(line, sizeof line, fp)) {
            char* l = trimmed(line);
            if (strcmp(l, "outer loop") == 0) { vi = 0; continue; }
            if (strcmp(l, "endloop") == 0) continue;
            if (strcmp(l, "endfacet") == 0) {
                if (vi == 3) {
                    if (n == cap) { cap = cap ? cap * 2 : 1024; t = (tri_t*)realloc(t, cap * sizeof(tri_t)); }
                    t[n++] = cur;
                }
                continue;
            }
            if (strncmp(l, "vertex", 6) == 0) {
                float x, y, z;
                if (sscanf(l + 6, "%f %f %f", &x, &y, &z) == 3) {
                    vec3 v = { x, y, z };
                    if (vi == 0) cur.a = v; else if (vi == 1) cur.b = v; else if (vi == 2) cur.c = v;
                    vi = (vi + 1) % 4;                          /* :166 */
                }
            }
        }
    }
    fclose(fp);
    *out = t;
    return n;
}

/* object_loader.cpp:177-202: R = Rx*Ry*Rz with float trig */
static vec3 spin(vec3 p, float dx, float dy, float dz)
{
    float rx = dx * M_PI / 180.0f, ry = dy * M_PI / 180.0f, rz = dz * M_PI / 180.0f;   /* float*double/float -> float */
    float cx = cosf(rx), sx = sinf(rx), cy = cosf(ry), sy = sinf(ry), cz = cosf(rz), sz = sinf(rz);
    vec3 o;
    o.x = (cy * cz) * p.x + (-cy * sz) * p.y + (sy) * p.z;
    o.y = (sx * sy * cz + cx * sz) * p.x + (-sx * sy * sz + cx * cz) * p.y + (-sx * cy) * p.z;
    o.z = (-cx * sy * cz + sx * sz) * p.x + (cx * sy * sz + sx * cz) * p.y + (cx * cy) * p.z;
    return o;
}

/* object_loader.cpp:205-233 */
static int ray_hits(vec3 o, vec3 d, const tri_t* t)
{
    const float EPS = 1e-6f;
    vec3 e1 = { t->b.x - t->a.x, t->b.y - t->a.y, t->b.z - t->a.z };
    vec3 e2 = { t->c.x - t->a.x, t->c.y - t->a.y, t->c.z - t->a.z };
    vec3 h = { d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x };
    float det = e1.x * h.x + e1.y * h.y + e1.z * h.z;
    if (fabsf(det) < EPS) return 0;
    float f = 1.0f / det;
    vec3 sv = { o.x - t->a.x, o.y - t->a.y, o.z - t->a.z };
    float u = f * (sv.x * h.x + sv.y * h.y + sv.z * h.z);
    if (u < 0.0f || u > 1.0f) return 0;
    vec3 q = { sv.y * e1.z - sv.z * e1.y, sv.z * e1.x - sv.x * e1.z, sv.x * e1.y - sv.y * e1.x };
    float v = f * (d.x * q.x + d.y * q.y + d.z * q.z);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float tt = f * (e2.x * q.x + e2.y * q.y + e2.z * q.z);
    return tt > 1e-3f;
}

/* std::minstd_rand: x <- 48271*x mod (2^31-1); seed 0 maps to 1 */
static uint32_t lcg_seed(unsigned seed) { uint32_t v = seed % 2147483647u; return v ? v : 1u; }
static uint32_t lcg_next(uint32_t* st) { *st = (uint32_t)(((uint64_t)*st * 48271u) % 2147483647u); return *st; }

/* libstdc++ uniform_real_distribution<float>(0.1f,1.0f) over minstd_rand: one draw,
 * generate_canonical<float,24> = float(x-1)/float(2^31-2 -> 2^31), clamped below 1. */
static float lcg_unit(uint32_t* st)
{
    float r = (float)(uint64_t)(lcg_next(st) - 1u) / 2147483648.0f;
    if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
    return r * (1.0f - 0.1f) + 0.1f;
}

static float max3f(float a, float b, float c) { float m = a > b ? a : b; return m > c ? m : c; }
static float min3f(float a, float b, float c) { float m = a < b ? a : b; return m < c ? m : c; }

/* Returns number of accepted sample points ("Added N obstacle points"), or -1 if the
 * STL could not be read (the reference then leaves the tunnel empty, :282-285). */
long cr_load_stl(cr_sim* s, const char* path, float scale, float rot_x, float rot_y, float rot_z,
                 float tr_x, float tr_y, float tr_z, unsigned seed)
{
    tri_t* raw = NULL;
    size_t nt = stl_read(path, &raw);
    if (nt == 0) { free(raw); return -1; }

    /* object centre is always the origin (orig_min/max are never updated, :288-296):
     * (1e6f + -1e6f) * 0.5f = 0 */
    tri_t* rt = (tri_t*)malloc(nt * sizeof(tri_t));
    float r2 = 0.0f;
    for (size_t i = 0; i < nt; ++i) {
        rt[i].a = spin(raw[i].a, rot_x, rot_y, rot_z);
        rt[i].b = spin(raw[i].b, rot_x, rot_y, rot_z);
        rt[i].c = spin(raw[i].c, rot_x, rot_y, rot_z);
        /* radius from the UNROTATED vertices (:328-333) */
        const vec3* v[3] = { &raw[i].a, &raw[i].b, &raw[i].c };
        for (int k = 0; k < 3; ++k) {
            float d2 = v[k]->x * v[k]->x + v[k]->y * v[k]->y + v[k]->z * v[k]->z;
            if (d2 > r2) r2 = d2;
        }
    }
    float radius = sqrtf(r2);
    float pad = radius * 0.05f;
    float lo = (0.0f - radius) - pad, hi = (0.0f + radius) + pad;   /* same on all three axes */
    float span = hi - lo;                                          /* objSize, :362-366 */
    float res = span / 200.0f; if (res < 0.02f) res = 0.02f;        /* :368 */
    int ns = (int)(span / res);                                    /* nx = ny = nz, :370-372 */

    /* coarse occupancy grid, :380-389 and :54-77 */
    const int G = 64;
    const float cell = res * 5.0f;
    unsigned char* occ = (unsigned char*)calloc((size_t)G * G * G, 1);
    for (size_t i = 0; i < nt; ++i) {
        const tri_t* t = &rt[i];
        float mnx = min3f(t->a.x, t->b.x, t->c.x), mxx = max3f(t->a.x, t->b.x, t->c.x);
        float mny = min3f(t->a.y, t->b.y, t->c.y), mxy = max3f(t->a.y, t->b.y, t->c.y);
        float mnz = min3f(t->a.z, t->b.z, t->c.z), mxz = max3f(t->a.z, t->b.z, t->c.z);
        int x0 = (int)((mnx - lo) / cell), x1 = (int)((mxx - lo) / cell);
        int y0 = (int)((mny - lo) / cell), y1 = (int)((mxy - lo) / cell);
        int z0 = (int)((mnz - lo) / cell), z1 = (int)((mxz - lo) / cell);
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (z0 < 0) z0 = 0;
        if (x1 > G - 1) x1 = G - 1; if (y1 > G - 1) y1 = G - 1; if (z1 > G - 1) z1 = G - 1;
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x) occ[x + y * G + z * G * G] = 1;
    }

    const int W = s->W, H = s->H, D = s->D;
    float gmin = (float)W; if ((float)H < gmin) gmin = (float)H; if ((float)D < gmin) gmin = (float)D;
    const float gscale = scale * gmin / span;                      /* :429 */
    const float cx = (float)W / 2, cy = (float)H / 2, cz = (float)D / 2;

    uint32_t st = lcg_seed(seed);
    long added = 0;
    for (int i = 0; i < ns; ++i)
        for (int j = 0; j < ns; ++j)
            for (int k = 0; k < ns; ++k) {
                vec3 p = { lo + i * res, lo + j * res, lo + k * res };   /* :407-409 */
                /* :79-87 (p >= lo always holds here except through rounding) */
                if (p.x < lo || p.y < lo || p.z < lo) continue;
                int gx = (int)((p.x - lo) / cell), gy = (int)((p.y - lo) / cell), gz = (int)((p.z - lo) / cell);
                if (gx < 0 || gx >= G || gy < 0 || gy >= G || gz < 0 || gz >= G) continue;
                if (!occ[gx + gy * G + gz * G * G]) continue;
                p.x += (float)(uint64_t)(lcg_next(&st) % 1000u) * 1e-6f - 5e-4f;   /* :417-419 */
                p.y += (float)(uint64_t)(lcg_next(&st) % 1000u) * 1e-6f - 5e-4f;
                p.z += (float)(uint64_t)(lcg_next(&st) % 1000u) * 1e-6f - 5e-4f;
                vec3 dir; dir.x = lcg_unit(&st); dir.y = lcg_unit(&st); dir.z = lcg_unit(&st);   /* :422 */
                int crossings = 0;
                for (size_t t = 0; t < nt; ++t) crossings += ray_hits(p, dir, &rt[t]);
                if ((crossings & 1) == 0) continue;
                int sx = (int)((p.x - 0.0f) * gscale + cx + tr_x);   /* :432-434, C truncation */
                int sy = (int)((p.y - 0.0f) * gscale + cy + tr_y);
                int sz = (int)((p.z - 0.0f) * gscale + cz + tr_z);
                if (sx >= 1 && sx <= W && sy >= 1 && sy <= H && sz >= 1 && sz <= D) {
                    cr_add_obstacle(s, sx, sy, sz);
                    ++added;
                }
            }
    free(occ); free(rt); free(raw);
    return added;
}

/* oracle/cpu_ref_mg.h -- TEST INFRASTRUCTURE ONLY; included by cpu_ref.c.
 *
 * CR_MG: the build's optional multigrid pressure solver.  NOT in the reference (which only ever
 * relaxes: simulation.cpp:251-273, 40-80 in-place sweeps that leave the pressure equation far from
 * solved at these grid sizes); it has no counterpart there and no golden vectors: this file DEFINES
 * the mode, and the HIP implementation (fluid_simulation_amd/csrc/multigrid.hip) is compared with it
 * bit for bit.  Parity of this mode with the reference is therefore not a question; what the tests
 * check besides GPU == oracle is that it solves the reference's own equation (residual of
 * simulation.cpp:263-269's fixed point).
 *
 * The equation (fixed point of linearSolver(0, p, div, 1, 6) + setBounds(0, p), simulation.cpp:320):
 *   fluid cell c:  6 p_c - sum of its six neighbours = div_c,  where a solid neighbour counts 0
 *   (setBounds zeroes solids, :219-223) and a wall ghost mirrors p_c (:187-215).
 * V-cycles on cell-centred 2x2x2 coarsening.  Level 0 is the simulation grid and keeps the
 * reference's arithmetic (the update of :263-269 as damped Jacobi sweeps, then setBounds; mg_smooth0).
 * Coarser levels (red-black Gauss-Seidel) carry per-face weights w and a Dirichlet term d:
 *   (sum_f w_f + d_C) e_C - sum_f w_f e_nbr(f) = b_C
 * with  w(coarse face) = 1/4 * sum of the four fine face weights it covers  (fine level 0: 1 where
 * both cells are fluid, else 0; wall faces 0) and  d_C = 1/2 * sum of its children's d  (level 0:
 * number of solid neighbours of a fluid cell).  That is the Galerkin operator of piecewise-constant
 * transfer with the fluid-fluid part halved (the usual correction for its factor-2 stiffness) and the
 * solid links kept; transfers: restriction = 1/2 * sum of the eight children's residuals,
 * prolongation = trilinear (weights 3/4, 1/4 per axis, index clamped at the walls), dead cells
 * (no links at all) stay 0.  All coefficients are small dyadic rationals, exact in float.
 * Every expression below fixes its association order; the HIP kernels use the same. */

typedef struct mg_level {
    int W, H, D;
    size_t sy, sz, n;
    real *wx, *wy, *wz;   /* weight of the face on the MINUS side of a cell: wx[c] couples c-1 and c (x = 1..W+1) */
    real *d, *dg;         /* Dirichlet term; diagonal = wx[c]+wx[c+1]+wy[c]+wy[c+sy]+wz[c]+wz[c+sz]+d[c] (0: dead cell) */
    real *e, *b;          /* unknown (correction) and right-hand side */
} mg_level;

#define MG_MIN_DIM 4
#define LAT(l, x, y, z) ((size_t)(x) + (size_t)(y) * (l)->sy + (size_t)(z) * (l)->sz)

static int mg_alloc_level(mg_level* l, int W, int H, int D)
{
    l->W = W; l->H = H; l->D = D;
    l->sy = (size_t)W + 2;
    l->sz = l->sy * ((size_t)H + 2);
    l->n = l->sz * ((size_t)D + 2);
    real** arrs[] = { &l->wx, &l->wy, &l->wz, &l->d, &l->dg, &l->e, &l->b };
    for (int i = 0; i < 7; ++i) {
        *arrs[i] = (real*)calloc(l->n, sizeof(real));
        if (!*arrs[i]) return -1;
    }
    return 0;
}

static void mg_free_level(mg_level* l)
{
    free(l->wx); free(l->wy); free(l->wz); free(l->d); free(l->dg); free(l->e); free(l->b);
}

static int mg_fluid0(const cr_sim* s, int x, int y, int z)
{
    return x >= 1 && x <= s->W && y >= 1 && y <= s->H && z >= 1 && z <= s->D && s->f[CR_OBS][AT(s, x, y, z)] != (real)1;
}
static int mg_solid0(const cr_sim* s, int x, int y, int z)
{
    return x >= 1 && x <= s->W && y >= 1 && y <= s->H && z >= 1 && z <= s->D && s->f[CR_OBS][AT(s, x, y, z)] == (real)1;
}
/* level-0 coefficients, never stored */
static real mg_w0(const cr_sim* s, int x, int y, int z, int axis)   /* face between (x,y,z) - e_axis and (x,y,z) */
{
    const int xm = x - (axis == 0), ym = y - (axis == 1), zm = z - (axis == 2);
    return (mg_fluid0(s, x, y, z) && mg_fluid0(s, xm, ym, zm)) ? (real)1 : (real)0;
}
static real mg_d0(const cr_sim* s, int x, int y, int z)
{
    if (!mg_fluid0(s, x, y, z)) return (real)0;
    return (real)(mg_solid0(s, x + 1, y, z) + mg_solid0(s, x - 1, y, z) + mg_solid0(s, x, y + 1, z) + mg_solid0(s, x, y - 1, z) +
                  mg_solid0(s, x, y, z + 1) + mg_solid0(s, x, y, z - 1));
}

/* coefficients of level `c` from the level below: `f` (stored) or, for f == NULL, level 0 of `s` */
static void mg_coarsen(const cr_sim* s, const mg_level* f, mg_level* c)
{
    const real q = (real)0.25, hf = (real)0.5;
#define FW(axis, x, y, z) (f ? (axis == 0 ? f->wx : axis == 1 ? f->wy : f->wz)[LAT(f, x, y, z)] : mg_w0(s, x, y, z, axis))
#define FD(x, y, z) (f ? f->d[LAT(f, x, y, z)] : mg_d0(s, x, y, z))
    for (int Z = 1; Z <= c->D + 1; ++Z)
        for (int Y = 1; Y <= c->H + 1; ++Y)
            for (int X = 1; X <= c->W + 1; ++X) {
                const size_t C = LAT(c, X, Y, Z);
                const int x = 2 * X - 1, y = 2 * Y - 1, z = 2 * Z - 1;     /* first child */
                if (Y <= c->H && Z <= c->D)
                    c->wx[C] = q * (((FW(0, x, y, z) + FW(0, x, y + 1, z)) + FW(0, x, y, z + 1)) + FW(0, x, y + 1, z + 1));
                if (X <= c->W && Z <= c->D)
                    c->wy[C] = q * (((FW(1, x, y, z) + FW(1, x + 1, y, z)) + FW(1, x, y, z + 1)) + FW(1, x + 1, y, z + 1));
                if (X <= c->W && Y <= c->H)
                    c->wz[C] = q * (((FW(2, x, y, z) + FW(2, x + 1, y, z)) + FW(2, x, y + 1, z)) + FW(2, x + 1, y + 1, z));
                if (X <= c->W && Y <= c->H && Z <= c->D)
                    c->d[C] = hf * (((((((FD(x, y, z) + FD(x + 1, y, z)) + FD(x, y + 1, z)) + FD(x + 1, y + 1, z)) + FD(x, y, z + 1)) +
                                      FD(x + 1, y, z + 1)) + FD(x, y + 1, z + 1)) + FD(x + 1, y + 1, z + 1));
            }
#undef FW
#undef FD
    for (int Z = 1; Z <= c->D; ++Z)
        for (int Y = 1; Y <= c->H; ++Y)
            for (int X = 1; X <= c->W; ++X) {
                const size_t C = LAT(c, X, Y, Z);
                c->dg[C] = (((((c->wx[C] + c->wx[C + 1]) + c->wy[C]) + c->wy[C + c->sy]) + c->wz[C]) + c->wz[C + c->sz]) + c->d[C];
            }
}

/* red-black Gauss-Seidel on a coarse level: cells with even x+y+z first */
static void mg_smooth(mg_level* l, int n)
{
    const size_t sy = l->sy, sz = l->sz;
    for (int it = 0; it < n; ++it)
        for (int colour = 0; colour < 2; ++colour) {
#pragma omp parallel for collapse(2) schedule(static)
            for (int z = 1; z <= l->D; ++z)
                for (int y = 1; y <= l->H; ++y)
                    for (int x = 1; x <= l->W; ++x) {
                        if (((x + y + z) & 1) != colour) continue;
                        const size_t c = LAT(l, x, y, z);
                        if (!(l->dg[c] > (real)0)) continue;
                        const real* e = l->e;
                        const real nb = ((((l->wx[c] * e[c - 1] + l->wx[c + 1] * e[c + 1]) + l->wy[c] * e[c - sy]) +
                                          l->wy[c + sy] * e[c + sy]) + l->wz[c] * e[c - sz]) + l->wz[c + sz] * e[c + sz];
                        l->e[c] = (l->b[c] + nb) / l->dg[c];
                    }
        }
}

static real mg_residual_at(const mg_level* l, size_t c)
{
    const size_t sy = l->sy, sz = l->sz;
    const real* e = l->e;
    if (!(l->dg[c] > (real)0)) return (real)0;
    const real nb = ((((l->wx[c] * e[c - 1] + l->wx[c + 1] * e[c + 1]) + l->wy[c] * e[c - sy]) + l->wy[c + sy] * e[c + sy]) +
                     l->wz[c] * e[c - sz]) + l->wz[c + sz] * e[c + sz];
    return (l->b[c] + nb) - l->dg[c] * e[c];
}

/* level-0 residual of the reference's fixed point; ghosts of p hold what setBounds(0, p) left */
static real mg_residual0_at(const cr_sim* s, const real* p, const real* rhs, int x, int y, int z)
{
    const size_t c = AT(s, x, y, z), sy = s->sy, sz = s->sz;
    if (s->f[CR_OBS][c] == (real)1) return (real)0;
    const real nb = p[c + 1] + p[c - 1] + p[c + sy] + p[c - sy] + p[c + sz] + p[c - sz];   /* order of :264-268 */
    return (rhs[c] + nb) - (real)6 * p[c];
}

/* trilinear interpolation of the coarse correction at fine cell (x, y, z); neighbour index clamped at the walls */
static real mg_interp(const mg_level* c, int x, int y, int z)
{
    const int X = (x + 1) / 2, Y = (y + 1) / 2, Z = (z + 1) / 2;
    int Xn = (x & 1) ? X - 1 : X + 1, Yn = (y & 1) ? Y - 1 : Y + 1, Zn = (z & 1) ? Z - 1 : Z + 1;
    if (Xn < 1) Xn = 1;
    if (Xn > c->W) Xn = c->W;
    if (Yn < 1) Yn = 1;
    if (Yn > c->H) Yn = c->H;
    if (Zn < 1) Zn = 1;
    if (Zn > c->D) Zn = c->D;
    const real a = (real)0.75, q = (real)0.25;
    const real* e = c->e;
    const real x00 = a * e[LAT(c, X, Y, Z)] + q * e[LAT(c, Xn, Y, Z)];
    const real x10 = a * e[LAT(c, X, Yn, Z)] + q * e[LAT(c, Xn, Yn, Z)];
    const real x01 = a * e[LAT(c, X, Y, Zn)] + q * e[LAT(c, Xn, Y, Zn)];
    const real x11 = a * e[LAT(c, X, Yn, Zn)] + q * e[LAT(c, Xn, Yn, Zn)];
    const real y0 = a * x00 + q * x10;
    const real y1 = a * x01 + q * x11;
    return a * y0 + q * y1;
}

/* Level-0 smoothing step: two Jacobi sweeps of the reference's update (simulation.cpp:263-269, from the previous
 * iterate), each damped -- q + (6/7)(r - q) in every interior cell -- and followed by setBounds.  (6/7 is the damping
 * that makes Jacobi a smoother for the 7-point operator; undamped it leaves the checkerboard mode alone.) */
static void mg_smooth0(cr_sim* s, real* q, const real* rhs, int steps)
{
    const int W = s->W, H = s->H, D = s->D;
    const size_t sy = s->sy, sz = s->sz;
    const real inv_c = (real)1 / (real)6, om = (real)6 / (real)7;
    real* nxt = s->scratch;
    for (int it = 0; it < 2 * steps; ++it) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int z = 1; z <= D; ++z)
            for (int y = 1; y <= H; ++y)
                for (int x = 1; x <= W; ++x) {
                    size_t c = AT(s, x, y, z);
                    real nb = q[c + 1] + q[c - 1] + q[c + sy] + q[c - sy] + q[c + sz] + q[c - sz];
                    real r = (rhs[c] + (real)1 * nb) * inv_c;
                    nxt[c] = q[c] + om * (r - q[c]);
                }
#pragma omp parallel for collapse(2) schedule(static)
        for (int z = 1; z <= D; ++z)
            for (int y = 1; y <= H; ++y)
                memcpy(&q[AT(s, 1, y, z)], &nxt[AT(s, 1, y, z)], (size_t)W * sizeof(real));
        enforce_bounds(s, 0, q);
    }
}

typedef struct mg_ctx {
    cr_sim* s;
    mg_level* lv;    /* lv[1..nl-1]; level 0 is the simulation grid */
    int nl;
} mg_ctx;

static void mg_vcycle_coarse(mg_ctx* m, int l)
{
    cr_sim* s = m->s;
    mg_level* L = &m->lv[l];
    if (l == m->nl - 1) {
        mg_smooth(L, s->mg_coarse);
        return;
    }
    mg_level* C = &m->lv[l + 1];
    mg_smooth(L, s->mg_pre);
#pragma omp parallel for collapse(2) schedule(static)
    for (int Z = 1; Z <= C->D; ++Z)
        for (int Y = 1; Y <= C->H; ++Y)
            for (int X = 1; X <= C->W; ++X) {
                const size_t cc = LAT(C, X, Y, Z);
                const int x = 2 * X - 1, y = 2 * Y - 1, z = 2 * Z - 1;
                real r = mg_residual_at(L, LAT(L, x, y, z));
                r = r + mg_residual_at(L, LAT(L, x + 1, y, z));
                r = r + mg_residual_at(L, LAT(L, x, y + 1, z));
                r = r + mg_residual_at(L, LAT(L, x + 1, y + 1, z));
                r = r + mg_residual_at(L, LAT(L, x, y, z + 1));
                r = r + mg_residual_at(L, LAT(L, x + 1, y, z + 1));
                r = r + mg_residual_at(L, LAT(L, x, y + 1, z + 1));
                r = r + mg_residual_at(L, LAT(L, x + 1, y + 1, z + 1));
                C->b[cc] = (C->dg[cc] > (real)0) ? (real)0.5 * r : (real)0;
                C->e[cc] = (real)0;
            }
    mg_vcycle_coarse(m, l + 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 1; z <= L->D; ++z)
        for (int y = 1; y <= L->H; ++y)
            for (int x = 1; x <= L->W; ++x) {
                const size_t c = LAT(L, x, y, z);
                if (L->dg[c] > (real)0) L->e[c] = L->e[c] + mg_interp(C, x, y, z);
            }
    mg_smooth(L, s->mg_post);
}

/* mg_cycles V-cycles on  p  (initial guess as found; the projection hands in zeros) */
static int mg_solve(cr_sim* s, real* p, const real* rhs)
{
    mg_ctx m;
    m.s = s;
    m.nl = 1;
    {
        int W = s->W, H = s->H, D = s->D;
        while (W % 2 == 0 && H % 2 == 0 && D % 2 == 0 && W / 2 >= MG_MIN_DIM && H / 2 >= MG_MIN_DIM && D / 2 >= MG_MIN_DIM) {
            W /= 2; H /= 2; D /= 2;
            ++m.nl;
        }
    }
    m.lv = (mg_level*)calloc((size_t)m.nl, sizeof(mg_level));
    if (!m.lv) return -1;
    {
        int W = s->W, H = s->H, D = s->D;
        for (int l = 1; l < m.nl; ++l) {
            W /= 2; H /= 2; D /= 2;
            if (mg_alloc_level(&m.lv[l], W, H, D)) return -1;
            mg_coarsen(s, l == 1 ? NULL : &m.lv[l - 1], &m.lv[l]);
        }
    }
    for (int cyc = 0; cyc < s->mg_cycles; ++cyc) {
        if (m.nl == 1) {
            mg_smooth0(s, p, rhs, s->mg_coarse);
            continue;
        }
        mg_level* C = &m.lv[1];
        mg_smooth0(s, p, rhs, s->mg_pre);
#pragma omp parallel for collapse(2) schedule(static)
        for (int Z = 1; Z <= C->D; ++Z)
            for (int Y = 1; Y <= C->H; ++Y)
                for (int X = 1; X <= C->W; ++X) {
                    const size_t cc = LAT(C, X, Y, Z);
                    const int x = 2 * X - 1, y = 2 * Y - 1, z = 2 * Z - 1;
                    real r = mg_residual0_at(s, p, rhs, x, y, z);
                    r = r + mg_residual0_at(s, p, rhs, x + 1, y, z);
                    r = r + mg_residual0_at(s, p, rhs, x, y + 1, z);
                    r = r + mg_residual0_at(s, p, rhs, x + 1, y + 1, z);
                    r = r + mg_residual0_at(s, p, rhs, x, y, z + 1);
                    r = r + mg_residual0_at(s, p, rhs, x + 1, y, z + 1);
                    r = r + mg_residual0_at(s, p, rhs, x, y + 1, z + 1);
                    r = r + mg_residual0_at(s, p, rhs, x + 1, y + 1, z + 1);
                    C->b[cc] = (C->dg[cc] > (real)0) ? (real)0.5 * r : (real)0;
                    C->e[cc] = (real)0;
                }
        mg_vcycle_coarse(&m, 1);
#pragma omp parallel for collapse(2) schedule(static)
        for (int z = 1; z <= s->D; ++z)
            for (int y = 1; y <= s->H; ++y)
                for (int x = 1; x <= s->W; ++x) {
                    const size_t c = AT(s, x, y, z);
                    if (s->f[CR_OBS][c] != (real)1) p[c] = p[c] + mg_interp(C, x, y, z);
                }
        enforce_bounds(s, 0, p);
        mg_smooth0(s, p, rhs, s->mg_post);
    }
    for (int l = 1; l < m.nl; ++l) mg_free_level(&m.lv[l]);
    free(m.lv);
    return 0;
}

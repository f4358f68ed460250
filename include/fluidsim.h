/* include/fluidsim.h -- C ABI of libfluidsim.so, the MI355X-native wind-tunnel solver.
 *
 * The reference (Ghundi/fluid_simulation) has no FFI or plugin interface; its seams are a
 * C++ class, one free function, a no-argument executable and a file layout (SURVEY.md
 * section 8b).  This header is the drop-in boundary for the first two: one handle type
 * and one function per public member of `class Simulation` (simulation.h:42-91) and for
 * loadSTLIntoObstacles (object_loader.h:7-17), with the same argument order, the same
 * 1-based interior coordinates and the same defaults.  Every entry point cites the
 * reference declaration it replaces.  Plain C types only; no device pointers cross.
 *
 * All compute happens in hand-written HIP kernels for gfx950; there is no CPU fallback.
 * fs_create fails (NULL + fs_last_error) when no HIP device is usable.
 *
 * Return convention: 0 on success, a negative FS_E* code on failure; fs_last_error()
 * describes the most recent failure on the calling thread.  The reference itself has no
 * error reporting (out-of-range mutators are UB there, simulation.cpp:157-177); here
 * they are FS_EINVAL.  A handle is driven by one host thread at a time.
 */
#ifndef FLUIDSIM_H
#define FLUIDSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_sim fs_sim;

enum {
    FS_OK = 0,
    FS_EINVAL = -1,   /* bad argument / out-of-range cell / call not legal in this state */
    FS_EIO = -2,      /* STL or dump-file I/O */
    FS_EHIP = -3,     /* HIP runtime error (no device, allocation, launch) */
    FS_ECOMM = -4,    /* RCCL error (multi-GPU slabs) */
    FS_ENOMEM = -5
};

/* Field selectors for fs_get_field / fs_set_field and the per-pass entry points.  The
 * order of 0..4 is the frame-dump order of simulation.cpp:143-147 with obs at 4; 5..10
 * are the reference's private scratch arrays (simulation.h:16-27). */
enum {
    FS_DENS = 0, FS_VX = 1, FS_VY = 2, FS_VZ = 3, FS_OBS = 4,
    FS_PRESSURE = 5, FS_DIVERGENCE = 6,
    FS_VX_PREV = 7, FS_VY_PREV = 8, FS_VZ_PREV = 9, FS_BUFFER = 10,
    FS_NFIELDS = 11
};

/* Solver used by linearSolver (simulation.cpp:251-273). */
enum {
    FS_SOLVER_JACOBI = 0,  /* ping-pong Jacobi, any grid, multi-GPU capable (default) */
    FS_SOLVER_GS_LEX = 1,  /* the reference's in-place sweep in its one-thread order; verification mode */
    FS_SOLVER_RBSOR = 2,   /* NOT the reference's arithmetic: red-black successive over-relaxation, `acc`
                            * iterations (even x+y+z cells, then odd, setBounds after each half), relaxation
                            * factor "sor_omega"; converges far faster per iteration; SURVEY.md 8f rank 4 */
    FS_SOLVER_MG = 3       /* NOT the reference's arithmetic: the projection's pressure equation (simulation.cpp:320) is
                            * solved by "mg_cycles" multigrid V-cycles instead of `acc` relaxation sweeps (2x2x2
                            * cell-centred coarsening, red-black smoothing, obstacle-aware coarse operators; defined in
                            * oracle/cpu_ref_mg.h); so does fs_linear_solver when called with that equation's coefficients
                            * (b = 0, a = 1, c = 6); every other solve (diffusion) runs Jacobi.  One GPU or z-slabs
                            * (a rank must then hold an even number of planes; option "mg_min_planes": coarse levels stay
                            * distributed while every rank keeps that many planes, default 32, and are held whole by every
                            * rank below); grids whose extents cannot be halved get no coarse levels.  SURVEY.md 8f rank 4 */
};

/* ---- construction -------------------------------------------------------------- */

/* Simulation::Simulation(w,h,d,iter,speed=30,dt=0.05f,diff=2.0e-5f,visc=1.5e-5f,acc=15)
 * -- simulation.h:59-64, simulation.cpp:17-44.  Uses the calling thread's current HIP
 * device.  Fields are zero-initialised; storage is allocated on first use so that
 * fs_set_option("precision", ...) can still be applied. */
fs_sim* fs_create(int w, int h, int d, int iter, int speed, float dt, float diff, float visc, int acc);

/* Reference defaults for the trailing constructor arguments (simulation.h:60-64). */
#define FS_DEFAULT_SPEED 30
#define FS_DEFAULT_DT 0.05f
#define FS_DEFAULT_DIFF 2.0e-5f
#define FS_DEFAULT_VISC 1.5e-5f
#define FS_DEFAULT_ACC 15

int fs_destroy(fs_sim* s);

/* Options (string key/value), legal before the first step unless noted:
 *   "precision"   "fp32" (default) | "fp64"       field storage + arithmetic; before first use only
 *   "solver"      "jacobi" (default; out-of-place sweeps, deterministic, multi-GPU) | "gs_lex" (the
 *                 reference's in-place lexicographic sweep, simulation.cpp:259-271, bit-identical with
 *                 the reference at one OpenMP thread; single GPU) | "rbsor" (optional red-black SOR,
 *                 different arithmetic from the reference by design; defined by oracle/cpu_ref.c CR_RBSOR)
 *                 | "mg" (optional: the projection's pressure equation solved by multigrid V-cycles, every
 *                 other system relaxed as under "jacobi"; different arithmetic from the reference by design,
 *                 defined by oracle/cpu_ref_mg.h CR_MG; see FS_SOLVER_MG above)
 *   "sor_omega"   relaxation factor of "rbsor", in (0, 2), default 1 (= red-black Gauss-Seidel)
 *   "dump_dir"    directory for frame dumps, default "data" (simulation.cpp:56-60)
 *   "dump_every"  N>=1 dump every Nth step (default 1 = reference behaviour), 0 = never,
 *                 -1 = last step of fs_run only.  May be changed at any time.
 *   "dump_async"  "1" (default): frames go through a pinned double buffer and a writer thread while
 *                 the next step computes; "0": every dump completes before fs_step returns
 *   "voxel_seed"  seed of the voxelizer's minstd_rand stream (object_loader.cpp:399 uses a
 *                 thread-id hash; default here is 1)
 *   "quiet"       "1" suppresses the reference's console lines
 *   "profile"     "1" brackets each kernel family with HIP events (see fs_get_timing)
 *   "elide_dead_density_solve" "1" skips diffuse(0,dens,buffer) whose result the next
 *                 advect overwrites (simulation.cpp:135-136); default "0" = do it
 * Per-handle tuning keys that never change results (kernel selection and launch shapes):
 *   "sweep_fuse"  "1" one solver sweep per pass over memory, "2" two, "3" (default) two or three: the
 *                 three-sweep kernel (fp32, rows up to 512 cells) is timed against the two-sweep one
 *                 once per grid and used where a sweep costs less (z-slab ranks: always, so that all
 *                 ranks keep one exchange schedule), "4" three wherever that kernel exists;
 *   "two_sweep_kernel" "auto" (default: timed once per grid) | "pair" | "fused" -- which of the two
 *                 two-sweep kernels (jacobi_pair_kernel / jacobi_fused_kernel<NL=2>) runs those passes;
 *   "advect_kernels" "cell" (default: one thread per cell) | "celltab" (the same reading clamped traces from the
 *                 column tables) | "tile" (the tables' window around an 8 x 8 tile of rows staged in LDS, "advect_window"
 *                 rows / planes wide) | "row" (four cells per lane, clamp tables); all bit-identical, none faster on the
 *                 benchmark flow;
 *   "mg_cycles" (default 4: 75 % of the time of the 80 sweeps of config 3, residual 34x smaller), "mg_pre", "mg_post" (smoothing steps before / after the coarse correction, default 1),
 *                 "mg_coarse_iters" (iterations on the coarsest level, default 30): solver "mg" only;
 *   "launch_plans" "<two-sweep plan id>,<three-sweep plan id>" (what fs_get_int "pair_shape" / "triple_plan" reported
 *                 in another run; -1 = none): replay those launch plans instead of timing candidates (profiling);
 *   "wall_free"   "auto" (default) | "0" | "1": whether workgroups of the three-sweep kernel that touch no wall run its
 *                 wall-free second body (auto: when a launch has more than 256 workgroups);
 *   "sweep_ry" "sweep_zc" "sweep_blocks" "pair_zc" "pair_shape" "project_kernels" "fuse_advect"
 *                 -- see csrc/kernels.h (SweepTune), csrc/fluidsim.cpp and tools/tune_*.py.
 * z-slab handles only (never change results either):
 *   "overlap"     how a solver pass and the exchange of its boundary planes are scheduled: "0" the pass, then the exchange;
 *                 "1" boundary planes first, their exchange beside the interior launch; "2" boundary launch + exchange on
 *                 the communication stream beside the interior launch; "3" (FSIPC transport only, elsewhere = "0") the
 *                 kernels store the boundary planes straight into the neighbours' halo planes; "auto" (default) times
 *                 them once over the real transport, the slowest rank's time decides, every rank agrees.  Before the
 *                 first solve.  fs_get_int "overlap_plan" / fs_get_float "overlap<k>_ms" report the choice and the times;
 *   "comm_cus"    "0" (default) | N | "auto": CUs kept free of solver workgroups (CU-masked compute stream) for the
 *                 transport's kernels; "auto" adds "8 free" to the timed candidates.  Before first use;
 *   "split_density_solve" "1" (default) | "0": run half of the density solve (simulation.cpp:135) between the first
 *                 projection and the velocity advection, so that the reach of each advection gather reaches the host
 *                 without stalling the device (same passes, same order, same bits);
 *   "debug_poison_gather" "1": fill the gathered advection source with NaN patterns before each gather (tests).
 * fs_get_int also answers "local_depth" "z_offset" "halo_depth" "last_advect_reach" "pair_shape" "triple_plan"
 * "two_sweep_fused" "mg_levels" and, for slab handles, "stream_syncs" (compute-stream synchronisations issued by slab
 * steps; 0 on the step path) "reach_waits" "reach_waits_blocked" "reach_wait_us" "reach_hidden" "reach_exposed".
 */
int fs_set_option(fs_sim* s, const char* key, const char* value);

/* Public data members of class Simulation (simulation.h:44-54), by name:
 * "width" "height" "depth" "speed" "acc" "iter" (int) and "dt" "diff" "visc" (float). */
int fs_get_int(fs_sim* s, const char* name, int* out);
int fs_set_int(fs_sim* s, const char* name, int value);      /* speed, acc, iter only */
int fs_get_float(fs_sim* s, const char* name, float* out);
int fs_set_float(fs_sim* s, const char* name, float value);

/* ---- mutators (1-based interior coordinates, like the reference) ---------------- */

int fs_add_obstacle(fs_sim* s, int x, int y, int z);                          /* Simulation::addObstacle  simulation.h:79, .cpp:155-158 */
int fs_add_density(fs_sim* s, int x, int y, int z, float amount);             /* Simulation::addDensity   simulation.h:84, .cpp:163-166 */
int fs_set_velocity(fs_sim* s, int x, int y, int z, float ax, float ay, float az); /* Simulation::setVelocity simulation.h:89, .cpp:171-178 */

/* loadSTLIntoObstacles(stlFile, sim, scale=0.8f, rot_x, rot_y, rot_z, translate_x/y/z)
 * -- object_loader.h:7-17, object_loader.cpp:270-452.  Ray-parity voxelisation runs on
 * the GPU.  A missing or empty STL prints the reference's message, leaves the tunnel
 * unchanged and returns FS_EIO (the reference returns void and carries on,
 * object_loader.cpp:282-285; callers that want that behaviour ignore the code).
 * On success *added (may be NULL) receives the "Added N obstacle points" count. */
int fs_load_stl(fs_sim* s, const char* stl_file, float scale, float rot_x, float rot_y, float rot_z,
                float translate_x, float translate_y, float translate_z, long* added);

/* Whole-mask injection (golden masks, analytic shapes): `mask` is a padded x-fastest
 * array of (w+2)(h+2)(d+2) bytes, non-zero = solid; ghost cells must be zero. */
int fs_set_obstacle_mask(fs_sim* s, const uint8_t* mask, size_t n);

/* ---- time stepping -------------------------------------------------------------- */

int fs_step(fs_sim* s);     /* Simulation::step()  simulation.h:74, .cpp:96-150 (incl. the frame dump, subject to dump_every) */
int fs_run_one(fs_sim* s);  /* one iteration of the loop in Simulation::run(): inlet density, buffer=dens, step()  .cpp:63-78 */
int fs_run(fs_sim* s);      /* Simulation::run()   simulation.h:69, .cpp:49-91: `iter` iterations + the console statistics */
int fs_sync(fs_sim* s);     /* wait for all queued GPU work of this handle */

/* The private passes of the reference, exposed for per-kernel parity tests.  `field` and
 * `prev` are FS_* selectors; b is the boundary code (0 scalar, 1/2/3 velocity component). */
int fs_set_bounds(fs_sim* s, int b, int field);                                  /* setBounds     simulation.cpp:183-246 */
int fs_linear_solver(fs_sim* s, int b, int field, int prev, float a, float c);   /* linearSolver  simulation.cpp:251-273 */
int fs_diffuse(fs_sim* s, int b, int field, int prev);                           /* diffuse       simulation.cpp:278-284 */
int fs_project(fs_sim* s);                                                       /* project       simulation.cpp:289-362 */
int fs_advect(fs_sim* s, int b, int field, int prev);                            /* advect        simulation.cpp:367-424 */

/* ---- data access ---------------------------------------------------------------- */

/* Copies one field in the reference's own layout: padded (w+2)(h+2)(d+2), x fastest
 * (simulation.h:9).  elem_size selects the host element type (4 = float, 8 = double);
 * conversion happens on the device.  n is the element count of the host buffer. */
int fs_get_field(fs_sim* s, int which, void* dst, size_t n, int elem_size);
int fs_set_field(fs_sim* s, int which, const void* src, size_t n, int elem_size);
size_t fs_padded_size(fs_sim* s);   /* Simulation::size  simulation.cpp:35 */

/* Append one frame to <dump_dir>/{data,obs,v_x,v_y,v_z}.bin exactly as simulation.cpp:140-148. */
int fs_dump_frame(fs_sim* s);

/* Diagnostics of run(): sum/min/max over the whole padded array (simulation.cpp:73-90). */
int fs_field_stats(fs_sim* s, int which, double* sum, double* min, double* max);

/* ---- measurement ---------------------------------------------------------------- */

/* With option "profile"="1": accumulated HIP-event time and launch count of one kernel
 * family since the last fs_reset_timing: "sweep" (one solver iteration per launch)
 * "sweep_pair" (two iterations per launch) "sweep_triple" (three iterations per launch)
 * "divergence" "gradient" "advect" "bounds" "misc" "comm" (z-slab exchanges and gathers)
 * "multigrid" (the coarse-level work of solver "mg"; its level-0 smoothing passes count as
 * "sweep_pair").  Events are recorded on the handle's own stream. */
int fs_get_timing(fs_sim* s, const char* family, double* total_ms, long* launches);
int fs_reset_timing(fs_sim* s);

/* Times `reps` back-to-back linearSolver sweeps (b, a, c as fs_linear_solver) on the
 * current state with HIP events on the handle's stream, without changing the state
 * (scratch output).  Writes the mean milliseconds per sweep. */
int fs_time_sweeps(fs_sim* s, int b, int field, int prev, float a, float c, int reps, double* ms_per_sweep);

/* ---- viewer post-processing on the device -------------------- */

/* The streamlines the reference's viewer computes on the CPU for the frame it shows --
 * generate_streamlines, GUI/utils.py:118-213, called from GUI/main_window.py:227-233 -- from the
 * fields as they are on the device now (a dumped frame holds the same values).  Parameters are
 * GUI/config.py:18-23: density = STREAMLINE_DENSITY (30), proximity = STREAMLINE_PROXIMITY (2),
 * max_length = INTEGRATION_STEPS (100), step_size = INTEGRATION_STEP_SIZE (0.2),
 * vel_change_threshold = VELOCITY_CHANGE_THRESHOLD (0.1).  Coordinates are the viewer's: indices
 * into the padded arrays, x first.  Lines come in the reference's seed order (z, y, x loops).
 * The result stays in the handle until the next call; *n_lines / *n_points (may be NULL) receive
 * its size.  Single-GPU handles only. */
int fs_streamlines(fs_sim* s, int density, double proximity, int max_length, double step_size,
                   double vel_change_threshold, long* n_lines, long* n_points);
/* Copies the last result: offsets[n_lines + 1] (index of each line's first point), points
 * [3 * n_points] (x, y, z per point), norm_speed[n_lines] -- the number the viewer hands to its
 * colour map, min(max speed along the line / (max(vx, vy, vz) + 1e-6), 1)  (utils.py:198-205).
 * Any of the three may be NULL. */
int fs_streamlines_fetch(fs_sim* s, long* offsets, double* points, double* norm_speed);

/* The obstacle mesh the reference's viewer builds on the CPU for the frame it shows --
 * generate_obstacle_mesh, GUI/utils.py:10-38 (scikit-image marching cubes of `obs` at level 0.5),
 * called from GUI/main_window.py:204-218 -- from `obs` as it is on the device: an indexed triangle mesh,
 * one vertex per grid edge on which obs crosses 0.5 (linear interpolation; for a 0/1 mask the edge
 * midpoint), in the viewer's coordinates (indices into the padded array, x first).  Closed, oriented
 * with normals from solid to fluid.  PARITY UNPINNED against scikit-image (vertex / triangle order and
 * the cut of ambiguous cubes may differ; see csrc/surface.h).  The result stays in the handle until
 * the next call.  Single-GPU handles only. */
int fs_obstacle_surface(fs_sim* s, long* n_vertices, long* n_triangles);
/* Copies the last result: vertices[3 * n_vertices] (x, y, z), triangles[3 * n_triangles] (vertex
 * indices).  Either may be NULL. */
int fs_obstacle_surface_fetch(fs_sim* s, float* vertices, int* triangles);
/* The triangle table behind it, for one cube configuration (bit i set: corner
 * (i & 1, (i >> 1) & 1, (i >> 2) & 1) is solid): writes 3 cube-edge ids per triangle into
 * edges[24] and returns the triangle count (0..8); edge id = 4 * axis + 2 * (offset on the higher
 * other axis) + (offset on the lower other axis).  Needs neither a handle nor a GPU. */
int fs_surface_case_table(int config, int* edges);

/* ---- multi-GPU z-slabs (one process per GPU; RCCL halo exchange over xGMI) -------- */

/* Size of the opaque RCCL unique id; rank 0 fills it with fs_comm_unique_id and the
 * host layer broadcasts it to the other ranks (e.g. through torch.distributed).  An id that starts with
 * "FSIPC:" + a POSIX shared-memory name instead selects the stream-ordered device-to-device transport between rank
 * processes of one host (csrc/ipc.h: hipIpc-mapped arrays, copy engines, device-side handshakes; ranks may share a
 * GPU; at most 8 ranks); "FSSHM:" + name the host-staged synchronous development transport; "FSNULL:" none (timing). */
#define FS_COMM_ID_BYTES 128
int fs_comm_unique_id(void* id_out);
/* Turns the handle into the owner of z-slab `rank` of `nranks` of the global grid given
 * to fs_create (depth must divide evenly).  Must precede first use. */
int fs_comm_init(fs_sim* s, int rank, int nranks, const void* id);

/* Loads RCCL, builds a one-rank communicator on the current device and pushes data through
 * every collective the slab path uses (grouped send/recv, all-gather, broadcast, all-reduce).
 * A plumbing check for machines with a single GPU. */
int fs_comm_selftest(void);

/* What carries the halo planes of this handle: "single GPU", or the path of the RCCL library that was
 * loaded (it must be the one next to the HIP runtime the process runs on), or the name of a
 * development transport.  The string belongs to the handle. */
const char* fs_comm_transport(fs_sim* s);

const char* fs_last_error(void);
const char* fs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDSIM_H */

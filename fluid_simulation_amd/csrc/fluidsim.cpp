// fluidsim.cpp -- C-ABI host driver of libfluidsim.so (include/fluidsim.h).
//
// Owns device storage, orders the kernels of one time step on a HIP stream, converts
// between the device layout and the reference's dump layout, and (multi-GPU) exchanges
// z-slab halo planes over RCCL.  No arithmetic on field data happens on the host and
// there is no CPU fallback: without a usable HIP device fs_create fails.
//
// Orchestration follows Simulation::step()/run() of the reference (simulation.cpp:49-150);
// every numeric expression lives in kernels.hip.
#include "../../include/fluidsim.h"
#include "kernels.h"
#include "voxelize.h"
#include "comm.h"
#include "dump.h"
#include "streamlines.h"
#include "surface.h"
#include "multigrid.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(FS_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum Family { FAM_SWEEP = 0, FAM_PAIR, FAM_TRIPLE, FAM_DIV, FAM_GRAD, FAM_ADVECT, FAM_BOUNDS, FAM_MISC, FAM_COMM, FAM_MG, FAM_COUNT };
const char* const kFamilyNames[FAM_COUNT] = { "sweep", "sweep_pair", "sweep_triple", "divergence", "gradient", "advect", "bounds", "misc", "comm", "multigrid" };

constexpr int NPOOL = FS_NFIELDS + 3;   // named fields + ping-pong scratch

struct Span {
    hipEvent_t a, b;
    int fam;
    long launches;
};

}  // namespace

// ---------------------------------------------------------------------------------------
struct EngineBase {
    virtual ~EngineBase() {}
    virtual int step() = 0;
    virtual int run_one() = 0;
    virtual int set_bounds(int b, int field) = 0;
    virtual int linear_solver(int b, int field, int prev, float a, float c) = 0;
    virtual int diffuse(int b, int field, int prev) = 0;
    virtual int project() = 0;
    virtual int advect(int b, int field, int prev) = 0;
    virtual int get_field(int which, void* dst, size_t n, int elem) = 0;
    virtual int set_field(int which, const void* src, size_t n, int elem) = 0;
    virtual int set_mask(const uint8_t* mask, size_t n) = 0;
    virtual int point(int which, int x, int y, int z, float v, int set_instead) = 0;
    virtual int stats(int which, double* out3) = 0;
    virtual int dump_frame() = 0;
    virtual int time_sweeps(int b, int field, int prev, float a, float c, int reps, double* ms) = 0;
    virtual int apply_solid_cells(const int* cells, long n) = 0;
    virtual int tuned_shape() const = 0;
    virtual int tuned_triple() const = 0;
    virtual int halo_depth() const = 0;
    virtual int streamlines(int density, double proximity, int max_length, double step_size, double threshold) = 0;
    virtual int obstacle_surface() = 0;
    virtual int reference_order_sum(int which, double* out) = 0;
    virtual int multigrid_levels() const = 0;
};

struct fs_sim {
    // Simulation's public members (simulation.h:44-54); W/H/D are the GLOBAL extents
    int W = 0, H = 0, D = 0, iter = 0, speed = 0, acc = 0;
    float dt = 0, diff = 0, visc = 0;
    // options
    bool fp64 = false;
    int solver = FS_SOLVER_JACOBI;
    float omega = 1.0f;          // relaxation factor of solver=rbsor
    int plan_two = -2, plan_three = -2;   // "launch_plans": replay these instead of timing (-2: not set)
    int mg_cycles = 4, mg_pre = 1, mg_post = 1, mg_coarse = 30;   // solver=mg: V-cycles per pressure solve, smoothing steps, coarsest-level iterations
    int mg_min_planes = 32;      // z-slabs: a coarse level stays distributed while every rank keeps at least this many of its planes
                                 // (measured, 512^3 as four slabs: 4 -> 203, 16 -> 186, 32 -> 175, 64 -> 152 ms per step; below 32 the
                                 // exchanges of a level cost more than computing it whole on every rank, above it the whole-held level
                                 // of an 8-way split is as large as a rank's own slab)
    std::string dump_dir = "data";
    int dump_every = 1;
    unsigned voxel_seed = 1;
    bool quiet = false, profile = false, elide_dead = false;
    bool fuse_advect = true;     // one kernel for the three velocity advections of a step (single GPU)
    int overlap = -1;            // z-slabs, how a pass and its halo exchange are scheduled: 0 the pass, then the exchange; 1 boundary
                                 // planes first, their exchange on the communication stream while the interior is computed; 2 boundary
                                 // launch + exchange on the communication stream beside the interior launch; -1 (default) = "auto":
                                 // the three are timed once over the real transport, the slowest rank's time decides (all ranks agree)
    int comm_cus = 0;            // z-slabs: CUs kept free of solver workgroups (the compute stream gets a CU mask) so that the
                                 // transport's kernels find room beside a launch that fills the chip; -1 = "auto": 0 and 8 are timed
    bool split_dens = true;      // z-slabs: run half of the (dead) density solve between the first projection and the velocity
                                 // advection, so that the reach of the back-trace arrives on the host without stalling the device
    // slab-step bookkeeping readable through fs_get_int
    long n_stream_syncs = 0;     // host synchronisations of the compute stream issued by the slab step (reach fallback path)
    long n_alloc_syncs = 0;      // ... by one-off allocations inside a step (the gathered sources at the first gather)
    long n_reach_waits = 0, n_reach_blocked = 0;   // waits for an asynchronously delivered reach; those that found it not yet there
    long n_reach_hidden = 0, n_reach_exposed = 0;  // gathers + advections queued while the device was still busy with the half density
                                                   // solve placed before them (hidden) / after it had run dry (exposed: a bubble)
    double reach_wait_ms = 0.0;  // host time spent in them
    int overlap_plan = -1, cus_plan = -1;          // what "auto" chose (or the forced values), -1 before the first slab solve
    double overlap_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // slowest rank's ms per pass of each timed candidate (overlap 0..3 x cu mask off/on)
    bool debug_poison = false;   // fill the gathered advection source with NaN bit patterns before each gather
    int last_reach = 0;          // planes of reach used by the most recent slab advection
    // device
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_full = nullptr, stream_masked = nullptr;   // z-slabs with comm_cus: `stream` is one of these two
    EngineBase* eng = nullptr;
    // z-slab partition (comm.h)
    fs::Comm comm;
    // timing
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    double fam_ms[FAM_COUNT] = {0};
    long fam_launches[FAM_COUNT] = {0};
    // dumps
    FILE* dump_fp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool dump_open = false, dump_warned = false, in_run = false;
    long step_no = 0;
    long dump_frames = 0;
    fs::FrameWriter writer;      // pinned double-buffered D2H + writer thread
    // result of the last fs_streamlines call
    std::vector<long> sl_offsets;
    std::vector<double> sl_points, sl_norm;
    // result of the last fs_obstacle_surface call
    std::vector<float> surf_verts;
    std::vector<int> surf_tris;
    bool surf_valid = false;
    bool dump_async = true;
    fs::SweepTune tune;          // launch tunables of this handle (fs_set_option sweep_* / pair_* / project_kernels)

    int span_begin(int fam)
    {
        if (!profile) return -1;
        Span sp;
        sp.fam = fam;
        sp.launches = 0;
        for (hipEvent_t* ev : { &sp.a, &sp.b }) {
            if (!event_pool.empty()) { *ev = event_pool.back(); event_pool.pop_back(); }
            else if (hipEventCreate(ev) != hipSuccess) return -1;
        }
        hipEventRecord(sp.a, stream);
        spans.push_back(sp);
        return (int)spans.size() - 1;
    }
    void span_end(int id, long launches)
    {
        if (id < 0) return;
        spans[id].launches = launches;
        hipEventRecord(spans[id].b, stream);
    }
    void resolve_spans()
    {
        if (spans.empty()) return;
        hipStreamSynchronize(stream);
        for (Span& sp : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
                fam_ms[sp.fam] += ms;
                fam_launches[sp.fam] += sp.launches;
            }
            event_pool.push_back(sp.a);
            event_pool.push_back(sp.b);
        }
        spans.clear();
    }
};

namespace {

struct ScopedSpan {
    fs_sim* s; int id; long n;
    ScopedSpan(fs_sim* s_, int fam, long launches = 1) : s(s_), id(s_->span_begin(fam)), n(launches) {}
    ~ScopedSpan() { s->span_end(id, n); }
};

template <class T> T host_cbrt(T v);
template <> float host_cbrt<float>(float v) { return std::cbrt(v); }     // std::cbrt(float), simulation.cpp:295
template <> double host_cbrt<double>(double v) { return std::cbrt(v); }

// ---------------------------------------------------------------------------------------
template <class T>
struct Engine : EngineBase {
    fs_sim* S;
    fs::GridDesc g;       // local slab
    fs::SlabCtx sc;
    static constexpr size_t ARENA_CHUNK_BYTES = (size_t)1 << 30;
    std::vector<T*> pool_chunks;        // the allocations behind arr[]: pool_per arrays each, pool_stride elements apart
    std::vector<size_t> pool_chunk_bytes;
    size_t pool_stride = 0;
    int pool_per = 1;
    std::vector<T*> gather_chunks;      // the same behind gathered / gathered3 (z-slabs, allocated at the first gather)
    T* arr[NPOOL] = {nullptr};          // LEAD-shifted pointers
    int slot[FS_NFIELDS];               // field -> array id (aliases allowed inside a step)
    bool held[NPOOL] = {false};         // temporaries owned by a running solve
    uint8_t* flags = nullptr;           // shifted like the fields
    uint8_t* kill = nullptr;            // one byte per four cells for the sweeps; byte (cell+3)/4
    bool flags_dirty = true;
    bool halos_dirty = false;           // a host-side mutation may have changed a slab boundary plane
    void* dense = nullptr;              // device staging of fs_get_field / fs_set_field (dense local slab), on demand
    size_t dense_bytes = 0;
    T* gathered = nullptr;              // gathered advection source (z-slabs only), LEAD-shifted global array
    T* gathered3[3] = {nullptr, nullptr, nullptr};   // the same for the three sources of the fused velocity advection
    double* red = nullptr;              // stats scratch
    T rb_omega = (T)1;                  // relaxation factor of the red-black / damped passes of the running solve
    bool rb_damped = false;             // those passes are two damped Jacobi sweeps (solver=mg level 0) instead of one red-black iteration
    fs::Multigrid<T> mg;                // coarse levels of solver=mg; rebuilt when the flag bytes change
    bool mg_current = false;
    T* coltab = nullptr;                // clamp tables of the advection row kernels: 6 x (H+2)(D+2) (single GPU only)
    static constexpr int FUSED2 = 64;   // pair_shape >= FUSED2: the two-sweep passes run jacobi_fused_kernel<NL = 2>, plan id - FUSED2
    int pair_shape = -1;                // fastest two-sweep launch plan for this grid (timed once)
    int pair_plan_rb = 0;               // fastest plan of jacobi_pair_kernel itself: its red-black / damped passes (rbsor, mg level 0)
                                        // always run that kernel, also where the plain two-sweep passes went to the fused one
    int tuned_plan_two = -3, tuned_plan_three = -3;   // "launch_plans" values the choices were made under
    int tuned_fuse = -1, tuned_pair_shape_opt = -1;   // option values the two choices here were timed under
    int triple_alt = -1;                // >= 0: three sweeps per pass beat two on this grid (launch plan id)
    hipStream_t comm_stream = nullptr;  // z-slabs: EVERY transport call runs on this one stream (a communicator is never driven from
                                        // two streams); events order it against the compute stream
    hipEvent_t ev_edges = nullptr, ev_halo = nullptr, ev_int = nullptr, ev_c2x = nullptr;
    // z-slabs: the reach of the advection back-trace without a host synchronisation.  max |v_z| after each of the step's two
    // projections is reduced over the ranks and copied to pinned memory asynchronously; the host waits for the EVENT behind the copy
    // when it sizes the gather, by which time the device has long passed it (see step()).
    double* reach_pinned = nullptr;     // 3 x {sum, min, max}: after the first / second projection, at the start of the step
    hipEvent_t ev_reach[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_slack = nullptr;      // behind the independent work queued between a projection and the advection that needs its reach
    bool reach_posted[3] = {false, false, false};
    double vzmax_prev = -1.0;           // max |v_z| at the end of the previous step (= v_z_prev of this one), -1 = unknown
    double vzmax_end = -1.0;            // the same for the step that is running
    bool in_step = false;               // inside step(): the data flow between the solver's calls is known
    static constexpr int SLOT_POOL = 0, SLOT_GATHER = NPOOL, SLOT_MG = NPOOL + 4;   // FSIPC export slots: one per arena chunk
    static constexpr int NRED = 3 * 1024 + 18;   // reduction scratch + up to six {sum, min, max} results (0, 1: stats / trace_reach; 2..4: post_vzmax)

    explicit Engine(fs_sim* s) : S(s) {}

    long dense_cells() const { return (long)(g.W + 2) * (g.H + 2) * (g.D + 2); }

    int need_dense(size_t bytes)
    {
        if (bytes <= dense_bytes) return FS_OK;
        if (dense) HIP_TRY(hipFree(dense));
        dense = nullptr;
        dense_bytes = 0;
        HIP_TRY(hipMalloc(&dense, bytes));
        dense_bytes = bytes;
        return FS_OK;
    }

    int init()
    {
        const fs::Comm& cm = S->comm;
        g.W = S->W; g.H = S->H;
        g.D = cm.active() ? cm.local_depth(S->D) : S->D;
        g.sy = ((long)(g.W + 5) + 3) / 4 * 4;
        g.sz = g.sy * (g.H + 2);
        // z-slabs keep as many halo planes per side as the deepest fused pass has levels (it recomputes the lower
        // levels of the neighbours' boundary planes): three where the three-sweep kernel exists, else two
        g.zh = !cm.active() ? 1 : (std::is_same<T, float>::value && g.W <= 512 && g.D >= 3) ? 3 : 2;
        g.lead = fs::LEAD + (long)(g.zh - 1) * g.sz;
        g.n = g.sz * (g.D + 2 * g.zh) + 8;   // lead + tail so that a dwordx4 at the last ghost stays in bounds
        g.n = (g.n + 3) / 4 * 4;
        sc.zoff = cm.active() ? cm.z_offset(S->D) : 0;
        sc.Dglobal = S->D;
        sc.lo_wall = (!cm.active() || cm.rank == 0) ? 1 : 0;
        sc.hi_wall = (!cm.active() || cm.rank == cm.nranks - 1) ? 1 : 0;
        // The field arrays live in a few large allocations ("arena chunks"), not one hipMalloc each: a slab rank exports
        // every chunk to its neighbours as one hipIpc handle.  Many small exports alias after a few handles of one process,
        // and an export beyond 2 GiB never returns on the HIP runtime PyTorch bundles (both seen in round 3's bench
        // rehearsals), so a chunk holds as many arrays as fit 1 GiB.
        pool_stride = (g.n + 63) / 64 * 64;              // keeps every array's first interior cell 16-byte aligned
        pool_per = (int)std::max<size_t>(1, ARENA_CHUNK_BYTES / (pool_stride * sizeof(T)));
        if (pool_per > NPOOL) pool_per = NPOOL;
        for (int i = 0; i < NPOOL; i += pool_per) {
            const size_t cnt = (size_t)std::min(pool_per, NPOOL - i) * pool_stride;
            T* chunk = nullptr;
            HIP_TRY(hipMalloc((void**)&chunk, cnt * sizeof(T)));
            pool_chunks.push_back(chunk);
            pool_chunk_bytes.push_back(cnt * sizeof(T));
            HIP_TRY(hipMemsetAsync(chunk, 0, cnt * sizeof(T), S->stream));   // simulation.cpp:38-43
        }
        for (int i = 0; i < NPOOL; ++i) arr[i] = pool_chunks[(size_t)(i / pool_per)] + (size_t)(i % pool_per) * pool_stride + g.lead;
        for (int f = 0; f < FS_NFIELDS; ++f) slot[f] = f;
        uint8_t* fb = nullptr;
        HIP_TRY(hipMalloc((void**)&fb, g.n));
        HIP_TRY(hipMemsetAsync(fb, 0, g.n, S->stream));
        flags = fb + g.lead;
        uint8_t* kb = nullptr;
        HIP_TRY(hipMalloc((void**)&kb, g.n / 4 + 16));
        HIP_TRY(hipMemsetAsync(kb, 0, g.n / 4 + 16, S->stream));
        kill = kb + (g.lead - fs::LEAD) / 4;
        HIP_TRY(hipMalloc((void**)&red, NRED * sizeof(double)));
        if (!cm.active()) HIP_TRY(hipMalloc((void**)&coltab, sizeof(T) * 6 * (size_t)(g.H + 2) * (size_t)(g.D + 2)));
        if (cm.active()) {
            int lo_pri = 0, hi_pri = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
            HIP_TRY(hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, hi_pri));
            for (hipEvent_t* ev : { &ev_edges, &ev_halo, &ev_int, &ev_c2x, &ev_reach[0], &ev_reach[1], &ev_reach[2], &ev_slack })
                HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
            HIP_TRY(hipHostMalloc((void**)&reach_pinned, 9 * sizeof(double), hipHostMallocDefault));
            // FSIPC: the neighbours write straight into these arrays
            HIP_TRY(hipStreamSynchronize(S->stream));
            for (size_t k = 0; k < pool_chunks.size(); ++k)
                if (S->comm.register_buffer(SLOT_POOL + (int)k, pool_chunks[k], pool_chunk_bytes[k], false))
                    return fail(FS_ECOMM, "exporting the field arrays: %s", S->comm.last_error());
            if (S->comm_cus != 0) {
                // a second compute stream whose CU mask leaves CUs to the transport; mask bit i is CU i / 8 of XCD i % 8 (the
                // driver deals the bits round-robin over the XCDs), so clearing the top bits takes the same number from every XCD
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, S->device));
                const int ncu = prop.multiProcessorCount;
                int keep_free = S->comm_cus > 0 ? S->comm_cus : 8;
                if (keep_free > ncu / 2) keep_free = ncu / 2;
                std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
                for (int i = 0; i < ncu - keep_free; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
                HIP_TRY(hipExtStreamCreateWithCUMask(&S->stream_masked, (uint32_t)mask.size(), mask.data()));
                S->stream_full = S->stream;
                masked_cus = ncu - keep_free;
                if (S->comm_cus > 0) use_masked_stream(true);      // forced; "auto" decides with the overlap plan
            }
        }
        return FS_OK;
    }

    int masked_cus = 0;
    // switch the compute stream (everything queued on the old one first completes)
    int use_masked_stream(bool on)
    {
        hipStream_t want = on ? S->stream_masked : S->stream_full;
        if (!want || want == S->stream) return FS_OK;
        HIP_TRY(hipStreamSynchronize(S->stream));
        S->resolve_spans();
        S->stream = want;
        S->tune.cu_slots = on ? masked_cus : 256;
        S->cus_plan = on ? (S->comm_cus > 0 ? S->comm_cus : 8) : 0;
        pair_shape = -1;                                 // the launch plans depend on how many CUs a launch can fill
        return FS_OK;
    }

    ~Engine() override
    {
        for (T* c : pool_chunks) hipFree(c);
        if (flags) hipFree(flags - g.lead);
        if (kill) hipFree(kill - (g.lead - fs::LEAD) / 4);
        for (T* c : gather_chunks) hipFree(c);        // the four gathered advection sources
        if (dense) hipFree(dense);
        if (red) hipFree(red);
        if (coltab) hipFree(coltab);
        mg.release();
        for (hipEvent_t ev : { ev_edges, ev_halo, ev_int, ev_c2x, ev_reach[0], ev_reach[1], ev_reach[2], ev_slack })
            if (ev) hipEventDestroy(ev);
        if (reach_pinned) hipHostFree(reach_pinned);
        if (comm_stream) hipStreamDestroy(comm_stream);
    }

    // ---- array pool ------------------------------------------------------------------
    bool referenced(int id) const
    {
        for (int f = 0; f < FS_NFIELDS; ++f)
            if (slot[f] == id) return true;
        return held[id];
    }
    int acquire(int not_a = -1, int not_b = -1)
    {
        for (int i = 0; i < NPOOL; ++i)
            if (i != not_a && i != not_b && !referenced(i)) { held[i] = true; return i; }
        return -1;   // cannot happen: NPOOL covers the worst case of a step
    }
    bool shared_slot(int f) const
    {
        for (int k = 0; k < FS_NFIELDS; ++k)
            if (k != f && slot[k] == slot[f]) return true;
        return false;
    }
    // give field f an array of its own (same contents)
    int unalias(int f)
    {
        if (!shared_slot(f)) return FS_OK;
        int id = acquire();
        if (id < 0) return fail(FS_ENOMEM, "array pool exhausted");
        fs::launch_copy<T>(S->stream, g, arr[slot[f]], arr[id]);
        slot[f] = id;
        held[id] = false;
        return FS_OK;
    }

    // Host-side writes (fs_set_field, fs_add_density, ...) reach only this rank's planes; before
    // the next kernel every rank refreshes the halo copies of all fields (collective).
    int ensure_halos()
    {
        if (!halos_dirty || !S->comm.active()) { halos_dirty = false; return FS_OK; }
        halos_dirty = false;
        bool done[NPOOL] = {false};
        for (int f = 0; f < FS_NFIELDS; ++f) {
            if (f == FS_OBS || done[slot[f]]) continue;
            done[slot[f]] = true;
            int rc = halo(arr[slot[f]]);
            if (rc) return rc;
        }
        return FS_OK;
    }

    int ensure_flags()
    {
        {
            int rc = ensure_halos();
            if (rc) return rc;
        }
        if (!flags_dirty) return FS_OK;
        int rc = unalias(FS_OBS);
        if (rc) return rc;
        if (S->comm.active() && (rc = halo(arr[slot[FS_OBS]]))) return rc;
        ScopedSpan sp(S, FAM_MISC);
        fs::launch_build_flags<T>(S->stream, g, sc, arr[slot[FS_OBS]], flags);
        fs::launch_build_kill(S->stream, g, sc, flags, kill);
        flags_dirty = false;
        mg_current = false;
        return FS_OK;
    }

    // Every transport call of a slab rank goes through here: queued on the communication stream behind everything the
    // compute stream holds so far, and the compute stream continues behind it.  One stream per communicator: RCCL orders a
    // communicator's operations by issue order, and two streams sharing one would be serialised in ways the schedule
    // does not show (round-2 verdict).
    // Schedule 0 (a pass, then its exchange) overlaps nothing, so there the one stream is the compute stream itself and no
    // event is needed: a cross-stream dependency costs about 14 us each way on this runtime (measured: 240 passes per step
    // with an event pair each = 6.7 ms of a 26 ms slab step, profiles/r3e_*).
    bool comm_on_compute_stream() const { return S->overlap_plan == 0 || S->overlap_plan == 3; }
    template <class F>
    int comm_op(F&& op, const char* what)
    {
        if (comm_on_compute_stream()) {
            if (op(S->stream)) return fail(FS_ECOMM, "%s failed: %s", what, S->comm.last_error());
            return FS_OK;
        }
        HIP_TRY(hipEventRecord(ev_c2x, S->stream));
        HIP_TRY(hipStreamWaitEvent(comm_stream, ev_c2x, 0));
        if (op(comm_stream)) return fail(FS_ECOMM, "%s failed: %s", what, S->comm.last_error());
        HIP_TRY(hipEventRecord(ev_halo, comm_stream));
        HIP_TRY(hipStreamWaitEvent(S->stream, ev_halo, 0));
        return FS_OK;
    }

    int halo(T* a)
    {
        if (!S->comm.active()) return FS_OK;
        ScopedSpan sp(S, FAM_COMM);
        return comm_op([&](hipStream_t st) { return S->comm.exchange_halo(st, a, g, sizeof(T), S->D, g.zh); }, "halo exchange");
    }

    // ---- linearSolver (simulation.cpp:251-273) -----------------------------------------
    // One pass over memory that applies `levels` (1, 2 or 3) Jacobi sweeps to planes zf..zl (and, with
    // second >= 0, to the equally long range starting there).
    // push: the pass also stores the planes its z neighbours need into their halo planes (plain Jacobi passes, FSIPC)
    void launch_pass(hipStream_t st, int levels, bool rb, const T* src_, const T* rhs_, T* dst_, int b, T a, T inv_c, int zf,
                     int zl, int second = -1, const fs::PeerPush* push = nullptr)
    {
        const T omega = rb ? rb_omega : (T)0;
        if (levels == 3)
            fs::launch_jacobi_fused<T>(st, S->tune, g, sc, 3, src_, rhs_, dst_, kill, b, a, inv_c, zf, zl, triple_alt, second, push);
        else if (levels == 2 && !rb && pair_shape >= FUSED2)
            fs::launch_jacobi_fused<T>(st, S->tune, g, sc, 2, src_, rhs_, dst_, kill, b, a, inv_c, zf, zl, pair_shape - FUSED2, second, push);
        else if (levels == 2)
            fs::launch_jacobi_pair<T>(st, S->tune, g, sc, src_, rhs_, dst_, kill, b, a, inv_c, zf, zl,
                                      pair_shape >= FUSED2 ? pair_plan_rb : pair_shape, second, omega, rb_damped, push);
        else
            fs::launch_jacobi<T>(st, S->tune, g, sc, src_, rhs_, dst_, kill, b, a, inv_c, zf, zl, second, push);
    }
    bool two_sweep_kernels() const
    {
        return fs::pair_supported<T>(S->tune, g, sc) || fs::fused_supported<T>(S->tune, g, sc, 2);
    }
    int ensure_tuned(int cur, int rhs, int b, T a, T inv_c)
    {
        if (two_sweep_kernels() || fs::fused_supported<T>(S->tune, g, sc, 3)) {
            const int opt = S->tune.pair_shape + 16 * S->tune.two_kind;
            if (!(pair_shape >= 0 && tuned_fuse == S->tune.fuse && tuned_pair_shape_opt == opt && tuned_plan_two == S->plan_two &&
                  tuned_plan_three == S->plan_three)) {
                tuned_fuse = S->tune.fuse;
                tuned_pair_shape_opt = opt;
                tuned_plan_two = S->plan_two;
                tuned_plan_three = S->plan_three;
                int rc = choose_pair_shape(cur, rhs, b, a, inv_c);
                if (rc) return rc;
            }
        }
        if (S->comm.active() && S->overlap_plan < 0) return choose_overlap(cur, rhs, b, a, inv_c);
        return FS_OK;
    }

    // all ranks have finished everything queued so far (host-blocking; tuning and dumps only)
    int slab_barrier()
    {
        HIP_TRY(hipStreamSynchronize(S->stream));
        if (S->comm.barrier(comm_on_compute_stream() ? S->stream : comm_stream, red)) return fail(FS_ECOMM, "barrier: %s", S->comm.last_error());
        return FS_OK;
    }
    // the largest `v` of any rank, the same bits on every rank (host-blocking; tuning only)
    int rank_max(double v, double* out)
    {
        double h[3] = { v, v, v };
        double* d3 = red + 3 * 1024;
        HIP_TRY(hipMemcpyAsync(d3, h, sizeof h, hipMemcpyHostToDevice, S->stream));
        int rc = comm_op([&](hipStream_t st) { return S->comm.reduce_stats(st, d3, g, S->D); }, "all-reduce");
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(h, d3, sizeof h, hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipStreamSynchronize(S->stream));
        *out = h[2];
        return FS_OK;
    }

    // "overlap" = "auto" (and "comm_cus" = "auto"): the communication schedules are chosen the way launch plans are -- by the
    // clock, once, on the transport the run really uses.  Every candidate runs a chain of the solver's deepest passes with
    // their exchanges between two barriers; what counts is the slowest rank's time (all-reduced, so that every rank takes
    // the same decision: the schedule must not diverge).  A later candidate has to win by 1.5 %.  The bits do not depend on
    // the choice (tests/test_gpu_slabs.py runs every candidate).
    int choose_overlap(int src, int rhs, int b, T a, T inv_c)
    {
        const bool auto_cus = S->comm_cus < 0 && S->stream_masked;
        const bool can2 = two_sweep_kernels(), can3 = triple_alt >= 0;
        const int lv = can3 ? 3 : can2 ? 2 : 1;
        std::vector<int> modes;
        if (S->overlap >= 0) modes.push_back(S->overlap);
        else if (g.D < 2 * lv + 8) modes.push_back(0);                 // too thin for a boundary/interior split: one schedule
        else modes = { 1, 0, 2 };
        if (S->overlap < 0 && S->comm.can_push()) modes.push_back(3);   // FSIPC: passes that store into the neighbours' halos
        if (modes.size() == 1 && !auto_cus) {
            S->overlap_plan = modes[0];
            if (S->cus_plan < 0) S->cus_plan = 0;
            return FS_OK;
        }
        const int t0 = acquire(src, rhs), t1 = acquire(src, rhs);
        struct Release {
            bool* held; int a, b; hipEvent_t e0 = nullptr, e1 = nullptr;
            ~Release() { if (a >= 0) held[a] = false; if (b >= 0) held[b] = false; if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
        } rel{held, t0, t1};
        if (t0 < 0 || t1 < 0) return fail(FS_ENOMEM, "array pool exhausted");
        HIP_TRY(hipEventCreate(&rel.e0));
        HIP_TRY(hipEventCreate(&rel.e1));
        const bool tune_log = getenv("FS_TUNE_LOG") != nullptr;
        const int NP = 6;
        double best = 1e300;
        int best_mode = modes[0], best_mask = 0;
        int rc = FS_OK;
        for (int mask = 0; mask < (auto_cus ? 2 : 1); ++mask) {
            if (auto_cus) {
                if ((rc = use_masked_stream(mask == 1))) return rc;
                if ((rc = choose_pair_shape(src, rhs, b, a, inv_c))) return rc;   // launch plans for that many CUs
            }
            for (int mode : modes) {
                double ms = 0.0;
                HIP_TRY(hipDeviceSynchronize());         // the transport's stream changes with the schedule: nothing may be in flight
                S->overlap_plan = mode;
                for (int rep = 0; rep < 2; ++rep) {          // the second chain is the timed one
                    if ((rc = slab_barrier())) return rc;
                    HIP_TRY(hipEventRecord(rel.e0, S->stream));
                    int from = src;
                    for (int p = 0; p < NP; ++p) {
                        const int to = (p & 1) ? t1 : t0;
                        if ((rc = slab_pass(mode, lv, lv, p == 0, p + 1 == NP, false, arr[from], arr[rhs], arr[to], b, a, inv_c))) return rc;
                        from = to;
                    }
                    HIP_TRY(hipEventRecord(rel.e1, S->stream));
                    HIP_TRY(hipEventSynchronize(rel.e1));
                    float t = 0;
                    HIP_TRY(hipEventElapsedTime(&t, rel.e0, rel.e1));
                    ms = (double)t / NP;
                }
                double worst = ms;
                if ((rc = rank_max(ms, &worst))) return rc;
                S->overlap_ms[4 * mask + mode] = worst;
                if (tune_log)
                    fprintf(stderr, "fluidsim tune: rank %d overlap=%d cu mask %s: %.4f ms per pass here, %.4f on the slowest rank\n",
                            S->comm.rank, mode, mask ? "on" : "off", ms, worst);
                if (worst < best * 0.985) { best = worst; best_mode = mode; best_mask = mask; }
            }
        }
        if (auto_cus) {
            if ((rc = use_masked_stream(best_mask == 1))) return rc;
            if ((rc = choose_pair_shape(src, rhs, b, a, inv_c))) return rc;
        }
        if (S->cus_plan < 0) S->cus_plan = 0;
        HIP_TRY(hipDeviceSynchronize());
        S->overlap_plan = best_mode;
        if (S->comm.rank == 0 && !S->quiet)
            fprintf(stderr, "fluidsim: communication schedule overlap=%d, %d CUs kept free (timed: slowest rank %.3f ms per %d-sweep pass)\n",
                    best_mode, S->cus_plan, best, lv);
        return FS_OK;
    }

    // A solve in progress: the list of its passes and where it stands.  step() interleaves the (dead) density solve of
    // a slab run with the velocity advection, so a solve can be paused between two passes.
    struct SolveRun {
        std::vector<int> plan;       // levels of each pass
        int next = 0;                // first pass not yet launched
        int b = 0, rhs = -1, src = -1;
        bool src_temp = false, rb = false, damped = false;
        T a = (T)0, inv_c = (T)1, omega = (T)1;
    };

    // One pass of a slab rank and the exchange of its boundary planes: `e` planes per side go to the neighbours (what their
    // next pass needs).  `first` / `last`: first / last pass of a run of passes queued back to back (the two-stream
    // schedule chains its events from pass to pass).
    int slab_pass(int mode, int lv, int e, bool first, bool last, bool rb, const T* src, const T* rhs, T* dst, int b, T a, T inv_c)
    {
        auto exchange = [&](hipStream_t st) { return S->comm.exchange_halo(st, dst, g, sizeof(T), S->D, e); };
        if (mode == 3 && !rb && S->comm.can_push()) {
            // Push (FSIPC): the pass stores the `e` outermost planes per side straight into the neighbours' halo planes; a
            // handshake on the same stream ("my pass is complete" both ways) is all that separates it from the next
            // pass -- no boundary launch, no copy, no second stream, no event.  The neighbour's halo of this array is
            // free: its last reader there was the neighbour's previous pass, which the previous handshake covered.
            fs::PeerPush pp;
            if (S->comm.peer_push(dst, g, sizeof(T), e, &pp)) return fail(FS_ECOMM, "push exchange: %s", S->comm.last_error());
            launch_pass(S->stream, lv, false, src, rhs, dst, b, a, inv_c, 1, g.D, -1, &pp);
            if (S->comm.handshake(S->stream)) return fail(FS_ECOMM, "handshake: %s", S->comm.last_error());
            return FS_OK;
        }
        if (mode == 3) mode = 0;                          // a pass the push kernels do not cover (red-black): pass, then copy exchange
        if (g.D < 2 * e + 8) mode = 0;                    // too thin to split into boundary and interior
        const int in_lo = sc.lo_wall ? 1 : e + 1, in_hi = sc.hi_wall ? g.D : g.D - e;
        auto boundary = [&](hipStream_t st) {
            if (!sc.lo_wall && !sc.hi_wall) launch_pass(st, lv, rb, src, rhs, dst, b, a, inv_c, 1, e, g.D - e + 1);
            else if (!sc.lo_wall) launch_pass(st, lv, rb, src, rhs, dst, b, a, inv_c, 1, e);
            else if (!sc.hi_wall) launch_pass(st, lv, rb, src, rhs, dst, b, a, inv_c, g.D - e + 1, g.D);
        };
        if (mode == 2) {
            // Two streams: the boundary regions of pass k and its interior only depend on pass k-1, not on each other, so
            // they are queued side by side -- boundary launch + exchange on the communication stream, interior on the
            // compute stream.  Interior k reads the halo planes that exchange k-1 fills (a pass of lv levels reads lv
            // planes beyond its range) and overwrites what boundary k-1 read: it waits for the event behind exchange k-1
            // (round-2 advice: waiting for boundary k-1 alone was a race whenever a pass had more levels than the next).
            if (first) HIP_TRY(hipEventRecord(ev_int, S->stream));                 // everything queued so far
            else HIP_TRY(hipStreamWaitEvent(S->stream, ev_halo, 0));               // boundary k-1 and exchange k-1
            HIP_TRY(hipStreamWaitEvent(comm_stream, ev_int, 0));                   // boundary k reads interior k-1
            boundary(comm_stream);
            launch_pass(S->stream, lv, rb, src, rhs, dst, b, a, inv_c, in_lo, in_hi);
            HIP_TRY(hipEventRecord(ev_int, S->stream));
            if (exchange(comm_stream)) return fail(FS_ECOMM, "halo exchange failed: %s", S->comm.last_error());
            HIP_TRY(hipEventRecord(ev_halo, comm_stream));
            if (last) HIP_TRY(hipStreamWaitEvent(S->stream, ev_halo, 0));          // whoever reads the result next runs on the compute stream
        } else if (mode == 1) {
            // Boundary planes first (both regions in one launch); their exchange travels on the communication stream
            // while the interior planes are computed; the next launch waits for it.
            boundary(S->stream);
            HIP_TRY(hipEventRecord(ev_edges, S->stream));
            launch_pass(S->stream, lv, rb, src, rhs, dst, b, a, inv_c, in_lo, in_hi);
            HIP_TRY(hipStreamWaitEvent(comm_stream, ev_edges, 0));
            if (exchange(comm_stream)) return fail(FS_ECOMM, "halo exchange failed: %s", S->comm.last_error());
            HIP_TRY(hipEventRecord(ev_halo, comm_stream));
            HIP_TRY(hipStreamWaitEvent(S->stream, ev_halo, 0));
        } else {
            launch_pass(S->stream, lv, rb, src, rhs, dst, b, a, inv_c, 1, g.D);
            return comm_op(exchange, "halo exchange");   // schedule 0: on the compute stream; a thin slab of schedule 1 / 2: through the events
        }
        return FS_OK;
    }

    // `cur` holds the initial iterate (may equal rhs when the caller aliased a snapshot).
    // smoother = true: `sweeps` passes of two 6/7-damped Jacobi sweeps each (the level-0 smoothing step of solver=mg)
    int solve_begin(SolveRun& r, int b, int cur, int rhs, T a, T c, int sweeps, bool smoother = false)
    {
        r = SolveRun();
        r.b = b; r.rhs = rhs; r.src = cur; r.a = a;
        r.inv_c = (T)1 / c;                              // cRecip, :257
        // solver=rbsor: every iteration is one pass of the pair kernel (its two levels are the two colours)
        r.rb = smoother || (S->solver == FS_SOLVER_RBSOR);
        r.omega = smoother ? (T)6 / (T)7 : (T)S->omega;
        r.damped = smoother;
        if (r.rb && !fs::pair_supported<T>(S->tune, g, sc))
            return fail(FS_EINVAL, "solver=rbsor / mg needs rows of at most 1024 cells and sweep_fuse >= 2");
        int rc = ensure_tuned(cur, rhs, b, a, r.inv_c);
        if (rc) return rc;
        // The passes of this solve: three sweeps per pass while at least three remain (where that kernel
        // exists and was found faster), then two, then one.  Under z-slabs every rank derives the same list
        // (it fixes the depth of every halo exchange).
        const bool can2 = two_sweep_kernels(), can3 = triple_alt >= 0;
        for (int left = sweeps; left > 0;) {
            if (r.rb) { r.plan.push_back(2); left -= 1; continue; }      // an rbsor iteration runs as a two-level pass
            const int lv = (can3 && left >= 3) ? 3 : (can2 && left >= 2) ? 2 : 1;
            r.plan.push_back(lv);
            left -= lv;
        }
        const int npass = (int)r.plan.size();
        if (npass > 0 && S->comm.active() && (r.plan[0] > 1 || npass > 1)) {
            // a fused pass recomputes the lower levels of the neighbours' boundary planes: it reads the
            // right-hand side there, so its halo planes must be current
            if ((rc = halo(arr[rhs]))) return rc;
        } else if (npass > 0 && S->overlap_plan == 3 && S->comm.can_push()) {
            // push schedule: the first pass writes into the neighbours' arrays, so whatever they queued before this solve
            // has to be complete (the exchange above is such a handshake; without it, one is issued)
            if (S->comm.handshake(S->stream)) return fail(FS_ECOMM, "handshake: %s", S->comm.last_error());
        }
        return FS_OK;
    }

    // launch passes next .. upto-1
    int solve_passes(SolveRun& r, int upto)
    {
        const int npass = (int)r.plan.size();
        if (upto > npass) upto = npass;
        if (r.next >= upto) return FS_OK;
        rb_omega = r.omega;
        rb_damped = r.damped;
        int span = -1, span_fam = -1;
        long span_launches = 0;
        auto close_span = [&]() {
            if (span >= 0) S->span_end(span, span_launches);
            span = -1;
            span_launches = 0;
        };
        const int first = r.next;
        for (int i = first; i < upto; ++i) {
            const int lv = r.plan[i];
            int dst = acquire(r.src, r.rhs);
            if (dst < 0) return fail(FS_ENOMEM, "array pool exhausted");
            // one event pair around each run of equal passes (an event pair per launch costs 2 % at 512^3 and
            // 16 % at 256^3); launches are counted so that time / launches is the mean launch time.  On slabs
            // the exchanges fall inside the span.
            const int fam = lv == 3 ? FAM_TRIPLE : lv == 2 ? FAM_PAIR : FAM_SWEEP;
            if (span >= 0 && fam != span_fam) close_span();
            if (span < 0) { span = S->span_begin(fam); span_fam = fam; }
            ++span_launches;
            if (!S->comm.active()) {
                launch_pass(S->stream, lv, r.rb, arr[r.src], arr[r.rhs], arr[dst], r.b, r.a, r.inv_c, 1, g.D);
            } else {
                // planes a neighbour needs of this pass's result: as many as its next pass has levels; after
                // the last pass the halos are brought to their full depth (what every other kernel assumes)
                const int e = (i + 1 < npass) ? r.plan[i + 1] : g.zh;
                int rc = slab_pass(S->overlap_plan, lv, e, i == first, i + 1 == upto, r.rb, arr[r.src], arr[r.rhs], arr[dst], r.b, r.a,
                                   r.inv_c);
                if (rc) { held[dst] = false; return rc; }
            }
            if (r.src_temp) held[r.src] = false;
            r.src = dst;
            r.src_temp = true;
        }
        close_span();
        r.next = upto;
        return FS_OK;
    }

    // the id of the array holding the result (held)
    int solve_end(SolveRun& r, int* result)
    {
        int rc = solve_passes(r, (int)r.plan.size());
        if (rc) return rc;
        if (!r.src_temp) held[r.src] = true;
        *result = r.src;
        return FS_OK;
    }

    int solve(int b, int cur, int rhs, T a, T c, int sweeps, int* result, bool smoother = false)
    {
        if (S->solver == FS_SOLVER_GS_LEX) {
            if (S->comm.active()) return fail(FS_EINVAL, "gs_lex is a single-GPU verification mode");
            if (cur == rhs) return fail(FS_EINVAL, "gs_lex needs distinct field and prev arrays");
            ScopedSpan sp(S, FAM_SWEEP, sweeps);
            if (sweeps > 0) fs::launch_gs_lex<T>(S->stream, g, arr[cur], arr[rhs], flags, b, a, (T)1 / c, sweeps);
            held[cur] = true;
            *result = cur;
            return FS_OK;
        }
        SolveRun r;
        int rc = solve_begin(r, b, cur, rhs, a, c, sweeps, smoother);
        if (rc) return rc;
        return solve_end(r, result);
    }

    // Times the candidate launch plans of the two-sweep kernels on this grid -- the pair kernel's workgroup
    // shapes and the fused kernel's (ids FUSED2 + shape), each x the three best z-chunk counts of the
    // launcher's model -- two launches each into a scratch array, the second one timed, and keeps the
    // fastest; then the same for the three-sweep kernel, which is used where a sweep costs less that way.
    // Every plan computes the same bits, so this only ever changes speed.
    int choose_pair_shape(int src, int rhs, int b, T a, T inv_c)
    {
        pair_shape = 0;
        pair_plan_rb = 0;
        triple_alt = -1;
        // "launch_plans" = "<two-sweep id>,<three-sweep id>" (as fs_get_int "pair_shape" / "triple_plan" report them): replay
        // the plans of another run instead of timing (tools/make_profiles.sh: the counter passes must run the plans the
        // bench line ran, and the clock is different under counter collection); -1 = time as usual / no such kernel.
        // An id that names a kernel or shape this grid does not have is refused, not run.
        const bool replay = S->plan_two >= -1 && S->plan_three >= -1 && (S->plan_two >= 0 || S->plan_three >= 0);
        if (replay) {
            if (S->plan_two >= FUSED2) {
                const int id = S->plan_two - FUSED2;
                if (!fs::fused_supported<T>(S->tune, g, sc, 2) || (id & 7) >= fs::fused_shape_count<T>(g, 2) || (id >> 3) > 2)
                    return fail(FS_EINVAL, "launch_plans: two-sweep plan %d names a fused-kernel shape this grid does not have", S->plan_two);
            } else if (S->plan_two >= 0) {
                if (!fs::pair_supported<T>(S->tune, g, sc) || (S->plan_two & 7) >= fs::pair_shape_count<T>(g) || (S->plan_two >> 3) > 2)
                    return fail(FS_EINVAL, "launch_plans: two-sweep plan %d names a pair-kernel shape this grid does not have", S->plan_two);
            }
            if (S->plan_three >= 0 && fs::fused_supported<T>(S->tune, g, sc, 3) &&
                ((S->plan_three & 7) >= fs::fused_shape_count<T>(g, 3) || (S->plan_three >> 3) > 2))
                return fail(FS_EINVAL, "launch_plans: three-sweep plan %d names a shape this grid does not have", S->plan_three);
            if (S->plan_two >= 0) pair_shape = S->plan_two;
            if (pair_shape < FUSED2) pair_plan_rb = pair_shape;
            triple_alt = (S->plan_three >= 0 && fs::fused_supported<T>(S->tune, g, sc, 3)) ? S->plan_three : -1;
            return FS_OK;
        }
        int tmp = acquire(src, rhs);
        if (tmp < 0) return fail(FS_ENOMEM, "array pool exhausted");
        struct Release {                                   // error paths must not leak the scratch arrays or the events
            bool* held; int id, id2 = -1; hipEvent_t e0 = nullptr, e1 = nullptr;
            ~Release() { held[id] = false; if (id2 >= 0) held[id2] = false; if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
        } rel{held, tmp};
        if (src == rhs) {
            // the first solve of a step reads iterate and right-hand side from ONE array (the snapshot alias): a
            // third less HBM traffic than every later pass, which would rank the candidates for the wrong regime
            const int copy = acquire(src, tmp);
            if (copy < 0) return fail(FS_ENOMEM, "array pool exhausted");
            rel.id2 = copy;
            fs::launch_copy<T>(S->stream, g, arr[src], arr[copy]);
            rhs = copy;
        }
        HIP_TRY(hipEventCreate(&rel.e0));
        HIP_TRY(hipEventCreate(&rel.e1));
        hipEvent_t e0 = rel.e0, e1 = rel.e1;
        const bool tune_log = getenv("FS_TUNE_LOG") != nullptr;   // development: print every candidate's time
        auto timed = [&](int levels, int cand, float* ms) -> int {
            const int keep_pair = pair_shape, keep_triple = triple_alt;
            if (levels == 3) triple_alt = cand; else pair_shape = cand;
            int rc = FS_OK;
            for (int rep = 0; rep < 2 && !rc; ++rep) {
                if (hipEventRecord(e0, S->stream) != hipSuccess) { rc = fail(FS_EHIP, "hipEventRecord"); break; }
                launch_pass(S->stream, levels, false, arr[src], arr[rhs], arr[tmp], b, a, inv_c, 1, g.D);
                if (hipEventRecord(e1, S->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                    hipEventElapsedTime(ms, e0, e1) != hipSuccess)
                    rc = fail(FS_EHIP, "timing a sweep launch plan failed: %s", hipGetErrorString(hipGetLastError()));
            }
            pair_shape = keep_pair;
            triple_alt = keep_triple;
            if (tune_log && !rc)
                fprintf(stderr, "fluidsim tune: %dx%dx%d %s levels=%d plan=%d  %.4f ms per pass\n", g.W, g.H, g.D,
                        sizeof(T) == 8 ? "fp64" : "fp32", levels, cand, *ms);
            return rc;
        };
        float best = 1e30f, best_pair = 1e30f;
        int best_cand = -1;
        auto consider2 = [&](int cand) -> int {
            float ms = 1e30f;
            int rc = timed(2, cand, &ms);
            if (rc) return rc;
            // a later candidate has to win by 1.5 %: plans within the noise of each other must not flip from run to run
            // (the chosen plan is part of what profiles/sweep_traffic.json is stamped with)
            if (ms < best * 0.985f) { best = ms; best_cand = cand; }
            if (cand < FUSED2 && ms < best_pair * 0.985f) { best_pair = ms; pair_plan_rb = cand; }
            return FS_OK;
        };
        // options: pair_shape > 0 forces a workgroup shape of the pair kernel, two_sweep_kernel one of the two kernels
        const bool have_fused2 = fs::fused_supported<T>(S->tune, g, sc, 2);
        const bool forced_pair = S->tune.pair_shape > 0 || S->tune.two_kind == 1 || !have_fused2;
        const bool forced_fused = !forced_pair && S->tune.two_kind == 2;
        if (fs::pair_supported<T>(S->tune, g, sc) && !forced_fused)
            for (int shape = 0; shape < fs::pair_shape_count<T>(g); ++shape)
                for (int alt = 0; alt < 3; ++alt) {      // candidate id = shape + 8 * (rank of the chunk count)
                    int rc = consider2(shape + 8 * alt);
                    if (rc) return rc;
                }
        if (have_fused2 && !forced_pair)
            for (int shape = 0; shape < fs::fused_shape_count<T>(g, 2); ++shape)
                for (int alt = 0; alt < 3; ++alt) {
                    int rc = consider2(FUSED2 + shape + 8 * alt);
                    if (rc) return rc;
                }
        if (best_cand >= 0) pair_shape = best_cand;
        // three sweeps per pass, where the kernel exists for this grid: keep it if a sweep costs less
        if (fs::fused_supported<T>(S->tune, g, sc, 3)) {
            float best3 = 1e30f;
            int alt3 = -1;
            for (int shape = 0; shape < fs::fused_shape_count<T>(g, 3); ++shape)
                for (int alt = 0; alt < 3; ++alt) {
                    const int cand = shape + 8 * alt;
                    float ms = 1e30f;
                    int rc = timed(3, cand, &ms);
                    if (rc) return rc;
                    if (ms < best3 * 0.985f) { best3 = ms; alt3 = cand; }
                }
            // fuse 4 forces it (tests, tuning); z-slab ranks must all take the same decision (it fixes the
            // exchange schedule), so there it is not left to each rank's clock
            if (S->tune.fuse >= 4 || S->comm.active() || best_cand < 0 || best3 / 3.0f < best / 2.0f) triple_alt = alt3;
        }
        return FS_OK;
    }

    // assign the result of a solve to a field slot
    void adopt(int field, int id)
    {
        slot[field] = id;
        held[id] = false;
    }

    T diffusion_a() const
    {
        // simulation.cpp:282: dt * diff * width * height * depth, left to right
        return (T)S->dt * (T)S->diff * (T)S->W * (T)S->H * (T)S->D;
    }

    int linear_solver(int b, int field, int prev, float a, float c) override
    {
        if (!in_step) vzmax_prev = -1.0;             // a call from outside step(): what is known about v_z is void
        int rc = ensure_flags();
        if (rc) return rc;
        if (S->solver == FS_SOLVER_GS_LEX && (rc = unalias(field))) return rc;
        int res;
        if (S->solver == FS_SOLVER_MG && b == 0 && a == 1.0f && c == 6.0f) {
            // the pressure equation's coefficients (:320): V-cycles; every other system (diffusion) is relaxed as under jacobi
            if ((rc = unalias(field)) || (rc = unalias(prev))) return rc;
            rc = multigrid_solve(field, prev, &res);
        } else {
            rc = solve(b, slot[field], slot[prev], (T)a, (T)c, S->acc, &res);
        }
        if (rc) return rc;
        adopt(field, res);
        return FS_OK;
    }

    int diffuse_T(int b, int field, int prev)
    {
        const T a = diffusion_a();
        int rc = ensure_flags();
        if (rc) return rc;
        if (S->solver == FS_SOLVER_GS_LEX && (rc = unalias(field))) return rc;
        int res;
        rc = solve(b, slot[field], slot[prev], a, (T)1 + (T)6 * a, S->acc, &res);   // :283
        if (rc) return rc;
        adopt(field, res);
        return FS_OK;
    }
    int diffuse(int b, int field, int prev) override { if (!in_step) vzmax_prev = -1.0; return diffuse_T(b, field, prev); }

    int set_bounds(int b, int field) override
    {
        if (!in_step) vzmax_prev = -1.0;             // a call from outside step(): what is known about v_z is void
        int rc = ensure_flags();
        if (rc) return rc;
        if ((rc = unalias(field))) return rc;
        ScopedSpan sp(S, FAM_BOUNDS, 2);
        fs::launch_set_bounds<T>(S->stream, g, sc, arr[slot[field]], flags, b);
        return halo(arr[slot[field]]);
    }

    // solver=mg (NOT the reference's arithmetic; defined in oracle/cpu_ref_mg.h): mg_cycles V-cycles on the pressure
    // equation of :320.  Level 0 is smoothed by the reference's update as damped Jacobi sweeps, two per pass of the pair
    // kernel (its red-black instantiation is 1.6x slower and smooths no better here), the coarse levels live in
    // multigrid.hip.  Single GPU.
    int multigrid_levels() const override { return mg.levels(); }
    // the transport's part in the coarse levels of a slab run (multigrid.h): one-plane halo refreshes of distributed levels,
    // all-gathers at the seam to the levels every rank holds whole
    fs::MgHooks<T> mg_hooks()
    {
        fs::MgHooks<T> h;
        h.halo = [this](const fs::MgLevel<T>& l, T* a) -> int {
            fs::GridDesc lg = g;
            lg.sz = l.sz; lg.sy = l.sy; lg.D = l.D; lg.W = l.W; lg.H = l.H;
            return comm_op([&](hipStream_t st) { return S->comm.exchange_halo(st, a, lg, sizeof(T), l.D * S->comm.nranks, 1); },
                           "halo exchange of a multigrid level");
        };
        h.gather = [this](const fs::MgLevel<T>& l, T* a, int dl, int zoff) -> int {
            fs::GridDesc lg = g;
            lg.sz = l.sz; lg.sy = l.sy; lg.D = dl; lg.W = l.W; lg.H = l.H;
            return comm_op([&](hipStream_t st) { return S->comm.all_gather_planes(st, a + (long)zoff * l.sz, a, lg, l.D, sizeof(T)); },
                           "all-gather of a multigrid level");
        };
        return h;
    }
    int multigrid_solve(int field, int prev, int* result)
    {
        if (S->mg_cycles > 0 && !fs::pair_supported<T>(S->tune, g, sc))
            return fail(FS_EINVAL, "solver=mg needs rows of at most 1024 cells and sweep_fuse >= 2");
        const bool slabs = S->comm.active();
        fs::MgHooks<T> hooks;
        if (slabs) hooks = mg_hooks();
        if (!mg_current) {
            ScopedSpan sp(S, FAM_MG);
            int brc = mg.build(S->stream, g, sc, flags, slabs ? S->comm.nranks : 1, slabs ? S->comm.rank : 0, S->mg_min_planes,
                               slabs ? &hooks : nullptr);
            if (brc == -1) {
                // a fresh allocation on a slab rank: the neighbours (halo planes) and, at the seam, all ranks write into it
                if (S->comm.register_buffer(SLOT_MG, mg.pool, mg.pool_elems * sizeof(T), true))
                    return fail(FS_ECOMM, "exporting the multigrid levels: %s", S->comm.last_error());
                brc = mg.build(S->stream, g, sc, flags, S->comm.nranks, S->comm.rank, S->mg_min_planes, &hooks);
            }
            if (brc == 2) return fail(FS_EINVAL, "solver=mg on z-slabs needs an even number of planes per rank (%d)", g.D);
            if (brc == 3) return FS_ECOMM;               // the hook has set the message
            if (brc) return fail(FS_EHIP, "multigrid levels: %s", hipGetErrorString(hipGetLastError()));
            mg_current = true;
        }
        const int own = slot[field], rhs = slot[prev];
        if (own == rhs) return fail(FS_EINVAL, "solver=mg needs distinct field and prev arrays");
        int cur = own;
        auto smooth = [&](int n) -> int {
            if (n <= 0) return FS_OK;
            int res;
            int rc = solve(0, cur, rhs, (T)1, (T)6, n, &res, true);
            if (rc) return rc;
            if (cur != own && cur != res) held[cur] = false;
            cur = res;
            return FS_OK;
        };
        for (int cyc = 0; cyc < S->mg_cycles; ++cyc) {
            int rc;
            if (mg.levels() < 2) {                           // a grid that cannot be halved: the "coarsest level" is level 0
                if ((rc = smooth(S->mg_coarse))) return rc;
                continue;
            }
            if ((rc = smooth(S->mg_pre))) return rc;
            {
                ScopedSpan sp(S, FAM_MG);
                const int crc = mg.coarse_correction(S->stream, g, sc, flags, arr[cur], arr[rhs], S->mg_pre, S->mg_post, S->mg_coarse,
                                                     slabs ? &hooks : nullptr);
                if (crc == 3) return FS_ECOMM;
                if (crc) return fail(FS_EHIP, "multigrid cycle: %s", hipGetErrorString(hipGetLastError()));
            }
            // the correction changed this rank's planes of p: the neighbours' copies of its boundary planes are stale
            if (slabs && (rc = halo(arr[cur]))) return rc;
            if ((rc = smooth(S->mg_post))) return rc;
        }
        if (cur == own) held[own] = true;                 // adopt() releases it again
        *result = cur;
        return FS_OK;
    }

    // ---- project (simulation.cpp:289-362) ----------------------------------------------
    int project() override
    {
        if (!in_step) vzmax_prev = -1.0;             // a call from outside step(): what is known about v_z is void
        int rc = ensure_flags();
        if (rc) return rc;
        for (int f : { FS_VX, FS_VY, FS_VZ, FS_PRESSURE, FS_DIVERGENCE })
            if ((rc = unalias(f))) return rc;
        const T h = (T)1 / host_cbrt<T>((T)(S->W * S->H * S->D));   // :295 (int product, like the reference)
        {
            ScopedSpan sp(S, FAM_DIV);
            fs::launch_divergence<T>(S->stream, S->tune, g, sc, arr[slot[FS_VX]], arr[slot[FS_VY]], arr[slot[FS_VZ]],
                                     arr[slot[FS_DIVERGENCE]], arr[slot[FS_PRESSURE]], flags, (T)(-0.5) * h);
        }
        // divergence of the neighbouring slabs' boundary planes is never read (the solve only reads
        // rhs at the cell itself); the pressure halo planes still hold the previous projection and
        // must become the neighbours' freshly zeroed planes before the first sweep reads them.
        if ((rc = halo(arr[slot[FS_PRESSURE]]))) return rc;
        int res;
        if (S->solver == FS_SOLVER_MG) rc = multigrid_solve(FS_PRESSURE, FS_DIVERGENCE, &res);
        else rc = solve(0, slot[FS_PRESSURE], slot[FS_DIVERGENCE], (T)1, (T)6, S->acc, &res);   // :320
        if (rc) return rc;
        adopt(FS_PRESSURE, res);
        {
            ScopedSpan sp(S, FAM_GRAD);
            fs::launch_gradient<T>(S->stream, S->tune, g, sc, arr[slot[FS_PRESSURE]], arr[slot[FS_VX]], arr[slot[FS_VY]],
                                   arr[slot[FS_VZ]], flags, h, (T)2 * h);
        }
        // the next consumer of v's z-halo planes is the divergence of the second projection
        // (v_z[z+-1]) and the advection back-trace; refresh them now.
        for (int f : { FS_VX, FS_VY, FS_VZ })
            if ((rc = halo(arr[slot[f]]))) return rc;
        if (in_step && S->comm.active() && (rc = post_vzmax(projections_this_step++))) return rc;
        return FS_OK;
    }

    // ---- reach of the back-trace on a slab, without stalling the device ----------------------------------------
    // Inside step() the z velocity that carries each advection is known: after the first projection for v_x / v_y,
    // v_z_prev (= the end of the previous step) for v_z, after the second projection for the density.  Its global
    // max |.| is queued right behind the projection -- device reduction, all-reduce, copy into pinned memory, an event
    // -- and step() puts half of the density solve (independent work) between that and the gather that needs it.
    int projections_this_step = 0;
    int post_vzmax(int which)
    {
        if (which < 0 || which > 2) return FS_OK;
        ScopedSpan sp(S, FAM_COMM);
        double* out3 = red + 3 * 1024 + 3 * (2 + which);
        const int zlo = sc.lo_wall ? 0 : 1, zhi = sc.hi_wall ? g.D + 1 : g.D;
        fs::launch_stats<T>(S->stream, g, arr[slot[FS_VZ]], out3, red, 3 * 1024, zlo, zhi);
        int rc = comm_op([&](hipStream_t st) { return S->comm.reduce_stats(st, out3, g, S->D); }, "all-reduce of max |v_z|");
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(reach_pinned + 3 * which, out3, 3 * sizeof(double), hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipEventRecord(ev_reach[which], S->stream));
        reach_posted[which] = true;
        return FS_OK;
    }
    // max |v_z| posted by post_vzmax(which); false if nothing was posted (caller falls back to trace_reach)
    bool take_vzmax(int which, double* umax)
    {
        if (!reach_posted[which]) return false;
        ++S->n_reach_waits;
        if (hipEventQuery(ev_reach[which]) != hipSuccess) {
            ++S->n_reach_blocked;
            const auto t0 = std::chrono::steady_clock::now();
            if (hipEventSynchronize(ev_reach[which]) != hipSuccess) return false;
            S->reach_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        const double* st = reach_pinned + 3 * which;
        *umax = std::fmax(std::fabs(st[1]), std::fabs(st[2]));
        return true;
    }
    // max |v_z_prev| of the running step: carried over from the previous step, or posted at the start of this one
    bool prev_vzmax_known(double* umax)
    {
        if (vzmax_prev < 0.0 && !take_vzmax(2, &vzmax_prev)) return false;
        *umax = vzmax_prev;
        return true;
    }
    int reach_of(double umax) const
    {
        const double planes = std::ceil(std::fabs((double)S->dt * (double)S->D) * umax) + 2.0;
        return planes >= (double)S->D ? S->D : (int)planes;
    }

    // ---- advect (simulation.cpp:367-424) ------------------------------------------------
    int advect(int b, int field, int prev) override
    {
        if (!in_step) vzmax_prev = -1.0;             // a call from outside step(): what is known about v_z is void
        int rc = ensure_flags();
        if (rc) return rc;
        if (slot[field] == slot[prev]) {
            // in-place transport would read its own output: give the field a fresh array
            int id = acquire();
            if (id < 0) return fail(FS_ENOMEM, "array pool exhausted");
            adopt(field, id);
        }
        for (int f : { FS_VX, FS_VY, FS_VZ })
            if (f != field && slot[f] == slot[field]) return fail(FS_EINVAL, "advect target aliases a velocity array");
        const T kx = (T)S->dt * (T)S->W, ky = (T)S->dt * (T)S->H, kz = (T)S->dt * (T)S->D;   // :384-386
        const T* src = arr[slot[prev]];
        long zshift = 0;
        if (S->comm.active()) {
            // The back-trace may leave the slab by dt*D*|u_z| planes.  Bound it: the
            // carrying z velocity is `prev` for b == 3 and the current v_z otherwise
            // (simulation.cpp:382); its global max |.| gives the reach in planes, and only planes
            // within that reach of the slab are fetched from their owners.
            ScopedSpan sp(S, FAM_COMM);
            int reach = 0;
            double umax = -1.0;
            // inside step(): v_x / v_y are carried by v_z after the first projection, the density by v_z after the second,
            // v_z by v_z_prev (the end of the previous step, where known)
            const bool known = in_step && (b == 3 ? prev_vzmax_known(&umax) : take_vzmax(b == 0 ? 1 : 0, &umax));
            if (known && b == 0) vzmax_end = umax;       // v_z does not change any more in this step: next step's v_z_prev
            if (known) S->last_reach = reach = reach_of(umax);
            else if ((rc = trace_reach({ b == 3 ? prev : FS_VZ }, &reach))) return rc;
            if ((rc = gather_source(src, &gathered, reach, 0))) return rc;
            src = gathered;
            zshift = (long)sc.zoff * g.sz;
        }
        {
            ScopedSpan sp(S, FAM_ADVECT);
            fs::launch_advect<T>(S->stream, S->tune, g, sc, b, arr[slot[field]], src, arr[slot[FS_VX]], arr[slot[FS_VY]],
                                 arr[slot[FS_VZ]], flags, kill, coltab, kx, ky, kz, zshift);
        }
        return halo(arr[slot[field]]);
    }

    // Reach, in planes, of any back-trace whose carrying z velocity is one of `fields`: global
    // max |u_z| over them (device reduction + all-reduce), times dt*D, plus the floor()/corner margin.
    int trace_reach(std::initializer_list<int> fields, int* reach)
    {
        // every field's reduction and all-reduce is queued first; ONE copy and ONE host synchronisation fetch them all
        // (the window sizes are arguments of host-side send/recv calls, so the host has to know the reach)
        double st[4][3];
        int k = 0;
        for (int f : fields) {
            if (k >= 4) break;
            double* out3 = red + 3 * 1024 + 3 * k;
            const int zlo = sc.lo_wall ? 0 : 1, zhi = sc.hi_wall ? g.D + 1 : g.D;
            fs::launch_stats<T>(S->stream, g, arr[slot[f]], out3, red, 3 * 1024, zlo, zhi);
            if (S->comm.active()) {
                int rc = comm_op([&](hipStream_t cs) { return S->comm.reduce_stats(cs, out3, g, S->D); }, "stats all-reduce");
                if (rc) return rc;
            }
            ++k;
        }
        HIP_TRY(hipMemcpyAsync(&st[0][0], red + 3 * 1024, 3 * k * sizeof(double), hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipStreamSynchronize(S->stream));
        ++S->n_stream_syncs;
        double umax = 0.0;
        for (int i = 0; i < k; ++i) umax = std::fmax(umax, std::fmax(std::fabs(st[i][1]), std::fabs(st[i][2])));
        *reach = reach_of(umax);
        S->last_reach = *reach;
        return FS_OK;
    }

    // `which`: 0 = `gathered`, 1..3 = `gathered3` (the FSIPC export slot of the buffer)
    int gather_source(const T* src, T** buf, int reach, int which)
    {
        const long n = (g.sz * ((long)S->D + 2) + 8 + 63) / 64 * 64;
        if (gather_chunks.empty()) {
            // all four at the first gather, in arena chunks like the field arrays: every rank allocates them at the same
            // point of the step, so the export is collective (FSIPC: the owners of the planes write into them)
            const int per = (int)std::min<size_t>(4, std::max<size_t>(1, ARENA_CHUNK_BYTES / ((size_t)n * sizeof(T))));
            T* at[4];
            for (int i = 0; i < 4; i += per) {
                const size_t cnt = (size_t)std::min(per, 4 - i) * (size_t)n;
                T* chunk = nullptr;
                HIP_TRY(hipMalloc((void**)&chunk, cnt * sizeof(T)));
                gather_chunks.push_back(chunk);
                HIP_TRY(hipMemsetAsync(chunk, 0, cnt * sizeof(T), S->stream));
                for (int j = 0; j < per && i + j < 4; ++j) at[i + j] = chunk + (size_t)j * (size_t)n;
            }
            HIP_TRY(hipStreamSynchronize(S->stream));
            ++S->n_alloc_syncs;
            for (size_t k = 0; k < gather_chunks.size(); ++k)
                if (S->comm.register_buffer(SLOT_GATHER + (int)k, gather_chunks[k],
                                            (size_t)std::min(per, 4 - (int)k * per) * (size_t)n * sizeof(T), true))
                    return fail(FS_ECOMM, "exporting the gathered advection sources: %s", S->comm.last_error());
            gathered = at[0] + fs::LEAD;
            for (int k = 0; k < 3; ++k) gathered3[k] = at[1 + k] + fs::LEAD;
        }
        (void)which;
        if (S->debug_poison) HIP_TRY(hipMemsetAsync(*buf - fs::LEAD, 0xFF, (g.sz * ((long)S->D + 2) + 8) * sizeof(T), S->stream));
        T* dst = *buf;
        return comm_op([&](hipStream_t st) {
            return (reach >= S->D) ? S->comm.all_gather_planes(st, src, dst, g, S->D, sizeof(T))
                                   : S->comm.gather_window(st, src, dst, g, S->D, sizeof(T), reach); },
                       "gather of the advection source");
    }

    // advect(1,v_x,v_x_prev); advect(2,v_y,v_y_prev); advect(3,v_z,v_z_prev) in one kernel
    int advect_velocity_fused()
    {
        int rc = ensure_flags();
        if (rc) return rc;
        const T kx = (T)S->dt * (T)S->W, ky = (T)S->dt * (T)S->H, kz = (T)S->dt * (T)S->D;   // :384-386
        const T* p[3] = { arr[slot[FS_VX_PREV]], arr[slot[FS_VY_PREV]], arr[slot[FS_VZ_PREV]] };
        long zshift = 0;
        if (S->comm.active()) {
            // the z velocity carrying the three traces is the current v_z (x, y) or v_z_prev (z)
            ScopedSpan sp(S, FAM_COMM);
            int reach = 0;
            double umax = -1.0;
            double uprev = -1.0;
            if (in_step && prev_vzmax_known(&uprev) && take_vzmax(0, &umax)) S->last_reach = reach = reach_of(std::fmax(umax, uprev));
            else if ((rc = trace_reach({ FS_VZ, FS_VZ_PREV }, &reach))) return rc;
            for (int k = 0; k < 3; ++k) {
                if ((rc = gather_source(p[k], &gathered3[k], reach, 1 + k))) return rc;
                p[k] = gathered3[k];
            }
            zshift = (long)sc.zoff * g.sz;
        }
        {
            ScopedSpan sp(S, FAM_ADVECT);
            fs::launch_advect_velocity<T>(S->stream, S->tune, g, sc, arr[slot[FS_VX]], arr[slot[FS_VY]], arr[slot[FS_VZ]], p[0], p[1],
                                          p[2], flags, kill, coltab, kx, ky, kz, zshift);
        }
        for (int f : { FS_VX, FS_VY, FS_VZ })
            if ((rc = halo(arr[slot[f]]))) return rc;
        return FS_OK;
    }

    // ---- step (simulation.cpp:96-150) ---------------------------------------------------
    int step() override
    {
        struct InStep {                                  // the reach bookkeeping of this step; cleared on every exit
            Engine* e;
            explicit InStep(Engine* e_) : e(e_) { e->in_step = true; e->projections_this_step = 0; e->reach_posted[0] = e->reach_posted[1] = e->reach_posted[2] = false; e->vzmax_end = -1.0; }
            ~InStep() { e->in_step = false; e->vzmax_prev = e->vzmax_end; }
        } scope(this);
        int rc = ensure_flags();
        if (rc) return rc;
        const bool gs = (S->solver == FS_SOLVER_GS_LEX);
        for (int f : { FS_VX, FS_VY, FS_VZ })
            if ((rc = unalias(f))) return rc;
        {
            ScopedSpan sp(S, FAM_MISC);
            fs::launch_inlet_velocity<T>(S->stream, g, sc, arr[slot[FS_VX]], arr[slot[FS_VY]], arr[slot[FS_VZ]],
                                         (T)(float)S->speed);   // :103-105
        }
        // z-slabs: v_z as it is now becomes v_z_prev, which carries the advection of v_z; where its max |.| is not known from
        // the previous step (first step, host-side edits) it is queued here and has the whole diffusion to arrive
        if (S->comm.active() && vzmax_prev < 0.0 && (rc = post_vzmax(2))) return rc;
        // :108-110  v_*_prev = v_*  (pre-diffusion snapshot).  Jacobi never writes its input, so
        // the snapshot is an alias and the copy costs nothing; the in-place mode really copies.
        const int V[3] = { FS_VX, FS_VY, FS_VZ }, V0[3] = { FS_VX_PREV, FS_VY_PREV, FS_VZ_PREV };
        for (int k = 0; k < 3; ++k) {
            if (gs || S->acc <= 0) {
                if ((rc = unalias(V0[k]))) return rc;
                ScopedSpan sp(S, FAM_MISC);
                fs::launch_copy<T>(S->stream, g, arr[slot[V[k]]], arr[slot[V0[k]]]);
            } else {
                slot[V0[k]] = slot[V[k]];
            }
        }
        for (int k = 0; k < 3; ++k)                      // :115-117
            if ((rc = diffuse_T(k + 1, V[k], V0[k]))) return rc;
        if ((rc = project())) return rc;                 // :120
        // z-slabs: :135's density solve (independent of the velocities; its result is dead, :136 overwrites it) is the work
        // the device does while the reach of each advection travels to the host -- half of its passes here, between the
        // first projection and the velocity advection, the other half where the reference has it, between the second
        // projection and the density advection.  Same passes, same order, same bits.
        const bool split = S->comm.active() && S->split_dens && !S->elide_dead && !gs && S->acc > 0;
        SolveRun dens_run;
        if (split) {
            const T a = diffusion_a();
            if ((rc = solve_begin(dens_run, 0, slot[FS_DENS], slot[FS_BUFFER], a, (T)1 + (T)6 * a, S->acc))) return rc;   // :283
            if ((rc = solve_passes(dens_run, (int)dens_run.plan.size() / 2))) return rc;
            HIP_TRY(hipEventRecord(ev_slack, S->stream));
        }
        // was the device still busy with that work when the advection that waited for its reach had been queued?
        auto slack_check = [&]() {
            if (!split) return;
            if (hipEventQuery(ev_slack) == hipErrorNotReady) ++S->n_reach_hidden; else ++S->n_reach_exposed;
        };
        if (S->fuse_advect && slot[FS_VX] != slot[FS_VX_PREV] && slot[FS_VY] != slot[FS_VY_PREV] &&
            slot[FS_VZ] != slot[FS_VZ_PREV]) {
            // :125-127 in one pass (the three traces only chain through the cell's own values)
            if ((rc = advect_velocity_fused())) return rc;
        } else {
            for (int k = 0; k < 3; ++k)                  // :125-127
                if ((rc = advect(k + 1, V[k], V0[k]))) return rc;
        }
        slack_check();
        if ((rc = project())) return rc;                 // :130
        if (split) {
            int res;
            if ((rc = solve_end(dens_run, &res))) return rc;
            adopt(FS_DENS, res);
            HIP_TRY(hipEventRecord(ev_slack, S->stream));
        } else if (!S->elide_dead) {                     // :135 (its result is overwritten by :136)
            if ((rc = diffuse_T(0, FS_DENS, FS_BUFFER))) return rc;
        }
        if ((rc = advect(0, FS_DENS, FS_BUFFER))) return rc;   // :136
        slack_check();
        S->step_no++;
        if (S->in_run && S->dump_every > 0 && (S->step_no % S->dump_every) == 0) return dump_frame();   // :140-148
        return FS_OK;
    }

    // one iteration of Simulation::run()'s loop (simulation.cpp:63-71)
    int run_one() override
    {
        int rc = ensure_halos();
        if (rc) return rc;
        if ((rc = unalias(FS_DENS))) return rc;
        {
            ScopedSpan sp(S, FAM_MISC);
            fs::launch_inlet_density<T>(S->stream, g, sc, arr[slot[FS_DENS]], (T)0.001f);   // :65-67
        }
        if (S->solver == FS_SOLVER_GS_LEX || S->acc <= 0 || S->elide_dead) {
            if ((rc = unalias(FS_BUFFER))) return rc;
            ScopedSpan sp(S, FAM_MISC);
            fs::launch_copy<T>(S->stream, g, arr[slot[FS_DENS]], arr[slot[FS_BUFFER]]);    // :70
        } else {
            slot[FS_BUFFER] = slot[FS_DENS];             // :70 as an alias (see step())
        }
        return step();
    }

    // ---- data access ----------------------------------------------------------------
    int get_field(int which, void* dst, size_t n, int elem) override
    {
        if ((long)n != dense_cells()) return fail(FS_EINVAL, "get_field: expected %ld elements, got %zu", dense_cells(), n);
        const T* f = arr[slot[which]];
        if (elem != 1 && elem != 4 && elem != 8) return fail(FS_EINVAL, "elem_size must be 1, 4 or 8");
        {
            int rc = need_dense(n * (size_t)elem);
            if (rc) return rc;
        }
        if (elem == 4) fs::launch_pack<T, float>(S->stream, g, f, (float*)dense, 0, g.D + 1);
        else if (elem == 8) fs::launch_pack<T, double>(S->stream, g, f, (double*)dense, 0, g.D + 1);
        else if (elem == 1) fs::launch_pack<T, uint8_t>(S->stream, g, f, (uint8_t*)dense, 0, g.D + 1);
        else return fail(FS_EINVAL, "elem_size must be 1, 4 or 8");
        HIP_TRY(hipMemcpyAsync(dst, dense, n * elem, hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipStreamSynchronize(S->stream));
        return FS_OK;
    }

    int set_field(int which, const void* src, size_t n, int elem) override
    {
        if (!in_step) vzmax_prev = -1.0;             // a call from outside step(): what is known about v_z is void
        if ((long)n != dense_cells()) return fail(FS_EINVAL, "set_field: expected %ld elements, got %zu", dense_cells(), n);
        if (elem != 4 && elem != 8 && elem != 1) return fail(FS_EINVAL, "elem_size must be 1, 4 or 8");
        // a slot that shares its array gets a fresh one; contents are fully overwritten below
        if (shared_slot(which)) {
            int id = acquire();
            if (id < 0) return fail(FS_ENOMEM, "array pool exhausted");
            adopt(which, id);
        }
        {
            int rc = need_dense(n * (size_t)elem);
            if (rc) return rc;
        }
        HIP_TRY(hipMemcpyAsync(dense, src, n * elem, hipMemcpyHostToDevice, S->stream));
        T* f = arr[slot[which]];
        if (elem == 4) fs::launch_unpack<T, float>(S->stream, g, (const float*)dense, f, 0, g.D + 1);
        else if (elem == 8) fs::launch_unpack<T, double>(S->stream, g, (const double*)dense, f, 0, g.D + 1);
        else fs::launch_unpack<T, uint8_t>(S->stream, g, (const uint8_t*)dense, f, 0, g.D + 1);
        HIP_TRY(hipStreamSynchronize(S->stream));       // `src` may be freed by the caller
        if (which == FS_OBS) flags_dirty = true;
        else halos_dirty = true;
        return FS_OK;
    }

    int set_mask(const uint8_t* mask, size_t n) override { return set_field(FS_OBS, mask, n, 1); }

    int point(int which, int x, int y, int z, float v, int set_instead) override
    {
        if (!in_step) vzmax_prev = -1.0;             // a call from outside step(): what is known about v_z is void
        // x,y are global = local; z is global and must fall into this slab to have an effect.
        // The bookkeeping is the same on EVERY rank (every rank issues the same call): the slot maps
        // must not diverge, and ensure_halos() / ensure_flags() are collectives gated on these bits --
        // only the one-cell kernel launch depends on who owns plane z.
        int rc = unalias(which);
        if (rc) return rc;
        if (which == FS_OBS) flags_dirty = true;
        else halos_dirty = true;
        int zl = z - sc.zoff;
        if (zl < 1 || zl > g.D) return FS_OK;
        long idx = (long)x + (long)y * g.sy + (long)zl * g.sz;
        fs::launch_point_add<T>(S->stream, arr[slot[which]], idx, (T)v, set_instead);
        return FS_OK;
    }

    int tuned_shape() const override { return pair_shape; }
    int halo_depth() const override { return g.zh; }
    int tuned_triple() const override { return triple_alt; }

    // ---- the viewer's streamlines (GUI/utils.py:118-213) -------------------------------------
    // The device integrates every seed in both directions; what is left for the host is the
    // reference's bookkeeping per seed: joining the two parts and the three filters.
    int streamlines(int density, double proximity, int max_length, double step_size, double threshold) override
    {
        S->sl_offsets.assign(1, 0);
        S->sl_points.clear();
        S->sl_norm.clear();
        if (S->comm.active()) return fail(FS_EINVAL, "streamlines are computed on a single-GPU handle");
        if (density < 0 || max_length < 0) return fail(FS_EINVAL, "density and max_length must be >= 0");
        if (density > 4096 || max_length > 1000000) return fail(FS_EINVAL, "density <= 4096 and max_length <= 1e6, please");
        if (!(step_size == step_size) || !(proximity == proximity) || !(threshold == threshold))
            return fail(FS_EINVAL, "streamline parameters must not be NaN");
        const T* obs = arr[slot[FS_OBS]];
        fs::StreamParams p;
        p.nx = density; p.ny = density / 2; p.nz = density / 2;          // utils.py:136-138
        p.half = max_length / 2;
        p.step_size = step_size;
        const int dims[3] = { g.W + 2, g.H + 2, g.D + 2 };               // config.width/height/depth
        for (int k = 0; k < 3; ++k) {
            p.clip_hi[k] = (double)dims[k] - 1.001;
            p.bound_hi[k] = (double)(dims[k] - 1);
        }
        int* d_box = nullptr;
        int box[6];
        HIP_TRY(hipMalloc((void**)&d_box, sizeof box));
        fs::launch_obs_bbox<T>(S->stream, g, obs, d_box);
        HIP_TRY(hipMemcpyAsync(box, d_box, sizeof box, hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipStreamSynchronize(S->stream));
        hipFree(d_box);
        if (box[0] > box[3]) return FS_OK;                               // no obstacles: no streamlines (:134-135)
        for (int k = 0; k < 3; ++k) {
            p.lo[k] = (double)box[k] - proximity / 10;
            p.hi[k] = (double)box[3 + k] + proximity / 10;
        }
        const long nseed = (long)p.nx * p.ny * p.nz;
        if (nseed == 0) return FS_OK;
        // np.linspace(1, dim - 2, n): arange(n) * step + start, last element = stop
        std::vector<double> seeds((size_t)p.nx + p.ny + p.nz);
        {
            const int cnt[3] = { p.nx, p.ny, p.nz };
            size_t o = 0;
            for (int k = 0; k < 3; ++k) {
                const double start = 1.0, stop = (double)(dims[k] - 2);
                const int n = cnt[k];
                const double step = n > 1 ? (stop - start) / (double)(n - 1) : 0.0;
                for (int i = 0; i < n; ++i) seeds[o + i] = (double)i * step + start;
                if (n > 1) seeds[o + n - 1] = stop;
                o += n;
            }
        }
        // utils.py:147-150: seeds outside the widened bounding box are skipped before anything else
        std::vector<int> cand;
        for (int iz = 0; iz < p.nz; ++iz)
            for (int iy = 0; iy < p.ny; ++iy)
                for (int ix = 0; ix < p.nx; ++ix) {
                    const double sx = seeds[ix], sy = seeds[p.nx + iy], sz = seeds[(size_t)p.nx + p.ny + iz];
                    if (sx < p.lo[0] || sx > p.hi[0] || sy < p.lo[1] || sy > p.hi[1] || sz < p.lo[2] || sz > p.hi[2]) continue;
                    cand.push_back((iz * p.ny + iy) * p.nx + ix);
                }
        const long ncand = (long)cand.size();
        if (ncand == 0) return FS_OK;
        const size_t per = (size_t)(p.half + 1) * 3, npts = (size_t)ncand * 2 * per;
        if (npts > ((size_t)1 << 29))                    // 2 x 4 GiB of points and velocities: split the call instead
            return fail(FS_ENOMEM, "%ld seeds x %d steps is more than one call should integrate; lower density or max_length", ncand, max_length);
        double *d_seeds = nullptr, *d_pts = nullptr, *d_vel = nullptr;
        int *d_count = nullptr, *d_cand = nullptr;
        int rc = FS_OK;
        std::vector<double> pts, vel;
        std::vector<int> count((size_t)ncand * 2);
        double mx[3][3];
        do {
            if (hipMalloc((void**)&d_seeds, seeds.size() * 8) != hipSuccess || hipMalloc((void**)&d_pts, npts * 8) != hipSuccess ||
                hipMalloc((void**)&d_vel, npts * 8) != hipSuccess || hipMalloc((void**)&d_count, count.size() * 4) != hipSuccess ||
                hipMalloc((void**)&d_cand, cand.size() * 4) != hipSuccess) {
                rc = fail(FS_ENOMEM, "streamline buffers");
                break;
            }
            pts.resize(npts);
            vel.resize(npts);
            if (hipMemcpyAsync(d_seeds, seeds.data(), seeds.size() * 8, hipMemcpyHostToDevice, S->stream) != hipSuccess ||
                hipMemcpyAsync(d_cand, cand.data(), cand.size() * 4, hipMemcpyHostToDevice, S->stream) != hipSuccess) { rc = fail(FS_EHIP, "seed upload"); break; }
            fs::launch_streamlines<T>(S->stream, g, arr[slot[FS_VX]], arr[slot[FS_VY]], arr[slot[FS_VZ]], obs, p, d_seeds, d_cand,
                                      (int)ncand, d_count, d_pts, d_vel);
            if (hipMemcpyAsync(count.data(), d_count, count.size() * 4, hipMemcpyDeviceToHost, S->stream) != hipSuccess ||
                hipMemcpyAsync(pts.data(), d_pts, npts * 8, hipMemcpyDeviceToHost, S->stream) != hipSuccess ||
                hipMemcpyAsync(vel.data(), d_vel, npts * 8, hipMemcpyDeviceToHost, S->stream) != hipSuccess ||
                hipStreamSynchronize(S->stream) != hipSuccess) { rc = fail(FS_EHIP, "streamline kernel or copy failed"); break; }
            for (int k = 0; k < 3; ++k)
                if ((rc = stats_of(arr[slot[FS_VX + k]], mx[k]))) break;
        } while (0);
        hipFree(d_seeds); hipFree(d_pts); hipFree(d_vel); hipFree(d_count); hipFree(d_cand);
        if (rc) return rc;
        // np.max([vx, vy, vz]) + 1e-6 in the arrays' own precision (float32 for the reference's dumps)
        const T vmax = (T)std::fmax(std::fmax(mx[0][2], mx[1][2]), mx[2][2]);
        const double denom = (double)(T)(vmax + (T)1e-6);
        auto norm3 = [](double a, double b, double c) { return std::sqrt((a * a + b * b) + c * c); };
        std::vector<double> line, lvel;
        for (long sd = 0; sd < ncand; ++sd) {
            const int nb = count[2 * sd], nf = count[2 * sd + 1];
            if (nb == 0) continue;                                       // seed inside an obstacle
            const double* B = &pts[(size_t)(2 * sd) * per], *F = &pts[(size_t)(2 * sd + 1) * per];
            const double* VB = &vel[(size_t)(2 * sd) * per], *VF = &vel[(size_t)(2 * sd + 1) * per];
            line.clear();
            lvel.clear();
            for (int i = nb - 1; i >= 1; --i)                            // backward[::-1][:-1]  (:167-168)
                for (int c = 0; c < 3; ++c) { line.push_back(B[3 * i + c]); lvel.push_back(VB[3 * i + c]); }
            for (int i = 0; i < nf; ++i)
                for (int c = 0; c < 3; ++c) { line.push_back(F[3 * i + c]); lvel.push_back(VF[3 * i + c]); }
            const int n = (int)(line.size() / 3);
            if (n <= 5) continue;                                        // :171-172
            double max_change = 0.0;                                     // :175-181
            for (int i = 1; i < n; ++i) {
                const double ch = norm3(lvel[3 * i] - lvel[3 * i - 3], lvel[3 * i + 1] - lvel[3 * i - 2], lvel[3 * i + 2] - lvel[3 * i - 1]);
                if (ch > max_change) max_change = ch;
            }
            if (max_change < threshold) continue;
            bool near = false;                                           // :184-195, every third point
            for (int i = 0; i < n && !near; i += 3)
                near = p.lo[0] <= line[3 * i] && line[3 * i] <= p.hi[0] && p.lo[1] <= line[3 * i + 1] && line[3 * i + 1] <= p.hi[1] &&
                       p.lo[2] <= line[3 * i + 2] && line[3 * i + 2] <= p.hi[2];
            if (!near) continue;
            double max_speed = 0.0;                                      // :198-205
            for (int i = 0; i < n; ++i) max_speed = std::fmax(max_speed, norm3(lvel[3 * i], lvel[3 * i + 1], lvel[3 * i + 2]));
            S->sl_norm.push_back(std::fmin(max_speed / denom, 1.0));
            S->sl_points.insert(S->sl_points.end(), line.begin(), line.end());
            S->sl_offsets.push_back((long)(S->sl_points.size() / 3));
        }
        return FS_OK;
    }

    // ---- the viewer's obstacle mesh (GUI/utils.py:10-38) ------------------------------------------
    int obstacle_surface() override
    {
        S->surf_verts.clear();
        S->surf_tris.clear();
        S->surf_valid = false;
        if (S->comm.active()) return fail(FS_EINVAL, "the obstacle surface is extracted on a single-GPU handle");
        fs::SurfaceResult r;
        const char* msg = "";
        int rc = fs::extract_surface<T>(S->stream, g, arr[slot[FS_OBS]], &r, &msg);
        if (rc) return fail(rc, "%s", msg);
        S->surf_verts.resize((size_t)r.nverts * 3);
        S->surf_tris.resize((size_t)r.ntris * 3);
        hipError_t e = hipSuccess;
        if (r.nverts > 0) {
            e = hipMemcpyAsync(S->surf_verts.data(), r.d_verts, S->surf_verts.size() * sizeof(float), hipMemcpyDeviceToHost, S->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(S->surf_tris.data(), r.d_tris, S->surf_tris.size() * sizeof(int), hipMemcpyDeviceToHost, S->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(S->stream);
        }
        fs::surface_free(&r);
        if (e != hipSuccess) return fail(FS_EHIP, "copying the obstacle surface: %s", hipGetErrorString(e));
        S->surf_valid = true;
        return FS_OK;
    }

    int apply_solid_cells(const int* cells, long n) override
    {
        // cells: device array of packed global cell ids x + y*(W+2) + z*(W+2)*(H+2)
        int rc = unalias(FS_OBS);
        if (rc) return rc;
        fs::launch_mark_cells<T>(S->stream, g, sc, arr[slot[FS_OBS]], cells, n);
        flags_dirty = true;
        return FS_OK;
    }

    int stats(int which, double* out3) override { return stats_of(arr[slot[which]], out3); }

    // The number Simulation::run() prints every 100 steps: std::reduce(dens.begin(), dens.end()) (simulation.cpp:76),
    // i.e. a sum in the field's own precision in libstdc++'s order -- groups of four as (a0 + a1) + (a2 + a3), added to the
    // running sum one group at a time (<numeric>, random-access branch).  A rounding chain cannot be reordered, so the
    // dense array comes to the host for it (once per 100 steps, console output only).
    int reference_order_sum(int which, double* out) override
    {
        const long n = dense_cells();
        std::vector<T> h((size_t)n);
        int rc = get_field(which, h.data(), (size_t)n, (int)sizeof(T));
        if (rc) return rc;
        T init = (T)0;
        long i = 0;
        for (; n - i >= 4; i += 4) {
            const T v1 = h[i] + h[i + 1], v2 = h[i + 2] + h[i + 3];
            const T v3 = v1 + v2;
            init = init + v3;
        }
        for (; i < n; ++i) init = init + h[i];
        *out = (double)init;
        return FS_OK;
    }

    int stats_of(const T* field, double* out3)
    {
        // whole padded array (simulation.cpp:76, :82-89); a slab counts its own planes plus
        // the physical ghost planes it holds, and the partial results are all-reduced
        const int zlo = sc.lo_wall ? 0 : 1, zhi = sc.hi_wall ? g.D + 1 : g.D;
        fs::launch_stats<T>(S->stream, g, field, red + 3 * 1024, red, 3 * 1024, zlo, zhi);
        if (S->comm.active()) {
            int rc = comm_op([&](hipStream_t cs) { return S->comm.reduce_stats(cs, red + 3 * 1024, g, S->D); }, "stats all-reduce");
            if (rc) return rc;
        }
        HIP_TRY(hipMemcpyAsync(out3, red + 3 * 1024, 3 * sizeof(double), hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipStreamSynchronize(S->stream));
        return FS_OK;
    }

    // ---- frame dump (simulation.cpp:56-60, 140-148) ---------------------------------------
    int dump_frame() override
    {
        static const char* const names[5] = { "data.bin", "obs.bin", "v_x.bin", "v_y.bin", "v_z.bin" };
        static const int which[5] = { FS_DENS, FS_OBS, FS_VX, FS_VY, FS_VZ };
        if (!S->dump_open) {
            // rank 0 truncates like the reference's ofstream::open (simulation.cpp:56-60); the other
            // slab ranks open the same files for update once they exist
            const bool lead = !S->comm.active() || S->comm.rank == 0;
            bool ok = true;
            for (int pass = 0; pass < 2; ++pass) {
                if ((pass == 0) == lead) {
                    for (int k = 0; k < 5; ++k) {
                        std::string path = S->dump_dir + "/" + names[k];
                        S->dump_fp[k] = fopen(path.c_str(), lead ? "wb" : "r+b");
                        if (!S->dump_fp[k]) ok = false;
                    }
                }
                if (pass == 0 && S->comm.active()) {
                    if (S->comm.shm && S->comm.shm_ready(g, S->D)) return fail(FS_ECOMM, "%s", S->comm.last_error());
                    { int brc = slab_barrier(); if (brc) return brc; }
                }
            }
            if (!ok) {
                for (int k = 0; k < 5; ++k)
                    if (S->dump_fp[k]) { fclose(S->dump_fp[k]); S->dump_fp[k] = nullptr; }
                if (!S->dump_warned) {
                    // the reference silently writes nothing when data/ is missing (simulation.cpp:56-60)
                    fprintf(stderr, "fluidsim: cannot open frame dumps under '%s' -- continuing without dumps\n",
                            S->dump_dir.c_str());
                    S->dump_warned = true;
                }
                return FS_OK;
            }
            S->dump_open = true;
            S->dump_frames = 0;
        }
        // local planes written by this rank: its interior planes, plus the physical ghost planes
        const int zlo = sc.lo_wall ? 0 : 1, zhi = sc.hi_wall ? g.D + 1 : g.D;
        const long plane = (long)(g.W + 2) * (g.H + 2);
        const long frame_cells = plane * ((long)S->D + 2);
        const long nloc = plane * (zhi - zlo + 1);
        std::string werr;
        if (S->writer.init(S->device, nloc * 5, S->dump_fp, &werr)) return fail(FS_EHIP, "frame writer: %s", werr.c_str());
        const int k = (int)(S->dump_frames % fs::FrameWriter::NSLOT);
        if (S->writer.acquire(k, &werr)) return fail(FS_EIO, "%s (%s)", werr.c_str(), S->dump_dir.c_str());
        for (int f = 0; f < 5; ++f)
            fs::launch_pack<T, float>(S->stream, g, arr[slot[which[f]]], S->writer.dev[k] + (size_t)f * nloc, zlo, zhi);
        const long off = S->comm.active()
                             ? (S->dump_frames * frame_cells + plane * (long)(sc.zoff + zlo)) * (long)sizeof(float)
                             : -1;
        if (S->writer.submit(k, S->stream, nloc, off, &werr)) return fail(FS_EHIP, "frame writer: %s", werr.c_str());
        S->dump_frames++;
        if (!S->dump_async && S->writer.flush(&werr)) return fail(FS_EIO, "%s (%s)", werr.c_str(), S->dump_dir.c_str());
        return FS_OK;
    }

    // ---- measurement ----------------------------------------------------------------
    int time_sweeps(int b, int field, int prev, float a, float c, int reps, double* ms) override
    {
        int rc = ensure_flags();
        if (rc) return rc;
        if (reps < 1) return fail(FS_EINVAL, "reps must be >= 1");
        int s1 = acquire(slot[field], slot[prev]);
        int s2 = acquire(slot[field], slot[prev]);
        if (s1 < 0 || s2 < 0) return fail(FS_ENOMEM, "array pool exhausted");
        const T inv_c = (T)1 / (T)c;
        struct Release {                                   // error paths must not leak the scratch arrays or the events
            bool* held; int a, b; hipEvent_t e0 = nullptr, e1 = nullptr;
            ~Release() { held[a] = held[b] = false; if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
        } rel{held, s1, s2};
        {                                                  // same launch plans as solve(); tuned BEFORE the clock starts
            int rc2 = ensure_tuned(slot[field], slot[prev], b, (T)a, inv_c);
            if (rc2) return rc2;
        }
        const bool can2 = two_sweep_kernels(), can3 = triple_alt >= 0;
        HIP_TRY(hipEventCreate(&rel.e0));
        HIP_TRY(hipEventCreate(&rel.e1));
        hipEvent_t e0 = rel.e0, e1 = rel.e1;
        // one untimed sweep to fault in code and scratch
        launch_pass(S->stream, 1, false, arr[slot[field]], arr[slot[prev]], arr[s1], b, (T)a, inv_c, 1, g.D);
        HIP_TRY(hipEventRecord(e0, S->stream));
        int src = s1, dst = s2;
        for (int r = 0; r < reps; ++r) {
            const int lv = (can3 && r + 2 < reps) ? 3 : (can2 && r + 1 < reps) ? 2 : 1;
            launch_pass(S->stream, lv, false, arr[src], arr[slot[prev]], arr[dst], b, (T)a, inv_c, 1, g.D);
            r += lv - 1;
            int t = src; src = dst; dst = t;
        }
        HIP_TRY(hipEventRecord(e1, S->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        *ms = (double)t / reps;
        return FS_OK;
    }
};

int ensure_engine(fs_sim* s)
{
    if (s->eng) return FS_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc;
    if (s->fp64) {
        auto* e = new Engine<double>(s);
        rc = e->init();
        if (rc) { delete e; return rc; }
        s->eng = e;
    } else {
        auto* e = new Engine<float>(s);
        rc = e->init();
        if (rc) { delete e; return rc; }
        s->eng = e;
    }
    return FS_OK;
}

bool in_box(fs_sim* s, int x, int y, int z) { return x >= 1 && x <= s->W && y >= 1 && y <= s->H && z >= 1 && z <= s->D; }

}  // namespace

// =======================================================================================
extern "C" {

const char* fs_last_error(void) { return g_err.c_str(); }
const char* fs_version(void) { return "fluidsim-amd 0.1 (gfx950)"; }

fs_sim* fs_create(int w, int h, int d, int iter, int speed, float dt, float diff, float visc, int acc)
{
    if (w < 1 || h < 1 || d < 1 || acc < 0 || iter < 0) {
        fail(FS_EINVAL, "fs_create: bad extents %dx%dx%d / iter %d / acc %d", w, h, d, iter, acc);
        return nullptr;
    }
    if ((double)(w + 2) * (h + 2) * (d + 2) >= 2147483647.0) {
        // the reference indexes with int (simulation.h:9,13); keep its limit on the dense layout
        fail(FS_EINVAL, "fs_create: padded grid exceeds 2^31 cells");
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        fail(FS_EHIP, "fs_create: no HIP device available (%s); this library has no CPU path",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    fs_sim* s = new fs_sim;
    s->W = w; s->H = h; s->D = d; s->iter = iter; s->speed = speed; s->acc = acc;
    s->dt = dt; s->diff = diff; s->visc = visc;
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
    e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        fail(FS_EHIP, "fs_create: hipStreamCreate: %s", hipGetErrorString(e));
        delete s;
        return nullptr;
    }
    return s;
}

int fs_destroy(fs_sim* s)
{
    if (!s) return FS_OK;
    hipSetDevice(s->device);
    if (s->stream) hipStreamSynchronize(s->stream);
    s->resolve_spans();
    for (hipEvent_t ev : s->event_pool) hipEventDestroy(ev);
    {
        std::string werr;
        s->writer.flush(&werr);
        s->writer.shutdown();
    }
    for (int k = 0; k < 5; ++k)
        if (s->dump_fp[k]) fclose(s->dump_fp[k]);
    if (s->comm.active()) hipDeviceSynchronize();   // the communication stream too
    s->comm.release_buffers();       // FSIPC: collective; no peer maps (or still writes) this rank's arrays once it returns
    delete s->eng;
    s->comm.destroy();
    if (s->stream_full || s->stream_masked) {
        if (s->stream_full) hipStreamDestroy(s->stream_full);
        if (s->stream_masked) hipStreamDestroy(s->stream_masked);
    } else if (s->stream) {
        hipStreamDestroy(s->stream);
    }
    delete s;
    return FS_OK;
}

int fs_set_option(fs_sim* s, const char* key, const char* value)
{
    if (!s || !key || !value) return fail(FS_EINVAL, "fs_set_option: null argument");
    std::string k = key, v = value;
    if (k == "precision") {
        if (s->eng) return fail(FS_EINVAL, "precision must be set before first use");
        if (v == "fp32") s->fp64 = false;
        else if (v == "fp64") s->fp64 = true;
        else return fail(FS_EINVAL, "precision: fp32 | fp64");
    } else if (k == "solver") {
        if (v == "jacobi") s->solver = FS_SOLVER_JACOBI;
        else if (v == "gs_lex") s->solver = FS_SOLVER_GS_LEX;
        else if (v == "rbsor") s->solver = FS_SOLVER_RBSOR;
        else if (v == "mg") s->solver = FS_SOLVER_MG;
        else return fail(FS_EINVAL, "solver: jacobi | gs_lex | rbsor | mg");
    } else if (k == "mg_cycles" || k == "mg_pre" || k == "mg_post" || k == "mg_coarse_iters") {
        const int n = atoi(value);
        if (n < (k == "mg_cycles" ? 0 : 1) || n > 1000) return fail(FS_EINVAL, "%s out of range", key);
        (k == "mg_cycles" ? s->mg_cycles : k == "mg_pre" ? s->mg_pre : k == "mg_post" ? s->mg_post : s->mg_coarse) = n;
    } else if (k == "mg_min_planes") {
        const int n = atoi(value);
        if (n < 1 || n > 1024) return fail(FS_EINVAL, "mg_min_planes: 1 .. 1024");
        s->mg_min_planes = n;
    } else if (k == "launch_plans") {
        int two = -2, three = -2;
        if (sscanf(value, "%d,%d", &two, &three) != 2 || two < -1 || three < -1 || two > 127 || three > 31)
            return fail(FS_EINVAL, "launch_plans: \"<two-sweep plan id>,<three-sweep plan id>\" (-1 = none)");
        s->plan_two = two;
        s->plan_three = three;
    } else if (k == "sor_omega") {
        const float om = (float)atof(value);
        if (!(om > 0.0f && om < 2.0f)) return fail(FS_EINVAL, "sor_omega must lie in (0, 2)");
        s->omega = om;
    } else if (k == "dump_dir") {
        s->dump_dir = v;
    } else if (k == "dump_every") {
        s->dump_every = atoi(value);
    } else if (k == "voxel_seed") {
        s->voxel_seed = (unsigned)strtoul(value, nullptr, 10);
    } else if (k == "quiet") {
        s->quiet = (v != "0");
    } else if (k == "profile") {
        s->profile = (v != "0");
    } else if (k == "elide_dead_density_solve") {
        s->elide_dead = (v != "0");
    } else if (k == "dump_async") {
        s->dump_async = (v != "0");
    } else if (k == "fuse_advect") {
        s->fuse_advect = (v != "0");
    } else if (k == "overlap") {
        if (s->eng && s->overlap_plan >= 0) return fail(FS_EINVAL, "overlap must be set before the first solve");
        s->overlap = (v == "auto") ? -1 : atoi(value);
        if (s->overlap < -1 || s->overlap > 3 || (v != "auto" && v != "0" && v != "1" && v != "2" && v != "3"))
            return fail(FS_EINVAL, "overlap: auto | 0 | 1 | 2 | 3");
    } else if (k == "comm_cus") {
        if (s->eng) return fail(FS_EINVAL, "comm_cus must be set before first use");
        s->comm_cus = (v == "auto") ? -1 : atoi(value);
        if (s->comm_cus < -1 || s->comm_cus > 128) return fail(FS_EINVAL, "comm_cus: auto | 0 .. 128");
    } else if (k == "split_density_solve") {
        s->split_dens = (v != "0");
    } else if (k == "debug_poison_gather") {
        s->debug_poison = (v != "0");
    } else if (k == "sweep_ry") {
        int r = atoi(value);
        if (r != 2 && r != 4) return fail(FS_EINVAL, "sweep_ry: 2 | 4");
        s->tune.ry = r;
    } else if (k == "sweep_zc") {
        s->tune.zc_len = atoi(value);
    } else if (k == "sweep_blocks") {
        s->tune.target_blocks = atoi(value) > 0 ? atoi(value) : 2048;
    } else if (k == "sweep_abl") {
        s->tune.abl = atoi(value);
    } else if (k == "sweep_fuse") {
        int f = atoi(value);
        if (f < 1 || f > 4) return fail(FS_EINVAL, "sweep_fuse: 1 | 2 | 3 | 4");
        s->tune.fuse = f;
    } else if (k == "project_kernels") {
        if (v == "cell") s->tune.project_cell = 1;
        else if (v == "march") s->tune.project_cell = 0;
        else return fail(FS_EINVAL, "project_kernels: march | cell");
    } else if (k == "advect_kernels") {
        if (v == "cell") s->tune.advect_cell = 1;
        else if (v == "celltab") s->tune.advect_cell = 2;
        else if (v == "row") s->tune.advect_cell = 0;
        else if (v == "tile") s->tune.advect_cell = 3;
        else return fail(FS_EINVAL, "advect_kernels: cell | celltab | tile | row");
    } else if (k == "advect_window") {
        const int n = atoi(value);
        if (n < 1 || n > 128) return fail(FS_EINVAL, "advect_window: 1 .. 128");
        s->tune.advect_window = n;
    } else if (k == "wall_free") {
        if (v == "0") s->tune.wall_free = 0;
        else if (v == "auto") s->tune.wall_free = 1;
        else if (v == "1") s->tune.wall_free = 2;
        else return fail(FS_EINVAL, "wall_free: 0 | auto | 1");
    } else if (k == "pair_zc") {
        s->tune.pair_zc = atoi(value);
    } else if (k == "pair_shape") {
        s->tune.pair_shape = atoi(value);
    } else if (k == "two_sweep_kernel") {
        if (v == "auto") s->tune.two_kind = 0;
        else if (v == "pair") s->tune.two_kind = 1;
        else if (v == "fused") s->tune.two_kind = 2;
        else return fail(FS_EINVAL, "two_sweep_kernel: auto | pair | fused");
    } else {
        return fail(FS_EINVAL, "unknown option '%s'", key);
    }
    return FS_OK;
}

int fs_get_int(fs_sim* s, const char* name, int* out)
{
    if (!s || !name || !out) return fail(FS_EINVAL, "null argument");
    std::string n = name;
    if (n == "width") *out = s->W; else if (n == "height") *out = s->H; else if (n == "depth") *out = s->D;
    else if (n == "speed") *out = s->speed; else if (n == "acc") *out = s->acc; else if (n == "iter") *out = s->iter;
    else if (n == "local_depth") *out = s->comm.active() ? s->comm.local_depth(s->D) : s->D;
    else if (n == "z_offset") *out = s->comm.active() ? s->comm.z_offset(s->D) : 0;
    else if (n == "last_advect_reach") *out = s->last_reach;
    else if (n == "pair_shape") *out = s->eng ? s->eng->tuned_shape() : -1;
    else if (n == "triple_plan") *out = s->eng ? s->eng->tuned_triple() : -1;
    else if (n == "two_sweep_fused") *out = (s->eng && s->eng->tuned_shape() >= 64) ? 1 : 0;   // 1: jacobi_fused_kernel<NL=2>, 0: jacobi_pair_kernel
    else if (n == "halo_depth") *out = s->eng ? s->eng->halo_depth() : 0;
    else if (n == "mg_levels") *out = s->eng ? s->eng->multigrid_levels() : 0;          // levels of the last solver=mg solve, level 0 included
    // z-slab runs: the communication schedule in force (what "auto" chose), and the slab step's host-side waits
    else if (n == "overlap_plan") *out = s->overlap_plan;
    else if (n == "comm_cus_plan") *out = s->cus_plan;
    else if (n == "stream_syncs") *out = (int)s->n_stream_syncs;          // hipStreamSynchronize calls issued by slab steps (reach fallback)
    else if (n == "reach_waits") *out = (int)s->n_reach_waits;            // waits for an asynchronously delivered reach ...
    else if (n == "reach_waits_blocked") *out = (int)s->n_reach_blocked;  // ... that found it not yet delivered
    else if (n == "reach_wait_us") *out = (int)(s->reach_wait_ms * 1e3);  // host time spent blocked in them
    else if (n == "reach_hidden") *out = (int)s->n_reach_hidden;          // advections queued while the device still had the work placed before them ...
    else if (n == "reach_exposed") *out = (int)s->n_reach_exposed;        // ... and after it had run dry (a bubble on the device)
    else return fail(FS_EINVAL, "unknown int member '%s'", name);
    return FS_OK;
}
int fs_set_int(fs_sim* s, const char* name, int value)
{
    if (!s || !name) return fail(FS_EINVAL, "null argument");
    std::string n = name;
    if (n == "speed") s->speed = value;
    else if (n == "acc") { if (value < 0) return fail(FS_EINVAL, "acc < 0"); s->acc = value; }
    else if (n == "iter") { if (value < 0) return fail(FS_EINVAL, "iter < 0"); s->iter = value; }
    else return fail(FS_EINVAL, "member '%s' is fixed after construction", name);
    return FS_OK;
}
int fs_get_float(fs_sim* s, const char* name, float* out)
{
    if (!s || !name || !out) return fail(FS_EINVAL, "null argument");
    std::string n = name;
    if (n == "dt") *out = s->dt; else if (n == "diff") *out = s->diff; else if (n == "visc") *out = s->visc;
    else if (n.size() == 11 && n.compare(0, 7, "overlap") == 0 && n.compare(8, 3, "_ms") == 0 && n[7] >= '0' && n[7] <= '7')
        *out = (float)s->overlap_ms[n[7] - '0'];   // "overlap<k>_ms": slowest rank's ms per pass of candidate k = overlap mode + 4 * (CU mask on), 0 = not timed
    else return fail(FS_EINVAL, "unknown float member '%s'", name);
    return FS_OK;
}
int fs_set_float(fs_sim* s, const char* name, float value)
{
    if (!s || !name) return fail(FS_EINVAL, "null argument");
    std::string n = name;
    if (n == "dt") s->dt = value; else if (n == "diff") s->diff = value; else if (n == "visc") s->visc = value;
    else return fail(FS_EINVAL, "unknown float member '%s'", name);
    return FS_OK;
}

#define ENGINE_OR_RETURN(s)                                  \
    if (!(s)) return fail(FS_EINVAL, "null handle");         \
    { int rc_ = ensure_engine(s); if (rc_) return rc_; }     \
    hipSetDevice((s)->device);

int fs_add_obstacle(fs_sim* s, int x, int y, int z)
{
    ENGINE_OR_RETURN(s);
    if (!in_box(s, x, y, z)) return fail(FS_EINVAL, "addObstacle(%d,%d,%d) outside 1..%dx1..%dx1..%d", x, y, z, s->W, s->H, s->D);
    return s->eng->point(FS_OBS, x, y, z, 1.0f, 1);
}
int fs_add_density(fs_sim* s, int x, int y, int z, float amount)
{
    ENGINE_OR_RETURN(s);
    if (!in_box(s, x, y, z)) return fail(FS_EINVAL, "addDensity(%d,%d,%d) outside the grid", x, y, z);
    return s->eng->point(FS_DENS, x, y, z, amount, 0);
}
int fs_set_velocity(fs_sim* s, int x, int y, int z, float ax, float ay, float az)
{
    ENGINE_OR_RETURN(s);
    if (!in_box(s, x, y, z)) return fail(FS_EINVAL, "setVelocity(%d,%d,%d) outside the grid", x, y, z);
    int rc = s->eng->point(FS_VX, x, y, z, ax, 1);
    if (!rc) rc = s->eng->point(FS_VY, x, y, z, ay, 1);
    if (!rc) rc = s->eng->point(FS_VZ, x, y, z, az, 1);
    return rc;
}

int fs_set_obstacle_mask(fs_sim* s, const uint8_t* mask, size_t n)
{
    ENGINE_OR_RETURN(s);
    if (!mask) return fail(FS_EINVAL, "null mask");
    return s->eng->set_mask(mask, n);
}

int fs_load_stl(fs_sim* s, const char* stl_file, float scale, float rot_x, float rot_y, float rot_z,
                float translate_x, float translate_y, float translate_z, long* added)
{
    ENGINE_OR_RETURN(s);
    if (!stl_file) return fail(FS_EINVAL, "null path");
    fs::VoxelResult vr;
    int rc = fs::voxelize_stl(s->stream, stl_file, s->W, s->H, s->D, scale, rot_x, rot_y, rot_z, translate_x,
                              translate_y, translate_z, s->voxel_seed, s->quiet, &vr);
    if (rc) {                                              // every exit releases the voxelizer's device buffers
        const std::string msg = vr.error;
        fs::voxelize_free(&vr);
        return rc == FS_EIO ? fail(FS_EIO, "%s", msg.c_str()) : fail(rc, "voxelizer: %s", msg.c_str());
    }
    if (added) *added = vr.added;
    rc = s->eng->apply_solid_cells(vr.d_cells, vr.added);
    hipStreamSynchronize(s->stream);
    fs::voxelize_free(&vr);
    return rc;
}

int fs_step(fs_sim* s) { ENGINE_OR_RETURN(s); return s->eng->step(); }
int fs_run_one(fs_sim* s) { ENGINE_OR_RETURN(s); return s->eng->run_one(); }

int fs_sync(fs_sim* s)
{
    if (!s) return fail(FS_EINVAL, "null handle");
    hipSetDevice(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::string werr;
    if (s->writer.flush(&werr)) return fail(FS_EIO, "frame writer: %s", werr.c_str());
    if (s->comm.check()) return fail(FS_ECOMM, "%s", s->comm.last_error());
    return FS_OK;
}

int fs_run(fs_sim* s)
{
    ENGINE_OR_RETURN(s);
    const bool talk = !s->quiet && (!s->comm.active() || s->comm.rank == 0);
    if (talk) printf("starting 3-D simulation: %dx%dx%d  steps = %d\n", s->W, s->H, s->D, s->iter);   // simulation.cpp:51-53
    // run() re-opens (truncates) the five files (simulation.cpp:56-60)
    {
        std::string werr;
        if (s->writer.flush(&werr)) return fail(FS_EIO, "frame writer: %s", werr.c_str());
    }
    for (int k = 0; k < 5; ++k)
        if (s->dump_fp[k]) { fclose(s->dump_fp[k]); s->dump_fp[k] = nullptr; }
    s->dump_open = false;
    s->in_run = true;
    s->step_no = 0;
    int rc = FS_OK;
    for (int i = 0; i < s->iter && !rc; ++i) {
        rc = s->eng->run_one();
        if (!rc && (i + 1) % 100 == 0 && i > 0) {         // :73-77
            // one GPU: the reference's own float sum, digit for digit; z-slabs: the all-reduced double sum (a rounding
            // chain over the whole array does not split over ranks)
            double st[3] = {0, 0, 0};
            if (s->comm.active()) rc = s->eng->stats(FS_DENS, st);
            else if (talk) rc = s->eng->reference_order_sum(FS_DENS, &st[0]);
            if (!rc && talk) printf("step %d\n  density sum = %g\n", i + 1, st[0]);
        }
    }
    if (!rc && s->dump_every == -1) rc = s->eng->dump_frame();
    s->in_run = false;
    {
        std::string werr;
        if (s->writer.flush(&werr) && !rc) rc = fail(FS_EIO, "frame writer: %s", werr.c_str());
    }
    if (rc) return rc;
    static const int which[4] = { FS_DENS, FS_VX, FS_VY, FS_VZ };
    static const char* const label[4] = { "density ", "velocity x", "velocity y", "velocity z" };
    if (talk) printf("\n--- statistics -------------------------------------------------\n");   // :81
    for (int k = 0; k < 4; ++k) {
        double st[3];
        rc = s->eng->stats(which[k], st);
        if (rc) return rc;
        if (talk) printf("%s min = %g\n%s max = %g\n", label[k], st[1], label[k], st[2]);          // :82-89
    }
    if (talk) { printf("simulation finished\n"); fflush(stdout); }
    return FS_OK;
}

#define CHECK_FIELD(f) if ((f) < 0 || (f) >= FS_NFIELDS) return fail(FS_EINVAL, "bad field selector %d", (f));
#define CHECK_B(b) if ((b) < 0 || (b) > 3) return fail(FS_EINVAL, "bad boundary code %d", (b));

int fs_set_bounds(fs_sim* s, int b, int field) { ENGINE_OR_RETURN(s); CHECK_B(b); CHECK_FIELD(field); return s->eng->set_bounds(b, field); }
int fs_linear_solver(fs_sim* s, int b, int field, int prev, float a, float c)
{
    ENGINE_OR_RETURN(s); CHECK_B(b); CHECK_FIELD(field); CHECK_FIELD(prev);
    if (field == prev) return fail(FS_EINVAL, "field and prev must differ");
    return s->eng->linear_solver(b, field, prev, a, c);
}
int fs_diffuse(fs_sim* s, int b, int field, int prev)
{
    ENGINE_OR_RETURN(s); CHECK_B(b); CHECK_FIELD(field); CHECK_FIELD(prev);
    if (field == prev) return fail(FS_EINVAL, "field and prev must differ");
    return s->eng->diffuse(b, field, prev);
}
int fs_project(fs_sim* s) { ENGINE_OR_RETURN(s); return s->eng->project(); }
int fs_advect(fs_sim* s, int b, int field, int prev)
{
    ENGINE_OR_RETURN(s); CHECK_B(b); CHECK_FIELD(field); CHECK_FIELD(prev);
    if (field == prev) return fail(FS_EINVAL, "field and prev must differ");
    return s->eng->advect(b, field, prev);
}

int fs_get_field(fs_sim* s, int which, void* dst, size_t n, int elem_size)
{
    ENGINE_OR_RETURN(s); CHECK_FIELD(which);
    if (!dst) return fail(FS_EINVAL, "null buffer");
    return s->eng->get_field(which, dst, n, elem_size);
}
int fs_set_field(fs_sim* s, int which, const void* src, size_t n, int elem_size)
{
    ENGINE_OR_RETURN(s); CHECK_FIELD(which);
    if (!src) return fail(FS_EINVAL, "null buffer");
    return s->eng->set_field(which, src, n, elem_size);
}
size_t fs_padded_size(fs_sim* s)
{
    if (!s) return 0;
    int d = s->comm.active() ? s->comm.local_depth(s->D) : s->D;
    return (size_t)(s->W + 2) * (s->H + 2) * (d + 2);
}

int fs_dump_frame(fs_sim* s) { ENGINE_OR_RETURN(s); return s->eng->dump_frame(); }

int fs_field_stats(fs_sim* s, int which, double* sum, double* mn, double* mx)
{
    ENGINE_OR_RETURN(s); CHECK_FIELD(which);
    double st[3];
    int rc = s->eng->stats(which, st);
    if (rc) return rc;
    if (sum) *sum = st[0];
    if (mn) *mn = st[1];
    if (mx) *mx = st[2];
    return FS_OK;
}

int fs_get_timing(fs_sim* s, const char* family, double* total_ms, long* launches)
{
    if (!s || !family) return fail(FS_EINVAL, "null argument");
    hipSetDevice(s->device);
    s->resolve_spans();
    for (int f = 0; f < FAM_COUNT; ++f)
        if (strcmp(family, kFamilyNames[f]) == 0) {
            if (total_ms) *total_ms = s->fam_ms[f];
            if (launches) *launches = s->fam_launches[f];
            return FS_OK;
        }
    return fail(FS_EINVAL, "unknown kernel family '%s'", family);
}
int fs_reset_timing(fs_sim* s)
{
    if (!s) return fail(FS_EINVAL, "null handle");
    hipSetDevice(s->device);
    s->resolve_spans();
    for (int f = 0; f < FAM_COUNT; ++f) { s->fam_ms[f] = 0; s->fam_launches[f] = 0; }
    return FS_OK;
}

int fs_time_sweeps(fs_sim* s, int b, int field, int prev, float a, float c, int reps, double* ms_per_sweep)
{
    ENGINE_OR_RETURN(s); CHECK_B(b); CHECK_FIELD(field); CHECK_FIELD(prev);
    if (!ms_per_sweep) return fail(FS_EINVAL, "null output");
    return s->eng->time_sweeps(b, field, prev, a, c, reps, ms_per_sweep);
}

int fs_streamlines(fs_sim* s, int density, double proximity, int max_length, double step_size,
                   double vel_change_threshold, long* n_lines, long* n_points)
{
    ENGINE_OR_RETURN(s);
    int rc = s->eng->streamlines(density, proximity, max_length, step_size, vel_change_threshold);
    if (rc) return rc;
    if (n_lines) *n_lines = (long)s->sl_norm.size();
    if (n_points) *n_points = (long)(s->sl_points.size() / 3);
    return FS_OK;
}

int fs_streamlines_fetch(fs_sim* s, long* offsets, double* points, double* norm_speed)
{
    ENGINE_OR_RETURN(s);
    if (s->sl_offsets.empty()) return fail(FS_EINVAL, "fs_streamlines has not been called");
    if (offsets) memcpy(offsets, s->sl_offsets.data(), s->sl_offsets.size() * sizeof(long));
    if (points && !s->sl_points.empty()) memcpy(points, s->sl_points.data(), s->sl_points.size() * sizeof(double));
    if (norm_speed && !s->sl_norm.empty()) memcpy(norm_speed, s->sl_norm.data(), s->sl_norm.size() * sizeof(double));
    return FS_OK;
}

int fs_obstacle_surface(fs_sim* s, long* n_vertices, long* n_triangles)
{
    ENGINE_OR_RETURN(s);
    int rc = s->eng->obstacle_surface();
    if (rc) return rc;
    if (n_vertices) *n_vertices = (long)(s->surf_verts.size() / 3);
    if (n_triangles) *n_triangles = (long)(s->surf_tris.size() / 3);
    return FS_OK;
}

int fs_obstacle_surface_fetch(fs_sim* s, float* vertices, int* triangles)
{
    if (!s) return fail(FS_EINVAL, "null handle");
    if (!s->surf_valid) return fail(FS_EINVAL, "fs_obstacle_surface has not been called");
    if (vertices && !s->surf_verts.empty()) memcpy(vertices, s->surf_verts.data(), s->surf_verts.size() * sizeof(float));
    if (triangles && !s->surf_tris.empty()) memcpy(triangles, s->surf_tris.data(), s->surf_tris.size() * sizeof(int));
    return FS_OK;
}

int fs_surface_case_table(int config, int* edges)
{
    if (!edges) return fail(FS_EINVAL, "null buffer");
    int n = fs::surface_case(config, edges);
    if (n < 0) return fail(FS_EINVAL, "bad cube configuration %d (0..255)", config);
    return n;
}

int fs_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(FS_EINVAL, "null id buffer");
    std::string err;
    if (fs::Comm::unique_id(id_out, &err)) return fail(FS_ECOMM, "%s", err.c_str());
    return FS_OK;
}

int fs_comm_selftest(void)
{
    std::string err;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int rc = fs::Comm::selftest(st, &err);
    hipStreamDestroy(st);
    if (rc) return fail(FS_ECOMM, "RCCL self-test: %s", err.c_str());
    return FS_OK;
}

int fs_comm_init(fs_sim* s, int rank, int nranks, const void* id)
{
    if (!s || !id) return fail(FS_EINVAL, "null argument");
    if (s->eng) return fail(FS_EINVAL, "fs_comm_init must precede first use of the handle");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(FS_EINVAL, "bad rank %d of %d", rank, nranks);
    if (s->D % nranks) return fail(FS_EINVAL, "depth %d does not divide over %d slabs", s->D, nranks);
    if (nranks > 1 && s->D / nranks < 2) return fail(FS_EINVAL, "a slab needs at least 2 planes (two-deep halos), got %d", s->D / nranks);
    if (nranks == 1) return FS_OK;
    hipSetDevice(s->device);
    if (s->comm.init(rank, nranks, id)) return fail(FS_ECOMM, "%s", s->comm.last_error());
    if (rank == 0 && !s->quiet)          // one line of provenance for multi-GPU logs
        fprintf(stderr, "fluidsim: %d z-slabs of %d planes, halo transport: %s\n", nranks, s->D / nranks, s->comm.transport_name());
    return FS_OK;
}

const char* fs_comm_transport(fs_sim* s)
{
    if (!s) return "";
    return s->comm.active() ? s->comm.transport_name() : "single GPU";
}

}  // extern "C"

"""ctypes front-end for the CPU oracle (oracle/cpu_ref.c) and, when it has been built in
this container, the compiled reference itself (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (fluid_simulation_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# field selectors shared by cpu_ref.c, ref_harness.cpp and include/fluidsim.h
DENS, VX, VY, VZ, OBS, P, DIV, VX0, VY0, VZ0, BUF = range(11)
FIELD_NAMES = ["dens", "v_x", "v_y", "v_z", "obs", "pressure", "divergence",
               "v_x_prev", "v_y_prev", "v_z_prev", "buffer"]
GS_LEX, JACOBI, RBSOR, MG = 0, 1, 2, 3


def build(force=False):
    """Compile the oracle (and the reference shim if /root/reference is present)."""
    need = force or not all(os.path.exists(os.path.join(HERE, n)) for n in ("libcpu_ref.so", "libcpu_ref64.so"))
    ref_missing = os.path.exists("/root/reference/simulation.cpp") and not os.path.exists(
        os.path.join(HERE, "_ref", "libref.so"))
    if need or ref_missing:
        subprocess.check_call(["make", "-C", HERE] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)


_gomp = None


def _set_omp_threads(n):
    """The oracle and the reference shim share one libgomp: its thread count is process-global, so
    every call into either library sets the count it wants first."""
    global _gomp
    if _gomp is None:
        try:
            _gomp = C.CDLL("libgomp.so.1")
        except OSError:
            _gomp = False
    if _gomp:
        _gomp.omp_set_num_threads(C.c_int(int(n)))


def default_threads(cap=8):
    """Never the OpenMP default: a GPU box reports all 256 host cores but grants 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, cap))


def _load(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


class _Sim:
    """Common shape of both back-ends: create / mutate / step / read back."""
    prefix = None
    dtype = np.float32
    threads = 1

    def __init__(self, lib, w, h, d, iter=1, speed=30, dt=0.05, diff=2.0e-5, visc=1.5e-5, acc=15):
        self.lib = lib
        self.W, self.H, self.D = w, h, d
        self.shape = (d + 2, h + 2, w + 2)          # C-order view of x-fastest padded arrays
        self.n = (w + 2) * (h + 2) * (d + 2)
        fn = getattr(lib, self.prefix + "create")
        fn.restype = C.c_void_p
        fn.argtypes = [C.c_int] * 5 + [C.c_float] * 3 + [C.c_int]
        self.h = C.c_void_p(fn(w, h, d, iter, speed, dt, diff, visc, acc))
        if not self.h:
            raise MemoryError("oracle create failed")

    def _call(self, name, *args, restype=None):
        _set_omp_threads(self.threads)
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        return fn(self.h, *args)

    def close(self):
        if self.h:
            self._call("destroy")
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_obstacle(self, x, y, z):
        self._call("add_obstacle", C.c_int(x), C.c_int(y), C.c_int(z))

    def add_density(self, x, y, z, a):
        self._call("add_density", C.c_int(x), C.c_int(y), C.c_int(z), C.c_float(a))

    def set_velocity(self, x, y, z, ax, ay, az):
        self._call("set_velocity", C.c_int(x), C.c_int(y), C.c_int(z), C.c_float(ax), C.c_float(ay), C.c_float(az))

    def get(self, which):
        out = np.empty(self.n, dtype=self.dtype)
        rc = self._call("get_field", C.c_int(which), out.ctypes.data_as(C.c_void_p), restype=C.c_int)
        assert rc == 0
        return out.reshape(self.shape)

    def set(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype).reshape(-1)
        assert a.size == self.n
        rc = self._call("set_field", C.c_int(which), a.ctypes.data_as(C.c_void_p), restype=C.c_int)
        assert rc == 0

    def set_mask(self, mask):
        """mask: bool/0-1 array of padded shape (D+2,H+2,W+2); only interior cells may be set."""
        self.set(OBS, np.asarray(mask, dtype=self.dtype))

    def step_only(self):
        self._call("step_only")

    def run_one(self):
        self._call("run_one")

    def diffuse(self, b, field, prev):
        self._call("diffuse", C.c_int(b), C.c_int(field), C.c_int(prev))

    def project(self):
        self._call("project")

    def advect(self, b, field, prev):
        self._call("advect", C.c_int(b), C.c_int(field), C.c_int(prev))

    def set_bounds(self, b, field):
        self._call("set_bounds", C.c_int(b), C.c_int(field))

    def linear_solver(self, b, field, prev, a, c):
        self._call("linear_solver", C.c_int(b), C.c_int(field), C.c_int(prev), C.c_float(a), C.c_float(c))


class Oracle(_Sim):
    """The C restatement.  solver = GS_LEX (reference order), JACOBI, RBSOR (the build's own
    optional red-black SOR, relaxation factor `omega`; not in the reference) or MG (the build's own
    multigrid pressure solve, oracle/cpu_ref_mg.h; `mg` = (V-cycles, pre-, post-smoothing steps,
    coarsest-level iterations); not in the reference)."""
    prefix = "cr_"

    def __init__(self, w, h, d, solver=GS_LEX, fp64=False, threads=None, omega=1.0, mg=None, **kw):
        lib = _load("libcpu_ref64.so" if fp64 else "libcpu_ref.so")
        self.dtype = np.float64 if fp64 else np.float32
        super().__init__(lib, w, h, d, **kw)
        # GS_LEX is only the reference's one-thread order at one thread; Jacobi is thread-independent
        self.threads = threads if threads is not None else (1 if solver == GS_LEX else default_threads())
        self._call("set_solver", C.c_int(solver))
        if solver == RBSOR:
            self._call("set_omega", C.c_float(omega))
        if mg is not None:
            self._call("set_mg", *(C.c_int(int(v)) for v in mg))

    def load_stl(self, path, scale=0.8, rot=(0.0, 0.0, 0.0), translate=(0.0, 0.0, 0.0), seed=1):
        return self._call("load_stl", C.c_char_p(os.fsencode(path)), C.c_float(scale),
                          C.c_float(rot[0]), C.c_float(rot[1]), C.c_float(rot[2]),
                          C.c_float(translate[0]), C.c_float(translate[1]), C.c_float(translate[2]),
                          C.c_uint(seed), restype=C.c_long)

    def dump_frame(self, directory, append=True):
        rc = self._call("dump_frame", C.c_char_p(os.fsencode(directory)), C.c_int(1 if append else 0), restype=C.c_int)
        if rc:
            raise OSError("dump_frame failed rc=%d" % rc)


def have_reference():
    return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))


class Reference(_Sim):
    """The compiled, unmodified reference.  Deterministic only at one thread (the default here);
    bench.py's cpu_baseline passes threads=<cores> to time it the way a user would run it."""
    prefix = "ref_"

    def __init__(self, w, h, d, threads=1, **kw):
        super().__init__(C.CDLL(os.path.join(HERE, "_ref", "libref.so")), w, h, d, **kw)
        self.threads = threads

    def load_stl(self, path, scale=0.8, rot=(0.0, 0.0, 0.0), translate=(0.0, 0.0, 0.0)):
        self._call("load_stl", C.c_char_p(os.fsencode(path)), C.c_float(scale),
                   C.c_float(rot[0]), C.c_float(rot[1]), C.c_float(rot[2]),
                   C.c_float(translate[0]), C.c_float(translate[1]), C.c_float(translate[2]))

    def thread_seed(self):
        fn = self.lib.ref_thread_seed
        fn.restype = C.c_uint
        return int(fn())

    def run(self):
        self._call("run")

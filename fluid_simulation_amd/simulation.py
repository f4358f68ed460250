"""Host-side mirror of the reference's solver interface, on top of the C ABI.

`Simulation` has the constructor, public data members and methods of the reference's
`class Simulation` (simulation.h:42-91) under the same names, and `loadSTLIntoObstacles`
has the signature of the reference's free function (object_loader.h:7-17), so code and
tests written against the reference read the same here.  Everything numeric happens in
libfluidsim.so on the MI355X.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import FluidsimError, check  # noqa: F401


class Simulation:
    """Simulation(w, h, d, iter, speed=30, dt=0.05, diff=2.0e-5, visc=1.5e-5, acc=15)
    -- simulation.h:59-64.  Extra keyword options map to fs_set_option."""

    def __init__(self, w, h, d, iter, speed=30, dt=0.05, diff=2.0e-5, visc=1.5e-5, acc=15,
                 precision="fp32", solver="jacobi", **options):
        L = _lib.lib()
        self._L = L
        self._h = L.fs_create(int(w), int(h), int(d), int(iter), int(speed), float(dt), float(diff), float(visc),
                              int(acc))
        if not self._h:
            raise FluidsimError(_lib.EHIP, (L.fs_last_error() or b"").decode(errors="replace"))
        self._h = C.c_void_p(self._h)
        self.precision = precision
        self.dtype = np.float64 if precision == "fp64" else np.float32
        self.set_option("precision", precision)
        self.set_option("solver", solver)
        for k, v in options.items():
            self.set_option(k, v)

    # -- lifetime -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.fs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key, value):
        if isinstance(value, bool):
            value = "1" if value else "0"
        check(self._L.fs_set_option(self._h, key.encode(), str(value).encode()))

    # -- public data members of the reference class (simulation.h:44-54) ---------------
    def _geti(self, name):
        v = C.c_int()
        check(self._L.fs_get_int(self._h, name.encode(), C.byref(v)))
        return v.value

    def _getf(self, name):
        v = C.c_float()
        check(self._L.fs_get_float(self._h, name.encode(), C.byref(v)))
        return v.value

    width = property(lambda s: s._geti("width"))
    height = property(lambda s: s._geti("height"))
    depth = property(lambda s: s._geti("depth"))
    speed = property(lambda s: s._geti("speed"), lambda s, v: check(s._L.fs_set_int(s._h, b"speed", int(v))))
    acc = property(lambda s: s._geti("acc"), lambda s, v: check(s._L.fs_set_int(s._h, b"acc", int(v))))
    iter = property(lambda s: s._geti("iter"), lambda s, v: check(s._L.fs_set_int(s._h, b"iter", int(v))))
    dt = property(lambda s: s._getf("dt"), lambda s, v: check(s._L.fs_set_float(s._h, b"dt", float(v))))
    diff = property(lambda s: s._getf("diff"), lambda s, v: check(s._L.fs_set_float(s._h, b"diff", float(v))))
    visc = property(lambda s: s._getf("visc"), lambda s, v: check(s._L.fs_set_float(s._h, b"visc", float(v))))
    local_depth = property(lambda s: s._geti("local_depth"))
    z_offset = property(lambda s: s._geti("z_offset"))

    # -- methods of the reference class ------------------------------------------------
    def run(self):
        """Simulation::run(), simulation.cpp:49-91."""
        check(self._L.fs_run(self._h))

    def step(self):
        """Simulation::step(), simulation.cpp:96-150."""
        check(self._L.fs_step(self._h))

    def addObstacle(self, x, y, z):
        check(self._L.fs_add_obstacle(self._h, x, y, z))

    def addDensity(self, x, y, z, amount):
        check(self._L.fs_add_density(self._h, x, y, z, amount))

    def setVelocity(self, x, y, z, amount_x, amount_y, amount_z):
        check(self._L.fs_set_velocity(self._h, x, y, z, amount_x, amount_y, amount_z))

    # -- the rest of the C ABI ---------------------------------------------------------
    def run_one(self):
        """One iteration of run()'s time loop: inlet density, buffer = dens, step()."""
        check(self._L.fs_run_one(self._h))

    def sync(self):
        check(self._L.fs_sync(self._h))

    @property
    def shape(self):
        """C-order shape of a field as the viewers reshape it (gui.py:228-231): (D+2, H+2, W+2);
        under z-slabs D is this rank's local depth."""
        return (self.local_depth + 2, self.height + 2, self.width + 2)

    def get(self, which, dtype=None):
        dtype = np.dtype(dtype or self.dtype)
        n = self._L.fs_padded_size(self._h)
        out = np.empty(n, dtype=dtype)
        check(self._L.fs_get_field(self._h, which, out.ctypes.data_as(C.c_void_p), n, dtype.itemsize))
        return out.reshape(self.shape)

    def set(self, which, arr):
        a = np.ascontiguousarray(arr)
        if a.dtype not in (np.float32, np.float64, np.uint8):
            a = a.astype(self.dtype)
        a = a.reshape(-1)
        check(self._L.fs_set_field(self._h, which, a.ctypes.data_as(C.c_void_p), a.size, a.dtype.itemsize))

    def set_mask(self, mask):
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8).reshape(-1)
        check(self._L.fs_set_obstacle_mask(self._h, m.ctypes.data_as(C.c_void_p), m.size))

    def set_bounds(self, b, field):
        check(self._L.fs_set_bounds(self._h, b, field))

    def linear_solver(self, b, field, prev, a, c):
        check(self._L.fs_linear_solver(self._h, b, field, prev, a, c))

    def diffuse(self, b, field, prev):
        check(self._L.fs_diffuse(self._h, b, field, prev))

    def project(self):
        check(self._L.fs_project(self._h))

    def advect(self, b, field, prev):
        check(self._L.fs_advect(self._h, b, field, prev))

    def dump_frame(self):
        check(self._L.fs_dump_frame(self._h))

    def stats(self, which):
        s, lo, hi = C.c_double(), C.c_double(), C.c_double()
        check(self._L.fs_field_stats(self._h, which, C.byref(s), C.byref(lo), C.byref(hi)))
        return s.value, lo.value, hi.value

    def timing(self, family):
        ms, n = C.c_double(), C.c_long()
        check(self._L.fs_get_timing(self._h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_timing(self):
        check(self._L.fs_reset_timing(self._h))

    def streamlines(self, density=30, proximity=2, max_length=100, step_size=0.2, vel_change_threshold=0.1):
        """generate_streamlines of the reference's viewer (GUI/utils.py:118-213, defaults from
        GUI/config.py:18-23) from the fields on the device.  Returns (lines, norm_speeds): a list
        of (n_i, 3) float64 arrays in the viewer's (x, y, z) padded index space, in the reference's
        order, and for each line the value the viewer passes to `config.density_cmap`."""
        nl, npts = C.c_long(), C.c_long()
        check(self._L.fs_streamlines(self._h, int(density), float(proximity), int(max_length), float(step_size),
                                     float(vel_change_threshold), C.byref(nl), C.byref(npts)))
        off = np.zeros(nl.value + 1, dtype=np.int64)
        pts = np.zeros((npts.value, 3), dtype=np.float64)
        norm = np.zeros(nl.value, dtype=np.float64)
        check(self._L.fs_streamlines_fetch(self._h, off.ctypes.data, pts.ctypes.data, norm.ctypes.data))
        return [pts[off[i]:off[i + 1]].copy() for i in range(nl.value)], norm

    def obstacle_surface(self):
        """generate_obstacle_mesh of the reference's viewer (GUI/utils.py:10-38) from `obs` on the device:
        (vertices (n, 3) float32 in the viewer's padded (x, y, z) index space, faces (m, 3) int32)."""
        nv, nt = C.c_long(), C.c_long()
        check(self._L.fs_obstacle_surface(self._h, C.byref(nv), C.byref(nt)))
        verts = np.zeros((nv.value, 3), dtype=np.float32)
        faces = np.zeros((nt.value, 3), dtype=np.int32)
        check(self._L.fs_obstacle_surface_fetch(self._h, verts.ctypes.data, faces.ctypes.data))
        return verts, faces

    def time_sweeps(self, b, field, prev, a, c, reps):
        ms = C.c_double()
        check(self._L.fs_time_sweeps(self._h, b, field, prev, a, c, reps, C.byref(ms)))
        return ms.value

    def comm_transport(self):
        return (self._L.fs_comm_transport(self._h) or b"").decode(errors="replace")

    def comm_init(self, rank, nranks, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        check(self._L.fs_comm_init(self._h, rank, nranks, buf))


def comm_unique_id(transport="rccl"):
    """128-byte id for fs_comm_init.  "rccl": an ncclUniqueId (RCCL over xGMI, one GPU per rank).
    "shm": a host-staged development transport through POSIX shared memory, for ranks that
    are processes on one host and may share a GPU (tests on a 1-GPU box).
    "ipc": a stream-ordered device-to-device transport between rank processes of one host (hipIpc-mapped
    arrays, copy engines, device-side handshakes; ranks may share a GPU) -- csrc/ipc.h."""
    if transport in ("shm", "ipc"):
        name = "%s:/fs_slab_%d_%s" % ("FSSHM" if transport == "shm" else "FSIPC", os.getpid(), os.urandom(4).hex())
        return name.encode().ljust(_lib.COMM_ID_BYTES, b"\0")
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    check(_lib.lib().fs_comm_unique_id(buf))
    return buf.raw


def loadSTLIntoObstacles(stlFile, sim, scale=0.8, rot_x=0.0, rot_y=0.0, rot_z=0.0,
                         translate_x=0.0, translate_y=0.0, translate_z=0.0):
    """loadSTLIntoObstacles -- object_loader.h:7-17.  Like the reference, a file that cannot
    be read leaves the tunnel empty and is not an exception; returns the number of accepted
    sample points ("Added N obstacle points"), or None when the file could not be loaded."""
    added = C.c_long(0)
    rc = sim._L.fs_load_stl(sim._h, os.fsencode(stlFile), scale, rot_x, rot_y, rot_z,
                            translate_x, translate_y, translate_z, C.byref(added))
    if rc == _lib.EIO:
        return None
    check(rc)
    return added.value

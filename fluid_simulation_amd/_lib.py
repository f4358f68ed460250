"""ctypes binding of libfluidsim.so (C ABI: include/fluidsim.h).

The shared library is the product; this module only declares its signatures.  There is no
Python or CPU implementation behind it: if the library is missing or no MI355X is usable,
the error is raised to the caller.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLUIDSIM_LIB: development override (e.g. a host-sanitizer build of the same sources)
LIB_PATH = os.environ.get("FLUIDSIM_LIB") or os.path.join(_HERE, "libfluidsim.so")

OK, EINVAL, EIO, EHIP, ECOMM, ENOMEM = 0, -1, -2, -3, -4, -5
DENS, VX, VY, VZ, OBS, PRESSURE, DIVERGENCE, VX_PREV, VY_PREV, VZ_PREV, BUFFER = range(11)
FIELD_NAMES = ["dens", "v_x", "v_y", "v_z", "obs", "pressure", "divergence",
               "v_x_prev", "v_y_prev", "v_z_prev", "buffer"]
COMM_ID_BYTES = 128


class FluidsimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fluidsim error %d: %s" % (code, msg))
        self.code = code


_lib = None

_SIGNATURES = {
    "fs_create": (C.c_void_p, [C.c_int] * 5 + [C.c_float] * 3 + [C.c_int]),
    "fs_destroy": (C.c_int, [C.c_void_p]),
    "fs_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "fs_get_int": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "fs_set_int": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "fs_get_float": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float)]),
    "fs_set_float": (C.c_int, [C.c_void_p, C.c_char_p, C.c_float]),
    "fs_add_obstacle": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fs_add_density": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]),
    "fs_set_velocity": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]),
    "fs_load_stl": (C.c_int, [C.c_void_p, C.c_char_p] + [C.c_float] * 7 + [C.POINTER(C.c_long)]),
    "fs_set_obstacle_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "fs_step": (C.c_int, [C.c_void_p]),
    "fs_run_one": (C.c_int, [C.c_void_p]),
    "fs_run": (C.c_int, [C.c_void_p]),
    "fs_sync": (C.c_int, [C.c_void_p]),
    "fs_set_bounds": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "fs_linear_solver": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
    "fs_diffuse": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fs_project": (C.c_int, [C.c_void_p]),
    "fs_advect": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fs_get_field": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]),
    "fs_set_field": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]),
    "fs_padded_size": (C.c_size_t, [C.c_void_p]),
    "fs_dump_frame": (C.c_int, [C.c_void_p]),
    "fs_field_stats": (C.c_int, [C.c_void_p, C.c_int] + [C.POINTER(C.c_double)] * 3),
    "fs_get_timing": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "fs_reset_timing": (C.c_int, [C.c_void_p]),
    "fs_time_sweeps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                 C.POINTER(C.c_double)]),
    "fs_streamlines": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double,
                                 C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "fs_streamlines_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fs_obstacle_surface": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "fs_obstacle_surface_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fs_surface_case_table": (C.c_int, [C.c_int, C.c_void_p]),
    "fs_comm_unique_id": (C.c_int, [C.c_void_p]),
    "fs_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fs_comm_selftest": (C.c_int, []),
    "fs_comm_transport": (C.c_char_p, [C.c_void_p]),
    "fs_last_error": (C.c_char_p, []),
    "fs_version": (C.c_char_p, []),
}


def exported_symbols():
    """Every entry point include/fluidsim.h declares."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FluidsimError(EHIP, "%s is missing: build it with `python -c 'import __graft_entry__ as g; "
                                "g.build()'` or `make -C fluid_simulation_amd/csrc` (no CPU fallback exists)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("FLUIDSIM_LIB"):       # development: an older build named explicitly (tools/ab_lib.py)
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise FluidsimError(rc, (lib().fs_last_error() or b"").decode(errors="replace"))
    return rc

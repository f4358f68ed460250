"""CPU checks of the drop-in boundary: the shared library loads, exports every symbol that
include/fluidsim.h declares, the ctypes table covers all of them, and without a GPU the product
fails loudly instead of falling back to anything."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "fluidsim.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def libpath():
    from fluid_simulation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "fluid_simulation_amd", "csrc"), "-j4"])
    return _lib.LIB_PATH


def test_header_declares_the_reference_surface():
    syms = declared_symbols()
    for need in ("fs_create", "fs_destroy", "fs_run", "fs_step", "fs_add_obstacle", "fs_add_density",
                 "fs_set_velocity", "fs_load_stl", "fs_get_field", "fs_set_field", "fs_last_error"):
        assert need in syms
    text = open(HEADER).read()
    for cite in ("simulation.h:59-64", "simulation.h:69", "simulation.h:74", "simulation.h:79", "simulation.h:84",
                 "simulation.h:89", "object_loader.h:7-17"):
        assert cite in text, cite


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from fluid_simulation_amd import _lib
    assert sorted(_lib.exported_symbols()) == declared_symbols()


def test_no_cpu_fallback(libpath):
    """In a container without a GPU the product refuses to run (and never touches the oracle)."""
    import torch
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    import fluid_simulation_amd as F
    with pytest.raises(F.FluidsimError) as e:
        F.Simulation(8, 8, 8, 1)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)
    src = "".join(open(os.path.join(ROOT, "fluid_simulation_amd", f)).read()
                  for f in ("__init__.py", "_lib.py", "simulation.py", "shapes.py", "viewer.py", "dist.py"))
    assert "oracle" not in src and "cpu_ref" not in src


def test_version_string(libpath):
    lib = ctypes.CDLL(libpath)
    lib.fs_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.fs_version()


def test_integration_md_cpp_shim_compiles(tmp_path):
    """The drop-in `Simulation` shim shown in INTEGRATION.md must compile against include/fluidsim.h
    (and, being the reference's main() verbatim on top of it, link against the library)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", text, flags=re.S)
    assert blocks, "no C++ block in INTEGRATION.md"
    src = tmp_path / "shim.cpp"
    src.write_text(blocks[0] + """
int main() {
    Simulation sim(128, 64, 64, 100, 30);
    loadSTLIntoObstacles("/nonexistent.stl", sim, 2.0f, 90.0f, 0.0f, 0.0f, -16.0f, 0.0f, 0.0f);
    sim.addObstacle(1, 1, 1); sim.addDensity(1, 1, 1, 0.5f); sim.setVelocity(1, 1, 1, 1.f, 0.f, 0.f);
    sim.run();
    return sim.width + sim.acc;
}
""")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "shim.o")])


def test_header_is_plain_c(tmp_path):
    """The ABI header must be usable from C (no C++-only constructs outside the extern "C" guard)."""
    src = tmp_path / "use.c"
    src.write_text('#include "fluidsim.h"\nint main(void) { fs_sim* s = 0; (void)s; return FS_NFIELDS == 11 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c",
                           str(src), "-o", str(tmp_path / "use.o")])

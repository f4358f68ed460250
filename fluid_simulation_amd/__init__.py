"""MI355X-native 3-D wind-tunnel solver: the per-step hot path of Ghundi/fluid_simulation
(`Simulation::step()`), rebuilt as hand-written HIP kernels behind a C ABI
(include/fluidsim.h).  Importing the package loads nothing; the shared library is bound
when the first `Simulation` is created and there is no CPU fallback.
"""
from ._lib import (BUFFER, DENS, DIVERGENCE, FIELD_NAMES, OBS, PRESSURE, VX, VX_PREV, VY, VY_PREV, VZ, VZ_PREV,
                   FluidsimError)
from .simulation import Simulation, comm_unique_id, loadSTLIntoObstacles

__all__ = ["Simulation", "loadSTLIntoObstacles", "comm_unique_id", "FluidsimError", "FIELD_NAMES",
           "DENS", "VX", "VY", "VZ", "OBS", "PRESSURE", "DIVERGENCE", "VX_PREV", "VY_PREV", "VZ_PREV", "BUFFER"]

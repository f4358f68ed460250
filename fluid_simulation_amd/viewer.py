"""Viewer-side helper: the streamlines of the reference's GUI, computed on the GPU.

`generate_streamlines` takes what GUI/utils.py:118 `generate_streamlines` takes -- the last frame's
`v_x, v_y, v_z, obs` arrays including padding, transposed to (x, y, z) as GUI/main_window.py:227-230
does -- and returns the same `(streamlines, streamline_colors)` pair, so a maintainer of the viewer
swaps one call:

    # streamlines, colors = utils.generate_streamlines(vx=vx, vy=vy, vz=vz, obs_data=obs)
    from fluid_simulation_amd.viewer import generate_streamlines
    streamlines, colors = generate_streamlines(vx, vy, vz, obs, cmap=config.density_cmap)

The parameters default to GUI/config.py:18-23.  Without `cmap` the second value is the list of
numbers the reference passes to its colour map (utils.py:202-205)."""
import numpy as np

from . import _lib
from .simulation import Simulation

_handles = {}


def generate_streamlines(vx, vy, vz, obs_data, max_length=100, density=30, proximity=2, step_size=0.2,
                         vel_change_threshold=0.1, cmap=None):
    shape = tuple(int(n) - 2 for n in vx.shape)          # (W, H, D): the arrays carry the padding
    if min(shape) < 1 or vy.shape != vx.shape or vz.shape != vx.shape or obs_data.shape != vx.shape:
        raise ValueError("expected four padded arrays of one shape (W+2, H+2, D+2)")
    sim = _handles.get(shape)
    if sim is None:
        sim = _handles[shape] = Simulation(shape[0], shape[1], shape[2], 1, quiet=1, dump_every=0)
    back = (2, 1, 0)                                     # the library's arrays are (z, y, x), like the dump files
    sim.set(_lib.VX, np.ascontiguousarray(np.transpose(vx, back), dtype=np.float32))
    sim.set(_lib.VY, np.ascontiguousarray(np.transpose(vy, back), dtype=np.float32))
    sim.set(_lib.VZ, np.ascontiguousarray(np.transpose(vz, back), dtype=np.float32))
    sim.set_mask(np.ascontiguousarray(np.transpose(obs_data, back) > 0.5))
    lines, norm = sim.streamlines(density=density, proximity=proximity, max_length=max_length, step_size=step_size,
                                  vel_change_threshold=vel_change_threshold)
    if cmap is None:
        return lines, list(norm)
    return lines, [np.array(cmap(float(v))) for v in norm]


def generate_obstacle_mesh(obs_data):
    """GUI/utils.py:10 `generate_obstacle_mesh` on the GPU: takes the padded obstacle array transposed to
    (x, y, z) as GUI/main_window.py:204 passes it and returns the same dictionary ('vertexes', 'faces',
    'vertex_colors' -- solid gray, utils.py:19-24; three empty arrays when there is no obstacle).  The mesh is
    the 0.5 iso-surface like scikit-image's, but vertex / face order and the cut of ambiguous cubes are this
    library's (parity unpinned: scikit-image is not available to compare with)."""
    shape = tuple(int(n) - 2 for n in obs_data.shape)
    if len(shape) != 3 or min(shape) < 1:
        raise ValueError("expected the padded obstacle array (W+2, H+2, D+2)")
    sim = _handles.get(shape)
    if sim is None:
        sim = _handles[shape] = Simulation(shape[0], shape[1], shape[2], 1, quiet=1, dump_every=0)
    sim.set(_lib.OBS, np.ascontiguousarray(np.transpose(obs_data, (2, 1, 0)), dtype=np.float32))
    verts, faces = sim.obstacle_surface()
    if verts.shape[0] == 0:
        return {"vertexes": np.array([]), "faces": np.array([]), "vertex_colors": np.array([])}
    colors = np.ones((verts.shape[0], 4))
    colors[:, :3] = 0.5
    return {"vertexes": verts.astype(np.float64), "faces": faces, "vertex_colors": colors}

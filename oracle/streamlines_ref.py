"""Test infrastructure (only tests/ may import this): CPU restatement of the streamline code of
the reference's viewer, GUI/utils.py:40-213, with the parameters of GUI/config.py:18-23 passed in.

PARITY UNPINNED: the reference module itself cannot be imported in this image -- it imports
scikit-image at module level (utils.py:6) and PyQt6 further up, neither of which is installed --
and the reference holds no fixtures for it.  The functions below follow the source statement by
statement (cited per function) with numpy scalars exactly where the reference has them, so that
float32 grid values meet float64 coordinates the same way; colours are left as the number the
reference feeds its colour map (utils.py:202-205), because that map lives in GUI/config.py."""
import numpy as np


def interpolate_scalar(grid, x, y, z):
    """utils.py:40-76"""
    x = np.clip(x, 0, grid.shape[0] - 1.001)
    y = np.clip(y, 0, grid.shape[1] - 1.001)
    z = np.clip(z, 0, grid.shape[2] - 1.001)
    x0, y0, z0 = int(x), int(y), int(z)
    x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
    xd, yd, zd = x - x0, y - y0, z - z0
    c000 = grid[x0, y0, z0]
    c100 = grid[x1, y0, z0]
    c010 = grid[x0, y1, z0]
    c110 = grid[x1, y1, z0]
    c001 = grid[x0, y0, z1]
    c101 = grid[x1, y0, z1]
    c011 = grid[x0, y1, z1]
    c111 = grid[x1, y1, z1]
    c00 = c000 * (1 - xd) + c100 * xd
    c01 = c001 * (1 - xd) + c101 * xd
    c10 = c010 * (1 - xd) + c110 * xd
    c11 = c011 * (1 - xd) + c111 * xd
    c0 = c00 * (1 - yd) + c10 * yd
    c1 = c01 * (1 - yd) + c11 * yd
    return c0 * (1 - zd) + c1 * zd


def interpolate_vector(vx, vy, vz, x, y, z):
    """utils.py:78-83"""
    return np.array([interpolate_scalar(vx, x, y, z), interpolate_scalar(vy, x, y, z), interpolate_scalar(vz, x, y, z)])


def integrate_part(start_pos, vx, vy, vz, obs, max_steps, direction, step_size, dims):
    """utils.py:85-115 (_integrate_streamline_part); dims = (config.width, height, depth)"""
    points = [start_pos]
    velocities = [interpolate_vector(vx, vy, vz, *start_pos)]
    pos = start_pos.copy()
    for _ in range(max_steps):
        vec = interpolate_vector(vx, vy, vz, pos[0], pos[1], pos[2])
        speed = np.linalg.norm(vec)
        if speed < 1e-6:
            break
        step = direction * (vec / speed) * step_size
        pos = pos + step
        if np.any(np.isnan(pos)) or np.any(np.isinf(pos)):
            break
        if not (1 <= pos[0] < dims[0] - 1 and 1 <= pos[1] < dims[1] - 1 and 1 <= pos[2] < dims[2] - 1):
            break
        if interpolate_scalar(obs, pos[0], pos[1], pos[2]) > 0.5:
            break
        points.append(pos.copy())
        velocities.append(vec)
    return points, velocities


def generate_streamlines(vx, vy, vz, obs, density=30, proximity=2, max_length=100, step_size=0.2, threshold=0.1):
    """utils.py:118-213.  Arrays are indexed [x, y, z] and include the padding, as main_window.py:227-231
    hands them over.  Returns (list of (n, 3) arrays, list of the colour-map arguments)."""
    dims = vx.shape
    lines, norms = [], []
    idx = np.where(obs > 0.5)
    if len(idx[0]) == 0:
        return lines, norms
    omin = np.array([np.min(idx[0]), np.min(idx[1]), np.min(idx[2])]) - (proximity / 10)
    omax = np.array([np.max(idx[0]), np.max(idx[1]), np.max(idx[2])]) + (proximity / 10)
    x_seeds = np.linspace(1, dims[0] - 2, density)
    y_seeds = np.linspace(1, dims[1] - 2, density // 2)
    z_seeds = np.linspace(1, dims[2] - 2, density // 2)
    denom = np.max([vx, vy, vz]) + 1e-6
    for zs in z_seeds:
        for ys in y_seeds:
            for xs in x_seeds:
                start = np.array([xs, ys, zs])
                if (start[0] < omin[0] or start[0] > omax[0] or start[1] < omin[1] or start[1] > omax[1] or
                        start[2] < omin[2] or start[2] > omax[2]):
                    continue
                if obs[int(xs), int(ys), int(zs)] > 0.5:
                    continue
                bp, bv = integrate_part(start, vx, vy, vz, obs, max_length // 2, -1.0, step_size, dims)
                fp, fv = integrate_part(start, vx, vy, vz, obs, max_length // 2, 1.0, step_size, dims)
                full = bp[::-1][:-1] + fp
                fvel = bv[::-1][:-1] + fv
                if len(full) <= 5:
                    continue
                max_change = 0.0
                for i in range(1, len(fvel)):
                    change = np.linalg.norm(fvel[i] - fvel[i - 1])
                    if change > max_change:
                        max_change = change
                if max_change < threshold:
                    continue
                near = False
                for i in range(0, len(full), 3):
                    pt = full[i]
                    if omin[0] <= pt[0] <= omax[0] and omin[1] <= pt[1] <= omax[1] and omin[2] <= pt[2] <= omax[2]:
                        near = True
                        break
                if not near:
                    continue
                speeds = [np.linalg.norm(v) for v in fvel]
                max_speed = max(speeds) if speeds else 0.0
                norms.append(min(max_speed / denom, 1.0))
                lines.append(np.array(full))
    return lines, norms

"""CPU checks of oracle/streamlines_ref.py (the restatement of GUI/utils.py:40-213 the GPU test
compares fs_streamlines with).  The reference holds no fixtures for this code and its module does
not import here (scikit-image, PyQt6), so these are known-answer cases worked out by hand from the
source: parity with the reference itself is unpinned and says so in the oracle's header."""
import numpy as np

from oracle import streamlines_ref as R


def tunnel(nx=20, ny=12, nz=10, block=True):
    vx = np.ones((nx, ny, nz), dtype=np.float32)
    vy = np.zeros_like(vx)
    vz = np.zeros_like(vx)
    obs = np.zeros_like(vx)
    if block:
        obs[9:11, 5:7, 4:6] = 1.0
    else:                                   # two separate cells: their bounding box is mostly free
        obs[9, 5, 4] = obs[11, 7, 6] = 1.0
    return vx, vy, vz, obs


def test_trilinear_interpolation_known_answers():
    g = np.arange(4 * 3 * 2, dtype=np.float32).reshape(4, 3, 2)      # g[x, y, z] = 6x + 2y + z: trilinear is exact
    assert R.interpolate_scalar(g, 1.25, 0.5, 0.75) == 6 * 1.25 + 2 * 0.5 + 0.75
    # coordinates are clipped to [0, n - 1.001] (utils.py:43-45)
    assert np.isclose(R.interpolate_scalar(g, 99.0, -5.0, 0.0), 6 * 2.999)
    assert R.interpolate_scalar(g, 0.0, 0.0, 0.0).dtype == np.float64


def test_uniform_flow_gives_straight_lines_with_the_configured_step():
    vx, vy, vz, obs = tunnel()
    start = np.array([3.0, 2.5, 2.5])
    pts, vel = R.integrate_part(start, vx, vy, vz, obs, 10, 1.0, 0.2, vx.shape)
    assert len(pts) == 11 and len(vel) == 11
    assert np.allclose(np.array(pts)[:, 0], 3.0 + 0.2 * np.arange(11)) and np.all(np.array(pts)[:, 1:] == 2.5)
    back, _ = R.integrate_part(start, vx, vy, vz, obs, 50, -1.0, 0.2, vx.shape)
    # ten steps of 0.2 back from 3.0 land on 0.9999999999999996 in float64, which fails 1 <= x (utils.py:104-107)
    assert np.isclose(back[-1][0], 1.2) and len(back) == 10
    # a line heading into the obstacle stops where the interpolated mask passes 0.5 (utils.py:110-111)
    hit, _ = R.integrate_part(np.array([7.0, 5.5, 4.5]), vx, vy, vz, obs, 50, 1.0, 0.2, vx.shape)
    assert 8.3 < hit[-1][0] < 8.5 and len(hit) == 8                      # 7.0, 7.2 .. 8.4; at 8.6 the mask reads 0.6


def test_filters_and_order():
    vx, vy, vz, obs = tunnel(block=False)
    # uniform flow: no velocity change anywhere, so the default threshold removes every line (utils.py:175-182)
    assert R.generate_streamlines(vx, vy, vz, obs, density=20) == ([], [])
    lines, norms = R.generate_streamlines(vx, vy, vz, obs, density=20, threshold=0.0)
    assert len(lines) > 0 and len(lines) == len(norms)
    assert all(len(l) > 5 for l in lines)
    assert np.allclose(norms, 1.0 / (np.float32(1.0) + 1e-6))           # max speed / (max(v) + 1e-6)
    keys = [(l[0][2], l[0][1]) for l in lines]                          # seeds run z outermost, then y, then x
    assert keys == sorted(keys)
    # without obstacles the reference returns nothing at all (utils.py:134-135)
    assert R.generate_streamlines(vx, vy, vz, np.zeros_like(obs), density=20, threshold=0.0) == ([], [])

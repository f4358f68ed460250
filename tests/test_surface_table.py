"""The marching-cubes case table behind fs_obstacle_surface (csrc/surface.hip builds it from the cube's
geometry at first use).  CPU only: fs_surface_case_table needs neither a handle nor a GPU.  The reference's
mesh comes from scikit-image (GUI/utils.py:17), which is not installed here -- parity unpinned; what is
checked is what makes a mesh usable: no cracks between neighbouring cubes, consistent orientation."""
import ctypes as C
import itertools
from collections import Counter

import numpy as np

from fluid_simulation_amd import _lib


def edge_ends(e):
    """cube-edge id -> its two corner coordinates (include/fluidsim.h: id = 4*axis + 2*ov + ou)."""
    a, ov, ou = e >> 2, (e >> 1) & 1, e & 1
    u, v = [k for k in range(3) if k != a]
    p = [0, 0, 0]
    p[u], p[v] = ou, ov
    q = list(p)
    q[a] = 1
    return tuple(p), tuple(q)


def corner_bit(p):
    return p[0] | (p[1] << 1) | (p[2] << 2)


def table():
    L = _lib.lib()
    buf = (C.c_int * 24)()
    out = []
    for cfg in range(256):
        n = L.fs_surface_case_table(cfg, buf)
        assert 0 <= n <= 8
        out.append([tuple(buf[3 * i:3 * i + 3]) for i in range(n)])
    return out


def midpoint(e):
    p, q = edge_ends(e)
    return tuple((a + b) / 2.0 for a, b in zip(p, q))


def test_every_vertex_sits_on_a_crossing_edge_and_triangles_face_the_fluid():
    T = table()
    assert T[0] == [] and T[255] == []
    assert T[1] and len(T[1]) == 1 and sorted(T[1][0]) == [0, 4, 8]      # one solid corner: the three edges leaving it
    assert max(len(t) for t in T) <= 5
    for cfg, tris in enumerate(T):
        inside = lambda p: (cfg >> corner_bit(p)) & 1      # noqa: E731
        used = set()
        for tri in tris:
            assert len(set(tri)) == 3
            P = np.array([midpoint(e) for e in tri])
            n = np.cross(P[1] - P[0], P[2] - P[0])
            assert np.linalg.norm(n) > 1e-9                               # no degenerate triangle
            for e in tri:
                p, q = edge_ends(e)
                assert inside(p) != inside(q), (cfg, e)
                used.add(e)
        crossing = {e for e in range(12) if inside(edge_ends(e)[0]) != inside(edge_ends(e)[1])}
        assert used == crossing, cfg                                      # every crossing edge carries a vertex that is used
        # orientation: every triangle's normal points from the solid ends of the edges it touches towards the fluid ends
        for t in tris:
            n = np.cross(np.subtract(midpoint(t[1]), midpoint(t[0])), np.subtract(midpoint(t[2]), midpoint(t[0])))
            R = np.zeros(3)
            for e in t:
                p, q = edge_ends(e)
                R += (np.subtract(q, p) if inside(p) else np.subtract(p, q))
            assert float(np.dot(n, R)) > 0, (cfg, t)


def face_boundary(tris, d, side, shift=0.0):
    """Net directed segments of a cube's triangles that lie in its face (axis d, side): pairs cancelling
    inside the cube removed.  Points are physical coordinates (the cube moved by `shift` along d)."""
    seg = Counter()
    for tri in tris:
        P = [midpoint(e) for e in tri]
        for i in range(3):
            a, b = P[i], P[(i + 1) % 3]
            if a[d] == side and b[d] == side:
                a = tuple(x + (shift if k == d else 0.0) for k, x in enumerate(a))
                b = tuple(x + (shift if k == d else 0.0) for k, x in enumerate(b))
                seg[(a, b)] += 1
    net = Counter()
    for (a, b), n in seg.items():
        m = n - seg.get((b, a), 0)
        if m > 0:
            net[(a, b)] = m
    return net


def test_neighbouring_cubes_meet_without_cracks_for_all_value_patterns():
    """Two cubes sharing a face, all 2^12 patterns of their twelve corners, all three axes: the segments the
    first leaves on the shared face are exactly the second's, traversed the other way."""
    T = table()
    for d in range(3):
        u, v = [k for k in range(3) if k != d]
        for bits in itertools.product((0, 1), repeat=12):
            val = {}
            for layer in range(3):
                for j, (pu, pv) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
                    p = [0, 0, 0]
                    p[d], p[u], p[v] = layer, pu, pv
                    val[tuple(p)] = bits[4 * layer + j]
            cfg_a = cfg_b = 0
            for c in range(8):
                p = (c & 1, (c >> 1) & 1, (c >> 2) & 1)
                cfg_a |= val[p] << c
                q = list(p)
                q[d] += 1
                cfg_b |= val[tuple(q)] << c
            fa = face_boundary(T[cfg_a], d, 1.0)
            fb = face_boundary(T[cfg_b], d, 0.0, shift=1.0)
            assert fa == Counter({(b, a): n for (a, b), n in fb.items()}), (d, bits)


def test_single_cube_meshes_close_up_with_their_six_neighbours_empty_or_full():
    """Inside one cube every triangle edge that is not on a cube face is shared by exactly two triangles,
    in opposite directions."""
    T = table()
    for cfg, tris in enumerate(T):
        seg = Counter()
        for tri in tris:
            for i in range(3):
                seg[(tri[i], tri[(i + 1) % 3])] += 1
        for (a, b), n in seg.items():
            assert n == 1, (cfg, a, b)                                    # no directed edge twice
            pa, pb = midpoint(a), midpoint(b)
            on_face = any(pa[k] == pb[k] and pa[k] in (0.0, 1.0) for k in range(3))
            if not on_face:
                assert seg.get((b, a), 0) == 1, (cfg, a, b)

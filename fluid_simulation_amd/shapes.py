"""Synthetic STL obstacles for the BASELINE configs.

The reference has no primitive shapes: every obstacle enters through
loadSTLIntoObstacles() (object_loader.cpp:270-452).  BASELINE.json's "sphere" and "plate"
are therefore small STL files authored here and voxelised through that same path.

The loader rotates about, and measures its bounding radius from, the mesh's own origin
(object_loader.cpp:288-296, :328-334), so both shapes are centred on the origin and are
positioned in the tunnel with the loader's translate arguments.
"""
import math
import struct

import numpy as np


def sphere_triangles(radius=1.0, n_lon=48, n_lat=24):
    """UV sphere about the origin: 2*n_lon*(n_lat-1) triangles (2208 for 48x24)."""
    tris = []

    def pt(i, j):
        th = math.pi * j / n_lat
        ph = 2.0 * math.pi * (i % n_lon) / n_lon
        return (radius * math.sin(th) * math.cos(ph), radius * math.sin(th) * math.sin(ph), radius * math.cos(th))

    for j in range(n_lat):
        for i in range(n_lon):
            a, b, c, d = pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)
            if j != 0:
                tris.append((a, b, c))
            if j != n_lat - 1:
                tris.append((a, c, d))
    return np.asarray(tris, dtype=np.float32)


def box_triangles(sx=0.1, sy=1.0, sz=1.0):
    """Axis-aligned box about the origin with half-extents (sx,sy,sz): 12 triangles.
    A thin sx makes the "plate" of BASELINE configs 3 and 4."""
    v = np.array([[x, y, z] for x in (-sx, sx) for y in (-sy, sy) for z in (-sz, sz)], dtype=np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, c, d in quads:
        tris.append((v[a], v[b], v[c]))
        tris.append((v[a], v[c], v[d]))
    return np.asarray(tris, dtype=np.float32)


def write_binary_stl(path, tris, header=b"fluid_simulation_amd synthetic mesh"):
    """80-byte header (must not start with 'solid', object_loader.cpp:105-107), uint32
    count, then 50-byte facets: normal, 3 vertices, attribute word."""
    tris = np.asarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    assert not header.lstrip().startswith(b"solid")
    with open(path, "wb") as f:
        f.write(header.ljust(80, b"\0")[:80])
        f.write(struct.pack("<I", len(tris)))
        for t in tris:
            n = np.cross(t[1] - t[0], t[2] - t[0])
            ln = float(np.linalg.norm(n))
            n = n / ln if ln > 0 else n
            f.write(struct.pack("<12fH", *n.astype(np.float32), *t.reshape(-1), 0))
    return path


def write_ascii_stl(path, tris, name="mesh"):
    tris = np.asarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    with open(path, "w") as f:
        f.write("solid %s\n" % name)
        for t in tris:
            f.write("  facet normal 0 0 0\n    outer loop\n")
            for v in t:
                f.write("      vertex %.9g %.9g %.9g\n" % tuple(float(c) for c in v))
            f.write("    endloop\n  endfacet\n")
        f.write("endsolid %s\n" % name)
    return path

// surface.h -- obstacle surface extraction of the viewer on the device.
//
// The reference's viewer turns the last dumped `obs` frame into a triangle mesh on the CPU,
// per displayed frame: generate_obstacle_mesh, GUI/utils.py:10-38 (scikit-image's
// measure.marching_cubes at level 0.5), called from GUI/main_window.py:204-218.  This is
// SURVEY.md section 8(f) rank 3, second half: the same kind of mesh -- the iso-surface of `obs`
// at 0.5, vertices on the grid edges where `obs` crosses 0.5, indexed triangles -- from the field
// while it is on the device.
//
// PARITY UNPINNED: scikit-image is not installed here and the reference holds no mesh fixtures, so
// vertex order, triangle order and the resolution of ambiguous cubes cannot be compared with the
// reference's output.  What is guaranteed (tests/test_surface.py): the mesh is closed and
// consistently oriented (every edge is used by exactly two triangles, in opposite directions,
// normals pointing from solid to fluid), every vertex lies on a grid edge at the 0.5 crossing,
// and the result is deterministic.
#pragma once
#include <hip/hip_runtime_api.h>

#include "kernels.h"

namespace fs {

// The 256-entry triangle table, built once on the host (surface.hip): for the cube whose corner i
// = (x + (i & 1), y + ((i >> 1) & 1), z + ((i >> 2) & 1)) is inside (obs > 0.5) iff bit i of `config`
// is set, `edges` receives 3 cube-edge ids per triangle and the count is returned (at most 8).
// Edge id = 4 * axis + 2 * (offset on the higher other axis) + (offset on the lower other axis):
// axis 0 = x edges, other axes (y, z); axis 1 = y edges, others (x, z); axis 2 = z edges, others (x, y).
constexpr int SURF_MAX_TRIS = 8;
int surface_case(int config, int* edges /* 3 * SURF_MAX_TRIS */);

struct SurfaceResult {
    long nverts = 0, ntris = 0;
    float* d_verts = nullptr;   // 3 * nverts: (x, y, z) in padded index coordinates (the viewer's, x first)
    int* d_tris = nullptr;      // 3 * ntris vertex indices
};

// Iso-surface of obs at 0.5 over the padded box (0..W+1) x (0..H+1) x (0..D+1).  Synchronises the
// stream (the host needs the counts to size the output).  Returns 0, or a negative FS_* code with
// `err` set.  Free the result with surface_free.
template <class T>
int extract_surface(hipStream_t st, const GridDesc& g, const T* obs, SurfaceResult* out, const char** err);
void surface_free(SurfaceResult* r);

}  // namespace fs

// streamlines.h -- streamline integration of the viewer on the device.
//
// The reference computes the streamlines it draws on the CPU, in Python, per displayed frame:
// GUI/utils.py:40-213 (generate_streamlines and helpers), called from GUI/main_window.py:227-233
// with the last dumped frame transposed to (x, y, z).  This is SURVEY.md section 8(f) rank 3: the
// same polylines from the fields while they are still on the device.
#pragma once
#include <hip/hip_runtime_api.h>

#include "kernels.h"

namespace fs {

struct StreamParams {
    int nx, ny, nz;          // seed counts: STREAMLINE_DENSITY, DENSITY // 2, DENSITY // 2   (utils.py:136-138)
    int half;                // steps per direction: max_length // 2                          (utils.py:157-164)
    double step_size;        // config.INTEGRATION_STEP_SIZE
    double lo[3], hi[3];     // obstacle bounding box widened by STREAMLINE_PROXIMITY / 10      (utils.py:127-133)
    double clip_hi[3];       // grid.shape[k] - 1.001                                         (utils.py:43-45)
    double bound_hi[3];      // config.width - 1 etc.                                         (utils.py:104-106)
};

// bounding box (inclusive, padded coordinates) of the cells with obs > 0.5; box[0..2] = min x,y,z,
// box[3..5] = max; min > max when there is none.  `box` is device memory, 6 ints.
template <class T>
void launch_obs_bbox(hipStream_t st, const GridDesc& g, const T* obs, int* box);

// One thread per candidate seed.  Seed id = (iz * ny + iy) * nx + ix, the order of the loops at
// utils.py:141-143; `cand` lists, in that order, the ids that pass the bounding-box cull (done by
// the caller: it is arithmetic on the seed coordinates only).  seeds: nx + ny + nz doubles (x seeds,
// then y, then z).  For candidate s the kernel writes
//   count[2*s], count[2*s+1]   points of the backward / forward part (0, 0 if the seed sits in an obstacle)
//   pts, vel [((2*s + part) * (half+1) + k) * 3 + c]   point k of that part and the velocity stored with it
template <class T>
void launch_streamlines(hipStream_t st, const GridDesc& g, const T* vx, const T* vy, const T* vz, const T* obs,
                        const StreamParams& p, const double* seeds, const int* cand, int ncand, int* count, double* pts,
                        double* vel);

}  // namespace fs

// ORACLE - test infrastructure only (tests/, __graft_entry__.smoke(), bench cpu_baseline); never linked into the product.
//
// CPU restatement of the offline dense map merge, /root/reference/utils/python/makeMergedMap.py:
//   :48-56    poses: one line of 12 numbers per keyframe = the top 3x4 of an SE(3) matrix (f64)
//   :95-96    local points of the keyframe (PCD f32, held as f64 by open3d)
//   :105      scan_pcd.transform(pose): open3d computes T * [x y z 1]^T in f64 and divides by the 4th component (= 1)
//   :109-116  near-range removal: keep points whose LOCAL range sqrt(x^2 + y^2 + z^2) (f64) is > 2 m
//   :129-133  concatenate xyz (f64) and intensity in keyframe order
//   :145-147  saved as f32 x, y, z, intensity (pypcdMyUtils.py:26 astype(float32))
// The reference needs open3d and pypcd, neither of which exists in this container: it cannot be run, and it holds no expected
// output for this step.  PARITY UNPINNED for the f64 summation order of the 4x4 product (restated here as Eigen's column-major
// gemv: ((T0*x + T1*y) + T2*z) + T3); after the cast to f32 a different order would move a coordinate by at most one f32 ulp.
#include "oracle.h"
#include <cmath>

extern "C" int orc_mapmerge_frame(const float* xyzi, int n, const double* pose12, double near_thres, float* out_xyzi) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const double x = xyzi[4 * i], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2];
        const double range = std::sqrt((x * x + y * y) + z * z);  // numpy: add.reduce over the three squares
        if (!(range > near_thres)) continue;
        for (int r = 0; r < 3; ++r) {
            const double* T = pose12 + 4 * r;
            const double g = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
            out_xyzi[4 * m + r] = static_cast<float>(g);
        }
        out_xyzi[4 * m + 3] = xyzi[4 * i + 3];
        ++m;
    }
    return m;
}

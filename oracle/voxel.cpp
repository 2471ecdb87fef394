// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp).
//
// Restatement of pcl::VoxelGrid<pcl::PointXYZI>::applyFilter with the reference's settings
// (default ctor: downsample_all_data=true, min_points_per_voxel=0, no filter field).
// Third-party code, NOT under /root/reference: PCL 1.8.0 is the pin (docker/Dockerfile:4).
// Restated from the published algorithm (pcl/filters/impl/voxel_grid.hpp, PCL 1.8):
//   min/max over the cloud -> overflow guard -> min_b/div_b -> per point
//   ijk = floor(p*inv_leaf) - min_b, idx = i + j*dx + k*dx*dy -> sort by idx ->
//   one output per distinct idx, ascending idx, = f32 mean of x,y,z,intensity accumulated in
//   sorted order (CentroidPoint<PointXYZI>).
// Call sites in the reference: scanRegistration.cpp:414-418 (0.2), laserMapping.cpp:543-551,
// :793-801, laserPosegraphOptimization.cpp:629-631 (0.4).
//
// PARITY UNPINNED by any reference test.  The order of points inside one voxel is decided by an
// unstable std::sort in PCL, i.e. it is libstdc++-version specific in the reference itself.
// order_mode 0 restates that literally (std::sort on idx only); order_mode 1 pins the order to
// (idx, arrival index) — the mode the HIP path is compared against bit-for-bit.
#include "orc_common.hpp"
#include "oracle.h"

namespace orc {

struct IdxPair {
    unsigned int idx;
    unsigned int cloud_point_index;
    bool operator<(const IdxPair& p) const { return idx < p.idx; }
};

// returns number of output points; out must hold n points. *guard_hit=1 when the INT_MAX guard
// fired (output = input unchanged, as PCL does).
int voxel_grid(const P4* in, int n, float leaf, int order_mode, P4* out, int* guard_hit) {
    if (guard_hit) *guard_hit = 0;
    if (n <= 0) return 0;
    // setLeafSize(float,float,float): inverse_leaf_size_ = 1 / leaf_size_ (Array4f)
    const float inv = 1.0f / leaf;
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {-mn[0], -mn[1], -mn[2]};
    for (int i = 0; i < n; ++i) {  // getMinMax3D (dense cloud)
        const float p[3] = {in[i].x, in[i].y, in[i].z};
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], p[a]);
            mx[a] = std::max(mx[a], p[a]);
        }
    }
    int64_t d[3];
    for (int a = 0; a < 3; ++a) d[a] = static_cast<int64_t>((mx[a] - mn[a]) * inv) + 1;
    if (d[0] * d[1] * d[2] > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) {
        if (guard_hit) *guard_hit = 1;
        if (out != in) std::memcpy(out, in, sizeof(P4) * n);
        return n;
    }
    int min_b[3], max_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = static_cast<int>(std::floor(mn[a] * inv));
        max_b[a] = static_cast<int>(std::floor(mx[a] * inv));
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    std::vector<IdxPair> iv;
    iv.reserve(n);
    for (int i = 0; i < n; ++i) {
        int ijk0 = static_cast<int>(std::floor(in[i].x * inv) - static_cast<float>(min_b[0]));
        int ijk1 = static_cast<int>(std::floor(in[i].y * inv) - static_cast<float>(min_b[1]));
        int ijk2 = static_cast<int>(std::floor(in[i].z * inv) - static_cast<float>(min_b[2]));
        int idx = ijk0 + ijk1 * mul1 + ijk2 * mul2;
        iv.push_back({static_cast<unsigned int>(idx), static_cast<unsigned int>(i)});
    }
    if (order_mode == 0)
        std::sort(iv.begin(), iv.end());
    else
        std::stable_sort(iv.begin(), iv.end());
    std::vector<P4> res;
    res.reserve(n);
    size_t index = 0;
    while (index < iv.size()) {
        size_t i = index + 1;
        while (i < iv.size() && iv[i].idx == iv[index].idx) ++i;
        // CentroidPoint<PointXYZI>: f32 sums in sorted order, divided by float(n)
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (size_t li = index; li < i; ++li) {
            const P4& p = in[iv[li].cloud_point_index];
            sx += p.x;
            sy += p.y;
            sz += p.z;
            si += p.i;
        }
        const float cnt = static_cast<float>(i - index);
        res.push_back({sx / cnt, sy / cnt, sz / cnt, si / cnt});
        index = i;
    }
    std::memcpy(out, res.data(), sizeof(P4) * res.size());
    return static_cast<int>(res.size());
}

}  // namespace orc

extern "C" int orc_voxel_grid(const float* xyzi, int n, float leaf, int order_mode, float* out_xyzi, int* n_out,
                              int* guard_hit) {
    std::vector<orc::P4> tmp(n > 0 ? n : 1);
    int m = orc::voxel_grid(reinterpret_cast<const orc::P4*>(xyzi), n, leaf, order_mode, tmp.data(), guard_hit);
    std::memcpy(out_xyzi, tmp.data(), sizeof(orc::P4) * m);
    *n_out = m;
    return 0;
}

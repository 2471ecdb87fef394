// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp).
//
// Stage A: literal CPU restatement of laserCloudHandler, /root/reference/src/scanRegistration.cpp:116-421.
// Every block cites the lines it follows.  f32/f64 types follow the C++ usual arithmetic conversions of
// the reference expressions exactly.
//
// Toolchain-dependent behaviour: `atan`/`sqrt` at :168 are called
// unqualified on floats.  Under the reference's only pinned toolchain (docker/Dockerfile:1,
// ros:kinetic = GCC 5.4, whose <cmath> leaves only ::atan(double) in the global namespace) they promote
// to double; under GCC >= 6 with <math.h> transitively included they resolve to the float overloads.
// float_math=0 restates the pinned behaviour (default), float_math=1 the other.
// `atan2` is `using std::atan2` (:56) on floats => atan2f in both.
// cr_libm=1 replaces atanf/atan2f by the correctly-rounded value (double libm rounded to float);
// glibc's own atanf/atan2f are not correctly rounded, so cr_libm=0 (literal libm) may differ from it by 1 ulp.
#include "orc_common.hpp"
#include "oracle.h"

namespace orc {

int voxel_grid(const P4* in, int n, float leaf, int order_mode, P4* out, int* guard_hit);

namespace {

struct Libm {
    int cr;
    float atan2_(float y, float x) const {
        return cr ? static_cast<float>(std::atan2(static_cast<double>(y), static_cast<double>(x))) : atan2f(y, x);
    }
    float atan_(float v) const { return cr ? static_cast<float>(std::atan(static_cast<double>(v))) : atanf(v); }
};

}  // namespace

int features_run(const OrcFeatureConfig& cfg, const float* xyz, int n, int stride, OrcFeatureOut& o) {
    const int N_SCANS = cfg.n_scans;
    if (N_SCANS != 16 && N_SCANS != 32 && N_SCANS != 64) return ORC_E_SCANLINE;  // :486-490
    const Libm lm{cfg.cr_libm};
    const double scanPeriod = 0.1;  // :62

    // :134-139  fromROSMsg -> PointXYZ; removeNaNFromPointCloud; removeClosedPointCloud(thres=float(MINIMUM_RANGE))
    struct P3 {
        float x, y, z;
        int src;
    };
    std::vector<P3> in;
    in.reserve(n);
    const float thres = static_cast<float>(cfg.minimum_range);
    for (int i = 0; i < n; ++i) {
        const float x = xyz[(size_t)i * stride + 0], y = xyz[(size_t)i * stride + 1], z = xyz[(size_t)i * stride + 2];
        if (cfg.check_finite && !(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) continue;  // :138
        if (x * x + y * y + z * z < thres * thres) continue;                                              // :101
        in.push_back({x, y, z, i});
    }
    int cloudSize = static_cast<int>(in.size());
    o.n_kept = 0;
    o.n_sharp = o.n_less_sharp = o.n_flat = o.n_less_flat = 0;
    o.n_ties = 0;
    for (int i = 0; i < N_SCANS; ++i) o.ring_start[i] = o.ring_end[i] = 0;
    if (cloudSize == 0) return ORC_E_EMPTY;  // reference would read points[0] of an empty cloud (:143)

    // :143-155
    float startOri = -lm.atan2_(in[0].y, in[0].x);
    float endOri = static_cast<float>(-lm.atan2_(in[cloudSize - 1].y, in[cloudSize - 1].x) + 2 * M_PI);
    if (endOri - startOri > 3 * M_PI)
        endOri = static_cast<float>(endOri - 2 * M_PI);
    else if (endOri - startOri < M_PI)
        endOri = static_cast<float>(endOri + 2 * M_PI);

    // :158-254
    bool halfPassed = false;
    int count = cloudSize;
    std::vector<std::vector<P4>> laserCloudScans(N_SCANS);
    std::vector<std::vector<int>> scanSrc(N_SCANS);
    for (int i = 0; i < cloudSize; i++) {
        const float px = in[i].x, py = in[i].y, pz = in[i].z;
        float angle;
        if (!cfg.float_math) {
            // ::sqrt(double) of the float sum, double divide, ::atan(double), * 180 / M_PI, -> float   (:168)
            angle = static_cast<float>(std::atan(pz / std::sqrt(static_cast<double>(px * px + py * py))) * 180 / M_PI);
        } else {
            const float a = lm.atan_(pz / sqrtf(px * px + py * py));
            angle = static_cast<float>((a * 180) / M_PI);
        }
        int scanID = 0;
        if (cfg.lidar_type == ORC_VLP16 && N_SCANS == 16) {
            scanID = int((angle + 15) / 2 + 0.5);  // :173
            if (scanID > (N_SCANS - 1) || scanID < 0) {
                count--;
                continue;
            }
        } else if (cfg.lidar_type == ORC_HDL32 && N_SCANS == 32) {
            scanID = int((angle + 92.0 / 3.0) * 3.0 / 4.0);  // :182
            if (scanID > (N_SCANS - 1) || scanID < 0) {
                count--;
                continue;
            }
        } else if (cfg.lidar_type == ORC_HDL64 && N_SCANS == 64) {
            if (angle >= -8.83)  // :192-195
                scanID = int((2 - angle) * 3.0 + 0.5);
            else
                scanID = N_SCANS / 2 + int((-8.83 - angle) * 2.0 + 0.5);
            if (angle > 2 || angle < -24.33 || scanID > 50 || scanID < 0) {  // :198
                count--;
                continue;
            }
        } else if (cfg.lidar_type == ORC_OS1_64 && N_SCANS == 64) {
            scanID = int((angle + 22.5) / 2 + 0.5);  // :207
            if (scanID > (N_SCANS - 1) || scanID < 0) {
                count--;
                continue;
            }
        } else {
            return ORC_E_LIDARTYPE;  // ROS_BREAK() :217
        }

        float ori = -lm.atan2_(py, px);  // :221
        if (!halfPassed) {
            if (ori < startOri - M_PI / 2)
                ori = static_cast<float>(ori + 2 * M_PI);
            else if (ori > startOri + M_PI * 3 / 2)
                ori = static_cast<float>(ori - 2 * M_PI);
            if (ori - startOri > M_PI) halfPassed = true;
        } else {
            ori = static_cast<float>(ori + 2 * M_PI);
            if (ori < endOri - M_PI * 3 / 2)
                ori = static_cast<float>(ori + 2 * M_PI);
            else if (ori > endOri + M_PI / 2)
                ori = static_cast<float>(ori - 2 * M_PI);
        }
        float relTime = (ori - startOri) / (endOri - startOri);  // :251
        P4 point{px, py, pz, static_cast<float>(scanID + scanPeriod * relTime)};  // :252
        laserCloudScans[scanID].push_back(point);
        scanSrc[scanID].push_back(in[i].src);
    }
    cloudSize = count;

    // :259-265
    std::vector<P4> laserCloud;
    laserCloud.reserve(cloudSize);
    std::vector<int> scanStartInd(N_SCANS, 0), scanEndInd(N_SCANS, 0);
    int pos = 0;
    for (int i = 0; i < N_SCANS; i++) {
        scanStartInd[i] = static_cast<int>(laserCloud.size()) + 5;
        laserCloud.insert(laserCloud.end(), laserCloudScans[i].begin(), laserCloudScans[i].end());
        if (o.src_index)
            for (int s : scanSrc[i]) o.src_index[pos++] = s;
        scanEndInd[i] = static_cast<int>(laserCloud.size()) - 6;
        o.ring_start[i] = scanStartInd[i];
        o.ring_end[i] = scanEndInd[i];
    }
    o.n_kept = cloudSize;
    std::memcpy(o.cloud, laserCloud.data(), sizeof(P4) * cloudSize);

    // :269-279
    std::vector<float> cloudCurvature(cloudSize, 0.f);
    std::vector<int> cloudSortInd(cloudSize, 0), cloudNeighborPicked(cloudSize, 0), cloudLabel(cloudSize, 0);
    const P4* L = laserCloud.data();
    for (int i = 5; i < cloudSize - 5; i++) {
        float diffX = L[i - 5].x + L[i - 4].x + L[i - 3].x + L[i - 2].x + L[i - 1].x - 10 * L[i].x + L[i + 1].x + L[i + 2].x + L[i + 3].x + L[i + 4].x + L[i + 5].x;
        float diffY = L[i - 5].y + L[i - 4].y + L[i - 3].y + L[i - 2].y + L[i - 1].y - 10 * L[i].y + L[i + 1].y + L[i + 2].y + L[i + 3].y + L[i + 4].y + L[i + 5].y;
        float diffZ = L[i - 5].z + L[i - 4].z + L[i - 3].z + L[i - 2].z + L[i - 1].z - 10 * L[i].z + L[i + 1].z + L[i + 2].z + L[i + 3].z + L[i + 4].z + L[i + 5].z;
        cloudCurvature[i] = diffX * diffX + diffY * diffY + diffZ * diffZ;
        cloudSortInd[i] = i;
    }

    // :284-421
    std::vector<P4> surfPointsLessFlat;
    std::vector<P4> lessFlatScan, lessFlatScanDS;
    const float* C = cloudCurvature.data();
    auto suppress = [&](int ind) {  // :332-355 == :378-401
        for (int l = 1; l <= 5; l++) {
            float diffX = L[ind + l].x - L[ind + l - 1].x;
            float diffY = L[ind + l].y - L[ind + l - 1].y;
            float diffZ = L[ind + l].z - L[ind + l - 1].z;
            if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.05) break;
            cloudNeighborPicked[ind + l] = 1;
        }
        for (int l = -1; l >= -5; l--) {
            float diffX = L[ind + l].x - L[ind + l + 1].x;
            float diffY = L[ind + l].y - L[ind + l + 1].y;
            float diffZ = L[ind + l].z - L[ind + l + 1].z;
            if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.05) break;
            cloudNeighborPicked[ind + l] = 1;
        }
    };
    for (int i = 0; i < N_SCANS; i++) {
        if (scanEndInd[i] - scanStartInd[i] < 6) continue;  // :292
        lessFlatScan.clear();
        for (int j = 0; j < 6; j++) {
            int sp = scanStartInd[i] + (scanEndInd[i] - scanStartInd[i]) * j / 6;
            int ep = scanStartInd[i] + (scanEndInd[i] - scanStartInd[i]) * (j + 1) / 6 - 1;
            if (cfg.sort_mode == 0)
                std::sort(cloudSortInd.begin() + sp, cloudSortInd.begin() + ep + 1, [&](int a, int b) { return C[a] < C[b]; });  // :73, :301
            else
                std::sort(cloudSortInd.begin() + sp, cloudSortInd.begin() + ep + 1,
                          [&](int a, int b) { return C[a] < C[b] || (C[a] == C[b] && a < b); });
            for (int k = sp; k < ep; ++k)
                if (C[cloudSortInd[k]] == C[cloudSortInd[k + 1]]) o.n_ties++;

            int largestPickedNum = 0;
            for (int k = ep; k >= sp; k--) {  // :305-357
                int ind = cloudSortInd[k];
                if (cloudNeighborPicked[ind] == 0 && C[ind] > 0.1) {
                    largestPickedNum++;
                    if (largestPickedNum <= 2) {
                        cloudLabel[ind] = 2;
                        o.sharp[o.n_sharp++] = ind;
                        o.less_sharp[o.n_less_sharp++] = ind;
                    } else if (largestPickedNum <= 20) {
                        cloudLabel[ind] = 1;
                        o.less_sharp[o.n_less_sharp++] = ind;
                    } else {
                        break;
                    }
                    cloudNeighborPicked[ind] = 1;
                    suppress(ind);
                }
            }
            int smallestPickedNum = 0;
            for (int k = sp; k <= ep; k++) {  // :360-403
                int ind = cloudSortInd[k];
                if (cloudNeighborPicked[ind] == 0 && C[ind] < 0.1) {
                    cloudLabel[ind] = -1;
                    o.flat[o.n_flat++] = ind;
                    smallestPickedNum++;
                    if (smallestPickedNum >= 4) break;  // before marking / suppressing (:372-375)
                    cloudNeighborPicked[ind] = 1;
                    suppress(ind);
                }
            }
            for (int k = sp; k <= ep; k++)  // :405-411
                if (cloudLabel[k] <= 0) lessFlatScan.push_back(L[k]);
        }
        // :414-420  VoxelGrid leaf 0.2 per ring, appended ring by ring
        lessFlatScanDS.resize(lessFlatScan.size() + 1);
        int m = voxel_grid(lessFlatScan.data(), static_cast<int>(lessFlatScan.size()), 0.2f, cfg.voxel_order, lessFlatScanDS.data(), nullptr);
        surfPointsLessFlat.insert(surfPointsLessFlat.end(), lessFlatScanDS.begin(), lessFlatScanDS.begin() + m);
    }
    o.n_less_flat = static_cast<int>(surfPointsLessFlat.size());
    std::memcpy(o.less_flat, surfPointsLessFlat.data(), sizeof(P4) * surfPointsLessFlat.size());
    if (o.curvature) std::memcpy(o.curvature, cloudCurvature.data(), sizeof(float) * cloudSize);
    if (o.label) std::memcpy(o.label, cloudLabel.data(), sizeof(int) * cloudSize);
    return 0;
}

}  // namespace orc

extern "C" int orc_features_run(const OrcFeatureConfig* cfg, const float* xyz, int n, int stride_floats, OrcFeatureOut* out) {
    return orc::features_run(*cfg, xyz, n, stride_floats, *out);
}

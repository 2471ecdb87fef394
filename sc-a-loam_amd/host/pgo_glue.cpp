// The three places where alaserPGO (laserPosegraphOptimization.cpp) touches the hot path, as drop-in functions (source only, see
// README.md).  Everything else of that node - queues, GTSAM factors and iSAM2 (:433-444, :646-690, :791-808), savers, rviz - stays
// the reference's own code and keeps calling these at the cited lines.
#include <cmath>
#include <vector>
#include <Eigen/Geometry>
#include "scal_common.hpp"

namespace scal_pgo {

struct Pose6D { double x, y, z, roll, pitch, yaw; };  // include/aloam_velodyne/common.h:43-62

// ---- keyframe gate (:598-617 with diffTransformation :325-336); the accumulators start large so the first pose is a keyframe (:67-68)
struct KeyframeGate {
    double meter_gap = 2.0, rad_gap = 10.0 * M_PI / 180.0;  // keyframe_meter_gap / keyframe_deg_gap (:874-876)
    double tr = 1000000.0, rot = 1000000.0;
    Pose6D curr{0, 0, 0, 0, 0, 0};
    static Eigen::Affine3f affine(const Pose6D& p) {
        return Eigen::Translation3f(p.x, p.y, p.z) * Eigen::AngleAxisf(p.yaw, Eigen::Vector3f::UnitZ()) * Eigen::AngleAxisf(p.pitch, Eigen::Vector3f::UnitY()) *
               Eigen::AngleAxisf(p.roll, Eigen::Vector3f::UnitX());  // pcl::getTransformation
    }
    bool operator()(const Pose6D& pose) {
        const Pose6D prev = curr;
        curr = pose;
        const Eigen::Matrix4f d = affine(prev).matrix().inverse() * affine(curr).matrix();
        const double dr = std::fabs(std::atan2(d(2, 1), d(2, 2))), dp = std::fabs(std::asin(-d(2, 0))), dy = std::fabs(std::atan2(d(1, 0), d(0, 0)));
        tr += std::sqrt(double(d(0, 3)) * d(0, 3) + double(d(1, 3)) * d(1, 3) + double(d(2, 3)) * d(2, 3));
        rot += dr + dp + dy;
        if (tr > meter_gap || rot > rad_gap) {
            tr = rot = 0.0;
            return true;
        }
        return false;
    }
};

// ---- :629-639: VoxelGrid 0.4 m + scManager.makeAndSaveScancontextAndKeys(*thisKeyFrameDS), under mKF in the reference; the
// context serialises insert and detect itself.  xyzi = the keyframe cloud (/velodyne_cloud_registered_local), packed.
inline void insert_keyframe(scal_voxel_t* vox, scal_sc_t* sc, const float* xyzi, int n, std::vector<float>& ds) {
    ds.resize(4 * static_cast<size_t>(n));
    int m = 0;
    SCAL_CHECK(scal_voxel_downsample(vox, xyzi, n, 0.4f, ds.data(), &m));
    ds.resize(4 * static_cast<size_t>(m));
    SCAL_CHECK(scal_sc_insert_cloud(sc, ds.data(), m));
}

// ---- performSCLoopClosure (:713-730): returns the loop keyframe id or -1; the caller pushes (id, latest) into scLoopICPBuf
inline int detect_loop(scal_sc_t* sc, int n_keyframes, float* yaw_diff_rad) {
    if (n_keyframes < 30) return -1;  // :715 (NUM_EXCLUDE_RECENT)
    scal_sc_result r;
    SCAL_CHECK(scal_sc_detect(sc, &r));
    *yaw_diff_rad = r.yaw_rad;
    return r.loop_id;
}

// ---- doICPVirtualRelative (:497-548): cureKeyframeCloud / targetKeyframeCloud are built as at :504-507 (the caller's
// loopFindNearKeyframesCloud, or scal_mapmerge_* + scal_voxel_*); returns false when the reference would reject the loop (:531-537)
inline bool verify_loop(scal_icp_t* icp, const float* src_xyzi, int n_src, const float* tgt_xyzi, int n_tgt, Eigen::Matrix4f* T) {
    scal_icp_result r;
    SCAL_CHECK(scal_icp_align(icp, src_xyzi, n_src, tgt_xyzi, n_tgt, &r));
    if (!r.converged || r.fitness > 0.3) return false;  // loopFitnessScoreThreshold (:530)
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) (*T)(i, j) = static_cast<float>(r.T[4 * i + j]);
    return true;  // the caller turns T into the gtsam::Pose3 pair of :539-547
}

}  // namespace scal_pgo

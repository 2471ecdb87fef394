// ORACLE - test infrastructure only (tests/, __graft_entry__.smoke(), bench cpu_baseline); never linked into the product.
//
// CPU restatement of the loop-closure verification ICP, laserPosegraphOptimization.cpp:497-548 (doICPVirtualRelative):
// pcl::IterativeClosestPoint<PointXYZI, PointXYZI> with setMaxCorrespondenceDistance(150), setMaximumIterations(100),
// setTransformationEpsilon(1e-6), setEuclideanFitnessEpsilon(1e-6), setRANSACIterations(0), then hasConverged() and
// getFitnessScore() <= 0.3 (:524-535).  PCL 1.8 is a third-party dependency that is absent here: its
// published algorithm is restated -
//   icp.hpp computeTransformation: per iteration nearest target point of every (already transformed) source point (kd-tree,
//     squared f32 distance, kept if <= max_dist^2), rigid transform by SVD (TransformationEstimationSVD = Eigen::umeyama without
//     scale, here as Horn's quaternion form of the same least-squares problem, evaluated in f64 and stored as the Matrix4f PCL
//     keeps), source cloud updated in place with that increment, final = increment * final, convergence test;
//   default_convergence_criteria.hpp hasConverged (1.8): iteration cap -> converged; rotation/translation of the increment below
//     (1 - eps, eps) -> converged; |mse - prev| < 1e-12 or relative change < eps -> converged (mse = mean of the squared
//     correspondence distances); max_iterations_similar_transforms_ = 0;
//   registration.hpp getFitnessScore: mean squared distance from each source point, moved by the final transform, to its nearest
//     target point.
// PARITY UNPINNED: nothing of the reference pins PCL's arithmetic (f32 JacobiSVD inside umeyama, kd-tree tie order).
#include "oracle.h"
#include "kdtree.hpp"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// largest eigenpair of a symmetric 4x4 by cyclic Jacobi
void eig4_max(double a[4][4], double q[4]) {
    double v[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) (i == j ? diag : off) += a[i][j] * a[i][j];
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                if (a[p][r] == 0.0) continue;
                const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double akp = a[k][p], akr = a[k][r];
                    a[k][p] = c * akp - s * akr, a[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = a[p][k], ark = a[r][k];
                    a[p][k] = c * apk - s * ark, a[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = v[k][p], vkr = v[k][r];
                    v[k][p] = c * vkp - s * vkr, v[k][r] = s * vkp + c * vkr;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (a[i][i] > a[best][best]) best = i;
    double nrm = 0;
    for (int k = 0; k < 4; ++k) nrm += v[k][best] * v[k][best];
    nrm = std::sqrt(nrm);
    for (int k = 0; k < 4; ++k) q[k] = v[k][best] / nrm;  // (w, x, y, z)
}

}  // namespace

// sums[0] = n, [1..3] = sum p (source), [4..6] = sum q (target), [7..15] = sum q_r * p_c (row-major) -> rigid increment as the
// Matrix4f PCL stores (values rounded to f32), row-major in T16
extern "C" void orc_icp_transform_from_sums(const double* sums, double* T16) {
    const double n = sums[0];
    double mp[3], mq[3], S[3][3];
    for (int k = 0; k < 3; ++k) mp[k] = sums[1 + k] / n, mq[k] = sums[4 + k] / n;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S[c][r] = sums[7 + 3 * r + c] / n - mq[r] * mp[c];  // S[a][b] = cov(p_a, q_b)
    // Horn 1987: N built from S = sum p q^T; the unit quaternion of the largest eigenvalue rotates p onto q
    double N[4][4] = {{S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
                      {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
                      {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
                      {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double q[4];
    eig4_max(N, q);
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                      {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                      {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int r = 0; r < 3; ++r) {
        double t = mq[r];
        for (int c = 0; c < 3; ++c) {
            T16[4 * r + c] = static_cast<double>(static_cast<float>(R[r][c]));
            t -= R[r][c] * mp[c];
        }
        T16[4 * r + 3] = static_cast<double>(static_cast<float>(t));
    }
    T16[12] = T16[13] = T16[14] = 0.0, T16[15] = 1.0;
}

static inline void apply_f32(const float* T, float& x, float& y, float& z) {
    const float nx = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    const float ny = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    const float nz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
    x = nx, y = ny, z = nz;
}

// returns hasConverged(); state: 1 iterations, 2 transform, 3 abs mse, 4 rel mse, 5 too few correspondences
extern "C" int orc_icp_align(const float* src_xyzi, int n_src, const float* tgt_xyzi, int n_tgt, double max_corr, int max_iter, double trans_eps,
                             double fit_eps, double* T16_out, double* fitness, int* iterations, int* state) {
    std::vector<orc::P4> tgt(n_tgt), cur(n_src);
    std::memcpy(tgt.data(), tgt_xyzi, sizeof(float) * 4 * n_tgt);
    std::memcpy(cur.data(), src_xyzi, sizeof(float) * 4 * n_src);
    orc::KdTree tree;
    tree.build(tgt.data(), n_tgt);
    float F[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const float max2 = static_cast<float>(max_corr * max_corr);
    int it = 0, st = 0;
    bool converged = false;
    double prev_mse = std::numeric_limits<double>::max();
    while (!converged) {
        double sums[16] = {0};
        double mse_sum = 0;
        for (int i = 0; i < n_src; ++i) {
            const float q[3] = {cur[i].x, cur[i].y, cur[i].z};
            int idx;
            float d;
            if (tree.knn(q, 1, &idx, &d) < 1 || d > max2) continue;
            const orc::P4& t = tgt[idx];
            sums[0] += 1;
            sums[1] += q[0], sums[2] += q[1], sums[3] += q[2];
            sums[4] += t.x, sums[5] += t.y, sums[6] += t.z;
            const double tq[3] = {t.x, t.y, t.z};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) sums[7 + 3 * r + c] += tq[r] * q[c];
            mse_sum += d;
        }
        if (sums[0] < 3) {
            st = 5;
            break;
        }
        double T16[16];
        orc_icp_transform_from_sums(sums, T16);
        float T[16];
        for (int k = 0; k < 16; ++k) T[k] = static_cast<float>(T16[k]);
        for (int i = 0; i < n_src; ++i) apply_f32(T, cur[i].x, cur[i].y, cur[i].z);
        float G[16];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) G[4 * r + c] = ((T[4 * r] * F[c] + T[4 * r + 1] * F[4 + c]) + T[4 * r + 2] * F[8 + c]) + T[4 * r + 3] * F[12 + c];
        std::memcpy(F, G, sizeof F);
        ++it;
        // DefaultConvergenceCriteria::hasConverged
        if (it >= max_iter) {
            st = 1, converged = true;
            break;
        }
        const double cos_angle = 0.5 * (static_cast<double>(T[0]) + T[5] + T[10] - 1.0);
        const double tr2 = static_cast<double>(T[3]) * T[3] + static_cast<double>(T[7]) * T[7] + static_cast<double>(T[11]) * T[11];
        if (cos_angle >= 1.0 - trans_eps && tr2 <= trans_eps) {
            st = 2, converged = true;
            break;
        }
        const double mse = mse_sum / sums[0];
        if (std::fabs(mse - prev_mse) < 1e-12) {
            st = 3, converged = true;
            break;
        }
        if (std::fabs(mse - prev_mse) / prev_mse < fit_eps) {
            st = 4, converged = true;
            break;
        }
        prev_mse = mse;
    }
    // getFitnessScore: the ORIGINAL source moved by the final transform
    double fs = 0;
    int nr = 0;
    for (int i = 0; i < n_src; ++i) {
        float x = src_xyzi[4 * i], y = src_xyzi[4 * i + 1], z = src_xyzi[4 * i + 2];
        apply_f32(F, x, y, z);
        const float q[3] = {x, y, z};
        int idx;
        float d;
        if (tree.knn(q, 1, &idx, &d) < 1) continue;
        fs += d;
        ++nr;
    }
    for (int k = 0; k < 16; ++k) T16_out[k] = F[k];
    *fitness = nr > 0 ? fs / nr : std::numeric_limits<double>::max();
    *iterations = it;
    *state = st;
    return converged ? 1 : 0;
}

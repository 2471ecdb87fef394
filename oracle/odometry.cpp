// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp).
//
// Stage B: literal CPU restatement of the laserOdometry main loop,
// /root/reference/src/laserOdometry.cpp:59-66 (constants), :93-129 (state, TransformToStart), :267-291,
// :299-384 (edge correspondences), :387-483 (plane correspondences), :494-506 (solve, integration),
// :554-568 (hand-over + kd-tree rebuild).  DISTORTION is 0 (:59) so s == 1 everywhere and
// Identity.slerp(1, q) == +-q (Eigen::Quaternion::slerp), which q*v cannot tell apart.
#include "orc_common.hpp"
#include "oracle.h"
#include "kdtree.hpp"
#include "lm.hpp"
#include <chrono>

namespace orc {

struct Odometry {
    static constexpr double DISTANCE_SQ_THRESHOLD = 25;  // :65
    static constexpr double NEARBY_SCAN = 2.5;           // :66
    bool systemInited = false;
    double para_q[4] = {0, 0, 0, 1};  // :97-98  (x,y,z,w) q_last_curr
    double para_t[3] = {0, 0, 0};
    Quat q_w_curr{0, 0, 0, 1};  // :93-94
    V3 t_w_curr{0, 0, 0};
    std::vector<P4> cornerLast, surfLast;
    KdTree kdCorner, kdSurf;

    void TransformToStart(const P4& pi, P4& po) const {  // :111-129, s = 1
        Quat q{para_q[0], para_q[1], para_q[2], para_q[3]};
        V3 un = rotate(q, V3{pi.x, pi.y, pi.z}) + V3{1.0 * para_t[0], 1.0 * para_t[1], 1.0 * para_t[2]};
        po.x = static_cast<float>(un.x);
        po.y = static_cast<float>(un.y);
        po.z = static_cast<float>(un.z);
        po.i = pi.i;
    }

    int step(const P4* sharp, int nSharp, const P4* lessSharp, int nLessSharp, const P4* flat, int nFlat, const P4* lessFlat,
             int nLessFlat, OrcOdomStats* st) {
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        OrcOdomStats S;
        std::memset(&S, 0, sizeof S);
        auto t0 = clk::now();
        double t_assoc = 0, t_solve = 0;
        if (!systemInited) {
            systemInited = true;  // :267-271
        } else {
            const std::vector<P4>& CL = cornerLast;
            const std::vector<P4>& SL = surfLast;
            std::vector<Factor> factors;
            for (int opti_counter = 0; opti_counter < 2; ++opti_counter) {  // :278
                auto ta = clk::now();
                factors.clear();
                int corner_correspondence = 0, plane_correspondence = 0;
                for (int i = 0; i < nSharp; ++i) {  // :299-384
                    P4 pointSel;
                    TransformToStart(sharp[i], pointSel);
                    int nnI = 0;
                    float nnD = std::numeric_limits<float>::max();
                    const float qv[3] = {pointSel.x, pointSel.y, pointSel.z};
                    kdCorner.knn(qv, 1, &nnI, &nnD);
                    int closestPointInd = -1, minPointInd2 = -1;
                    if (nnD < DISTANCE_SQ_THRESHOLD) {
                        closestPointInd = nnI;
                        int closestPointScanID = int(CL[closestPointInd].i);
                        double minPointSqDis2 = DISTANCE_SQ_THRESHOLD;
                        for (int j = closestPointInd + 1; j < (int)CL.size(); ++j) {
                            if (int(CL[j].i) <= closestPointScanID) continue;
                            if (int(CL[j].i) > (closestPointScanID + NEARBY_SCAN)) break;
                            double pointSqDis = (CL[j].x - pointSel.x) * (CL[j].x - pointSel.x) + (CL[j].y - pointSel.y) * (CL[j].y - pointSel.y) +
                                                (CL[j].z - pointSel.z) * (CL[j].z - pointSel.z);
                            if (pointSqDis < minPointSqDis2) {
                                minPointSqDis2 = pointSqDis;
                                minPointInd2 = j;
                            }
                        }
                        for (int j = closestPointInd - 1; j >= 0; --j) {
                            if (int(CL[j].i) >= closestPointScanID) continue;
                            if (int(CL[j].i) < (closestPointScanID - NEARBY_SCAN)) break;
                            double pointSqDis = (CL[j].x - pointSel.x) * (CL[j].x - pointSel.x) + (CL[j].y - pointSel.y) * (CL[j].y - pointSel.y) +
                                                (CL[j].z - pointSel.z) * (CL[j].z - pointSel.z);
                            if (pointSqDis < minPointSqDis2) {
                                minPointSqDis2 = pointSqDis;
                                minPointInd2 = j;
                            }
                        }
                    }
                    if (minPointInd2 >= 0) {
                        Factor f;
                        f.kind = 0;
                        f.cp = {sharp[i].x, sharp[i].y, sharp[i].z};
                        f.a = {CL[closestPointInd].x, CL[closestPointInd].y, CL[closestPointInd].z};
                        f.b = {CL[minPointInd2].x, CL[minPointInd2].y, CL[minPointInd2].z};
                        factors.push_back(f);
                        corner_correspondence++;
                    }
                }
                for (int i = 0; i < nFlat; ++i) {  // :387-483
                    P4 pointSel;
                    TransformToStart(flat[i], pointSel);
                    int nnI = 0;
                    float nnD = std::numeric_limits<float>::max();
                    const float qv[3] = {pointSel.x, pointSel.y, pointSel.z};
                    kdSurf.knn(qv, 1, &nnI, &nnD);
                    int closestPointInd = -1, minPointInd2 = -1, minPointInd3 = -1;
                    if (nnD < DISTANCE_SQ_THRESHOLD) {
                        closestPointInd = nnI;
                        int closestPointScanID = int(SL[closestPointInd].i);
                        double minPointSqDis2 = DISTANCE_SQ_THRESHOLD, minPointSqDis3 = DISTANCE_SQ_THRESHOLD;
                        for (int j = closestPointInd + 1; j < (int)SL.size(); ++j) {
                            if (int(SL[j].i) > (closestPointScanID + NEARBY_SCAN)) break;
                            double pointSqDis = (SL[j].x - pointSel.x) * (SL[j].x - pointSel.x) + (SL[j].y - pointSel.y) * (SL[j].y - pointSel.y) +
                                                (SL[j].z - pointSel.z) * (SL[j].z - pointSel.z);
                            if (int(SL[j].i) <= closestPointScanID && pointSqDis < minPointSqDis2) {
                                minPointSqDis2 = pointSqDis;
                                minPointInd2 = j;
                            } else if (int(SL[j].i) > closestPointScanID && pointSqDis < minPointSqDis3) {
                                minPointSqDis3 = pointSqDis;
                                minPointInd3 = j;
                            }
                        }
                        for (int j = closestPointInd - 1; j >= 0; --j) {
                            if (int(SL[j].i) < (closestPointScanID - NEARBY_SCAN)) break;
                            double pointSqDis = (SL[j].x - pointSel.x) * (SL[j].x - pointSel.x) + (SL[j].y - pointSel.y) * (SL[j].y - pointSel.y) +
                                                (SL[j].z - pointSel.z) * (SL[j].z - pointSel.z);
                            if (int(SL[j].i) >= closestPointScanID && pointSqDis < minPointSqDis2) {
                                minPointSqDis2 = pointSqDis;
                                minPointInd2 = j;
                            } else if (int(SL[j].i) < closestPointScanID && pointSqDis < minPointSqDis3) {
                                minPointSqDis3 = pointSqDis;
                                minPointInd3 = j;
                            }
                        }
                        if (minPointInd2 >= 0 && minPointInd3 >= 0) {
                            Factor f;
                            f.kind = 1;
                            f.cp = {flat[i].x, flat[i].y, flat[i].z};
                            const V3 pj{SL[closestPointInd].x, SL[closestPointInd].y, SL[closestPointInd].z};
                            const V3 pl{SL[minPointInd2].x, SL[minPointInd2].y, SL[minPointInd2].z};
                            const V3 pm{SL[minPointInd3].x, SL[minPointInd3].y, SL[minPointInd3].z};
                            V3 nrm = cross(pj - pl, pj - pm);  // lidarFactor.hpp:64-65
                            const double z = dot(nrm, nrm);
                            if (z > 0) nrm = nrm / std::sqrt(z);
                            f.a = pj;
                            f.b = nrm;
                            factors.push_back(f);
                            plane_correspondence++;
                        }
                    }
                }
                auto tb = clk::now();
                t_assoc += ms(ta, tb);
                S.n_edge[opti_counter] = corner_correspondence, S.n_plane[opti_counter] = plane_correspondence;
                double x[7] = {para_q[0], para_q[1], para_q[2], para_q[3], para_t[0], para_t[1], para_t[2]};
                LMSummary sum;
                ceres_solve(factors, x, &sum);  // :494-499
                for (int k = 0; k < 4; ++k) para_q[k] = x[k];
                for (int k = 0; k < 3; ++k) para_t[k] = x[4 + k];
                S.lm_iters[opti_counter] = sum.iterations;
                S.cost_init[opti_counter] = sum.initial_cost;
                S.cost_final[opti_counter] = sum.final_cost;
                t_solve += ms(tb, clk::now());
            }
            // :504-505
            const Quat q_last_curr{para_q[0], para_q[1], para_q[2], para_q[3]};
            t_w_curr = t_w_curr + rotate(q_w_curr, V3{para_t[0], para_t[1], para_t[2]});
            q_w_curr = qmul(q_w_curr, q_last_curr);
        }
        auto t1 = clk::now();
        // :554-568
        cornerLast.assign(lessSharp, lessSharp + nLessSharp);
        surfLast.assign(lessFlat, lessFlat + nLessFlat);
        kdCorner.build(cornerLast.data(), nLessSharp);
        kdSurf.build(surfLast.data(), nLessFlat);
        auto t2 = clk::now();
        S.t_ms[0] = t_assoc, S.t_ms[1] = t_solve, S.t_ms[2] = ms(t1, t2), S.t_ms[3] = ms(t0, t2);
        if (st) *st = S;
        return 0;
    }
};

}  // namespace orc

extern "C" {
void* orc_odom_create(void) { return new orc::Odometry(); }
void orc_odom_destroy(void* h) { delete static_cast<orc::Odometry*>(h); }
int orc_odom_step(void* h, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp, const float* flat, int n_flat,
                  const float* less_flat, int n_less_flat, double* q_lc, double* t_lc, double* q_w, double* t_w, OrcOdomStats* stats) {
    auto* o = static_cast<orc::Odometry*>(h);
    using orc::P4;
    int rc = o->step(reinterpret_cast<const P4*>(sharp), n_sharp, reinterpret_cast<const P4*>(less_sharp), n_less_sharp,
                     reinterpret_cast<const P4*>(flat), n_flat, reinterpret_cast<const P4*>(less_flat), n_less_flat, stats);
    for (int i = 0; i < 4; ++i) q_lc[i] = o->para_q[i];
    for (int i = 0; i < 3; ++i) t_lc[i] = o->para_t[i];
    q_w[0] = o->q_w_curr.x, q_w[1] = o->q_w_curr.y, q_w[2] = o->q_w_curr.z, q_w[3] = o->q_w_curr.w;
    t_w[0] = o->t_w_curr.x, t_w[1] = o->t_w_curr.y, t_w[2] = o->t_w_curr.z;
    return rc;
}
}

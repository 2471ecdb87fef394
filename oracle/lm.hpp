// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp).
//
// Restatement of what `ceres::Solve` does for the reference's two problems
// (laserOdometry.cpp:284-291, :494-499 and laserMapping.cpp:566-573, :713-721):
//   parameter blocks q (4, EigenQuaternionParameterization) and t (3); every residual block an
//   AutoDiffCostFunction over one lidarFactor.hpp functor with a shared HuberLoss(0.1);
//   options: DENSE_QR, max_num_iterations = 4, everything else default.
// Ceres is third-party and NOT under /root/reference; the pin is Ceres 1.12.0 (docker/Dockerfile:3).
// The published algorithm restated here (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
// dense_qr_solver.cc, residual_block.cc, corrector.cc, loss_function.cc, local_parameterization.cc):
//   * residual block: autodiff Jacobians (forward-mode Jets below, functors follow
//     lidarFactor.hpp:12-138 literally), projected through the 4x3 plus-Jacobian BEFORE the loss
//     correction; Huber rho(s) with rho'' <= 0 => residuals and Jacobian rows scaled by sqrt(rho');
//     block cost 0.5*rho(s).
//   * trust-region LM: Jacobi column scaling 1/(1+||col||) fixed at iteration 0; D = sqrt(clamp(
//     ||col||^2, 1e-6, 1e32)/radius) on the scaled Jacobian; step from a Householder QR of [J; D];
//     model_cost_change = -(J s).(r + J s / 2); candidate via Plus; parameter tolerance 1e-8 and
//     function tolerance 1e-6 tested BEFORE the step is accepted (the candidate is then not applied);
//     accept iff relative_decrease > 1e-3; radius /= max(1/3, 1-(2 rho-1)^3) on accept, /= 2,4,8..
//     on reject; initial radius 1e4, max 1e16; gradient tolerance 1e-10 on max|x - Plus(x,-g)|.
// PARITY UNPINNED: no reference test or fixture covers Ceres' trajectory; recalled from the Ceres
// 1.12-1.14 sources.
#pragma once
#include "orc_common.hpp"

namespace orc {

// One residual block.  kind 0: LidarEdgeFactor(cp, a, b, s=1)          lidarFactor.hpp:12-55
//                      kind 1: LidarPlaneFactor(cp, j, ljm_norm, s=1)  lidarFactor.hpp:57-104 (normal precomputed :64-65)
//                      kind 2: LidarPlaneNormFactor(cp, n, d)          lidarFactor.hpp:106-138
struct Factor {
    int kind;
    V3 cp;
    V3 a;      // edge: last_point_a | plane: last_point_j | planenorm: plane_unit_norm
    V3 b;      // edge: last_point_b | plane: ljm_norm     | planenorm: (negative_OA_dot_norm, -, -)
    int num_residuals() const { return kind == 0 ? 3 : 1; }
};

struct LMSummary {
    int iterations = 0;        // iterations attempted (<= 4)
    int successful_steps = 0;  // excluding iteration 0
    double initial_cost = 0, final_cost = 0;
    int termination = 0;  // 0 no-convergence (max iters), 1 gradient, 2 parameter, 3 function, 4 no blocks
    std::vector<double> cost_trace;  // cost after every iteration (x_cost_)
};

// residual (unrobustified) + 3x7 / 1x7 Jacobian wrt (qx,qy,qz,qw,tx,ty,tz) by forward-mode autodiff
void factor_eval(const Factor& f, const double* x7, double* residual, double* jac_rowmajor_7);

// Ceres-equivalent solve: x7 = (qx,qy,qz,qw,tx,ty,tz) in/out
void ceres_solve(const std::vector<Factor>& factors, double* x7, LMSummary* summary);

}  // namespace orc

// On-device Levenberg-Marquardt for the reference's two pose problems (laserOdometry.cpp:284-291, :494-499;
// laserMapping.cpp:566-573, :713-721): residual blocks of lidarFactor.hpp:12-138 over q (4, Eigen quaternion
// parameterisation) and t (3), HuberLoss(0.1), DENSE_QR, max_num_iterations 4, Ceres defaults otherwise.
//
// What runs where (k_lm_solve, one launch per solve):
//   evaluation  one thread per residual block: residual, analytic Jacobian of the un-normalised Eigen rotation
//               (equal to Ceres' autodiff up to rounding), projection through the 4x3
//               plus-Jacobian, Huber weight per block; reduced per workgroup to LM_NACC doubles (cost, g[6], upper
//               H[21], live edge / plane blocks) through a fixed-order LDS stage => bitwise run-to-run reproducible,
//               no float atomics.
//   step        after a grid barrier every workgroup sums the partials in workgroup order and advances Ceres'
//               trust-region state machine (Jacobi scaling fixed at iteration 0, D = sqrt(clamp(diag)/radius), step
//               acceptance, parameter/function/gradient tolerances, radius update).  The damped 6x6 system is solved by
//               Cholesky on the normal equations, which is algebraically what Ceres' DENSE_QR on [J; D] solves.
// Every data-dependent decision stays on the GPU; the host reads the final state once per stage.
#pragma once
#include <hip/hip_runtime.h>
#include "device_utils.hpp"
#include "common.hpp"
#include <cstdlib>

namespace scal {

// residual blocks, structure of arrays; slot i is live iff valid[i] != 0
struct FactorSoA {
    int* valid;
    int* kind;    // 0 LidarEdgeFactor(a,b)  1 LidarPlaneFactor(j, unit normal)  2 LidarPlaneNormFactor(n, d)
    double* cp;   // [3][cap]
    double* pa;   // [3][cap]
    double* pb;   // [3][cap]
    int cap;
};

struct LMState {
    double x[7];     // qx qy qz qw tx ty tz (accepted point)
    double cand[7];  // candidate point being evaluated
    double x_cost, mcc, radius, decrease_factor, x_norm;
    double H[21], g[6], scale[6];
    int iteration, done, successful, started, enabled;
    int termination;  // 0 max iterations, 1 gradient, 2 parameter, 3 function, 4 no residual blocks
    double cost_init, cost_final;
    int log_iters[2], log_success[2];  // per outer iteration, for the caller's statistics
    int log_n_edge[2], log_n_plane[2];
    double log_cost_init[2], log_cost_final[2];
};

// The part of LMState the trust-region state machine works on, without the per-outer-iteration log arrays (those are indexed
// with a run-time value, which would push a private copy into scratch memory): the serial step runs on a register copy of this.
struct LMCore {
    double x[7], cand[7];
    double x_cost, mcc, radius, decrease_factor, x_norm;
    double H[21], g[6], scale[6];
    int iteration, done, successful, started, enabled, termination;
    double cost_init, cost_final;
};

__device__ __forceinline__ void quat_plus(const double* x, const double* delta, double* o) {
    // EigenQuaternionParameterization::Plus: [sin|d| d/|d|, cos|d|] (x) q, no half angle
    const double nd = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (nd > 0.0) {
        double sn, cs;
        sincos(nd, &sn, &cs);  // one range reduction for both
        const double s = sn / nd;
        const double ax = s * delta[0], ay = s * delta[1], az = s * delta[2], aw = cs;
        const double bx = x[0], by = x[1], bz = x[2], bw = x[3];
        o[0] = aw * bx + ax * bw + ay * bz - az * by;
        o[1] = aw * by + ay * bw + az * bx - ax * bz;
        o[2] = aw * bz + az * bw + ax * by - ay * bx;
        o[3] = aw * bw - ax * bx - ay * by - az * bz;
    } else {
        o[0] = x[0], o[1] = x[1], o[2] = x[2], o[3] = x[3];
    }
    o[4] = x[4] + delta[3], o[5] = x[5] + delta[4], o[6] = x[6] + delta[5];
}

// Eigen q*v: v + w*2(u x v) + u x 2(u x v)
__device__ __forceinline__ void quat_rotate(const double* q, double vx, double vy, double vz, double* o) {
    double ux = q[1] * vz - q[2] * vy, uy = q[2] * vx - q[0] * vz, uz = q[0] * vy - q[1] * vx;
    ux += ux, uy += uy, uz += uz;
    const double cx = q[1] * uz - q[2] * uy, cy = q[2] * ux - q[0] * uz, cz = q[0] * uy - q[1] * ux;
    o[0] = (vx + q[3] * ux) + cx;
    o[1] = (vy + q[3] * uy) + cy;
    o[2] = (vz + q[3] * uz) + cz;
}

// robustified contribution of one block to (cost, g, H); acc[0]=cost, acc[1..6]=g, acc[7..27]=upper H row-major
__device__ __forceinline__ void factor_accumulate(int kind, const double* cp, const double* pa, const double* pb, const double* x, double* acc) {
    const double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
    double lp[3];
    quat_rotate(x, cp[0], cp[1], cp[2], lp);
    lp[0] += x[4], lp[1] += x[5], lp[2] += x[6];
    // d lp / d(qx,qy,qz) = -2w[cp]x - 2[u x cp]x - 2[u]x[cp]x ;  d lp / d qw = 2 (u x cp)
    const double c0 = cp[0], c1 = cp[1], c2 = cp[2];
    const double ucx = qy * c2 - qz * c1, ucy = qz * c0 - qx * c2, ucz = qx * c1 - qy * c0;
    // [u]x[cp]x = cp u^T - (u.cp) I
    const double udc = qx * c0 + qy * c1 + qz * c2;
    double A[3][3];
    // -2w[cp]x
    A[0][0] = 0, A[0][1] = 2 * qw * c2, A[0][2] = -2 * qw * c1;
    A[1][0] = -2 * qw * c2, A[1][1] = 0, A[1][2] = 2 * qw * c0;
    A[2][0] = 2 * qw * c1, A[2][1] = -2 * qw * c0, A[2][2] = 0;
    // -2[u x cp]x
    A[0][1] += 2 * ucz, A[0][2] += -2 * ucy;
    A[1][0] += -2 * ucz, A[1][2] += 2 * ucx;
    A[2][0] += 2 * ucy, A[2][1] += -2 * ucx;
    // -2 (cp u^T - (u.cp) I)
    const double cpv[3] = {c0, c1, c2}, uv[3] = {qx, qy, qz};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[r][c] += -2 * cpv[r] * uv[c] + (r == c ? 2 * udc : 0.0);
    const double bq[3] = {2 * ucx, 2 * ucy, 2 * ucz};
    // plus-Jacobian P (4x3): rows [w,z,-y], [-z,w,x], [y,-x,w], [-x,-y,-z]
    const double P[4][3] = {{qw, qz, -qy}, {-qz, qw, qx}, {qy, -qx, qw}, {-qx, -qy, -qz}};
    double JL[3][6];  // d lp / d local
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) JL[r][c] = A[r][0] * P[0][c] + A[r][1] * P[1][c] + A[r][2] * P[2][c] + bq[r] * P[3][c];
        JL[r][3] = r == 0, JL[r][4] = r == 1, JL[r][5] = r == 2;
    }
    // One residual row -> (g, H); HuberLoss(0.1): rho = s | 2a sqrt(s) - a^2, rho' = 1 | a/sqrt(s); rho'' <= 0 => rows
    // scaled by sqrt(rho').  Everything below is straight-line code with compile-time indices: no private-memory spills.
    const double huber_a = 0.1, huber_b = huber_a * huber_a;  // HuberLoss(a): b_ = a*a = 0.010000000000000002
    auto huber = [&](double s, double& rho0, double& rho1) {
        if (s > huber_b) {
            const double rt = sqrt(s);
            rho0 = 2.0 * huber_a * rt - huber_b;
            rho1 = fmax(2.2250738585072014e-308, huber_a / rt);
        } else {
            rho0 = s, rho1 = 1.0;
        }
    };
    auto add_row = [&](const double* Jr, double res, double rho1) {
        int k = 7;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            acc[1 + a] += rho1 * Jr[a] * res;
#pragma unroll
            for (int b = a; b < 6; ++b) acc[k++] += rho1 * Jr[a] * Jr[b];
        }
    };
    if (kind == 0) {  // r = (lp-a) x (lp-b) / |a-b| ; dr/dlp = [b-a]x / |a-b|
        const double ax = lp[0] - pa[0], ay = lp[1] - pa[1], az = lp[2] - pa[2];
        const double bx = lp[0] - pb[0], by = lp[1] - pb[1], bz = lp[2] - pb[2];
        const double dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
        const double den = sqrt(dx * dx + dy * dy + dz * dz);
        const double r0 = (ay * bz - az * by) / den, r1 = (az * bx - ax * bz) / den, r2 = (ax * by - ay * bx) / den;
        const double ex = -dx / den, ey = -dy / den, ez = -dz / den;  // (b - a)/den
        double J0[6], J1[6], J2[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            J0[c] = -ez * JL[1][c] + ey * JL[2][c];
            J1[c] = ez * JL[0][c] - ex * JL[2][c];
            J2[c] = -ey * JL[0][c] + ex * JL[1][c];
        }
        double s = 0;
        s += r0 * r0;
        s += r1 * r1;
        s += r2 * r2;
        double rho0, rho1;
        huber(s, rho0, rho1);
        acc[0] += 0.5 * rho0;
        add_row(J0, r0, rho1);
        add_row(J1, r1, rho1);
        add_row(J2, r2, rho1);
    } else {
        double nx, ny, nz, r0;
        if (kind == 1) {  // (lp - j) . n
            nx = pb[0], ny = pb[1], nz = pb[2];
            r0 = (lp[0] - pa[0]) * nx + (lp[1] - pa[1]) * ny + (lp[2] - pa[2]) * nz;
        } else {  // n . lp + d
            nx = pa[0], ny = pa[1], nz = pa[2];
            r0 = (nx * lp[0] + ny * lp[1] + nz * lp[2]) + pb[0];
        }
        double J0[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) J0[c] = nx * JL[0][c] + ny * JL[1][c] + nz * JL[2][c];
        double s = 0;
        s += r0 * r0;
        double rho0, rho1;
        huber(s, rho0, rho1);
        acc[0] += 0.5 * rho0;
        add_row(J0, r0, rho1);
    }
}

// Plain residual and Jacobian of one block with respect to the AMBIENT parameters (qx qy qz qw tx ty tz), as the reference's
// ceres::AutoDiffCostFunction<F, kRes, 4, 3> hands them to Ceres (lidarFactor.hpp:48-50, :96-98, :130-132): no loss function,
// no local parameterisation - in the Ceres-adapter mode Ceres applies both itself.  Returns the number of residual rows (3 / 1).
__device__ __forceinline__ int factor_residual_jacobian(int kind, const double* cp, const double* pa, const double* pb, const double* x, double* r,
                                                        double (*J)[7]) {
    const double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
    double lp[3];
    quat_rotate(x, cp[0], cp[1], cp[2], lp);
    lp[0] += x[4], lp[1] += x[5], lp[2] += x[6];
    const double c0 = cp[0], c1 = cp[1], c2 = cp[2];
    const double ucx = qy * c2 - qz * c1, ucy = qz * c0 - qx * c2, ucz = qx * c1 - qy * c0;
    const double udc = qx * c0 + qy * c1 + qz * c2;
    double A[3][7];  // d lp / d (qx qy qz qw tx ty tz)
    A[0][0] = 0, A[0][1] = 2 * qw * c2, A[0][2] = -2 * qw * c1;
    A[1][0] = -2 * qw * c2, A[1][1] = 0, A[1][2] = 2 * qw * c0;
    A[2][0] = 2 * qw * c1, A[2][1] = -2 * qw * c0, A[2][2] = 0;
    A[0][1] += 2 * ucz, A[0][2] += -2 * ucy;
    A[1][0] += -2 * ucz, A[1][2] += 2 * ucx;
    A[2][0] += 2 * ucy, A[2][1] += -2 * ucx;
    const double cpv[3] = {c0, c1, c2}, uv[3] = {qx, qy, qz};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) A[a][b] += -2 * cpv[a] * uv[b] + (a == b ? 2 * udc : 0.0);
    A[0][3] = 2 * ucx, A[1][3] = 2 * ucy, A[2][3] = 2 * ucz;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) A[a][4 + b] = a == b ? 1.0 : 0.0;
    if (kind == 0) {  // r = (lp-a) x (lp-b) / |a-b| ; dr/dlp = [b-a]x / |a-b|
        const double ax = lp[0] - pa[0], ay = lp[1] - pa[1], az = lp[2] - pa[2];
        const double bx = lp[0] - pb[0], by = lp[1] - pb[1], bz = lp[2] - pb[2];
        const double dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
        const double den = sqrt(dx * dx + dy * dy + dz * dz);
        r[0] = (ay * bz - az * by) / den, r[1] = (az * bx - ax * bz) / den, r[2] = (ax * by - ay * bx) / den;
        const double ex = -dx / den, ey = -dy / den, ez = -dz / den;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            J[0][c] = -ez * A[1][c] + ey * A[2][c];
            J[1][c] = ez * A[0][c] - ex * A[2][c];
            J[2][c] = -ey * A[0][c] + ex * A[1][c];
        }
        return 3;
    }
    double nx, ny, nz;
    if (kind == 1) {  // (lp - j) . n
        nx = pb[0], ny = pb[1], nz = pb[2];
        r[0] = (lp[0] - pa[0]) * nx + (lp[1] - pa[1]) * ny + (lp[2] - pa[2]) * nz;
    } else {  // n . lp + d
        nx = pa[0], ny = pa[1], nz = pa[2];
        r[0] = (nx * lp[0] + ny * lp[1] + nz * lp[2]) + pb[0];
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) J[0][c] = nx * A[0][c] + ny * A[1][c] + nz * A[2][c];
    return 1;
}

// ---- Ceres-adapter mode: the live blocks of a FactorSoA, compacted, and their batched evaluation
struct BlockList {
    int* live;      // [cap] slot of the i-th live block
    int* row_off;   // [cap + 1] first residual row of the i-th live block
    int* counts;    // [0] live blocks, [1] residual rows
};
// one workgroup: stable compaction of the valid slots (slot order = the order the reference adds its residual blocks in)
__device__ __forceinline__ void k_blocks_compact_body(const FactorSoA& f, const int* __restrict__ d_nslots, const BlockList& bl) {
    __shared__ int s_scan[17];
    __shared__ int s_run[2];
    const int n = min(*d_nslots, f.cap);
    if (threadIdx.x == 0) s_run[0] = 0, s_run[1] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < n && f.valid[i]) ? 1 : 0;
        const int rows = v ? (f.kind[i] == 0 ? 3 : 1) : 0;
        int tot_v, tot_r;
        const int pv = block_exclusive_scan(v, s_scan, &tot_v);
        const int pr = block_exclusive_scan(rows, s_scan, &tot_r);
        if (v) {
            bl.live[s_run[0] + pv] = i;
            bl.row_off[s_run[0] + pv] = s_run[1] + pr;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run[0] += tot_v, s_run[1] += tot_r;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bl.row_off[s_run[0]] = s_run[1];
        bl.counts[0] = s_run[0], bl.counts[1] = s_run[1];
    }
}
SCAL_KERNEL(1024, k_blocks_compact)
// one thread per live block: residual rows (and, if wanted, their 7 ambient Jacobian columns, row-major) at the pose x7
__device__ __forceinline__ void k_blocks_eval_body(const FactorSoA& f, const BlockList& bl, const double* __restrict__ x7, int want_jac,
                                                           double* __restrict__ residuals, double* __restrict__ jac) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bl.counts[0]) return;
    const int i = bl.live[b];
    double x[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = x7[k];
    const double cp[3] = {f.cp[i], f.cp[f.cap + i], f.cp[2 * f.cap + i]};
    const double pa[3] = {f.pa[i], f.pa[f.cap + i], f.pa[2 * f.cap + i]};
    const double pb[3] = {f.pb[i], f.pb[f.cap + i], f.pb[2 * f.cap + i]};
    double r[3], J[3][7];
    const int rows = factor_residual_jacobian(f.kind[i], cp, pa, pb, x, r, J);
    const int r0 = bl.row_off[b];
    for (int q = 0; q < rows; ++q) {
        residuals[r0 + q] = r[q];
        if (want_jac)
#pragma unroll
            for (int c = 0; c < 7; ++c) jac[static_cast<size_t>(r0 + q) * 7 + c] = J[q][c];
    }
}
SCAL_KERNEL(256, k_blocks_eval)
// blocks as records for the host (struct scal_block of the C-ABI: int kind, int pad, double cp[3], pa[3], pb[3] = 80 bytes)
__device__ __forceinline__ void k_blocks_export_body(const FactorSoA& f, const BlockList& bl, double* __restrict__ out10) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bl.counts[0]) return;
    const int i = bl.live[b];
    double* o = out10 + static_cast<size_t>(b) * 10;
    reinterpret_cast<int*>(o)[0] = f.kind[i], reinterpret_cast<int*>(o)[1] = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) o[1 + a] = f.cp[a * f.cap + i], o[4 + a] = f.pa[a * f.cap + i], o[7 + a] = f.pb[a * f.cap + i];
}
SCAL_KERNEL(256, k_blocks_export)

constexpr int LM_NACC = 30;  // cost, g[6], upper H[21], number of live edge blocks, number of live plane blocks

// which = 0: evaluate at st->x (iteration zero), 1: at st->cand
__device__ __forceinline__ void k_lm_eval_body(const FactorSoA& f, const int* __restrict__ d_nslots, const LMState* __restrict__ st, int which,
                                                 double* __restrict__ partials) {
    __shared__ double red[4][LM_NACC];
    const int n = d_nslots ? min(*d_nslots, f.cap) : f.cap;
    const int nb = (n + 255) / 256;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    if (!st->enabled || st->done) return;
    const double* x = which ? st->cand : st->x;
    double xl[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) xl[k] = x[k];
    double acc[LM_NACC];
#pragma unroll
    for (int k = 0; k < LM_NACC; ++k) acc[k] = 0.0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && f.valid[i]) {
        const double cp[3] = {f.cp[i], f.cp[f.cap + i], f.cp[2 * f.cap + i]};
        const double pa[3] = {f.pa[i], f.pa[f.cap + i], f.pa[2 * f.cap + i]};
        const double pb[3] = {f.pb[i], f.pb[f.cap + i], f.pb[2 * f.cap + i]};
        factor_accumulate(f.kind[i], cp, pa, pb, xl, acc);
        acc[28] += f.kind[i] == 0 ? 1.0 : 0.0;
        acc[29] += f.kind[i] == 0 ? 0.0 : 1.0;
    }
#pragma unroll
    for (int k = 0; k < LM_NACC; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane_id() == 0) red[wave_id()][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < LM_NACC) partials[blockIdx.x * LM_NACC + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
SCAL_KERNEL(256, k_lm_eval)

__device__ __forceinline__ int hidx(int a, int b) {  // index into upper-triangular row-major H[21], a <= b
    return a * 6 - a * (a - 1) / 2 + (b - a);
}

// solve (Hs + diag(d2)) y = gs by Cholesky; returns false when not positive definite / not finite
__device__ __forceinline__ bool chol_solve6(const double* Hs, const double* d2, const double* gs, double* y) {
    // one division per pivot: every other division by a diagonal entry is a multiplication by its reciprocal
    double L[6][6], inv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = Hs[hidx(j, i)] + (i == j ? d2[i] : 0.0);
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i][i] = sqrt(s);
                inv[i] = 1.0 / L[i][i];
            } else {
                L[i][j] = s * inv[j];
            }
        }
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = gs[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i][k] * z[k];
        z[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = z[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * y[k];
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (!isfinite(y[i])) return false;
    return true;
}

// LevenbergMarquardtStrategy::ComputeStep + TrustRegionMinimizer::ComputeTrustRegionStep; loops over invalid steps
template <class State>
__device__ __forceinline__ void lm_compute_candidate(State* st) {
    const int max_num_iterations = 4;
    while (!st->done) {
        if (st->iteration >= max_num_iterations) {
            st->done = 1, st->termination = 0;
            break;
        }
        st->iteration++;
        double Hs[21], gs[6], d2[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            gs[a] = st->g[a] * st->scale[a];
#pragma unroll
            for (int b = a; b < 6; ++b) Hs[hidx(a, b)] = st->H[hidx(a, b)] * st->scale[a] * st->scale[b];
        }
        const double inv_radius = 1.0 / st->radius;
#pragma unroll
        for (int a = 0; a < 6; ++a) d2[a] = fmin(fmax(Hs[hidx(a, a)], 1e-6), 1e32) * inv_radius;
        double y[6];
        if (blockIdx.x == 0) SCAL_STAMP(26);
        bool ok = chol_solve6(Hs, d2, gs, y);
        if (blockIdx.x == 0) SCAL_STAMP(27);
        double mcc = 0.0;
        if (ok) {
            // model_cost_change = -(J s).(r + J s/2) with s = -y  =  y.gs - y^T Hs y / 2
            double yg = 0.0, yhy = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                yg += y[a] * gs[a];
                double row = 0.0;
#pragma unroll
                for (int b = 0; b < 6; ++b) row += Hs[a <= b ? hidx(a, b) : hidx(b, a)] * y[b];
                yhy += y[a] * row;
            }
            mcc = yg - 0.5 * yhy;
            ok = mcc > 0.0;
        }
        if (!ok) {  // HandleInvalidStep -> StepIsInvalid
            st->radius = st->radius / st->decrease_factor;
            st->decrease_factor *= 2.0;
            continue;
        }
        double delta[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) delta[a] = -y[a] * st->scale[a];
        st->mcc = mcc;
        quat_plus(st->x, delta, st->cand);
        return;
    }
}

// Trust-region bookkeeping after one evaluation (TrustRegionMinimizer: IterationZero / candidate evaluation).
template <class State>
__device__ __forceinline__ void lm_tail(State* st, const double* tot, int phase) {
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8, min_relative_decrease = 1e-3;
    // max-norm of x - Plus(x, -g), only ever compared with gradient_tolerance.  The translation part of that difference is g_t
    // itself (up to the rounding of x - (x - g), far below the tolerance's scale), so when one of its components already exceeds
    // the tolerance by a factor of two the answer is known without the quaternion exponential (sqrt, sin, cos, a division).
    auto grad_max_norm = [&]() {
        const double gt = fmax(fmax(fabs(st->g[3]), fabs(st->g[4])), fabs(st->g[5]));
        if (gt > 2.0 * gradient_tolerance && gt > 1e-6 * (1.0 + fmax(fmax(fabs(st->x[4]), fabs(st->x[5])), fabs(st->x[6])))) return gt;
        double neg[6], proj[7], m = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) neg[a] = -st->g[a];
        quat_plus(st->x, neg, proj);
#pragma unroll
        for (int k = 0; k < 7; ++k) m = fmax(m, fabs(st->x[k] - proj[k]));
        return m;
    };
    if (phase == 0) {
        st->radius = 1e4, st->decrease_factor = 2.0;
        st->iteration = 0, st->done = 0, st->successful = 0, st->started = 1, st->termination = 0, st->mcc = 0;
        st->enabled = 1;
        st->x_cost = tot[0], st->cost_init = tot[0], st->cost_final = tot[0];
        if (tot[28] + tot[29] == 0.0) {  // no residual blocks: Ceres finds no non-constant parameter block, parameters stay untouched
            st->done = 1, st->termination = 4;
            return;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) st->g[a] = tot[1 + a];
#pragma unroll
        for (int k = 0; k < 21; ++k) st->H[k] = tot[7 + k];
#pragma unroll
        for (int a = 0; a < 6; ++a) st->scale[a] = 1.0 / (1.0 + sqrt(st->H[hidx(a, a)]));  // Jacobi scaling, iteration 0 only
        double xn = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) xn += st->x[k] * st->x[k];
        st->x_norm = sqrt(xn);
        if (grad_max_norm() <= gradient_tolerance) {
            st->done = 1, st->termination = 1;
            return;
        }
        lm_compute_candidate(st);
        return;
    }
    const double candidate_cost = tot[0];
    double sn = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) sn += (st->x[k] - st->cand[k]) * (st->x[k] - st->cand[k]);
    if (sqrt(sn) <= parameter_tolerance * (st->x_norm + parameter_tolerance)) {
        st->done = 1, st->termination = 2;
        return;
    }
    if (fabs(st->x_cost - candidate_cost) <= function_tolerance * st->x_cost) {
        st->done = 1, st->termination = 3;
        return;
    }
    const double relative_decrease = (st->x_cost - candidate_cost) / st->mcc;
    if (relative_decrease > min_relative_decrease) {
#pragma unroll
        for (int k = 0; k < 7; ++k) st->x[k] = st->cand[k];
        double xn = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) xn += st->x[k] * st->x[k];
        st->x_norm = sqrt(xn);
        st->x_cost = candidate_cost, st->cost_final = candidate_cost;
#pragma unroll
        for (int a = 0; a < 6; ++a) st->g[a] = tot[1 + a];
#pragma unroll
        for (int k = 0; k < 21; ++k) st->H[k] = tot[7 + k];
        st->successful++;
        const double t = 2.0 * relative_decrease - 1.0;
        st->radius = fmin(1e16, st->radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
        st->decrease_factor = 2.0;
        if (grad_max_norm() <= gradient_tolerance) {
            st->done = 1, st->termination = 1;
            return;
        }
    } else {
        st->radius = st->radius / st->decrease_factor;
        st->decrease_factor *= 2.0;
    }
    lm_compute_candidate(st);
}

// Hooks a caller can hang on the solve kernel so that work which must precede / follow it needs no launch of its own:
//   Pre::operator()(first, stride, n)   every thread, before round 0: prepare the residual-block slots first, first + stride, ...
//                                        (a thread only ever evaluates the slots it prepared itself, so no barrier is needed)
//   Post::operator()(x, first, stride)  every thread, after the last round, with the final accepted point
struct LMNoHook {
    __device__ __forceinline__ void operator()(int, int, int) const {}
    __device__ __forceinline__ void operator()(const double*, int, int, bool) const {}
};

// The whole <= 4-iteration solve in ONE launch.  A small grid (<= LM_GRID workgroups, all resident at once on 256 CUs)
// walks the rounds together: every workgroup evaluates its tiles of residual blocks at the current point (round 0: the
// accepted point, later: the candidate), publishes LM_NACC partial sums, collects everybody's, sums ALL
// partials in block order and advances Ceres' trust-region state machine on its own private copy of the state.  Same
// inputs, same instruction sequence: the copies stay bitwise identical, so every workgroup takes the same decisions
// (including when to stop) without a second exchange.  Workgroup 0 writes the state back.
//
// Exchange: there is no barrier of its own.  Every partial sum is published as two 64-bit words that carry 32 bits of the double
// and the 32-bit sequence number of the round (LMSync::epoch + round + 1; the epoch is read at kernel start and advanced by
// workgroup 0 after the last round, when every workgroup has long read it).  64-bit relaxed agent-scope atomics are single-copy
// atomic and go to L2, so a consumer that polls a word sees either an older sequence number or the complete word: the data is
// its own flag, and a round costs one store -> poll-load hop instead of store -> drain -> arrival counter -> poll -> load.
// Words are double buffered by round parity: a workgroup can be at most one round ahead of the slowest one.  A poll budget
// bounds the spin: on exhaustion the solve is abandoned with termination 5 and the host clears epoch and words.
constexpr int LM_GRID = 48;
constexpr int LM_THREADS = 256;   // (512-thread workgroups, i.e. half the arrivals and partial sums per round, measured slower: 64 vs 58 us)
constexpr int LM_WAVES = LM_THREADS / 64;
constexpr size_t LM_PARTIAL_WORDS = static_cast<size_t>(4) * LM_GRID * LM_NACC;
constexpr int LM_LDS_BYTES = LM_WAVES * LM_NACC * 65 * 8;
// poll budget of the grid barrier (per translation unit; scal_*_debug_set_lm_polls lowers it to force the give-up path in tests)
static __device__ int g_lm_poll_budget = 1 << 22;
struct LMSync {
    unsigned abandoned; // sticky: a workgroup of some solve ran out of polls.  Every later solve on this exchange returns at once
                        // (its caller's chain drains as no-ops) until the host has cleared the words and this struct.
    unsigned epoch;     // sequence numbers consumed by all earlier solves
};
constexpr int LM_EPOCH_STEP = 8;   // sequence numbers reserved per solve (it uses at most 5: one per round), whatever happened in it
constexpr int LM_ABORT_CODE = 4;   // written to the caller's abort word by a workgroup that gives up (stage C: MAP_ABORT_LM)

template <class Pre, class Post>
__device__ __forceinline__ void k_lm_solve_body(const FactorSoA& f, const int* __restrict__ d_nslots, LMState* st, int outer,
                                                 const int* __restrict__ d_enable, double* partials, LMSync* sync,
                                                 int* d_abort, const Pre& pre, const Post& post) {
    // uniform over the grid (both words are written by earlier kernels only): a stopped chain leaves the state alone, and so does
    // every solve queued behind an abandoned one - its exchange words may hold tags of workgroups that gave up at different rounds
    if ((d_abort && *d_abort) || sync->abandoned) {
        post(st->x, blockIdx.x * LM_THREADS + threadIdx.x, gridDim.x * LM_THREADS, true);
        return;
    }
    extern __shared__ __align__(16) unsigned char lm_lds[];  // per-wave transpose buffers (row stride 65: conflict-free column sums)
    double (*xch)[LM_NACC][65] = reinterpret_cast<double (*)[LM_NACC][65]>(lm_lds);
    __shared__ double red[LM_WAVES][LM_NACC];
    __shared__ double tot[LM_NACC];
    __shared__ LMState L;
    __shared__ int s_ok;
    const int G = gridDim.x;
    const int n = d_nslots ? min(*d_nslots, f.cap) : f.cap;
    const int tid = threadIdx.x;
    const int enabled = d_enable ? *d_enable : 1;
    if (!enabled) {  // uniform over the grid: nobody reaches a barrier
        if (blockIdx.x == 0 && tid == 0) {
            st->enabled = 0, st->done = 1, st->termination = 4, st->iteration = 0, st->successful = 0;
            st->cost_init = 0, st->cost_final = 0;
            st->log_iters[outer] = 0, st->log_success[outer] = 0, st->log_cost_init[outer] = 0, st->log_cost_final[outer] = 0;
            st->log_n_edge[outer] = 0, st->log_n_plane[outer] = 0;
        }
        __syncthreads();
        post(st->x, blockIdx.x * LM_THREADS + tid, G * LM_THREADS, false);  // x is untouched: the prior pose stands
        return;
    }
    pre(blockIdx.x * LM_THREADS + tid, G * LM_THREADS, n);
    const unsigned epoch = sync->epoch;
    if (tid == 0) L = *st;  // only x carries over from the previous solve; lm_tail(phase 0) re-arms the rest
    __syncthreads();
    for (int round = 0; round < 5; ++round) {
        const int phase = round ? 1 : 0;
        if (phase && L.done) break;  // identical in every workgroup
        if (blockIdx.x == 0 && tid == 0 && round < 4) SCAL_STAMP(round * 6 + 0);
        double xl[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) xl[k] = phase ? L.cand[k] : L.x[k];
        double acc[LM_NACC];
#pragma unroll
        for (int k = 0; k < LM_NACC; ++k) acc[k] = 0.0;
        for (int i = blockIdx.x * LM_THREADS + tid; i < n; i += G * LM_THREADS) {
            if (f.valid[i]) {
                const double cp[3] = {f.cp[i], f.cp[f.cap + i], f.cp[2 * f.cap + i]};
                const double pa[3] = {f.pa[i], f.pa[f.cap + i], f.pa[2 * f.cap + i]};
                const double pb[3] = {f.pb[i], f.pb[f.cap + i], f.pb[2 * f.cap + i]};
                factor_accumulate(f.kind[i], cp, pa, pb, xl, acc);
                acc[28] += f.kind[i] == 0 ? 1.0 : 0.0;  // live residual blocks are counted here: no contended atomics in the
                acc[29] += f.kind[i] == 0 ? 0.0 : 1.0;  // association kernels
            }
        }
        if (blockIdx.x == 0 && tid == 0 && round < 4) SCAL_STAMP(round * 6 + 1);
        // wave reduction through LDS: conflict-free stores per lane, then lane k sums row k in lane order (fixed order =>
        // reproducible).  Six dependent cross-lane shuffle steps for each of the doubles would be the slow part otherwise.
        {
            const int w = wave_id(), l = lane_id();
#pragma unroll
            for (int k = 0; k < LM_NACC; ++k) xch[w][k][l] = acc[k];
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (l < LM_NACC) {
                double sum = 0.0;
#pragma unroll 8
                for (int j = 0; j < 64; ++j) sum += xch[w][l][j];
                red[w][l] = sum;
            }
        }
        if (tid == 0) s_ok = 1;
        __syncthreads();
        // Exchange of the partial sums WITHOUT a separate barrier: every partial travels as two 64-bit words, each carrying 32
        // bits of the double and the 32-bit sequence number of this round (64-bit relaxed atomics are single-copy atomic, so a
        // word is either stale - older sequence number - or complete).  The consumers poll the words themselves: what used to be
        // store -> drain -> arrival counter -> poll -> load is store -> poll-load.  Buffers are double buffered by round parity: a
        // workgroup can be at most one round ahead of the slowest one (it needs everybody's partials of round r to leave round r).
        const unsigned tag = epoch + static_cast<unsigned>(round) + 1u;
        unsigned long long* words = reinterpret_cast<unsigned long long*>(partials) + static_cast<size_t>(round & 1) * LM_GRID * LM_NACC * 2;
        if (tid < LM_NACC) {
            double wsum = red[0][tid];
#pragma unroll
            for (int w2 = 1; w2 < LM_WAVES; ++w2) wsum += red[w2][tid];  // fixed wave order
            const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(wsum));
            unsigned long long* mine = words + (static_cast<size_t>(blockIdx.x) * LM_NACC + tid) * 2;
            __hip_atomic_store(&mine[0], (static_cast<unsigned long long>(tag) << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine[1], (static_cast<unsigned long long>(tag) << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (blockIdx.x == 0 && tid == 0 && round < 4) SCAL_STAMP(round * 6 + 2);
        {   // sum the per-workgroup partials: 8 groups of workgroups in parallel, fixed order inside a group and across groups
            double (*gsum)[LM_NACC][65] = xch;  // reuse the transpose buffer: gsum[g][k] = xch[0][k][g]
            constexpr int PER = (LM_GRID + 7) / 8;
            const int k = tid & 31, g = tid >> 5;
            if (k < LM_NACC && g < 8) {
                unsigned long long lo[PER], hi[PER];
                unsigned pending = 0;
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    if (g + 8 * i < G) pending |= 1u << i;
                const int budget = g_lm_poll_budget;
                for (int poll = 0; pending && poll < budget; ++poll) {
#pragma unroll
                    for (int i = 0; i < PER; ++i)
                        if (pending & (1u << i)) {
                            const unsigned long long* src = words + (static_cast<size_t>(g + 8 * i) * LM_NACC + k) * 2;
                            lo[i] = __hip_atomic_load(&src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            hi[i] = __hip_atomic_load(&src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                    for (int i = 0; i < PER; ++i)
                        if ((pending & (1u << i)) && static_cast<unsigned>(lo[i] >> 32) == tag && static_cast<unsigned>(hi[i] >> 32) == tag) pending &= ~(1u << i);
                    if (pending) __builtin_amdgcn_s_sleep(1);
                }
                if (pending) s_ok = 0;  // poll budget exhausted
                double sacc = 0.0;
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    if (g + 8 * i < G) sacc += __longlong_as_double(static_cast<long long>((hi[i] << 32) | (lo[i] & 0xffffffffull)));
                gsum[0][k][g] = sacc;
            }
            __syncthreads();
            if (!s_ok) {
                // Give up (termination 5).  Workgroups may reach this in different rounds (or, in the last round, not at all), so
                // the verdict is published where the kernels behind this one see it: the sticky word of the exchange and the
                // caller's abort word (stage C: nothing of this step is committed, queued steps drain as no-ops).  The host reports
                // the step as failed and clears words, epoch and both flags before anything else runs on this exchange.
                if (tid == 0) {
                    L.done = 1, L.termination = 5;
                    __hip_atomic_store(&sync->abandoned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d_abort) __hip_atomic_store(d_abort, LM_ABORT_CODE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                break;
            }
            if (blockIdx.x == 0 && tid == 0 && round < 4) SCAL_STAMP(round * 6 + 3);
            if (tid < LM_NACC) {
                double t8 = 0.0;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) t8 += gsum[0][tid][gg];
                tot[tid] = t8;
            }
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid == 0 && round < 4) SCAL_STAMP(round * 6 + 4);
        if (tid == 0) {
            if (blockIdx.x == 0 && round == 1) SCAL_STAMP(24);
            // The serial step runs on a register copy of the state (LMCore: no run-time indexed arrays, so it does not go to
            // scratch) and of the totals: on the LDS copy every field access was a dependent ~100-cycle round trip.
            if (phase == 0) L.log_n_edge[outer] = static_cast<int>(tot[28]), L.log_n_plane[outer] = static_cast<int>(tot[29]);
            LMCore R;
            double t[LM_NACC];
#pragma unroll
            for (int k = 0; k < LM_NACC; ++k) t[k] = tot[k];
#pragma unroll
            for (int k = 0; k < 7; ++k) R.x[k] = L.x[k], R.cand[k] = L.cand[k];
            R.x_cost = L.x_cost, R.mcc = L.mcc, R.radius = L.radius, R.decrease_factor = L.decrease_factor, R.x_norm = L.x_norm;
#pragma unroll
            for (int k = 0; k < 21; ++k) R.H[k] = L.H[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) R.g[k] = L.g[k], R.scale[k] = L.scale[k];
            R.iteration = L.iteration, R.done = L.done, R.successful = L.successful, R.started = L.started, R.enabled = L.enabled;
            R.termination = L.termination, R.cost_init = L.cost_init, R.cost_final = L.cost_final;
            if (blockIdx.x == 0 && round == 1) SCAL_STAMP(25);
            lm_tail(&R, t, phase);
            if (blockIdx.x == 0 && round == 1) SCAL_STAMP(30);
#pragma unroll
            for (int k = 0; k < 7; ++k) L.x[k] = R.x[k], L.cand[k] = R.cand[k];
            L.x_cost = R.x_cost, L.mcc = R.mcc, L.radius = R.radius, L.decrease_factor = R.decrease_factor, L.x_norm = R.x_norm;
#pragma unroll
            for (int k = 0; k < 21; ++k) L.H[k] = R.H[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) L.g[k] = R.g[k], L.scale[k] = R.scale[k];
            L.iteration = R.iteration, L.done = R.done, L.successful = R.successful, L.started = R.started, L.enabled = R.enabled;
            L.termination = R.termination, L.cost_init = R.cost_init, L.cost_final = R.cost_final;
            if (blockIdx.x == 0 && round == 1) SCAL_STAMP(31);
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid == 0 && round < 4) SCAL_STAMP(round * 6 + 5);
    }
    if (blockIdx.x == 0 && tid == 0) {
        L.log_iters[outer] = L.iteration, L.log_success[outer] = L.successful;
        L.log_cost_init[outer] = L.cost_init, L.log_cost_final[outer] = L.cost_final;
        *st = L;
        // A fixed stride per solve, not the rounds it took: a tag a workgroup published before some other one gave up can then
        // never equal a tag of a later solve.
        sync->epoch = epoch + static_cast<unsigned>(LM_EPOCH_STEP);
    }
    __syncthreads();  // block 0's hook may publish *st
    post(L.x, blockIdx.x * LM_THREADS + tid, G * LM_THREADS, L.termination == 5);
}

// the kernel proper: argument set blockIdx.z of a batch (batch.hpp); a sequence's workgroups are its own gridDim.x of them, with
// its own exchange words, so the solves of several sequences share a launch without knowing of each other
template <class Pre, class Post>
struct LMSolveKernel {
    using traits = KernelTraits<decltype(&k_lm_solve_body<Pre, Post>)>;
    struct tag {
        template <class... T>
        __device__ __forceinline__ static void call(const T&... t) {
            k_lm_solve_body<Pre, Post>(t...);
        }
    };
};
template <class Pre, class Post>
static __global__ void __launch_bounds__(LM_THREADS) k_lm_solve(Batch<typename LMSolveKernel<Pre, Post>::traits::pack> b) {
    batch_call<typename LMSolveKernel<Pre, Post>::tag>(b.p[blockIdx.z], std::make_index_sequence<LMSolveKernel<Pre, Post>::traits::n>());
}
// the same code compiled for TWO workgroups per CU (256 instead of 426 registers per lane, the rest spilled): 512 resident workgroups
// on the device, for launches of more than LM_WGS_PER_LAUNCH workgroups.  Same arithmetic, same results.
template <class Pre, class Post>
static __global__ void __launch_bounds__(LM_THREADS, 2) k_lm_solve_slim(Batch<typename LMSolveKernel<Pre, Post>::traits::pack> b) {
    batch_call<typename LMSolveKernel<Pre, Post>::tag>(b.p[blockIdx.z], std::make_index_sequence<LMSolveKernel<Pre, Post>::traits::n>());
}
// The solve keeps its <= LM_GRID workgroups per sequence spinning on each other's partial sums, and at 426 registers per lane only
// ONE of its workgroups fits a CU: 256 resident workgroups on the whole device.  Stage B's and stage C's solves run on different
// streams and may overlap, so one launch of the full-register kernel may bring at most LM_WGS_PER_LAUNCH = 128 workgroups (two
// sequences of stage C's 48, four of stage B's 9); anything larger goes to the two-workgroups-per-CU build of the same code (512
// resident workgroups).  (32 workgroups per solve, so that four of stage C's fit: their solves drop from 77 to 59 us, but four
// sequences run at 5,550 scans/s instead of 6,000 - 164 CUs held by spinning full-register workgroups starve the other stages.)  (Tried first: all solves ordered behind each other across the two streams with an event chain - it welds
// stage B's and stage C's chains into one, 5,500 scans/s for four sequences; two launches of two sequences each - 5,700; the slim
// build for every batch of more than two sequences of 48 workgroups - 6,000, its solve taking 77 us against 45.)
constexpr int LM_WGS_PER_LAUNCH = 128;
static_assert(2 * BATCH_MAX * LM_GRID <= 512, "two overlapping batches of solves must fit the device at two workgroups per CU");
template <class Pre, class Post>
static hipError_t k_lm_solve_launch(const char* name, dim3 grid, dim3 block, int lds, hipStream_t s, int n, const void* const* packs) {
    using P = typename LMSolveKernel<Pre, Post>::traits::pack;
    if (n * static_cast<int>(grid.x) <= LM_WGS_PER_LAUNCH) return batch_launch_impl<P>(k_lm_solve<Pre, Post>, name, grid, block, lds, s, n, packs);
    return batch_launch_impl<P>(k_lm_solve_slim<Pre, Post>, name, grid, block, lds, s, n, packs);
}

// Results for the host in ONE launch: the state (and a counters struct) are written straight into pinned, device-visible host
// memory instead of one blit kernel per hipMemcpyAsync.  The host reads them after waiting on an event recorded behind this.
__device__ __forceinline__ void k_publish_body(const void* a, void* host_a, int words_a, const void* b, void* host_b, int words_b) {
    const unsigned* sa = static_cast<const unsigned*>(a);
    unsigned* da = static_cast<unsigned*>(host_a);
    for (int i = threadIdx.x; i < words_a; i += blockDim.x) da[i] = sa[i];
    const unsigned* sb = static_cast<const unsigned*>(b);
    unsigned* db = static_cast<unsigned*>(host_b);
    for (int i = threadIdx.x; i < words_b; i += blockDim.x) db[i] = sb[i];
}
SCAL_KERNEL(256, k_publish)
template <class A, class B>
inline void launch_publish(hipStream_t s, const A* a, A* host_a, const B* b, B* host_b) {
    static_assert(sizeof(A) % 4 == 0 && sizeof(B) % 4 == 0, "word copies");
    SCAL_LAUNCH("k_publish", k_publish, dim3(1), dim3(256), 0, s, static_cast<const void*>(a), static_cast<void*>(host_a),
                       a ? static_cast<int>(sizeof(A) / 4) : 0, static_cast<const void*>(b), static_cast<void*>(host_b),
                       b ? static_cast<int>(sizeof(B) / 4) : 0);
}

// host helper: one launch per solve.  `partials` holds 4 * LM_GRID * LM_NACC zero-initialised 64-bit words (two round parities x two
// tagged words per partial, LM_PARTIAL_WORDS), `sync` one zero-initialised LMSync; both are cleared again after an abandoned solve.
// The grid barrier needs all (<= LM_GRID) workgroups of a solve resident at once.  A plain launch does not promise that, a
// cooperative launch would - but cooperative launches of different streams take turns on this runtime, and stage B's and stage
// C's solves must overlap.  So the contexts check at creation that the device can hold LM_GRID such workgroups many times over
// (one per CU suffices: 256 CUs against 64), and the barrier gives up after a bounded number of polls (termination 5) instead of
// hanging if something else ever occupied the machine.
template <class Pre, class Post>
inline int lm_check_residency(int device) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SCAL_E_HIP;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_lm_solve<Pre, Post>), hipFuncAttributeMaxDynamicSharedMemorySize, LM_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_lm_solve_slim<Pre, Post>), hipFuncAttributeMaxDynamicSharedMemorySize, LM_LDS_BYTES) != hipSuccess)
        return SCAL_E_HIP;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_lm_solve<Pre, Post>), LM_THREADS, LM_LDS_BYTES) != hipSuccess)
        return SCAL_E_HIP;
    if (per_cu * prop.multiProcessorCount < 4 * LM_GRID) {
        set_error("device %d cannot keep %d LM workgroups resident (%d per CU x %d CUs)", device, LM_GRID, per_cu, prop.multiProcessorCount);
        return SCAL_E_NO_DEVICE;
    }
    return SCAL_OK;
}
inline int lm_set_poll_budget(int polls) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lm_poll_budget), &polls, sizeof(int)) == hipSuccess ? SCAL_OK : SCAL_E_HIP;
}

template <class Pre = LMNoHook, class Post = LMNoHook>
inline void launch_lm_solve(hipStream_t s, FactorSoA f, const int* d_nslots, LMState* st, const int* d_enable, double* partials, LMSync* sync,
                            int outer, int* d_abort = nullptr, Pre pre = Pre(), Post post = Post(), const char* prof_name = "k_lm_solve") {
    int g = (f.cap + LM_THREADS - 1) / LM_THREADS;
    g = g < 1 ? 1 : (g > LM_GRID ? LM_GRID : g);
    launch_or_record<typename LMSolveKernel<Pre, Post>::traits>(k_lm_solve_launch<Pre, Post>, prof_name, dim3(g), dim3(LM_THREADS), LM_LDS_BYTES, s, f, d_nslots, st,
                                                                 outer, d_enable, partials, sync, d_abort, pre, post);

}

}  // namespace scal

// ORACLE — TEST INFRASTRUCTURE ONLY.  See lm.hpp for what is restated and the sources it follows.
#include "lm.hpp"
#include "oracle.h"

namespace orc {

// ---------------------------------------------------------------- forward-mode autodiff (ceres::Jet<double,7>)
struct Jet {
    double a;
    double v[7];
    Jet() : a(0) { std::memset(v, 0, sizeof v); }
    explicit Jet(double s) : a(s) { std::memset(v, 0, sizeof v); }
    Jet(double s, int k) : a(s) {
        std::memset(v, 0, sizeof v);
        v[k] = 1.0;
    }
};
static inline Jet operator+(const Jet& f, const Jet& g) {
    Jet h;
    h.a = f.a + g.a;
    for (int i = 0; i < 7; ++i) h.v[i] = f.v[i] + g.v[i];
    return h;
}
static inline Jet operator-(const Jet& f, const Jet& g) {
    Jet h;
    h.a = f.a - g.a;
    for (int i = 0; i < 7; ++i) h.v[i] = f.v[i] - g.v[i];
    return h;
}
static inline Jet operator*(const Jet& f, const Jet& g) {
    Jet h;
    h.a = f.a * g.a;
    for (int i = 0; i < 7; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
}
static inline Jet operator/(const Jet& f, const Jet& g) {  // ceres jet.h: (f.v - f.a/g.a * g.v) / g.a
    Jet h;
    const double g_a_inverse = 1.0 / g.a;
    const double f_a_by_g_a = f.a * g_a_inverse;
    h.a = f_a_by_g_a;
    for (int i = 0; i < 7; ++i) h.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
    return h;
}
static inline Jet sqrt(const Jet& f) {
    Jet h;
    h.a = std::sqrt(f.a);
    const double two_a_inverse = 1.0 / (2.0 * h.a);
    for (int i = 0; i < 7; ++i) h.v[i] = f.v[i] * two_a_inverse;
    return h;
}
static inline double sqrt(double f) { return std::sqrt(f); }

template <class T>
struct Vec3T {
    T x, y, z;
};
template <class T>
static inline Vec3T<T> vsub(const Vec3T<T>& a, const Vec3T<T>& b) {
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <class T>
static inline Vec3T<T> vadd(const Vec3T<T>& a, const Vec3T<T>& b) {
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
template <class T>
static inline Vec3T<T> vcross(const Vec3T<T>& a, const Vec3T<T>& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T>
static inline T vdot(const Vec3T<T>& a, const Vec3T<T>& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <class T>
static inline T make(double c) {
    return T(c);
}

// Eigen Quaternion<T>{w,x,y,z} * Vector3<T> (_transformVector); q given in Ceres order (x,y,z,w).
// Eigen::Quaternion::slerp(T(1), q) from the identity returns +-q with a zero derivative through the
// interpolation weights, and q*v is even in q, so it is the identity here
// (lidarFactor.hpp:27-29, :79-81 with s = 1, DISTORTION 0).
template <class T>
static inline Vec3T<T> qrot(const T* q, const Vec3T<T>& v) {
    Vec3T<T> u{q[0], q[1], q[2]};
    Vec3T<T> uv = vcross(u, v);
    uv = vadd(uv, uv);
    Vec3T<T> wuv{q[3] * uv.x, q[3] * uv.y, q[3] * uv.z};
    return vadd(vadd(v, wuv), vcross(u, uv));
}

template <class T>
static void factor_functor(const Factor& f, const T* q, const T* t, T* residual) {
    Vec3T<T> cp{make<T>(f.cp.x), make<T>(f.cp.y), make<T>(f.cp.z)};
    if (f.kind == 0) {  // LidarEdgeFactor::operator(), lidarFactor.hpp:18-43
        Vec3T<T> lpa{make<T>(f.a.x), make<T>(f.a.y), make<T>(f.a.z)};
        Vec3T<T> lpb{make<T>(f.b.x), make<T>(f.b.y), make<T>(f.b.z)};
        Vec3T<T> tl{make<T>(1.0) * t[0], make<T>(1.0) * t[1], make<T>(1.0) * t[2]};
        Vec3T<T> lp = vadd(qrot(q, cp), tl);
        Vec3T<T> nu = vcross(vsub(lp, lpa), vsub(lp, lpb));
        Vec3T<T> de = vsub(lpa, lpb);
        residual[0] = nu.x / sqrt(vdot(de, de));
        residual[1] = nu.y / sqrt(vdot(de, de));
        residual[2] = nu.z / sqrt(vdot(de, de));
    } else if (f.kind == 1) {  // LidarPlaneFactor::operator(), lidarFactor.hpp:68-90
        Vec3T<T> lpj{make<T>(f.a.x), make<T>(f.a.y), make<T>(f.a.z)};
        Vec3T<T> ljm{make<T>(f.b.x), make<T>(f.b.y), make<T>(f.b.z)};
        Vec3T<T> tl{make<T>(1.0) * t[0], make<T>(1.0) * t[1], make<T>(1.0) * t[2]};
        Vec3T<T> lp = vadd(qrot(q, cp), tl);
        residual[0] = vdot(vsub(lp, lpj), ljm);
    } else {  // LidarPlaneNormFactor::operator(), lidarFactor.hpp:113-125
        Vec3T<T> tw{t[0], t[1], t[2]};
        Vec3T<T> point_w = vadd(qrot(q, cp), tw);
        Vec3T<T> nrm{make<T>(f.a.x), make<T>(f.a.y), make<T>(f.a.z)};
        residual[0] = vdot(nrm, point_w) + make<T>(f.b.x);
    }
}

void factor_eval(const Factor& f, const double* x7, double* residual, double* jac) {
    Jet x[7];
    for (int i = 0; i < 7; ++i) x[i] = Jet(x7[i], i);
    Jet r[3];
    factor_functor<Jet>(f, x, x + 4, r);
    const int nr = f.num_residuals();
    for (int i = 0; i < nr; ++i) {
        residual[i] = r[i].a;
        if (jac)
            for (int k = 0; k < 7; ++k) jac[7 * i + k] = r[i].v[k];
    }
}

static void factor_residual_only(const Factor& f, const double* x7, double* residual) {
    factor_functor<double>(f, x7, x7 + 4, residual);
}

// ---------------------------------------------------------------- Ceres pieces
// EigenQuaternionParameterization::Plus (local_parameterization.cc): delta_q (x) x, no half angle.
static void plus(const double* x, const double* delta, double* xpd) {
    const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (norm_delta > 0.0) {
        const double s = std::sin(norm_delta) / norm_delta;
        Quat dq{s * delta[0], s * delta[1], s * delta[2], std::cos(norm_delta)};
        Quat r = qmul(dq, Quat{x[0], x[1], x[2], x[3]});
        xpd[0] = r.x, xpd[1] = r.y, xpd[2] = r.z, xpd[3] = r.w;
    } else {
        for (int i = 0; i < 4; ++i) xpd[i] = x[i];
    }
    for (int i = 0; i < 3; ++i) xpd[4 + i] = x[4 + i] + delta[3 + i];
}

// HuberLoss(a = 0.1)::Evaluate (loss_function.cc)
static inline void huber(double s, double* rho) {
    const double a_ = 0.1, b_ = a_ * a_;
    if (s > b_) {
        const double r = std::sqrt(s);
        rho[0] = 2.0 * a_ * r - b_;
        rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s, rho[1] = 1.0, rho[2] = 0.0;
    }
}

struct Evaluator {
    const std::vector<Factor>& F;
    int num_rows;
    explicit Evaluator(const std::vector<Factor>& f) : F(f), num_rows(0) {
        for (const Factor& k : F) num_rows += k.num_residuals();
    }
    // ProgramEvaluator::Evaluate + ResidualBlock::Evaluate.  jacobian: num_rows x 6 row-major (local size).
    void evaluate(const double* x, double* cost, double* residuals, double* gradient, double* jacobian) const {
        *cost = 0.0;
        if (gradient)
            for (int i = 0; i < 6; ++i) gradient[i] = 0.0;
        // EigenQuaternionParameterization::ComputeJacobian, 4x3 row-major
        const double P[12] = {x[3], x[2], -x[1], -x[2], x[3], x[0], x[1], -x[0], x[3], -x[0], -x[1], -x[2]};
        int row = 0;
        for (const Factor& f : F) {
            const int nr = f.num_residuals();
            double r[3], Jg[21], Jl[18];
            const bool want_j = jacobian != nullptr || gradient != nullptr;
            if (want_j)
                factor_eval(f, x, r, Jg);
            else
                factor_residual_only(f, x, r);
            double squared_norm = 0.0;
            for (int i = 0; i < nr; ++i) squared_norm += r[i] * r[i];
            if (want_j) {
                for (int i = 0; i < nr; ++i) {
                    for (int c = 0; c < 3; ++c) {  // global (nr x 4) * plus-Jacobian (4 x 3)
                        double acc = 0.0;
                        for (int k = 0; k < 4; ++k) acc += Jg[7 * i + k] * P[3 * k + c];
                        Jl[6 * i + c] = acc;
                    }
                    for (int c = 0; c < 3; ++c) Jl[6 * i + 3 + c] = Jg[7 * i + 4 + c];
                }
            }
            double rho[3];
            huber(squared_norm, rho);
            *cost += 0.5 * rho[0];
            if (want_j || residuals) {
                // Corrector with rho[2] <= 0 (always for Huber): scale by sqrt(rho[1])
                const double sqrt_rho1 = std::sqrt(rho[1]);
                if (want_j)
                    for (int i = 0; i < nr * 6; ++i) Jl[i] *= sqrt_rho1;
                for (int i = 0; i < nr; ++i) r[i] *= sqrt_rho1;
            }
            if (residuals)
                for (int i = 0; i < nr; ++i) residuals[row + i] = r[i];
            if (jacobian)
                for (int i = 0; i < nr * 6; ++i) jacobian[6 * row + i] = Jl[i];
            if (gradient)
                for (int c = 0; c < 6; ++c) {
                    double acc = 0.0;
                    for (int i = 0; i < nr; ++i) acc += Jl[6 * i + c] * r[i];
                    gradient[c] += acc;
                }
            row += nr;
        }
    }
};

// min || A y - b ||, A (m x 6, row-major, destroyed), Householder QR as Eigen::HouseholderQR does.
static void householder_qr_solve(std::vector<double>& A, std::vector<double>& b, int m, double* y) {
    const int n = 6;
    for (int k = 0; k < n; ++k) {
        double tail = 0.0;
        for (int i = k + 1; i < m; ++i) tail += A[(size_t)i * n + k] * A[(size_t)i * n + k];
        const double c0 = A[(size_t)k * n + k];
        double tau, beta;
        if (tail <= std::numeric_limits<double>::min()) {
            tau = 0.0;
            beta = c0;
            for (int i = k + 1; i < m; ++i) A[(size_t)i * n + k] = 0.0;
        } else {
            beta = std::sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            const double inv = 1.0 / (c0 - beta);
            for (int i = k + 1; i < m; ++i) A[(size_t)i * n + k] *= inv;  // essential part
            tau = (beta - c0) / beta;
        }
        A[(size_t)k * n + k] = beta;
        if (tau != 0.0) {
            for (int c = k + 1; c < n; ++c) {  // apply H = I - tau v v^T from the left
                double s = A[(size_t)k * n + c];
                for (int i = k + 1; i < m; ++i) s += A[(size_t)i * n + k] * A[(size_t)i * n + c];
                s *= tau;
                A[(size_t)k * n + c] -= s;
                for (int i = k + 1; i < m; ++i) A[(size_t)i * n + c] -= s * A[(size_t)i * n + k];
            }
            double s = b[k];
            for (int i = k + 1; i < m; ++i) s += A[(size_t)i * n + k] * b[i];
            s *= tau;
            b[k] -= s;
            for (int i = k + 1; i < m; ++i) b[i] -= s * A[(size_t)i * n + k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int c = k + 1; c < n; ++c) s -= A[(size_t)k * n + c] * y[c];
        y[k] = s / A[(size_t)k * n + k];
    }
}

void ceres_solve(const std::vector<Factor>& factors, double* x_user, LMSummary* sum) {
    LMSummary local;
    LMSummary& S = sum ? *sum : local;
    S = LMSummary();
    if (factors.empty()) {  // program reduces to nothing: "No non-constant parameter blocks found."
        S.termination = 4;
        return;
    }
    Evaluator ev(factors);
    const int m = ev.num_rows, n = 6;
    const int max_num_iterations = 4;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_diagonal = 1e-6, max_diagonal = 1e32, max_radius = 1e16;
    double radius = 1e4, decrease_factor = 2.0;

    double x[7], candidate_x[7], delta[6], step[6], grad[6], scale[6];
    for (int i = 0; i < 7; ++i) x[i] = x_user[i];
    double x_norm = 0;
    for (int i = 0; i < 7; ++i) x_norm += x[i] * x[i];
    x_norm = std::sqrt(x_norm);
    std::vector<double> residuals(m), jac((size_t)m * n), A, bvec;
    double x_cost = 0, minimum_cost = std::numeric_limits<double>::max();
    double gradient_max_norm = 0;

    auto evaluate_gradient_and_jacobian = [&](int iteration) {
        ev.evaluate(x, &x_cost, residuals.data(), grad, jac.data());
        if (iteration == 0) {
            for (int c = 0; c < n; ++c) {
                double s = 0;
                for (int r = 0; r < m; ++r) s += jac[(size_t)r * n + c] * jac[(size_t)r * n + c];
                scale[c] = 1.0 / (1.0 + std::sqrt(s));
            }
        }
        for (int r = 0; r < m; ++r)
            for (int c = 0; c < n; ++c) jac[(size_t)r * n + c] *= scale[c];
        double neg[6], proj[7];
        for (int i = 0; i < 6; ++i) neg[i] = -grad[i];
        plus(x, neg, proj);
        gradient_max_norm = 0;
        for (int i = 0; i < 7; ++i) gradient_max_norm = std::max(gradient_max_norm, std::fabs(x[i] - proj[i]));
    };

    // IterationZero
    evaluate_gradient_and_jacobian(0);
    S.initial_cost = x_cost;
    bool step_is_successful = true;
    int iteration = 0;
    double reference_cost = x_cost;  // TrustRegionStepEvaluator, monotonic (max_consecutive_nonmonotonic_steps = 0)
    S.termination = 0;
    while (true) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (step_is_successful) {
            if (iteration > 0) S.successful_steps++;
            if (x_cost < minimum_cost) {
                minimum_cost = x_cost;
                for (int i = 0; i < 7; ++i) x_user[i] = x[i];
            }
        }
        S.cost_trace.push_back(x_cost);
        if (iteration >= max_num_iterations) {
            S.termination = 0;
            break;
        }
        if (step_is_successful && gradient_max_norm <= gradient_tolerance) {
            S.termination = 1;
            break;
        }
        if (radius <= 1e-32) break;
        ++iteration;
        S.iterations = iteration;

        // ComputeTrustRegionStep: LevenbergMarquardtStrategy::ComputeStep + DenseQRSolver
        double lm_diag[6];
        for (int c = 0; c < n; ++c) {
            double s = 0;
            for (int r = 0; r < m; ++r) s += jac[(size_t)r * n + c] * jac[(size_t)r * n + c];
            s = std::min(std::max(s, min_diagonal), max_diagonal);
            lm_diag[c] = std::sqrt(s / radius);
        }
        A.assign((size_t)(m + n) * n, 0.0);
        std::copy(jac.begin(), jac.end(), A.begin());
        for (int c = 0; c < n; ++c) A[(size_t)(m + c) * n + c] = lm_diag[c];
        bvec.assign(m + n, 0.0);
        std::copy(residuals.begin(), residuals.end(), bvec.begin());
        double yv[6];
        householder_qr_solve(A, bvec, m + n, yv);
        bool valid = true;
        for (int c = 0; c < n; ++c) {
            if (!std::isfinite(yv[c])) valid = false;
            step[c] = -yv[c];
        }
        double model_cost_change = 0;
        if (valid) {
            // model_cost_change = -(J s) . (r + J s / 2)
            for (int r = 0; r < m; ++r) {
                double js = 0;
                for (int c = 0; c < n; ++c) js += jac[(size_t)r * n + c] * step[c];
                model_cost_change += -js * (residuals[r] + js / 2.0);
            }
            valid = model_cost_change > 0.0;
        }
        if (!valid) {  // HandleInvalidStep -> StepIsInvalid == StepRejected
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            step_is_successful = false;
            continue;
        }
        for (int c = 0; c < n; ++c) delta[c] = step[c] * scale[c];
        // ComputeCandidatePointAndEvaluateCost
        plus(x, delta, candidate_x);
        double candidate_cost;
        ev.evaluate(candidate_x, &candidate_cost, nullptr, nullptr, nullptr);
        // ParameterToleranceReached
        double step_norm = 0;
        for (int i = 0; i < 7; ++i) step_norm += (x[i] - candidate_x[i]) * (x[i] - candidate_x[i]);
        step_norm = std::sqrt(step_norm);
        if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) {
            S.termination = 2;
            break;
        }
        // FunctionToleranceReached
        const double cost_change = x_cost - candidate_cost;
        if (std::fabs(cost_change) <= function_tolerance * x_cost) {
            S.termination = 3;
            break;
        }
        // IsStepSuccessful
        const double relative_decrease = (reference_cost - candidate_cost) / model_cost_change;
        if (relative_decrease > min_relative_decrease) {
            // HandleSuccessfulStep
            for (int i = 0; i < 7; ++i) x[i] = candidate_x[i];
            x_norm = 0;
            for (int i = 0; i < 7; ++i) x_norm += x[i] * x[i];
            x_norm = std::sqrt(x_norm);
            evaluate_gradient_and_jacobian(iteration);
            step_is_successful = true;
            radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
            radius = std::min(max_radius, radius);
            decrease_factor = 2.0;
            reference_cost = candidate_cost;
        } else {
            step_is_successful = false;
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
        }
    }
    S.final_cost = minimum_cost;
}

}  // namespace orc

extern "C" void orc_factor_eval(int kind, const double* cp, const double* params, const double* x7, double* residual3,
                                double* jac3x7) {
    orc::Factor f;
    f.kind = kind;
    f.cp = {cp[0], cp[1], cp[2]};
    f.a = {params[0], params[1], params[2]};
    f.b = {params[3], params[4], params[5]};
    orc::factor_eval(f, x7, residual3, jac3x7);
}

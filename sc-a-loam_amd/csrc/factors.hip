// Batched evaluation of lidarFactor.hpp:12-138 residual blocks at one pose: robustified cost, gradient and
// Gauss-Newton Hessian in the 6-dof local parameterisation.  For hosts that keep ceres::Problem orchestration
// (INTEGRATION.md "Ceres adapter mode") and for unit tests of the kernel the on-device solver is built from.
#include "common.hpp"
#include "lm_dev.hpp"

using namespace scal;

namespace scal {
__global__ void k_factors_arm(LMState* st, const double* x7) {
    if (threadIdx.x < 7) st->x[threadIdx.x] = x7[threadIdx.x];
    if (threadIdx.x == 0) st->enabled = 1, st->done = 0;
}
}  // namespace scal

extern "C" int scal_factors_eval(int device, int n, const int* kind, const double* cp, const double* pa, const double* pb, const double* x7,
                                 double* cost, double* gradient6, double* hessian6x6) {
    if (n < 0 || !x7 || !cost || (n > 0 && (!kind || !cp || !pa || !pb))) {
        set_error("scal_factors_eval: bad argument");
        return SCAL_E_ARG;
    }
    SCAL_TRY(select_device(device));
    const int cap = std::max(n, 1);
    DevBuf<int> dvalid, dkind;
    DevBuf<double> dcp, dpa, dpb, dx, dpart;
    DevBuf<LMState> dst;
    SCAL_TRY(dvalid.alloc(cap));
    SCAL_TRY(dkind.alloc(cap));
    SCAL_TRY(dcp.alloc(3 * (size_t)cap));
    SCAL_TRY(dpa.alloc(3 * (size_t)cap));
    SCAL_TRY(dpb.alloc(3 * (size_t)cap));
    SCAL_TRY(dx.alloc(8));
    const int nb = std::max(1, div_up(cap, 256));
    SCAL_TRY(dpart.alloc((size_t)LM_NACC * nb));
    SCAL_TRY(dst.alloc(1));
    // [n][3] rows -> [3][cap] planes
    std::vector<double> t(3 * (size_t)cap, 0.0);
    std::vector<int> ones(cap, 0);
    auto planes = [&](const double* src, DevBuf<double>& dst_) -> int {
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) t[(size_t)a * cap + i] = src[3 * (size_t)i + a];
        SCAL_HIP(hipMemcpy(dst_.p, t.data(), sizeof(double) * 3 * cap, hipMemcpyHostToDevice));
        return SCAL_OK;
    };
    if (n > 0) {
        SCAL_TRY(planes(cp, dcp));
        SCAL_TRY(planes(pa, dpa));
        SCAL_TRY(planes(pb, dpb));
        for (int i = 0; i < n; ++i) {
            if (kind[i] < 0 || kind[i] > 2) {
                set_error("scal_factors_eval: kind[%d] = %d", i, kind[i]);
                return SCAL_E_ARG;
            }
            ones[i] = 1;
        }
        SCAL_HIP(hipMemcpy(dkind.p, kind, sizeof(int) * n, hipMemcpyHostToDevice));
    }
    SCAL_HIP(hipMemcpy(dvalid.p, ones.data(), sizeof(int) * cap, hipMemcpyHostToDevice));
    SCAL_HIP(hipMemcpy(dx.p, x7, sizeof(double) * 7, hipMemcpyHostToDevice));
    SCAL_HIP(hipMemset(dst.p, 0, sizeof(LMState)));
    SCAL_HIP(hipMemset(dpart.p, 0, sizeof(double) * LM_NACC * nb));
    hipLaunchKernelGGL(k_factors_arm, dim3(1), dim3(64), 0, 0, dst.p, dx.p);
    FactorSoA F{dvalid.p, dkind.p, dcp.p, dpa.p, dpb.p, cap};
    SCAL_LAUNCH("k_lm_eval", k_lm_eval, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(nullptr), F, static_cast<const int*>(nullptr), dst.p, 0, dpart.p);
    SCAL_HIP(hipGetLastError());
    std::vector<double> part((size_t)LM_NACC * nb);
    SCAL_HIP(hipMemcpy(part.data(), dpart.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost));
    double tot[LM_NACC] = {0};
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < LM_NACC; ++k) tot[k] += part[(size_t)b * LM_NACC + k];
    *cost = tot[0];
    if (gradient6)
        for (int a = 0; a < 6; ++a) gradient6[a] = tot[1 + a];
    if (hessian6x6) {
        int k = 7;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) {
                hessian6x6[a * 6 + b] = tot[k];
                hessian6x6[b * 6 + a] = tot[k];
                ++k;
            }
    }
    return SCAL_OK;
}

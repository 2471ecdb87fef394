// Issue-rate probe for v_mfma_f64_16x16x4_f64 on gfx950: every SIMD of the chip runs waves that issue back-to-back MFMAs into
// independent accumulators.  Prints TFLOP/s at three occupancies; tools/bench_sc_matrix.py quotes
// it next to the spec figure.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k_probe(double* out, int iters, double a0, double b0) {
    d4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static double run(double* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<NACC>, dim3(blocks), dim3(256), 0, 0, d, 160, 1e-3, 1e-3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-3, 1e-3);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return 2048.0 * NACC * iters * blocks * 4 / (ms * 1e-3) / 1e12;
}
int main() {
    double* d;
    if (hipMalloc(&d, sizeof(double) * 2048 * 256) != hipSuccess) return 1;
    // 8 waves per SIMD x 8 accumulators, then the shape of k_sc_gram: 2 waves per SIMD x 16 accumulators, then 1 x 16
    const double t8 = run<8>(d, 2048, 20000), t2 = run<16>(d, 512, 40000), t1 = run<16>(d, 256, 40000);
    printf("{\"mfma_f64_16x16x4_tflops\": {\"8_waves_per_simd_8_acc\": %.2f, \"2_waves_per_simd_16_acc\": %.2f, \"1_wave_per_simd_16_acc\": %.2f}}\n", t8, t2, t1);
    (void)hipFree(d);
    return 0;
}

// Host cost and device time of a dependent chain of N small kernels on one stream: launched one by one vs replayed as a captured
// hipGraph.  Planning data for DESIGN.md section 9 item 1 (the pipelined bench is bound by ~330 us of launch calls per scan).
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/graph_launch_probe.hip -o gpurun_out/graph_launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                          \
        }                                                                      \
    } while (0)
__global__ void __launch_bounds__(256) k_step(float* buf, int n, float a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) buf[i] = buf[i] * a + 1.0f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int n = 1 << 16, reps = 400;
    float* d;
    CK(hipMalloc(&d, sizeof(float) * n));
    CK(hipMemset(d, 0, sizeof(float) * n));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("{\"chains\": [");
    for (int len : {12, 24, 48}) {
        // direct launches
        for (int w = 0; w < 20; ++w)
            for (int k = 0; k < len; ++k) hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, s, d, n, 0.5f);
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        for (int r = 0; r < reps; ++r)
            for (int k = 0; k < len; ++k) hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, s, d, n, 0.5f);
        const double host_direct = (now_us() - t0) / reps;
        CK(hipStreamSynchronize(s));
        const double total_direct = (now_us() - t0) / reps;
        // captured graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < len; ++k) hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, s, d, n, 0.5f);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        const double host_graph = (now_us() - t0) / reps;
        CK(hipStreamSynchronize(s));
        const double total_graph = (now_us() - t0) / reps;
        printf("%s{\"kernels\": %d, \"direct_host_us\": %.1f, \"direct_total_us\": %.1f, \"graph_host_us\": %.1f, \"graph_total_us\": %.1f}",
               len == 12 ? "" : ", ", len, host_direct, total_direct, host_graph, total_graph);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    printf("]}\n");
    CK(hipFree(d));
    return 0;
}

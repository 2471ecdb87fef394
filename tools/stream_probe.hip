// How many dependent kernel chains can one MI355X advance at once?  S streams, each fed a chain of L dependent kernels of ~T us
// (one workgroup spinning on the clock), launched from S host threads (one per stream) or replayed as one captured graph per
// stream.  Aggregate kernels/s against S tells whether small dependent kernels on different streams really overlap or share a
// serial resource (command processor / dispatch).  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -pthread tools/stream_probe.hip -o gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)
__global__ void __launch_bounds__(256) k_spin(int* buf, int ticks, int blocks_work) {
    // ~ticks * 10 ns of wall clock per workgroup (100 MHz counter), then one store so the kernel is not empty
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0) buf[blockIdx.x % blocks_work] += 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int L = 400;
    int* d;
    CK(hipMalloc(&d, sizeof(int) * 4096));
    CK(hipMemset(d, 0, sizeof(int) * 4096));
    printf("[");
    bool first = true;
    for (int blocks : {1, 64}) {
        for (int ticks : {0, 500}) {  // 0 us and 5 us of work per kernel
            for (int S : {1, 2, 4, 6, 8}) {
                std::vector<hipStream_t> st(S);
                for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                auto run_direct = [&](int reps) {
                    std::vector<std::thread> th;
                    for (int q = 0; q < S; ++q)
                        th.emplace_back([&, q] {
                            for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[q], d + 64 * q, ticks, 64);
                            (void)hipStreamSynchronize(st[q]);
                        });
                    for (auto& t : th) t.join();
                };
                run_direct(50);
                double t0 = now_us();
                run_direct(L);
                const double direct = now_us() - t0;
                // one graph per stream
                std::vector<hipGraphExec_t> ge(S);
                for (int q = 0; q < S; ++q) {
                    hipGraph_t g;
                    CK(hipStreamBeginCapture(st[q], hipStreamCaptureModeThreadLocal));
                    for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[q], d + 64 * q, ticks, 64);
                    CK(hipStreamEndCapture(st[q], &g));
                    CK(hipGraphInstantiate(&ge[q], g, nullptr, nullptr, 0));
                    CK(hipGraphDestroy(g));
                }
                for (int q = 0; q < S; ++q) CK(hipGraphLaunch(ge[q], st[q]));
                for (int q = 0; q < S; ++q) CK(hipStreamSynchronize(st[q]));
                t0 = now_us();
                for (int r = 0; r < L / 40; ++r)
                    for (int q = 0; q < S; ++q) CK(hipGraphLaunch(ge[q], st[q]));
                for (int q = 0; q < S; ++q) CK(hipStreamSynchronize(st[q]));
                const double graph = now_us() - t0;
                printf("%s\n {\"blocks\": %d, \"work_us\": %d, \"streams\": %d, \"direct_us_per_kernel_per_stream\": %.2f, \"direct_Mkernels_s\": %.3f, "
                       "\"graph_us_per_kernel_per_stream\": %.2f, \"graph_Mkernels_s\": %.3f}",
                       first ? "" : ",", blocks, ticks / 100, S, direct / L, S * L / direct, graph / L, S * L / graph);
                first = false;
                for (int q = 0; q < S; ++q) CK(hipGraphExecDestroy(ge[q]));
                for (auto& s : st) CK(hipStreamDestroy(s));
            }
        }
    }
    printf("\n]\n");
    return 0;
}

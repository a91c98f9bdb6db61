// Per-kernel cost of a chain of dependent launches on one stream: plain launches vs one hipGraph of the same chain.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(float *p, int n) {   // a small dependent step: every workgroup reads what the previous launch wrote
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[(i + 1) % n] + 1.0f;
}

int main() {
    float *p;
    const int n = 64 * 256;
    CK(hipMalloc(&p, n * sizeof(float)));
    CK(hipMemset(p, 0, n * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int chain = 64, reps = 50;
    for (int blocks : {1, 16, 64}) {
        auto run_plain = [&]() { for (int i = 0; i < chain; i++) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, st, p, blocks * 256); };
        run_plain(); CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) run_plain();
        CK(hipStreamSynchronize(st));
        const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        run_plain();
        CK(hipStreamEndCapture(st, &g));
        auto ti = std::chrono::steady_clock::now();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double inst = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ti).count();
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
        printf("{\"blocks\": %d, \"plain_us_per_kernel\": %.2f, \"graph_us_per_kernel\": %.2f, \"graph_instantiate_us\": %.1f, \"chain\": %d}\n", blocks, plain, graph, inst, chain);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}

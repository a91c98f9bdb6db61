// Host launch cost vs device back-to-back cost for small kernels (tail regime).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { long a[40]; };
__global__ void k_small(Big b, int *out, int spin) {
    long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && b.a[0] == 12345) out[0] = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int *d; hipMalloc(&d, 4);
    hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    Big b{};
    for (int spin_us : {0, 5, 20}) {
        const int spin = spin_us * 100;   // wall_clock64 ticks at 100 MHz
        for (int ns = 1; ns <= 2; ns++) {
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_small, dim3(4), dim3(256), 0, s[i % ns], b, d, spin);
            hipDeviceSynchronize();
            const int N = 4000;
            double t0 = now();
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_small, dim3(4), dim3(256), 0, s[i % ns], b, d, spin);
            double t1 = now();
            hipDeviceSynchronize();
            double t2 = now();
            printf("spin %2d us, %d stream(s): host %.2f us/launch, end-to-end %.2f us/kernel\n", spin_us, ns, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
        }
    }
    // graph of 64 kernel nodes on one stream
    {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k_small, dim3(4), dim3(256), 0, s[0], b, d, 500);
        hipStreamEndCapture(s[0], &g);
        double t0 = now();
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        double t1 = now();
        hipGraphLaunch(ge, s[0]); hipStreamSynchronize(s[0]);
        const int R = 50;
        double t2 = now();
        for (int r = 0; r < R; r++) hipGraphLaunch(ge, s[0]);
        double t3 = now();
        hipStreamSynchronize(s[0]);
        double t4 = now();
        printf("graph (64 x 5 us kernels): instantiate %.1f us, host %.2f us/graph launch, end-to-end %.2f us/kernel\n", (t1 - t0) * 1e6, (t3 - t2) / R * 1e6, (t4 - t2) / R / 64 * 1e6);
    }
    return 0;
}

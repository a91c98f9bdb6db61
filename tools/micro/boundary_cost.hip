// What a kernel boundary costs in the tail regime: dispatch only, a dependent-load prologue, dirty lines to write back.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { long a[40]; };
struct ListArg { int m[64]; };
__global__ void k_empty(Big b, int *out) { if (b.a[0] == 12345) out[0] = 1; }
__global__ void k_chase(Big b, const int *list, const int *done, int *out) {
    const int m = list[blockIdx.x];
    if (done[m]) return;
    out[threadIdx.x] = 1;
}
__global__ void k_chase3(Big b, const int *list, const int *slot, const long *off, const float *noise, float *out) {
    const int m = list[blockIdx.x];
    const long o = off[slot[m]];
    if (noise[o] == 123.f) out[threadIdx.x] = 1;
}
__global__ void k_karg(Big b, ListArg l, const int *done, int *out) {
    const int m = l.m[blockIdx.x];
    if (done[m]) return;
    out[threadIdx.x] = 1;
}
__global__ void k_write(Big b, float *buf, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) buf[(size_t)blockIdx.x * n + i] = (float)i;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void run(const char *name, F f) {
    for (int i = 0; i < 200; i++) f();
    hipDeviceSynchronize();
    const int N = 4000;
    double t0 = now();
    for (int i = 0; i < N; i++) f();
    hipDeviceSynchronize();
    printf("%-44s %.2f us/kernel\n", name, (now() - t0) / N * 1e6);
}
int main() {
    int *list, *done, *out, *slot; long *off; float *noise, *buf;
    hipMalloc(&list, 4096); hipMalloc(&done, 4096); hipMalloc(&out, 4096); hipMalloc(&slot, 4096); hipMalloc(&off, 8192);
    hipMalloc(&noise, 1 << 20); hipMalloc(&buf, 64 << 20);
    hipMemset(list, 0, 4096); hipMemset(done, 0xff, 4096); hipMemset(slot, 0, 4096); hipMemset(off, 0, 8192); hipMemset(noise, 0, 1 << 20);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Big b{}; ListArg l{};
    run("empty, 8 blocks", [&] { hipLaunchKernelGGL(k_empty, dim3(8), dim3(256), 0, s, b, out); });
    run("empty, 4096 blocks", [&] { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(256), 0, s, b, out); });
    run("list -> done (2 dependent loads), exit", [&] { hipLaunchKernelGGL(k_chase, dim3(8), dim3(256), 0, s, b, list, done, out); });
    run("list -> slot -> off -> noise (4 dependent)", [&] { hipLaunchKernelGGL(k_chase3, dim3(8), dim3(256), 0, s, b, list, slot, off, noise, (float *)out); });
    run("kernarg list -> done (1 dependent load)", [&] { hipLaunchKernelGGL(k_karg, dim3(8), dim3(256), 0, s, b, l, done, out); });
    run("8 blocks write 4 KB each", [&] { hipLaunchKernelGGL(k_write, dim3(8), dim3(256), 0, s, b, buf, 1024); });
    run("8 blocks write 64 KB each", [&] { hipLaunchKernelGGL(k_write, dim3(8), dim3(256), 0, s, b, buf, 16384); });
    run("128 blocks write 64 KB each (8 MB)", [&] { hipLaunchKernelGGL(k_write, dim3(128), dim3(256), 0, s, b, buf, 16384); });
    return 0;
}

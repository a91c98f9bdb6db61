// Can an LDS-DMA (global_load_lds_dwordx4, destination base in M0) write beyond the first 64 KB of a workgroup's LDS on gfx950?
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/lds_dma_hi tools/micro/lds_dma_hi.hip && tools/micro/lds_dma_hi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *src, float *out, const unsigned *offs, int n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned base = (unsigned)(size_t)lds;
    for (int i = threadIdx.x; i < 160 * 1024 / 4 - 64; i += blockDim.x) ((float *)lds)[i] = -1.0f;
    __syncthreads();
    for (int j = 0; j < n; j++) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + offs[j])), voff = threadIdx.x * 16;
        const float *s = src + 256 * j;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(voff), "s"(s), "s"(dst) : "memory");
        __syncthreads();
        for (int q = 0; q < 4; q++) out[256 * j + threadIdx.x * 4 + q] = ((float *)(lds + offs[j]))[threadIdx.x * 4 + q];
        __syncthreads();
    }
}
int main() {
    std::vector<unsigned> offs = {0, 32768, 61440, 65536, 66560, 98304, 131072, 150 * 1024};
    const int n = (int)offs.size();
    std::vector<float> h(256 * n);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)i;
    float *src, *out; unsigned *doffs;
    (void)hipMalloc(&src, h.size() * 4); (void)hipMalloc(&out, h.size() * 4); (void)hipMalloc(&doffs, n * 4);
    (void)hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(doffs, offs.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024 - 256, 0, src, out, doffs, n);
    std::vector<float> r(h.size());
    hipError_t e = hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost);
    printf("{\"hip\": \"%s\"", hipGetErrorString(e));
    for (int j = 0; j < n; j++) {
        int bad = 0;
        for (int i = 0; i < 256; i++) bad += r[256 * j + i] != h[256 * j + i];
        printf(", \"lds_offset_%u\": \"%s\"", offs[j], bad ? "MISMATCH" : "ok");
    }
    printf("}\n");
    return 0;
}

// Issue cost of the vector instructions k_fc_duo's row loop is made of, on gfx950, at 1 / 2 / 3 / 4 waves per SIMD: shader cycles per
// wave-instruction of v_fma_f32, v_pk_fma_f32 (VGPR source and SGPR-broadcast source), v_pk_mul_f32, v_pk_add_f32, v_readlane_b32,
// and of the loop's own per-row-side mix in its packed form (2 readlane + 2 pk_mul + 4 pk_add + 4 pk_fma) and in a scalar form
// (2 readlane + 4 mul + 8 add + 8 fma).  s_memtime brackets a long loop; the slowest wave of the launch is reported.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4000;
template <class T> __device__ __forceinline__ void opaque(T &v) { asm volatile("" : "+v"(v)); }

template <int KIND> struct Info;
template <> struct Info<0> { static constexpr int n = 16; static constexpr const char *name = "v_fma_f32 (16 chains)"; };
template <> struct Info<1> { static constexpr int n = 16; static constexpr const char *name = "v_pk_fma_f32 vgpr sources (16 chains)"; };
template <> struct Info<2> { static constexpr int n = 16; static constexpr const char *name = "v_pk_fma_f32 sgpr-pair src0 op_sel_hi 0 (16 chains)"; };
template <> struct Info<3> { static constexpr int n = 16; static constexpr const char *name = "v_pk_mul_f32"; };
template <> struct Info<4> { static constexpr int n = 16; static constexpr const char *name = "v_pk_add_f32"; };
template <> struct Info<5> { static constexpr int n = 16; static constexpr const char *name = "v_readlane_b32"; };
template <> struct Info<6> { static constexpr int n = 24; static constexpr const char *name = "row-side mix, packed: 2 x (2 readlane + 2 pk_mul + 4 pk_add + 4 pk_fma)"; };
template <> struct Info<7> { static constexpr int n = 44; static constexpr const char *name = "row-side mix, scalar: 2 x (2 readlane + 4 mul + 8 add + 8 fma)"; };
template <> struct Info<8> { static constexpr int n = 16; static constexpr const char *name = "v_pk_fma_f32 one dependent chain x 16"; };

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(unsigned long long *out, float *sink, float seed) {
    f2 a[8], b[8], c[8];
    float s[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = f2{seed + i, seed - i}; b[i] = f2{seed * 0.5f + i, 1.0f}; c[i] = f2{0.f, 0.f};
        s[i] = seed + threadIdx.x * 1e-3f + i; t[i] = seed - i - threadIdx.x * 2e-3f;
    }
    const f2 sxx = {seed, seed * 2.0f};
    int lanesel = 3;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %3, %2, %1" : "+v"(s[i]), "+v"(t[i]) : "v"(a[i].x), "v"(b[i].x));
            }
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %3, %2, %1" : "+v"(c[i]), "+v"(a[i]) : "v"(b[(i + 1) & 7]), "v"(b[i]));
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %2, %3, %1 op_sel_hi:[0,1,1]" : "+v"(c[i]), "+v"(a[i]) : "s"(sxx), "v"(b[i]));
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) { f2 d0, d1; asm volatile("v_pk_mul_f32 %0, %2, %3\n v_pk_mul_f32 %1, %3, %2" : "=&v"(d0), "=&v"(d1) : "v"(a[i]), "v"(b[i])); c[i] = d0; if (it == ITERS) a[i] = d1; }
        } else if constexpr (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 8; i++) { f2 d0, d1; asm volatile("v_pk_add_f32 %0, %2, %3\n v_pk_add_f32 %1, %3, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(d0), "=&v"(d1) : "v"(a[i]), "v"(b[i])); c[i] = d0; if (it == ITERS) a[i] = d1; }
        } else if constexpr (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 8; i++) { int r0, r1; asm volatile("v_readlane_b32 %0, %2, %4\n v_readlane_b32 %1, %3, %4" : "=&s"(r0), "=&s"(r1) : "v"(s[i]), "v"(t[i]), "s"(lanesel)); if (it == ITERS) lanesel += r0 + r1; }
        } else if constexpr (KIND == 6) {
#pragma unroll
            for (int h = 0; h < 2; h++) {   // sides A and B: accumulators c[4h .. 4h+3], rows a[h] (noise) / b[h] (base), activations s[h], t[h]
                opaque(a[2 * h]); opaque(a[2 * h + 1]); opaque(b[2 * h]); opaque(b[2 * h + 1]);   // "freshly loaded" every row
                const f2 elo = a[2 * h], ehi = a[2 * h + 1], tlo = b[2 * h], thi = b[2 * h + 1];
                const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[h]), lanesel));
                const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t[h]), lanesel));
                const f2 sc = {sxx.x, sxx.x}, xx0 = {x0, x0}, xx1 = {x1, x1};
                const f2 pl = sc * elo, ph = sc * ehi;
                const f2 wl0 = tlo + pl, wh0 = thi + ph, wl1 = tlo - pl, wh1 = thi - ph;
                c[4 * h + 0] = __builtin_elementwise_fma(xx0, wl0, c[4 * h + 0]);
                c[4 * h + 1] = __builtin_elementwise_fma(xx0, wh0, c[4 * h + 1]);
                c[4 * h + 2] = __builtin_elementwise_fma(xx1, wl1, c[4 * h + 2]);
                c[4 * h + 3] = __builtin_elementwise_fma(xx1, wh1, c[4 * h + 3]);
            }
            lanesel = (lanesel + 1) & 63;
        } else if constexpr (KIND == 7) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                opaque(a[2 * h]); opaque(a[2 * h + 1]); opaque(b[2 * h]); opaque(b[2 * h + 1]);
                const f2 elo = a[2 * h], ehi = a[2 * h + 1], tlo = b[2 * h], thi = b[2 * h + 1];
                const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[h]), lanesel));
                const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t[h]), lanesel));
                const float e4[4] = {elo.x, elo.y, ehi.x, ehi.y}, t4[4] = {tlo.x, tlo.y, thi.x, thi.y};
                float *acc0 = reinterpret_cast<float *>(&c[4 * h]), *acc1 = reinterpret_cast<float *>(&c[4 * h + 2]);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float p;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "s"(sxx.x), "v"(e4[q]));   // scalar on purpose: keep the SLP vectoriser out
                    float w0, w1;
                    asm volatile("v_add_f32 %0, %2, %3\n v_sub_f32 %1, %2, %3" : "=&v"(w0), "=&v"(w1) : "v"(t4[q]), "v"(p));
                    asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %4, %5, %1" : "+v"(acc0[q]), "+v"(acc1[q]) : "s"(x0), "v"(w0), "s"(x1), "v"(w1));
                }
            }
            lanesel = (lanesel + 1) & 63;
        } else if constexpr (KIND == 8) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[0]) : "v"(a[i & 7]), "v"(b[i & 7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = (float)lanesel;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y + b[i].x + c[i].x + c[i].y + s[i] + t[i];
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
int run(unsigned long long *d_out, float *d_sink) {
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * wps;   // 256 CUs x 4 SIMDs: one 256-thread workgroup per CU puts one wave on every SIMD
        std::vector<unsigned long long> h(blocks * 4);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, 1.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, 1.5f);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double n = (double)ITERS * Info<KIND>::n;
        // s_memtime ticks at a fixed 100 MHz on this part; the kernel's wall time at the nominal 2.4 GHz gives cycles too
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ticks_median_per_instr\": %.4f, \"ticks_max_per_instr\": %.4f, "
               "\"kernel_ms\": %.4f, \"ns_per_wave_instr\": %.3f, \"cycles_at_2p4GHz_per_wave_instr\": %.2f, \"cycles_per_instr_per_simd\": %.2f}\n",
               Info<KIND>::name, wps, h[h.size() / 2] / n, h.back() / n, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4, ms * 1e6 / n * 2.4 / wps);
    }
    return 0;
}

int main() {
    unsigned long long *d_out; float *d_sink;
    CK(hipMalloc(&d_out, 256 * 4 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 64));
    if (run<0>(d_out, d_sink) || run<1>(d_out, d_sink) || run<2>(d_out, d_sink) || run<3>(d_out, d_sink) || run<4>(d_out, d_sink) ||
        run<5>(d_out, d_sink) || run<6>(d_out, d_sink) || run<7>(d_out, d_sink) || run<8>(d_out, d_sink)) return 1;
    return 0;
}

// forward_large.h -- the GPU tree's LargeModel (gpu_implementation/neuroevolution/models/dqn.py:39-47 over models/base.py:50-95):
// conv 32 8x8/4 -> relu -> conv 64 4x4/2 -> relu -> conv 64 3x3/1 -> relu -> fc 512 -> relu -> out, all SAME, all with bias,
// member-unique weights formed on the fly as base[p] + scale * noise[off + p] (a GA child = its parent's vector + one mutation).
//
// Same numerics contract as forward.h: every dot product is an fp32 fmaf chain in (kh, kw, ci) / k order starting at 0; the fc is
// 4 k-slices of 1936 rows combined ((s0+s1)+(s2+s3)) + bias; taps in the SAME padding multiply a stored zero, which leaves a chain
// value untouched bit for bit (an accumulator that started at +0 is never -0).  The oracle is orc_forward_large_debug.
//
// 98 % of a member's 16.2 MB are the 7744 x 512 fc matrix: the step is HBM-bound on k_lfc (15.9 MB of noise per member-step); the
// convolutions (12 M MAC per member-step) run on the fp32 matrix cores from LDS-staged images.
#pragma once
#include "forward.h"

namespace dne {

// conv1 (8x8 stride 4, 4 -> 32 channels): forward.h's matrix-core conv1, one workgroup per (member, 16-channel half).
__global__ __launch_bounds__(256) void k_lconv1(FwdArgs A, const int *__restrict__ list, const uint8_t *__restrict__ stacks,
                                                float *__restrict__ y1) {
    __shared__ Conv1Lds S;
    const Item it = decode_item(A, blockIdx.x >> 1, list, 1, 1, 0, stacks, nullptr, A.done);
    if (it.skip) return;
    conv1_body<32>(S, A, it, y1, 0, 1, blockIdx.x & 1);
}

// conv2 / conv3 on v_mfma_f32_16x16x4_f32 (bitwise a k-ordered fp32 fmaf chain): one workgroup = one member, its 16-channel
// output tiles one after the other over an image staged once.  GEMM view [HOUT^2 positions] x [K*K*CIN] x [16]: the four k-values of one MFMA are four consecutive input channels
// of one tap, so lane (lp = l & 15, kq = l >> 4) feeds x[position lp][ci0 + kq] and w[tap][ci0 + kq][co lp].  The relu'd input
// image sits in LDS with a zero border (SAME padding) and a pixel stride PS chosen so that the 64 operand reads of an MFMA hit
// 64 different banks (S * PS = 4 mod 64: bank = 4 lp + kq); the member's perturbed 16-column weight tile sits next to it.
// Each wave owns two position tiles (two independent accumulators cover the dependent-MFMA latency).
template <int CIN, int COUT, int K, int S, int HIN, int HOUT, int PAD, int PS, bool NOISE>
__global__ __launch_bounds__(256) void k_lconv_mfma(FwdArgs A, const int *__restrict__ list, int w_off, int b_off,
                                                    const float *__restrict__ in_all, float *__restrict__ out_all,
                                                    int nsplit /* workgroups per member (1, 2 or 4): each takes NT / nsplit of the 16-channel tiles */) {
    constexpr int HP = (HOUT - 1) * S + K, NPOS = HOUT * HOUT, KK = K * K * CIN, NT = COUT / 16;
    constexpr int NTILE = (NPOS + 15) / 16;
    static_assert(NTILE <= 8 && CIN % 4 == 0, "two position tiles per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float *wt = reinterpret_cast<float *>(lds_raw);   // [KK][16]: one 16-channel weight tile at a time
    float *xf = wt + KK * 16;                          // [HP][HP][PS]: the image, staged once for all NT tiles
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, kq = lane >> 4;
    const int item = blockIdx.x / nsplit, part = blockIdx.x % nsplit;
    const int m = list ? list[item] : item;
    if (A.done && A.done[m]) return;
    const float sc = A.m_scale[m];
    const int64_t off = A.m_off[m];
    const float *base = A.bases + (size_t)A.m_slot[m] * A.base_stride;
    const float *src = in_all + (size_t)m * (HIN * HIN * CIN);
    // (staging loops unrolled 8 deep: a rolled loop pays one memory round trip per element -- this was the kernel's whole time)
#pragma unroll 8
    for (int i = tid; i < HP * HP * CIN; i += 256) {
        const int ci = i % CIN, px = (i / CIN) % HP, py = i / (CIN * HP), x = px - PAD, y = py - PAD;
        const bool in = x >= 0 && x < HIN && y >= 0 && y < HIN;
        float v = src[in ? ((size_t)y * HIN + x) * CIN + ci : 0];
        v = in && v > 0.0f ? v : 0.0f;                 // zero border; the previous layer's relu
        xf[(py * HP + px) * PS + ci] = v;
    }
    const int tA = wv, tB = wv + 4;
    const bool hasB = tB < NTILE;
    const int pA = min(tA * 16 + lp, NPOS - 1), pB = min(tB * 16 + lp, NPOS - 1);
    const float *xA = xf + ((pA / HOUT) * S * HP + (pA % HOUT) * S) * PS + kq;
    const float *xB = xf + ((pB / HOUT) * S * HP + (pB % HOUT) * S) * PS + kq;
    const float *wl = wt + kq * 16 + lp;
    for (int tile = part * (NT / nsplit); tile < (part + 1) * (NT / nsplit); tile++) {
        __syncthreads();                               // the previous tile's readers are done (first pass: nothing to wait for)
#pragma unroll 8
        for (int i = tid; i < KK * 16; i += 256) {
            const size_t p = (size_t)w_off + (size_t)(i >> 4) * COUT + tile * 16 + (i & 15);
            float w = base[p];
            if (NOISE) {                               // (materialised members are plain vectors: no noise row to add)
                float pv = sc * A.noise[off + p];
                w = base[p] + pv;
            }
            wt[i] = w;
        }
        float bias = base[b_off + tile * 16 + lp];
        if (NOISE) {
            float pvb = sc * A.noise[off + b_off + tile * 16 + lp];
            bias = base[b_off + tile * 16 + lp] + pvb;
        }
        __syncthreads();
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kh = 0; kh < K; kh++)
#pragma unroll 1
            for (int kw = 0; kw < K; kw++) {
                const int xo = (kh * HP + kw) * PS, wo = (kh * K + kw) * CIN * 16;
#pragma unroll
                for (int c0 = 0; c0 < CIN; c0 += 4) {
                    const float b = wl[wo + c0 * 16];
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xA[xo + c0], b, accA, 0, 0, 0);
                    if (hasB) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xB[xo + c0], b, accB, 0, 0, 0);
                }
            }
        float *out = out_all + (size_t)m * (NPOS * COUT) + tile * 16 + lp;
#pragma unroll
        for (int r = 0; r < 4; r++) {   // D[row = 4 * (l >> 4) + r][col = l & 15]
            const int posA = tA * 16 + kq * 4 + r, posB = tB * 16 + kq * 4 + r;
            if (posA < NPOS) out[(size_t)posA * COUT] = accA[r] + bias;
            if (hasB && posB < NPOS) out[(size_t)posB * COUT] = accB[r] + bias;
        }
    }
}

template <int CIN, int K, int S, int HOUT, int PS>
constexpr size_t lconv_mfma_lds_bytes() {
    constexpr int HP = (HOUT - 1) * S + K;
    return ((size_t)K * K * CIN * 16 + (size_t)HP * HP * PS) * 4;
}

// fc 7744 -> 512, streamed: workgroup = (member, 256-column half), wave = k-slice of 1936 rows, lane = 4 columns; two row batches
// of 4 in flight per wave.  The activations are relu(conv3) read 64 rows at a time, one per lane, and broadcast with v_readlane.
// NOISE = false: every member's vector is materialised (a GA child = its parent + one mutation, written out once per
// generation by k_materialize_children): the rows are read as they are -- half the bytes of streaming parent and noise rows.
// PAD (round 4, the lesson of k_fc_duo's FAT form): at 96 registers five of these workgroups fit a CU and the windows' launches together
// fill every register file -- the convolutions (106 .. 189 registers), k_lout and the renderer of the OTHER windows then wait for a slot
// instead of running beside the HBM stream.  PAD = 1 / 2 touches a high accumulation register so that at most two / one of its
// workgroups fit a CU.
template <bool NOISE, int RB, int PAD = 0>
__global__ __launch_bounds__(256) void k_lfc(FwdArgs A, const int *__restrict__ list, int n_items, const float *__restrict__ y3,
                                             float *__restrict__ y4) {
    if constexpr (PAD == 1) asm volatile("v_accvgpr_write_b32 a79, %0" : : "v"(0) : "a79");
    if constexpr (PAD == 2) asm volatile("v_accvgpr_write_b32 a167, %0" : : "v"(0) : "a167");
    __shared__ float part[4][256];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    __builtin_amdgcn_s_setprio(3);
    constexpr int ROWS = 1936, PITCH = 512, NB = ROWS / RB, BPC = 64 / RB;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int mi = item >> 1, half = item & 1;
        const int m = list ? list[mi] : mi;
        if (A.done && A.done[m]) continue;
        const float sc = A.m_scale[m];
        const int64_t off = A.m_off[m];
        const float *base = A.bases + (size_t)A.m_slot[m] * A.base_stride;
        const int kbeg = ROWS * wv;
        const float *eps = A.noise + off + L.fcw + (size_t)kbeg * PITCH + half * 256 + lane * 4;
        const float *th = base + L.fcw + (size_t)kbeg * PITCH + half * 256 + lane * 4;
        const float *xs = y3 + (size_t)m * 7744 + kbeg;
        auto load_x = [&](int c) {   // rows 64c .. 64c+63 of the slice (the last chunk has 16)
            float t = 0.0f;
            if (64 * c + lane < ROWS) {
                t = xs[64 * c + lane];
                t = t > 0.0f ? t : 0.0f;
            }
            return t;
        };
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        f4u e_cur[RB] = {}, e_nxt[RB] = {};
        f4a t_cur[RB], t_nxt[RB];
        float xv = load_x(0), xn = load_x(1);
#pragma unroll
        for (int i = 0; i < RB; i++) {
            if (NOISE) e_cur[i] = *(const f4u *)(eps + (size_t)i * PITCH);
            t_cur[i] = *(const f4a *)(th + (size_t)i * PITCH);
        }
        for (int bt = 0; bt < NB; bt++) {
            if (bt + 1 < NB) {
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const size_t ro = (size_t)((bt + 1) * RB + i) * PITCH;
                    if (NOISE) e_nxt[i] = *(const f4u *)(eps + ro);
                    t_nxt[i] = *(const f4a *)(th + ro);
                }
            }
            const int li = (bt % BPC) * RB;
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const float x = lane_bcast(xv, li + i);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float w = t_cur[i][q];
                    if (NOISE) {
                        float pv = sc * e_cur[i][q];
                        w = t_cur[i][q] + pv;
                    }
                    acc[q] = __builtin_fmaf(x, w, acc[q]);
                }
            }
#pragma unroll
            for (int i = 0; i < RB; i++) { if (NOISE) e_cur[i] = e_nxt[i]; t_cur[i] = t_nxt[i]; }
            if (bt % BPC == BPC - 1) {
                xv = xn;
                xn = load_x(bt / BPC + 2);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) part[wv][lane * 4 + q] = acc[q];
        __syncthreads();
        {
            const int j = tid, col = half * 256 + j;
            const float s01 = part[0][j] + part[1][j];
            const float s23 = part[2][j] + part[3][j];
            float s = s01 + s23;
            float bias = base[L.fcb + col];
            if (NOISE) {
                float pv = sc * A.noise[off + L.fcb + col];
                bias = base[L.fcb + col] + pv;
            }
            y4[(size_t)m * 512 + col] = s + bias;
        }
        __syncthreads();   // part is reused by the next item
    }
}

// The same fc for a handful of members (the tail of a generation): eight workgroups per member, one per 64-column block,
// wave = k-slice, lane = ONE column with 2 x 16 rows in flight -- eight times the workgroups pulling on one member's 15.9 MB.
template <bool NOISE>
__global__ __launch_bounds__(256) void k_lfc_cols(FwdArgs A, const int *__restrict__ list, const float *__restrict__ y3,
                                                  float *__restrict__ y4) {
    __shared__ float part[4][64];
    __shared__ float xs[4][1936];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    constexpr int ROWS = 1936, PITCH = 512, RB = 16, NB = ROWS / RB;
    const int mi = blockIdx.x >> 3, cb = blockIdx.x & 7;
    const int m = list ? list[mi] : mi;
    if (A.done && A.done[m]) return;
    const float sc = A.m_scale[m];
    const int64_t off = A.m_off[m];
    const float *base = A.bases + (size_t)A.m_slot[m] * A.base_stride;
    const int kbeg = ROWS * wv, col = cb * 64 + lane;
    const float *eps = A.noise + off + L.fcw + (size_t)kbeg * PITCH + col;
    const float *th = base + L.fcw + (size_t)kbeg * PITCH + col;
    float e_cur[RB] = {}, e_nxt[RB] = {}, t_cur[RB], t_nxt[RB];
#pragma unroll
    for (int i = 0; i < RB; i++) {
        if (NOISE) e_cur[i] = eps[(size_t)i * PITCH];
        t_cur[i] = th[(size_t)i * PITCH];
    }
    for (int i = lane; i < ROWS; i += 64) {   // the slice's activations (relu of conv3), each wave its own
        const float t = y3[(size_t)m * 7744 + kbeg + i];
        xs[wv][i] = t > 0.0f ? t : 0.0f;
    }
    float acc = 0.0f;
    for (int bt = 0; bt < NB; bt++) {
        if (bt + 1 < NB) {
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const size_t ro = (size_t)((bt + 1) * RB + i) * PITCH;
                if (NOISE) e_nxt[i] = eps[ro];
                t_nxt[i] = th[ro];
            }
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
            float w = t_cur[i];
            if (NOISE) {
                float pv = sc * e_cur[i];
                w = t_cur[i] + pv;
            }
            acc = __builtin_fmaf(xs[wv][bt * RB + i], w, acc);
        }
#pragma unroll
        for (int i = 0; i < RB; i++) { if (NOISE) e_cur[i] = e_nxt[i]; t_cur[i] = t_nxt[i]; }
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (tid < 64) {
        const int c = cb * 64 + tid;
        const float s01 = part[0][tid] + part[1][tid];
        const float s23 = part[2][tid] + part[3][tid];
        float s = s01 + s23;
        float bias = base[L.fcb + c];
        if (NOISE) {
            float pv = sc * A.noise[off + L.fcb + c];
            bias = base[L.fcb + c] + pv;
        }
        y4[(size_t)m * 512 + c] = s + bias;
    }
}

// relu + out layer (512 x nact: thread = inputs k = tid and 256 + tid, forward.h's out_products / out_wave_sums; the eight groups
// of 64 combined ((S0+S1)+(S2+S3)) + ((S4+S5)+(S6+S7))) + first-max argmax, one workgroup per member
__global__ __launch_bounds__(256) void k_lout(FwdArgs A, const int *__restrict__ list, const float *__restrict__ y4,
                                              int32_t *__restrict__ actions, float *__restrict__ logits_out) {
    __shared__ float red[4][2][OUT_NA];   // [wave][half of k][action]
    __shared__ float lg[32];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    const int nact = L.nact;
    const int m = list ? list[blockIdx.x] : blockIdx.x;
    if (A.done && A.done[m]) return;
    const float sc[1] = {A.m_scale[m]};
    const int64_t off = A.m_off[m];
    const float *base = A.bases + (size_t)A.m_slot[m] * A.base_stride;
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const int k = hf * 256 + tid;
        const float t = y4[(size_t)m * 512 + k];
        const float x[1] = {t > 0.0f ? t : 0.0f};
        float p[1][OUT_NA];
        out_products<1>(p, x, base + L.ow + k * nact, A.noise + off + L.ow + k * nact, sc, nact);
        out_wave_sums<1>(p, nact, red[wv] + hf, lane);
    }
    __syncthreads();
    if (tid < nact) {
        const float s01 = red[0][0][tid] + red[1][0][tid], s23 = red[2][0][tid] + red[3][0][tid];
        const float s45 = red[0][1][tid] + red[1][1][tid], s67 = red[2][1][tid] + red[3][1][tid];
        const float lo = s01 + s23, hi = s45 + s67;
        const float t = lo + hi;
        float pv = sc[0] * A.noise[off + L.ob + tid];
        const float bias = base[L.ob + tid] + pv;
        lg[tid] = t + bias;
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0;
        for (int a = 1; a < nact; a++)
            if (lg[a] > lg[best]) best = a;   // tf.argmax: first maximum
        actions[m] = best;
        if (logits_out)
            for (int a = 0; a < nact; a++) logits_out[(size_t)m * nact + a] = lg[a];
    }
}

}  // namespace dne

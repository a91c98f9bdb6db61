// forward_variants.h -- kernels NO default path launches: the A/B library of DESIGN.md sections 4 / 4a.
// They stay compiled, reachable through the knobs named at each and bit-exact against the oracle
// (tests/test_gpu_edges.py runs every one of them), so that every "measured slower" in the design notes can be re-measured with
// tools/ab_inproc.py -- but the product's default evaluation never enters this file.  forward.h holds what it does enter.
//   k_conv1_ref<FPW>   reference-pass conv1, one member per workgroup            DNE_CONV1_SHARED=0   (default: k_conv1_ref_shared<16>)
//   k_fc2<BN, RB>      streaming fc, two antithetic pairs per work item          DNE_FC_DUO=0         (default: k_fc_duo / k_fc_sub)
#pragma once
#include "forward.h"

namespace dne {

// Reference-pass conv1 (policies.py:399: 128 reference frames through every member's perturbed net): one workgroup takes
// FPW consecutive frames of ONE member, so the perturbed weights are formed once per FPW frames, and the next frame's
// pixels are fetched into registers while the matrix cores work on the current one.  Same tiles, same MFMA order as k_conv1.
template <int FPW>
__global__ __launch_bounds__(256) void k_conv1_ref(FwdArgs A, int F, int member0, const uint8_t *__restrict__ ref,
                                                   float *__restrict__ y1 /*[n_local * F][441][16]*/,
                                                   float *__restrict__ fr /*[n_local * F][2][16] per-frame moments*/) {
    __shared__ float lut[256];
    __shared__ uint32_t img[88 * 88];
    __shared__ float wsum[4][2][16];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, ci = lane >> 4;
    const int gpm = F / FPW;                         // frame groups per member
    const int mloc = blockIdx.x / gpm, f0 = (blockIdx.x % gpm) * FPW;
    const int member = member0 + mloc;
    const float *base = A.bases + (size_t)A.m_slot[member] * A.base_stride + A.L.c1w;
    const float *eps = A.noise + A.m_off[member] + A.L.c1w;
    const float sc = A.m_scale[member];
    uint32_t px[28];
    auto fetch = [&](int f) {
        const uint32_t *ob = (const uint32_t *)(ref + (size_t)f * OB_BYTES);
#pragma unroll
        for (int j = 0; j < 28; j++) {
            const int e = tid + 256 * j;
            px[j] = e < 7056 ? ob[e] : 0u;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < 28; j++) {
            const int e = tid + 256 * j;
            if (e < 7056) img[(e / 84 + 2) * 88 + e % 84 + 2] = px[j];
        }
    };
    fetch(f0);
    float b[64];
#pragma unroll
    for (int kk = 0; kk < 64; kk++) {
        float v = sc * eps[64 * kk + lane];
        b[kk] = base[64 * kk + lane] + v;
    }
    float pb = sc * eps[4096 + lp];
    const float bias = base[4096 + lp] + pb;
    lut[tid] = (float)tid / 255.0f;
    for (int i = tid; i < 688; i += 256) {           // the 2-pixel zero border, once
        int r, c;
        if (i < 352) { r = i / 88; r = r < 2 ? r : 84 + r; c = i % 88; }
        else { const int j = i - 352; r = 2 + j / 4; c = j % 4; c = c < 2 ? c : 84 + c; }
        img[r * 88 + c] = 0u;
    }
    stage();
    __syncthreads();
    for (int fi = 0; fi < FPW; fi++) {
        if (fi + 1 < FPW) fetch(f0 + fi + 1);        // in flight under the MFMAs below
        float *out = y1 + ((size_t)mloc * F + f0 + fi) * 7056;
        float Ws = 0.0f, Wq = 0.0f;                  // this wave's tiles (wv, wv + 4, ...) of this frame, in tile order
        auto run = [&](int j, auto has_b) {
            constexpr bool HASB = decltype(has_b)::value;
            const int tA = wv + 4 * j, tB = wv + 4 * (j + 1);
            const int pA = min(tA * 16 + lp, 440), pB = HASB ? min(tB * 16 + lp, 440) : 0;
            const int oA = (pA / 21) * 4 * 88 + (pA % 21) * 4, oB = (pB / 21) * 4 * 88 + (pB % 21) * 4;
            f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 8; kh++) {
#pragma unroll
                for (int kw = 0; kw < 8; kw++) {
                    const float xA = lut[(img[oA + kh * 88 + kw] >> (8 * ci)) & 255u];
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xA, b[kh * 8 + kw], accA, 0, 0, 0);
                    if (HASB) {
                        const float xB = lut[(img[oB + kh * 88 + kw] >> (8 * ci)) & 255u];
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xB, b[kh * 8 + kw], accB, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int posA = tA * 16 + ci * 4 + r, posB = tB * 16 + ci * 4 + r;
                if (posA < 441) out[posA * 16 + lp] = accA[r] + bias;
                if (HASB && posB < 441) out[posB * 16 + lp] = accB[r] + bias;
            }
            tile_moments(accA, tA * 16 + ci * 4, 441, Ws, Wq);
            if (HASB) tile_moments(accB, tB * 16 + ci * 4, 441, Ws, Wq);
        };
        run(0, std::true_type{});
        run(2, std::true_type{});
        run(4, std::true_type{});
        run(6, std::false_type{});
        if (ci == 0) { wsum[wv][0][lp] = Ws; wsum[wv][1][lp] = Wq; }
        __syncthreads();                             // every wave is done reading this frame; its moments are in LDS
        if (tid < 32) {                              // frame moments: (W0 + W1) + (W2 + W3) per channel
            const int k = tid >> 4, c = tid & 15;
            const float lo = wsum[0][k][c] + wsum[1][k][c], hi = wsum[2][k][c] + wsum[3][k][c];
            fr[(((size_t)mloc * F + f0 + fi) * 2 + k) * 16 + c] = lo + hi;
        }
        if (fi + 1 < FPW) {
            stage();
            __syncthreads();
        }
    }
}

// Two antithetic pairs per work item (ES only: every member perturbs the same base vector).  The base rows of the
// fc matrix are loaded once and combined with both pairs' noise rows, so a quarter of the bytes that k_fc<2> pulls
// through L1 / L2 per pair disappears (3 row loads per 2 pairs instead of 4) -- at full width that traffic, not HBM,
// is what keeps k_fc<2> below the read-stream ceiling.  Per member the arithmetic (k-ordered fmaf chain per slice,
// ((s0+s1)+(s2+s3)) + bias, bn3, output layer, first-max argmax) is exactly k_fc's.
template <bool HAS_BN, int RB>
__global__ __launch_bounds__(256) void k_fc2(FwdArgs A, const int *__restrict__ list, int n_groups,
                                             const float *__restrict__ y2, float *__restrict__ y3,
                                             int32_t *__restrict__ actions) {
    constexpr int NM = 4;   // members per item: pair p = members 2p, 2p+1
    __shared__ float part[4][NM][256];
    __shared__ float red[4][NM][OUT_NA];
    __shared__ float lg[NM][32];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    __builtin_amdgcn_s_setprio(3);
    const int n_items = (n_groups + 1) >> 1;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int i0 = 2 * item, i1 = min(2 * item + 1, n_groups - 1);
    int member[NM];
    float scale[NM];
    int nm;
    {
        int g0 = list ? list[i0] : i0, g1 = list ? list[i1] : i1;
        if (A.done) {   // finished pairs still in the list (compaction runs every 32 lock-steps) are not streamed
            const bool d0 = A.done[2 * g0] && A.done[2 * g0 + 1], d1 = A.done[2 * g1] && A.done[2 * g1 + 1];
            if (d0 && d1) continue;
            if (d0) g0 = g1;
            if (d1) g1 = g0;
        }
        nm = g0 != g1 ? NM : 2;   // odd count or a finished partner: the item repeats its pair (same addresses) and writes it once
        member[0] = 2 * g0; member[1] = 2 * g0 + 1; member[2] = 2 * g1; member[3] = 2 * g1 + 1;
    }
#pragma unroll
    for (int v = 0; v < NM; v++) scale[v] = A.m_scale[member[v]];
    const int64_t off[2] = {A.m_off[member[0]], A.m_off[member[2]]};
    const float *base = A.bases + (size_t)A.m_slot[member[0]] * A.base_stride;
    const float *eps0 = A.noise + off[0] + L.fcw + lane * 4, *eps1 = A.noise + off[1] + L.fcw + lane * 4;
    const float *th = base + L.fcw + lane * 4;

    const int ch = (8 * wv + lane) & 31;   // bn2 channel of this lane's activation rows (968 = 8 mod 32)
    float s2[NM], h2[NM];
#pragma unroll
    for (int v = 0; v < NM; v++) {
        s2[v] = HAS_BN ? A.bn[(size_t)member[v] * 608 + 32 + ch] : 1.0f;
        h2[v] = HAS_BN ? A.bn[(size_t)member[v] * 608 + 64 + ch] : 0.0f;
    }
    float acc[NM][4];
#pragma unroll
    for (int v = 0; v < NM; v++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[v][e] = 0.0f;

    const int kbeg = 968 * wv;
    auto load_x = [&](int c, float (&dst)[NM]) {
        const int nr = c < 15 ? 64 : 8;
#pragma unroll
        for (int v = 0; v < NM; v++) {
            float t = 0.0f;
            if (c < 16 && lane < nr) {
                t = y2[(size_t)member[v] * 3872 + kbeg + 64 * c + lane];
                if (HAS_BN) {
                    t = t * s2[v];
                    t = t + h2[v];
                }
                t = t > 0.0f ? t : 0.0f;
            }
            dst[v] = t;
        }
    };
    f4u e0_cur[RB], e0_nxt[RB], e1_cur[RB], e1_nxt[RB];
    f4a t_cur[RB], t_nxt[RB];
    float xv[NM], xn[NM];
    load_x(0, xv);
    load_x(1, xn);
#pragma unroll
    for (int i = 0; i < RB; i++) {
        const size_t ro = (size_t)(kbeg + i) * 256;
        e0_cur[i] = *(const f4u *)(eps0 + ro);
        e1_cur[i] = *(const f4u *)(eps1 + ro);
        t_cur[i] = *(const f4a *)(th + ro);
    }
    constexpr int NB = 968 / RB, BPC = 64 / RB;
    float fold[NM][4];                            // the quarter's running left fold over its sub-slices
#pragma unroll
    for (int v = 0; v < NM; v++)
#pragma unroll
        for (int q = 0; q < 4; q++) fold[v][q] = 0.0f;
    int next_sub = FC_SUB0;
    for (int bt = 0; bt < NB; bt++) {
        if (bt + 1 < NB) {
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const size_t ro = (size_t)(kbeg + (bt + 1) * RB + i) * 256;
                e0_nxt[i] = *(const f4u *)(eps0 + ro);
                e1_nxt[i] = *(const f4u *)(eps1 + ro);
                t_nxt[i] = *(const f4a *)(th + ro);
            }
        }
        const int li = (bt % BPC) * RB;
#pragma unroll
        for (int i = 0; i < RB; i++) {
#pragma unroll
            for (int v = 0; v < NM; v++) {
                const float x = lane_bcast(xv[v], li + i);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float pv = scale[v] * (v < 2 ? e0_cur[i][q] : e1_cur[i][q]);
                    float w = t_cur[i][q] + pv;
                    acc[v][q] = __builtin_fmaf(x, w, acc[v][q]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; i++) { e0_cur[i] = e0_nxt[i]; e1_cur[i] = e1_nxt[i]; t_cur[i] = t_nxt[i]; }
        if (bt % BPC == BPC - 1) {
#pragma unroll
            for (int v = 0; v < NM; v++) xv[v] = xn[v];
            load_x(bt / BPC + 2, xn);
        }
        if ((bt + 1) * RB == next_sub) {   // end of a sub-slice (k_fc)
#pragma unroll
            for (int v = 0; v < NM; v++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    fold[v][q] = next_sub == FC_SUB0 ? acc[v][q] : fold[v][q] + acc[v][q];
                    acc[v][q] = 0.0f;
                }
            next_sub += FC_SUBN;
        }
    }
#pragma unroll
    for (int v = 0; v < NM; v++)
#pragma unroll
        for (int q = 0; q < 4; q++) part[wv][v][lane * 4 + q] = fold[v][q];
    __syncthreads();
    const int nact = L.nact;
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {   // the item's two pairs: each shares one noise slice, all four members the base vector
        if (2 * pr < nm) {
            float x3[2], sc2[2] = {scale[2 * pr], scale[2 * pr + 1]};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int v = 2 * pr + u, j = tid;
                const float s01 = part[0][v][j] + part[1][v][j];
                const float s23 = part[2][v][j] + part[3][v][j];
                float s = s01 + s23;
                float pv = scale[v] * A.noise[off[pr] + L.fcb + j];
                const float bias = base[L.fcb + j] + pv;
                s = s + bias;
                y3[(size_t)member[v] * 256 + j] = s;
                float t = s;
                if (HAS_BN) {
                    t = t * A.bn[(size_t)member[v] * 608 + 96 + j];
                    t = t + A.bn[(size_t)member[v] * 608 + 352 + j];
                }
                x3[u] = t > 0.0f ? t : 0.0f;
            }
            float p[2][OUT_NA];
            out_products<2>(p, x3, base + L.ow + tid * nact, A.noise + off[pr] + L.ow + tid * nact, sc2, nact);
            out_wave_sums<2>(p, nact, red[wv] + 2 * pr, lane);
        }
    }
    __syncthreads();
    if (tid < nm * nact) {
        const int v = tid / nact, a = tid % nact;
        const float s01 = red[0][v][a] + red[1][v][a];
        const float s23 = red[2][v][a] + red[3][v][a];
        const float t = s01 + s23;
        const int mv = (v >> 1 ? member[2] : member[0]) + (v & 1);
        float pv = A.m_scale[mv] * A.noise[(v >> 1 ? off[1] : off[0]) + L.ob + a];
        const float bias = base[L.ob + a] + pv;
        lg[v][a] = t + bias;
    }
    __syncthreads();
    if (tid < nm) {
        const int v = tid;
        int best = 0;
        for (int a = 1; a < nact; a++)
            if (lg[v][a] > lg[v][best]) best = a;   // tf.argmax: first maximum
        actions[(v >> 1 ? member[2] : member[0]) + (v & 1)] = best;
    }
    __syncthreads();   // LDS is reused by the next item
    }
}

}  // namespace dne

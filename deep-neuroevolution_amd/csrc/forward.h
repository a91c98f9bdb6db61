// forward.h -- batched Nature-CNN policy forward with member-unique, never-materialised weights.
//
// Replaces ESAtariPolicy._make_net / GAAtariPolicy._make_net + act (es_distributed/policies.py:319-330,
// 374-375, 449-459, 469-470) and the per-pair perturbation + 4 MB feed of es.py:412-419 /
// tf_util.py:224-240: every weight is formed in registers as  base[p] + scale * noise[off + p]
// (two fp32 roundings, exactly params + noise_stdev*noise.get(...)) while it streams from the
// SharedNoiseTable device buffer, so an antithetic pair reads its 4 MB noise slice ONCE per env-step.
//
// Numerics contract (identical to the CPU oracle): each dot product of the convolutions is an fp32 fmaf chain in
// (kh, kw, ci) order starting at 0; fc (oracle fc_raw) = 4 quarters of 968 rows, a quarter = the left fold of the 8 sub-slice
// chains of 128, 120 x 7 rows, quarters combined ((q0+q1)+(q2+q3)) + bias; the output layer is a fixed binary tree
// (out_raw_k); batch-norm is x*scale + shift with two roundings.  Built with -ffp-contract=off.
//
// Work decode shared by all kernels: step mode (F == 1) walks a list of active groups (ES: antithetic
// pairs, GA: single members) and reads each member's own frame stack; reference mode (F > 1) runs F
// reference frames through every member of a chunk (virtual batch norm, policies.py:399).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace dne {

struct Layout {
    int kind, nact, P;
    int c1w, c1b, bn1b, bn1g, c2w, c2b, bn2b, bn2g, fcw, fcb, bn3b, bn3g, ow, ob;
    int c3w, c3b;   // LargeModel only
};

// Tail of a generation (at most TT_MAX members left): a lock-step is a chain of short dependent launches, and what each of them
// does first -- list -> member -> (noise offset, base slot, scale) -> first weight address -- is two dependent memory round trips
// before the first useful load can be issued.  The host knows all of it (the active list comes back with the active count at
// every compaction), so it rides in the kernel arguments: position in the window -> member and descriptors, read with scalar
// loads from the argument segment.  n = 0: the table is not in use and the kernels read list / m_off / m_slot / m_scale.
constexpr int TT_MAX = 48;
struct TailTable {
    int n;
    int member[TT_MAX];
    int slot[TT_MAX];
    float scale[TT_MAX];
    long long off[TT_MAX];
};

struct FwdArgs {
    const float *noise;
    const float *bases;      // [slots][base_stride]
    size_t base_stride;
    const int32_t *m_slot;   // per member
    const int64_t *m_off;
    const float *m_scale;
    float *bn;               // [members][608]: s1[16] h1[16] s2[32] h2[32] s3[256] h3[256]
    float *bn_mom;           // [members][608]: the batch moments behind them, mean / variance in the same layout (snapshots)
    const int32_t *done;     // per member, step mode only
    int sub_sums;            // the fc left 32 chain sums per column (k_fc_sub: y3s[member][32][256]) instead of 4 quarter sums: the head folds them
    Layout L;
    TailTable tt;
};

constexpr int OB_BYTES = 84 * 84 * 4;
// fc sub-slices within a quarter of 968 rows (oracle ORC_FC_SUB): the first has 128 rows, the other seven 120 -- every boundary is
// a multiple of 8 rows.  A kernel that walks a quarter row by row keeps the running left fold T: at a boundary T (+)= chain, chain = 0.
constexpr int FC_SUB0 = 128, FC_SUBN = 120;
constexpr int FCREF4_LDS = 16 * 256 * 16;   // k_fc_ref<4>: the running fold over the sub-slices, [16 accumulators][256 threads] 16-byte words
// the fc output of column `col` before the bias, from what the fc kernels leave behind: 4 quarter sums [member][4][256], or -- behind
// k_fc_sub -- the 32 chain sums [member][32][256], a quarter being the LEFT FOLD of its 8 (oracle fc_raw); then (q0 + q1) + (q2 + q3)
__device__ __forceinline__ float fc_combine(const float *__restrict__ sums, int member, int col, bool sub) {
    float q[4];
    if (sub) {
        const float *p = sums + (size_t)member * 32 * 256 + col;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float f = p[(8 * k) * 256];
#pragma unroll
            for (int i = 1; i < 8; i++) f = f + p[(8 * k + i) * 256];
            q[k] = f;
        }
    } else {
        const float *p = sums + (size_t)member * 4 * 256 + col;
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = p[k * 256];
    }
    const float s01 = q[0] + q[1];
    const float s23 = q[2] + q[3];
    return s01 + s23;
}
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment
typedef float f4a __attribute__((ext_vector_type(4)));

// f(integral_constant<int, 0>{}) ... f(integral_constant<int, N - 1>{}): a loop whose index is a constant expression in the body
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct Item {
    int member, row;
    int pos;                 // index into the tail table, -1 when the descriptors come from memory
    const uint8_t *ob;
    bool skip;
};

// a member's base vector / noise slice / scale: from the tail table (TT: scalar loads from the kernel arguments) or from memory.
// TT is a compile-time choice -- a kernel branches once, on A.tt.n, into the body it needs -- so that the table form keeps its
// addresses in scalar registers and neither form pays for the other.
template <bool TT>
__device__ __forceinline__ const float *item_base(const FwdArgs &A, const Item &it) {
    const int slot = TT ? A.tt.slot[it.pos] : A.m_slot[it.member];
    return A.bases + (size_t)slot * A.base_stride;
}
template <bool TT>
__device__ __forceinline__ const float *item_eps(const FwdArgs &A, const Item &it) {
    const long long off = TT ? A.tt.off[it.pos] : (long long)A.m_off[it.member];
    return A.noise + off;
}
template <bool TT>
__device__ __forceinline__ float item_scale(const FwdArgs &A, const Item &it) {
    return TT ? A.tt.scale[it.pos] : A.m_scale[it.member];
}
// member at position b of the window (b = group index * gsize + member within the group)
template <bool TT>
__device__ __forceinline__ int window_member(const FwdArgs &A, const int *__restrict__ list, int gsize, int b) {
    if (TT) return A.tt.member[b];
    const int g = list ? list[b / gsize] : b / gsize;
    return g * gsize + b % gsize;
}

template <bool TT = false>
__device__ __forceinline__ Item decode_item(const FwdArgs &A, int b, const int *__restrict__ list, int gsize, int F, int member0,
                                            const uint8_t *__restrict__ stacks, const uint8_t *__restrict__ ref,
                                            const int32_t *__restrict__ done) {
    Item it;
    it.pos = -1;
    if (F == 1) {
        it.member = window_member<TT>(A, list, gsize, b);
        if (TT) it.pos = b;
        it.row = it.member;
        it.ob = stacks + (size_t)it.member * OB_BYTES;
        it.skip = done && done[it.member];
    } else {
        it.member = member0 + b / F;
        it.row = b;
        it.ob = ref + (size_t)(b % F) * OB_BYTES;
        it.skip = false;
    }
    return it;
}

// ------------------------------------------------------------------------------------------ conv1
// 8x8 stride 4 SAME(2,2) over the u8 [84][84][4] stack; /255 fused into the load through a 256-entry
// table (exactly float32(u8)/255.0, atari_wrappers.py:186).  y1[row][441][16] raw (pre-BN).
// GEMM view: [441 positions] x [256 = (kh,kw,ci)] x [16 co] on v_mfma_f32_16x16x4_f32, which is bitwise a
// k-ordered fp32 fmaf chain -- the oracle's order.  The 4 k-values of one MFMA are the 4 stacked frames
// (ci) of one tap, so lane (l>>4) extracts byte (l>>4) of the tap's dword.  Each wave keeps the whole
// perturbed weight set as B fragments in 64 VGPRs and walks 7 of the 28 position tiles, two at a time
// (two independent accumulators cover the 40-cycle dependent-MFMA latency).
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv1Lds {
    float lut[256];
    uint32_t img[88 * 88];
};

// one member-frame (or 1/4, 1/7 of its position tiles when nsplit = 4 / 7: few members left, several workgroups share one
// member's 28 tiles to cut the latency); called by all 256 threads of a workgroup
// CO = output channels of the layer (16; 32 for the LargeModel, whose two 16-channel halves are two calls with half = 0 / 1):
// the weights are [kh][kw][ci][CO], the output rows [441][CO].
template <int CO = 16, bool TT = false>
__device__ __forceinline__ void conv1_body(Conv1Lds &S, const FwdArgs &A, const Item &it, float *__restrict__ y1, int part, int nsplit,
                                           int half = 0) {
    float (&lut)[256] = S.lut;
    uint32_t (&img)[88 * 88] = S.img;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, ci = lane >> 4;
    const float *base = item_base<TT>(A, it) + A.L.c1w;
    const float *eps = item_eps<TT>(A, it) + A.L.c1w;
    const float sc = item_scale<TT>(A, it);
    const int wl = CO == 16 ? lane : ci * CO + half * 16 + lp;   // this lane's weight within a tap's [ci][CO] block
    float b[64];
#pragma unroll
    for (int kk = 0; kk < 64; kk++) {
        float v = sc * eps[4 * CO * kk + wl];
        b[kk] = base[4 * CO * kk + wl] + v;
    }
    float pb = sc * eps[256 * CO + half * 16 + lp];
    const float bias = base[256 * CO + half * 16 + lp] + pb;
    lut[tid] = (float)tid / 255.0f;
    {   // stage the frame stack: all 28 loads of a thread are issued before the first LDS write (a rolled loop
        // would pay one L2 round trip per element); the 2-pixel zero border is written separately
        uint32_t px[28];
#pragma unroll
        for (int j = 0; j < 28; j++) {
            const int e = tid + 256 * j;
            px[j] = e < 7056 ? ((const uint32_t *)it.ob)[e] : 0u;
        }
        for (int i = tid; i < 688; i += 256) {
            int r, c;
            if (i < 352) { r = i / 88; r = r < 2 ? r : 84 + r; c = i % 88; }
            else { const int j = i - 352; r = 2 + j / 4; c = j % 4; c = c < 2 ? c : 84 + c; }
            img[r * 88 + c] = 0u;
        }
#pragma unroll
        for (int j = 0; j < 28; j++) {
            const int e = tid + 256 * j;
            if (e < 7056) img[(e / 84 + 2) * 88 + e % 84 + 2] = px[j];
        }
    }
    __syncthreads();
    float *out = y1 + (size_t)it.row * (441 * CO) + half * 16;
    // 28 position tiles, 7 per wave: three pairs (two independent accumulators cover the dependent-MFMA latency) and
    // one single tile -- the single one runs on one accumulator instead of dragging an empty partner through the pipe
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
        for (int r = 0; r < 4; r++) {   // D[row = 4*(l>>4) + r][col = l&15]
            const int posA = tA * 16 + ci * 4 + r, posB = tB * 16 + ci * 4 + r;
            if (posA < 441) out[posA * CO + lp] = accA[r] + bias;
            if (HASB && posB < 441) out[posB * CO + lp] = accB[r] + bias;
        }
    };
    if (nsplit == 7) {   // the last handful of members: one tile per wave, seven workgroups per member
        run(part, std::false_type{});
        return;
    }
    for (int j = nsplit == 4 ? 2 * part : 0; j < (nsplit == 4 ? 2 * part + 2 : 8); j += 2) {
        if (j + 1 < 7) run(j, std::true_type{});
        else run(j, std::false_type{});
    }
}

__global__ __launch_bounds__(256) void k_conv1(FwdArgs A, const int *__restrict__ list, int gsize, int F, int member0,
                                               const uint8_t *__restrict__ stacks, const uint8_t *__restrict__ ref,
                                               float *__restrict__ y1, int nsplit) {
    __shared__ Conv1Lds S;
    if (A.tt.n > 0) {
        const Item it = decode_item<true>(A, blockIdx.x / nsplit, list, gsize, F, member0, stacks, ref, A.done);
        if (it.skip) return;
        conv1_body<16, true>(S, A, it, y1, blockIdx.x % nsplit, nsplit);
    } else {
        const Item it = decode_item(A, blockIdx.x / nsplit, list, gsize, F, member0, stacks, ref, A.done);
        if (it.skip) return;
        conv1_body(S, A, it, y1, blockIdx.x % nsplit, nsplit);
    }
}

// Batch-norm moments of the reference pass, formed in the convolution epilogue (the oracle's bn_finish_tiles order): one
// 16-position MFMA tile gives, per output channel, T = (s_0 + s_1) + (s_2 + s_3) over its four row groups, with
// s_g = ((a0 + a1) + a2) + a3 of the PRE-BIAS accumulators (positions past the layer's end count as zeros) and the same
// tree over q_g = fma(a3,a3, fma(a2,a2, fma(a1,a1, a0*a0))).  Lane (col = l & 15, g = l >> 4) holds rows 4g..4g+3 of
// column col; the row groups are combined with two xor-shuffles (fp add is commutative, so every lane ends with T).
__device__ __forceinline__ void tile_moments(const f32x4 &acc, int first_pos, int npos, float &Ws, float &Wq) {
    float a[4];
#pragma unroll
    for (int r = 0; r < 4; r++) a[r] = first_pos + r < npos ? acc[r] : 0.0f;
    float s = a[0] + a[1];
    s = s + a[2];
    s = s + a[3];
    float q = a[0] * a[0];
    q = __builtin_fmaf(a[1], a[1], q);
    q = __builtin_fmaf(a[2], a[2], q);
    q = __builtin_fmaf(a[3], a[3], q);
    const float s1 = s + __shfl_xor(s, 16), q1 = q + __shfl_xor(q, 16);
    const float Ts = s1 + __shfl_xor(s1, 32), Tq = q1 + __shfl_xor(q1, 32);
    Ws = Ws + Ts;
    Wq = Wq + Tq;
}


// Reference-pass conv1, shared-image form (round 2).  The 128 reference frames are the same for every member, so eight members
// (one wave each, its perturbed weights in 64 VGPRs for the whole workgroup) share ONE frame at a time in LDS -- stored as
// floats, planar per channel, already divided by 255 (k_ref_to_float, once per dne_set_ref_batch; the same (float)u8 / 255.0f as
// the table of k_conv1).  A lane's operands for four consecutive taps are then one 16-byte LDS read and no table lookup, no byte
// extraction: 0.25 LDS instructions per MFMA instead of 2.  Same taps in the same order, same tile sums, same bits as
// k_conv1_ref.  The next frame is fetched into registers under the MFMAs of the current one.
constexpr int RF_W = 88, RF_PLANE = 88 * 88, RF_FRAME = 4 * RF_PLANE;   // padded float frame: [4 channels][88][88]

__global__ __launch_bounds__(256) void k_ref_to_float(const uint8_t *__restrict__ ref, int F, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F * RF_FRAME) return;
    const int f = i / RF_FRAME, c = (i / RF_PLANE) % 4, r = (i / RF_W) % 88, x = i % RF_W;
    float v = 0.0f;
    if (r >= 2 && r < 86 && x >= 2 && x < 86) v = (float)ref[(size_t)f * OB_BYTES + ((r - 2) * 84 + (x - 2)) * 4 + c] / 255.0f;
    out[i] = v;
}

template <int FPW>
__global__ __launch_bounds__(512) void k_conv1_ref_shared(FwdArgs A, int F, int member0, int n_local, const float *__restrict__ reff,
                                                          float *__restrict__ y1 /*[n_local * F][441][16]*/,
                                                          float *__restrict__ fr /*[n_local * F][2][16]*/) {
    extern __shared__ __attribute__((aligned(16))) float imgf[];   // [4][88][88]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, ci = lane >> 4;
    const int gpm = F / FPW;
    const int mloc = (blockIdx.x / gpm) * 8 + wv, f0 = (blockIdx.x % gpm) * FPW;
    const bool live = mloc < n_local;
    const int member = member0 + (live ? mloc : 0);
    const float *base = A.bases + (size_t)A.m_slot[member] * A.base_stride + A.L.c1w;
    const float *eps = A.noise + A.m_off[member] + A.L.c1w;
    const float sc = A.m_scale[member];
    constexpr int NV4 = RF_FRAME / 4 / 512 + 1;   // 16-byte words of a frame per thread (7744 = 15.1 x 512)
    f32x4 pf[NV4];
    auto fetch = [&](int f) {
        const f32x4 *src = (const f32x4 *)(reff + (size_t)f * RF_FRAME);
#pragma unroll
        for (int j = 0; j < NV4; j++) {
            const int e = tid + 512 * j;
            pf[j] = e < RF_FRAME / 4 ? src[e] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NV4; j++) {
            const int e = tid + 512 * j;
            if (e < RF_FRAME / 4) ((f32x4 *)imgf)[e] = pf[j];
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
    stage();
    __syncthreads();
    const float *plane = imgf + ci * RF_PLANE;
    for (int fi = 0; fi < FPW; fi++) {
        if (fi + 1 < FPW) fetch(f0 + fi + 1);        // in flight under the MFMAs below
        float *out = y1 + ((size_t)mloc * F + f0 + fi) * 7056;
        float Ws[4] = {0.f, 0.f, 0.f, 0.f}, Wq[4] = {0.f, 0.f, 0.f, 0.f};   // tile t adds to group t % 4, in tile order (bn_finish_tiles)
        if (live) {
            auto round2 = [&](int tA, float &WsA, float &WqA, float &WsB, float &WqB) {   // tiles tA, tA + 1 on two accumulators
                const int tB = tA + 1;
                const int pA = min(tA * 16 + lp, 440), pB = min(tB * 16 + lp, 440);
                const float *qA = plane + (pA / 21) * 4 * RF_W + (pA % 21) * 4, *qB = plane + (pB / 21) * 4 * RF_W + (pB % 21) * 4;
                f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kh = 0; kh < 8; kh++) {
                    const f32x4 a0 = *(const f32x4 *)(qA + kh * RF_W), a1 = *(const f32x4 *)(qA + kh * RF_W + 4);
                    const f32x4 c0 = *(const f32x4 *)(qB + kh * RF_W), c1 = *(const f32x4 *)(qB + kh * RF_W + 4);
#pragma unroll
                    for (int kw = 0; kw < 4; kw++) {
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kw], b[kh * 8 + kw], accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[kw], b[kh * 8 + kw], accB, 0, 0, 0);
                    }
#pragma unroll
                    for (int kw = 0; kw < 4; kw++) {
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kw], b[kh * 8 + 4 + kw], accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[kw], b[kh * 8 + 4 + kw], accB, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int posA = tA * 16 + ci * 4 + r, posB = tB * 16 + ci * 4 + r;
                    if (posA < 441) out[posA * 16 + lp] = accA[r] + bias;
                    if (posB < 441) out[posB * 16 + lp] = accB[r] + bias;
                }
                tile_moments(accA, tA * 16 + ci * 4, 441, WsA, WqA);
                tile_moments(accB, tB * 16 + ci * 4, 441, WsB, WqB);
            };
#pragma unroll 1
            for (int r4 = 0; r4 < 7; r4++) {         // tiles 4 r4 .. 4 r4 + 3: one tile for each of the four groups, in tile order
                round2(4 * r4, Ws[0], Wq[0], Ws[1], Wq[1]);
                round2(4 * r4 + 2, Ws[2], Wq[2], Ws[3], Wq[3]);
            }
            if (ci == 0) {   // frame moments: (W0 + W1) + (W2 + W3) per channel
                const float slo = Ws[0] + Ws[1], shi = Ws[2] + Ws[3], qlo = Wq[0] + Wq[1], qhi = Wq[2] + Wq[3];
                float *dst = fr + ((size_t)mloc * F + f0 + fi) * 2 * 16;
                dst[lp] = slo + shi;
                dst[16 + lp] = qlo + qhi;
            }
        }
        if (fi + 1 < FPW) {
            __syncthreads();                         // every wave is done reading this frame
            stage();
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------ conv2
// 4x4 stride 2 SAME(1,2); input = relu(bn1(y1)) formed while staging; y2[row][121][32] raw.
// GEMM view [121 -> 128 positions] x [256 = (kh,kw,ci)] x [32 co] on the same fp32 MFMA: wave w owns
// co tile (w & 1) and position tiles 4*(w>>1) .. +3 (4 independent accumulators); its B fragments (the
// perturbed weights of 16 output channels) live in 64 VGPRs, A comes from the padded activation image in LDS.
constexpr int C2_PS = 17;   // LDS pixel stride (16 channels + 1 pad)
constexpr int C2_RW = 27;   // LDS row width in pixels (24 used): 2 * RW * PS = 22 mod 32 continues the 2-per-position bank
                            // sequence across output rows, so the 16 positions of an MFMA tile never share a bank
struct Conv2Lds {
    float a_s[24 * C2_RW * C2_PS];
    float wsum[4][2][16];
};

// nsplit = 2 / 4: two / four workgroups share one member's position tiles
template <bool HAS_BN, bool TT = false>
__device__ __forceinline__ void conv2_body(Conv2Lds &S, const FwdArgs &A, const Item &it, const float *__restrict__ y1,
                                           float *__restrict__ y2, int part, int nsplit, float *__restrict__ fr,
                                           const float *__restrict__ y1_row = nullptr /* the member's conv1 output when it is not row it.row of y1 */) {
    constexpr int PS = C2_PS, RW = C2_RW;
    float (&a_s)[24 * C2_RW * C2_PS] = S.a_s;
    float (&wsum)[4][2][16] = S.wsum;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, lk = lane >> 4;
    const int nt = wv & 1, mt0 = 4 * (wv >> 1);
    const float *base = item_base<TT>(A, it) + A.L.c2w;
    const float *eps = item_eps<TT>(A, it) + A.L.c2w;
    const float sc = item_scale<TT>(A, it);
    const float *bn = A.bn + (size_t)it.member * 608;
    const float *src = y1_row ? y1_row : y1 + (size_t)it.row * 7056;
    float yv[28];   // all of this thread's activation loads in flight at once (its channel is tid & 15 throughout)
#pragma unroll
    for (int j = 0; j < 28; j++) {
        const int e = tid + 256 * j;
        yv[j] = e < 7056 ? src[e] : 0.0f;
    }
    const float s1 = HAS_BN ? bn[tid & 15] : 1.0f, h1 = HAS_BN ? bn[16 + (tid & 15)] : 0.0f;
    float b[64];
#pragma unroll
    for (int kk = 0; kk < 64; kk++) {
        const int o = (4 * kk + lk) * 32 + nt * 16 + lp;
        float v = sc * eps[o];
        b[kk] = base[o] + v;
    }
    float pb = sc * eps[8192 + nt * 16 + lp];
    const float bias = base[8192 + nt * 16 + lp] + pb;
    for (int pix = tid; pix < 24 * 24; pix += 256) {   // zero only the SAME-padding ring (disjoint from the fill)
        const int y = pix / 24, x = pix % 24;
        if (y < 1 || y > 21 || x < 1 || x > 21)
#pragma unroll
            for (int c = 0; c < 16; c++) a_s[(y * RW + x) * PS + c] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 28; j++) {
        const int e = tid + 256 * j;
        if (e < 7056) {
            const int c = e & 15, pix = e >> 4;
            float t = yv[j];
            if (HAS_BN) {
                t = t * s1;
                t = t + h1;
            }
            t = t > 0.0f ? t : 0.0f;
            a_s[((pix / 21 + 1) * RW + pix % 21 + 1) * PS + c] = t;
        }
    }
    __syncthreads();
    auto run = [&](auto ntl_) {
        constexpr int NTL = decltype(ntl_)::value;                   // position tiles per wave: 4, or 2 / 1 when split
        const int mtb = mt0 + (NTL == 2 ? 2 * part : NTL == 1 ? part : 0);
        int off[NTL];
        f32x4 acc[NTL];
#pragma unroll
        for (int m = 0; m < NTL; m++) {
            const int p = min((mtb + m) * 16 + lp, 120);
            off[m] = ((p / 11) * 2 * RW + (p % 11) * 2) * PS + lk;
            acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kh = 0; kh < 4; kh++) {
#pragma unroll
            for (int kw = 0; kw < 4; kw++) {
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) {   // k = (kh*4+kw)*16 + c4*4 + (l>>4)
                    const int kk = (kh * 4 + kw) * 4 + c4;
#pragma unroll
                    for (int m = 0; m < NTL; m++) {
                        const float x = a_s[off[m] + (kh * RW + kw) * PS + c4 * 4];
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[kk], acc[m], 0, 0, 0);
                    }
                }
            }
        }
        float *o = y2 + (size_t)it.row * 3872;
#pragma unroll
        for (int m = 0; m < NTL; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {   // D[row = 4*(l>>4) + r][col = l&15]
                const int pos = (mtb + m) * 16 + lk * 4 + r;
                if (pos < 121) o[pos * 32 + nt * 16 + lp] = acc[m][r] + bias;
            }
        if (NTL == 4 && fr) {   // reference pass: this wave's four tiles in tile order, then (tiles 0-3) + (tiles 4-7)
            float Ws = 0.0f, Wq = 0.0f;
#pragma unroll
            for (int m = 0; m < NTL; m++) tile_moments(acc[m], (mtb + m) * 16 + lk * 4, 121, Ws, Wq);
            if (lk == 0) { wsum[wv][0][lp] = Ws; wsum[wv][1][lp] = Wq; }
            __syncthreads();
            if (tid < 64) {
                const int k = tid >> 5, c = tid & 31, h = c >> 4, l = c & 15;
                fr[((size_t)it.row * 2 + k) * 32 + c] = wsum[h][k][l] + wsum[h + 2][k][l];
            }
        }
    };
    if (nsplit == 4) run(std::integral_constant<int, 1>{});
    else if (nsplit == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 4>{});
}

template <bool HAS_BN>
__global__ __launch_bounds__(256) void k_conv2(FwdArgs A, const int *__restrict__ list, int gsize, int F, int member0,
                                               const float *__restrict__ y1, float *__restrict__ y2, int nsplit,
                                               float *__restrict__ fr /*reference pass: [rows][2][32] per-frame moments, else null*/) {
    __shared__ Conv2Lds S;
    if (A.tt.n > 0) {
        const Item it = decode_item<true>(A, blockIdx.x / nsplit, list, gsize, F, member0, nullptr, nullptr, A.done);
        if (it.skip) return;
        conv2_body<HAS_BN, true>(S, A, it, y1, y2, blockIdx.x % nsplit, nsplit, fr);
    } else {
        const Item it = decode_item(A, blockIdx.x / nsplit, list, gsize, F, member0, nullptr, nullptr, A.done);
        if (it.skip) return;
        conv2_body<HAS_BN>(S, A, it, y1, y2, blockIdx.x % nsplit, nsplit, fr);
    }
}

// Reference-pass conv2: one workgroup takes FPW consecutive reference frames of ONE member through the layer, so the member's
// perturbed weights (64 B fragments per lane) and bn1 scale / shift are formed once per FPW frames instead of once per frame
// (k_conv2<true> with one frame per workgroup re-read 64 KB of theta + eps for every 28 KB of activations), and the next
// frame's conv1 output is fetched into registers under the MFMAs of the current one.  Same tiles, same MFMA order, same
// moment tree as conv2_body -- same bits.
// Round 4 (the profiling build's phase clock, tools/ref_phase_clock.py): per frame the workgroup spent 4.3 us in its MFMAs, 2.7 us in
// the 180 instructions of the epilogue behind them (16 predicated stores, the tile moments) and 0.8 us staging the next image --
// beside a wave of the CU's other workgroup that streams MFMAs, an instruction outside one's own MFMA stream is issued about once
// per MFMA (32 cycles).  So the epilogue of frame f now travels INSIDE the MFMA stream of frame f + 1 (a copy of the accumulators,
// one basic block: the stores are unconditional -- PAD: y2 rows are 128 positions apart and the seven positions past the layer's
// end land in the padding --, every lane group writes the wave's tile sums; __builtin_amdgcn_sched_group_barrier deals one or two
// of those instructions behind each MFMA), and the staging moves 16 bytes per load and packs its arithmetic.  A schedule: same
// MFMAs in the same order per accumulator, same moments, same bits.
constexpr int Y2_PAD_ROW = 128 * 32;   // floats between the y2 rows of the reference pass when the fc runs on the matrix cores
template <int FPW, bool PAD>
__global__ __launch_bounds__(256, 2) void k_conv2_ref(FwdArgs A, int F, int member0, const float *__restrict__ y1 /*[n_local * F][441][16]*/,
                                                   float *__restrict__ y2 /*[n_local * F][121 (PAD: 128)][32]*/,
                                                   float *__restrict__ fr /*[n_local * F][2][32]*/) {
    // LDS image: pixel stride 20 floats, channel c at float (c % 4) * 4 + c / 4 of its pixel -- the four k-groups a lane feeds to four
    // consecutive MFMAs (channels lk, lk + 4, lk + 8, lk + 12) are then ONE 16-byte read (0.25 operand reads per MFMA instead of 0.5,
    // every tap an immediate offset from one address per position tile), conflict-free: consecutive positions are 40 floats apart,
    // eight of them start on eight different 8-bank groups, the second lane group of a ds_read_b128 cycle sits 4 banks on, and an
    // output row's wrap (2 * 27 * 20 floats) continues the sequence (1080 - 11 * 40 = 10 * 64)
    constexpr int PS = 20, RW = C2_RW, Y2S = PAD ? Y2_PAD_ROW : 3872;
    __shared__ __attribute__((aligned(16))) float a_s[24 * RW * PS];
    // two sets: the loop's frames use set 0 (a barrier lies between every read and the next write); the last frame's epilogue
    // writes set 1, because wave 0 may still be reading frame FPW - 2's sums from set 0 when the other waves get there
    __shared__ float wsum[2][4][2][16];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, lk = lane >> 4;
    const int nt = wv & 1, mt0 = 4 * (wv >> 1);
    const int gpm = F / FPW, mloc = blockIdx.x / gpm, f0 = (blockIdx.x % gpm) * FPW, member = member0 + mloc;
    const float *base = A.bases + (size_t)A.m_slot[member] * A.base_stride + A.L.c2w;
    const float *eps = A.noise + A.m_off[member] + A.L.c2w;
    const float sc = A.m_scale[member];
    const float *bn = A.bn + (size_t)member * 608;
    // staging: thread = four consecutive channels (c0 .. c0 + 3, fixed: 256 % 4 == 0) of pixels (tid >> 2) + 64 j
    f32x4 yv[7];
    auto fetch = [&](int f) {
        const f32x4 *src = (const f32x4 *)(y1 + ((size_t)mloc * F + f) * 7056);
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int e = tid + 256 * j;
            yv[j] = e < 1764 ? src[e] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(f0);
    const int c0 = (tid & 3) * 4;
    float s1[4], h1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { s1[i] = bn[c0 + i]; h1[i] = bn[16 + c0 + i]; }
    float b[64];
#pragma unroll
    for (int kk = 0; kk < 64; kk++) {
        const int o = (4 * kk + lk) * 32 + nt * 16 + lp;
        float v = sc * eps[o];
        b[kk] = base[o] + v;
    }
    float pb = sc * eps[8192 + nt * 16 + lp];
    const float bias = base[8192 + nt * 16 + lp] + pb;
    for (int pix = tid; pix < 24 * 24; pix += 256) {   // the SAME-padding ring, once: the fills never touch it
        const int y = pix / 24, x = pix % 24;
        if (y < 1 || y > 21 || x < 1 || x > 21)
#pragma unroll
            for (int c = 0; c < 16; c++) a_s[(y * RW + x) * PS + c] = 0.0f;   // (any order: all sixteen)
    }
    int off[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int p = min((mt0 + m) * 16 + lp, 120);
        off[m] = ((p / 11) * 2 * RW + (p % 11) * 2) * PS + lk * 4;
    }
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int e = tid + 256 * j;
            if (e < 1764) {
                const int pix = e >> 2;
                float *d = a_s + ((pix / 21 + 1) * RW + pix % 21 + 1) * PS + (c0 >> 2);   // channel c0 + i -> float 4 i + c0 / 4
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float t = yv[j][i] * s1[i];
                    t = t + h1[i];
                    d[4 * i] = t > 0.0f ? t : 0.0f;
                }
            }
        }
    };
    auto mfmas = [&](f32x4 (&acc)[4]) {
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 4; kh++) {
#pragma unroll
            for (int kw = 0; kw < 4; kw++) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const f32x4 x = *(const f32x4 *)(a_s + off[m] + (kh * RW + kw) * PS);
#pragma unroll
                    for (int c4 = 0; c4 < 4; c4++)   // k = (kh*4+kw)*16 + c4*4 + (l>>4), ascending per accumulator
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[c4], b[(kh * 4 + kw) * 4 + c4], acc[m], 0, 0, 0);
                }
            }
        }
    };
    // a frame's results out: y2 rows, and this wave's four tiles' moments in tile order into wsum (every lane group holds the sums)
    auto epilogue = [&](const f32x4 (&acc)[4], size_t row, int set) {
        float *o = y2 + row * Y2S;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {   // D[row = 4*(l>>4) + r][col = l&15]
                const int pos = (mt0 + m) * 16 + lk * 4 + r;
                if (PAD || pos < 121) o[pos * 32 + nt * 16 + lp] = acc[m][r] + bias;
            }
        float Ws = 0.0f, Wq = 0.0f;                  // this wave's four tiles in tile order, then (tiles 0-3) + (tiles 4-7)
#pragma unroll
        for (int m = 0; m < 4; m++) tile_moments(acc[m], (mt0 + m) * 16 + lk * 4, 121, Ws, Wq);
        wsum[set][wv][0][lp] = Ws;
        wsum[set][wv][1][lp] = Wq;
    };
    auto frame_moments = [&](size_t row, int set) {
        if (tid < 64) {
            const int k = tid >> 5, c = tid & 31, h = c >> 4, l = c & 15;
            fr[(row * 2 + k) * 32 + c] = wsum[set][h][k][l] + wsum[set][h + 2][k][l];
        }
    };
    DNE_ACC_DECL;   // profiling build: 0 = prologue + staging, 1 = barrier, 2 = MFMAs (+ the previous frame's epilogue), 3 = barrier + frame moments
    const size_t row0 = (size_t)mloc * F + f0;
    f32x4 acc[4], accp[4];
    for (int fi = 0; fi < FPW; fi++) {
        stage();
        DNE_ACC(0);
        __syncthreads();
        DNE_ACC(1);
        if (fi + 1 < FPW) fetch(f0 + fi + 1);        // in flight under the MFMAs below
        if (fi == 0) {
            mfmas(acc);
        } else {
            // 16 steps = the 16 taps: sixteen MFMAs (four k-groups x four position tiles) behind four 16-byte operand reads that were
            // requested one step earlier (two register sets); behind them two of the 32 pieces of the previous frame's epilogue (pieces
            // 0-15: a store; 16-31: a quarter of a tile's moments -- the exchange requested in one piece is read in the next).  One
            // fence per step.
            const size_t prow = row0 + fi - 1;
            float *o = y2 + prow * Y2S;
            float Ws = 0.0f, Wq = 0.0f;
            float ts[2] = {0.f, 0.f}, tq[2] = {0.f, 0.f}, ts1[2] = {0.f, 0.f}, tq1[2] = {0.f, 0.f}, xs_[2] = {0.f, 0.f}, xq_[2] = {0.f, 0.f};   // two tiles in flight
            f32x4 xr[2][4];
            auto rd = [&](int t, f32x4 (&x)[4]) {
#pragma unroll
                for (int m = 0; m < 4; m++) x[m] = *(const f32x4 *)(a_s + off[m] + ((t >> 2) * RW + (t & 3)) * PS);
            };
            auto store_piece = [&](auto P) {
                constexpr int p = decltype(P)::value, m = p / 4, r = p % 4;
                const int pos = (mt0 + m) * 16 + lk * 4 + r;
                if (PAD || pos < 121) o[pos * 32 + nt * 16 + lp] = accp[m][r] + bias;
            };
            auto moment_piece = [&](auto M, auto Q) {   // tile_moments(accp[m], ...) in four pieces; the tiles' sums join Ws / Wq in tile order
                constexpr int m = decltype(M)::value, q = decltype(Q)::value, i = m & 1;
                if constexpr (q == 0) {
                    float a[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) a[r] = (mt0 + m) * 16 + lk * 4 + r < 121 ? accp[m][r] : 0.0f;
                    ts[i] = a[0] + a[1];
                    ts[i] = ts[i] + a[2];
                    ts[i] = ts[i] + a[3];
                    tq[i] = a[0] * a[0];
                    tq[i] = __builtin_fmaf(a[1], a[1], tq[i]);
                    tq[i] = __builtin_fmaf(a[2], a[2], tq[i]);
                    tq[i] = __builtin_fmaf(a[3], a[3], tq[i]);
                } else if constexpr (q == 1) {
                    xs_[i] = __shfl_xor(ts[i], 16);
                    xq_[i] = __shfl_xor(tq[i], 16);
                } else if constexpr (q == 2) {
                    ts1[i] = ts[i] + xs_[i];
                    tq1[i] = tq[i] + xq_[i];
                    xs_[i] = __shfl_xor(ts1[i], 32);
                    xq_[i] = __shfl_xor(tq1[i], 32);
                } else {
                    const float Ts = ts1[i] + xs_[i], Tq = tq1[i] + xq_[i];
                    Ws = Ws + Ts;
                    Wq = Wq + Tq;
                }
            };
            // steps 0-7: two stores each; steps 8-15: the moments, two tiles in flight so that no exchange is read in the step that requests it
            constexpr int MSEQ[16][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 0}, {0, 3}, {1, 1}, {1, 2}, {2, 0}, {1, 3}, {2, 1}, {2, 2}, {3, 0}, {2, 3}, {3, 1}, {3, 2}, {-1, 0}};
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            rd(0, xr[0]);
            static_for<16>([&](auto T) {
                constexpr int t = decltype(T)::value;
                if constexpr (t + 1 < 16) rd(t + 1, xr[(t + 1) & 1]);
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int c4 = 0; c4 < 4; c4++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[t & 1][m][c4], b[4 * t + c4], acc[m], 0, 0, 0);
                if constexpr (t < 8) {
                    store_piece(std::integral_constant<int, 2 * t>{});
                    store_piece(std::integral_constant<int, 2 * t + 1>{});
                } else {
                    static_for<2>([&](auto H) {
                        constexpr int e = 2 * (t - 8) + decltype(H)::value;
                        if constexpr (MSEQ[e][0] >= 0) moment_piece(std::integral_constant<int, MSEQ[e][0]>{}, std::integral_constant<int, MSEQ[e][1]>{});
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            moment_piece(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
            wsum[0][wv][0][lp] = Ws;
            wsum[0][wv][1][lp] = Wq;
        }
#pragma unroll
        for (int m = 0; m < 4; m++) accp[m] = acc[m];
        DNE_ACC(2);
        __syncthreads();                             // every wave is done reading this frame's image; wsum of the previous frame is complete
        if (fi > 0) frame_moments(row0 + fi - 1, 0);
        DNE_ACC(3);
    }
    epilogue(accp, row0 + FPW - 1, 1);
    __syncthreads();
    frame_moments(row0 + FPW - 1, 1);
    DNE_ACC_STORE_EVERY(4, gridDim.x / 128);
}

// ------------------------------------------------------------------------------- conv1 -> conv2 in one kernel (lock-steps)
// At full width a window's lock-step is a serial chain conv1 -> conv2 -> fc -> emulator, each link stretched by the other
// windows' HBM streams; y1 (28 KB per member) made a round trip through the memory system between the first two.  Here one
// workgroup takes a member through both convolutions: conv1's epilogue applies bn1 + relu and writes straight into conv2's
// padded LDS image.  Same tiles, same MFMA order, same
// bits as k_conv1 + k_conv2; y1 is written out only when asked (dne_debug_activations, dne_act).
struct Conv12Lds {
    float lut[256];
    uint32_t img[88 * 88];
    float a_s[24 * C2_RW * C2_PS];
};

template <bool HAS_BN>
__global__ __launch_bounds__(256, 2) void k_conv12(FwdArgs A, const int *__restrict__ list, int gsize,
                                                const uint8_t *__restrict__ stacks, float *__restrict__ y1 /*may be null*/,
                                                float *__restrict__ y2, int act2 /* write relu(bn2(y2)) instead of y2: k_fc_ring's input */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char conv12_raw[];
    Conv12Lds &S = *reinterpret_cast<Conv12Lds *>(conv12_raw);
    const Item it = decode_item(A, blockIdx.x, list, gsize, 1, 0, stacks, nullptr, A.done);
    if (it.skip) return;
    DNE_WG_BEGIN;
    // profiling build: thread 0 of workgroups 1024 .. 1151 of the launch (the steady state, not the launch's first wave) stamps its phases (tools/conv12_phase_clock.py)
#ifdef DNE_PHASE_CLOCK
#define CONV12_MARK(I) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x - 1024u < 128u) dne::g_phase[4][blockIdx.x - 1024u][I] = (long long)wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CONV12_MARK(I) do { } while (0)
#endif
    CONV12_MARK(0);
    constexpr int PS = C2_PS, RW = C2_RW;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, ci = lane >> 4;
    const float *base = item_base<false>(A, it);     // more than 128 members per launch: the tail table is never in use here
    const float *eps = item_eps<false>(A, it);
    const float sc = item_scale<false>(A, it);
    const float *bn = A.bn + (size_t)it.member * 608;
    // ---- everything both convolutions need from memory, issued up front
    uint32_t px[28];
#pragma unroll
    for (int j = 0; j < 28; j++) {
        const int e = tid + 256 * j;
        px[j] = e < 7056 ? ((const uint32_t *)it.ob)[e] : 0u;
    }
    float b[64];
    {
        const float *b1 = base + A.L.c1w, *e1 = eps + A.L.c1w;
#pragma unroll
        for (int kk = 0; kk < 64; kk++) {
            float v = sc * e1[64 * kk + lane];
            b[kk] = b1[64 * kk + lane] + v;
        }
    }
    float pb1 = sc * eps[A.L.c1w + 4096 + lp];
    const float bias1 = base[A.L.c1w + 4096 + lp] + pb1;
    const float s1 = HAS_BN ? bn[lp] : 1.0f, h1 = HAS_BN ? bn[16 + lp] : 0.0f;      // conv1's output channel of this lane = lp
    const int nt = wv & 1, mt0 = 4 * (wv >> 1), lk = ci;

    float pb2 = sc * eps[A.L.c2w + 8192 + nt * 16 + lp];
    const float bias2 = base[A.L.c2w + 8192 + nt * 16 + lp] + pb2;
    S.lut[tid] = (float)tid / 255.0f;
    for (int i = tid; i < 688; i += 256) {                                            // conv1's 2-pixel zero border
        int r, c;
        if (i < 352) { r = i / 88; r = r < 2 ? r : 84 + r; c = i % 88; }
        else { const int j = i - 352; r = 2 + j / 4; c = j % 4; c = c < 2 ? c : 84 + c; }
        S.img[r * 88 + c] = 0u;
    }
    for (int pix = tid; pix < 24 * 24; pix += 256) {                                  // conv2's SAME-padding ring
        const int y = pix / 24, x = pix % 24;
        if (y < 1 || y > 21 || x < 1 || x > 21)
#pragma unroll
            for (int c = 0; c < 16; c++) S.a_s[(y * RW + x) * PS + c] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 28; j++) {
        const int e = tid + 256 * j;
        if (e < 7056) S.img[(e / 84 + 2) * 88 + e % 84 + 2] = px[j];
    }
    __syncthreads();
    CONV12_MARK(1);
    // ---- conv1: 28 position tiles, 7 per wave (three pairs + one single), exactly k_conv1's schedule
    float *out1 = y1 ? y1 + (size_t)it.row * 7056 : nullptr;
    auto run1 = [&](int j, auto has_b) {
        constexpr bool HASB = decltype(has_b)::value;
        const int tA = wv + 4 * j, tB = wv + 4 * (j + 1);
        const int pA = min(tA * 16 + lp, 440), pB = HASB ? min(tB * 16 + lp, 440) : 0;
        const int oA = (pA / 21) * 4 * 88 + (pA % 21) * 4, oB = (pB / 21) * 4 * 88 + (pB % 21) * 4;
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 8; kh++) {
#pragma unroll
            for (int kw = 0; kw < 8; kw++) {
                const float xA = S.lut[(S.img[oA + kh * 88 + kw] >> (8 * ci)) & 255u];
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xA, b[kh * 8 + kw], accA, 0, 0, 0);
                if (HASB) {
                    const float xB = S.lut[(S.img[oB + kh * 88 + kw] >> (8 * ci)) & 255u];
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xB, b[kh * 8 + kw], accB, 0, 0, 0);
                }
            }
        }
        auto emit = [&](const f32x4 &acc, int t) {
#pragma unroll
            for (int r = 0; r < 4; r++) {   // D[row = 4*(l>>4) + r][col = l&15]
                const int pos = t * 16 + ci * 4 + r;
                if (pos < 441) {
                    const float y = acc[r] + bias1;
                    if (out1) out1[pos * 16 + lp] = y;
                    float a = y;
                    if (HAS_BN) {
                        a = a * s1;
                        a = a + h1;
                    }
                    S.a_s[((pos / 21 + 1) * RW + pos % 21 + 1) * PS + lp] = a > 0.0f ? a : 0.0f;
                }
            }
        };
        emit(accA, tA);
        if (HASB) emit(accB, tB);
    };
    run1(0, std::true_type{});
    run1(2, std::true_type{});
    run1(4, std::true_type{});
    run1(6, std::false_type{});
    CONV12_MARK(2);
    // ---- conv2: its perturbed weights (fetching them before conv1 costs 69 spilled registers and was slower), then four
    //      position tiles per wave over the image conv1 just wrote
    float b2[64];
    {
        const float *w2 = base + A.L.c2w, *ee = eps + A.L.c2w;
#pragma unroll
        for (int kk = 0; kk < 64; kk++) {
            const int o = (4 * kk + lk) * 32 + nt * 16 + lp;
            float v = sc * ee[o];
            b2[kk] = w2[o] + v;
        }
    }
    __syncthreads();
    CONV12_MARK(3);
    int off[4];
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int p = min((mt0 + m) * 16 + lp, 120);
        off[m] = ((p / 11) * 2 * RW + (p % 11) * 2) * PS + lk;
        acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kh = 0; kh < 4; kh++) {
#pragma unroll
        for (int kw = 0; kw < 4; kw++) {
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {   // k = (kh*4+kw)*16 + c4*4 + (l>>4)
                const int kk = (kh * 4 + kw) * 4 + c4;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const float x = S.a_s[off[m] + (kh * RW + kw) * PS + c4 * 4];
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b2[kk], acc[m], 0, 0, 0);
                }
            }
        }
    }
    CONV12_MARK(4);
    float *o = y2 + (size_t)it.row * 3872;
    // k_fc_ring takes its activations as scalars and wants them finished: relu(fl(fl(y * scale) + shift)), the consumer's own operations
    const bool act = HAS_BN && act2 != 0;
    const float s2 = act ? bn[32 + nt * 16 + lp] : 1.0f, h2 = act ? bn[64 + nt * 16 + lp] : 0.0f;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pos = (mt0 + m) * 16 + lk * 4 + r;
            float y = acc[m][r] + bias2;
            if (act) {
                y = y * s2;
                y = y + h2;
                y = y > 0.0f ? y : 0.0f;
            }
            if (pos < 121) o[pos * 32 + nt * 16 + lp] = y;
        }
    CONV12_MARK(5);
    DNE_WG_END(0);
#undef CONV12_MARK
}

// conv1 -> conv2 in one launch for the tail of a generation (a few dozen members left: a lock-step is a chain of launches on a
// nearly idle chip, and every launch has a floor of 3-4 us whatever it does).  Four workgroups of eight waves per member; each
// owns two of conv2's eight position tiles (32 of the 121 positions), works out for itself the rows of conv1's output those
// positions read -- at most 15 of the 28 conv1 tiles, one or two per wave, some of them computed by a neighbour as well (48 tiles
// per member instead of 28; the matrix cores are idle anyway) -- and runs its conv2 tiles over them on four waves.  No
// workgroup waits for another; y1 never leaves LDS.  Same tiles, same MFMA order, same bits as k_conv1 + k_conv2.
template <bool HAS_BN, bool TT>
__device__ __forceinline__ void conv12t_body(Conv12Lds &S, const FwdArgs &A, const Item &it, int part, float *__restrict__ y1 /*may be null*/,
                                             float *__restrict__ y2) {
    constexpr int PS = C2_PS, RW = C2_RW, NT = 512;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, ci = lane >> 4;
    const float *base = item_base<TT>(A, it);
    const float *eps = item_eps<TT>(A, it);
    const float sc = item_scale<TT>(A, it);
    const float *bn = A.bn + (size_t)it.member * 608;
    // conv2 positions 32 part .. 32 part + 31 -> conv2 rows oy0 .. oy1 -> conv1 rows ra .. rb (4x4 stride 2, SAME(1,2)) -> tiles t0 .. t1
    DNE_PHASE(1, 0);
    const int oy0 = (32 * part) / 11, oy1 = min(32 * part + 31, 120) / 11;
    const int ra = max(2 * oy0 - 1, 0), rb = min(2 * oy1 + 2, 20);
    const int t0 = (21 * ra) / 16, t1 = (21 * rb + 20) / 16;
    uint32_t px[14];
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const int e = tid + NT * j;
        px[j] = e < 7056 ? ((const uint32_t *)it.ob)[e] : 0u;
    }
    float b[64];
    {
        const float *b1 = base + A.L.c1w, *e1 = eps + A.L.c1w;
#pragma unroll
        for (int kk = 0; kk < 64; kk++) {
            float v = sc * e1[64 * kk + lane];
            b[kk] = b1[64 * kk + lane] + v;
        }
    }
    float pb1 = sc * eps[A.L.c1w + 4096 + lp];
    const float bias1 = base[A.L.c1w + 4096 + lp] + pb1;
    const float s1 = HAS_BN ? bn[lp] : 1.0f, h1 = HAS_BN ? bn[16 + lp] : 0.0f;      // conv1's output channel of this lane = lp
    const int nt = wv & 1, mt = 2 * part + ((wv >> 1) & 1), lk = ci;                  // conv2: waves 0..3, one tile each
    float pb2 = sc * eps[A.L.c2w + 8192 + nt * 16 + lp];
    const float bias2 = base[A.L.c2w + 8192 + nt * 16 + lp] + pb2;
    if (tid < 256) S.lut[tid] = (float)tid / 255.0f;
    for (int i = tid; i < 688; i += NT) {                                             // conv1's 2-pixel zero border
        int r, c;
        if (i < 352) { r = i / 88; r = r < 2 ? r : 84 + r; c = i % 88; }
        else { const int j = i - 352; r = 2 + j / 4; c = j % 4; c = c < 2 ? c : 84 + c; }
        S.img[r * 88 + c] = 0u;
    }
    for (int pix = tid; pix < 24 * 24; pix += NT) {                                   // conv2's SAME-padding ring
        const int y = pix / 24, x = pix % 24;
        if (y < 1 || y > 21 || x < 1 || x > 21)
#pragma unroll
            for (int c = 0; c < 16; c++) S.a_s[(y * RW + x) * PS + c] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const int e = tid + NT * j;
        if (e < 7056) S.img[(e / 84 + 2) * 88 + e % 84 + 2] = px[j];
    }
    __syncthreads();
    DNE_PHASE(1, 1);
    float *out1 = y1 ? y1 + (size_t)it.row * 7056 : nullptr;
    auto run1 = [&](int tA, int tB, auto has_b) {
        constexpr bool HASB = decltype(has_b)::value;
        const int pA = min(tA * 16 + lp, 440), pB = HASB ? min(tB * 16 + lp, 440) : 0;
        const int oA = (pA / 21) * 4 * 88 + (pA % 21) * 4, oB = (pB / 21) * 4 * 88 + (pB % 21) * 4;
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 8; kh++) {
#pragma unroll
            for (int kw = 0; kw < 8; kw++) {
                const float xA = S.lut[(S.img[oA + kh * 88 + kw] >> (8 * ci)) & 255u];
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xA, b[kh * 8 + kw], accA, 0, 0, 0);
                if (HASB) {
                    const float xB = S.lut[(S.img[oB + kh * 88 + kw] >> (8 * ci)) & 255u];
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xB, b[kh * 8 + kw], accB, 0, 0, 0);
                }
            }
        }
        auto emit = [&](const f32x4 &acc, int t) {
#pragma unroll
            for (int r = 0; r < 4; r++) {   // D[row = 4*(l>>4) + r][col = l&15]
                const int pos = t * 16 + ci * 4 + r;
                if (pos < 441) {
                    const float y = acc[r] + bias1;
                    if (out1) out1[pos * 16 + lp] = y;
                    float a = y;
                    if (HAS_BN) {
                        a = a * s1;
                        a = a + h1;
                    }
                    S.a_s[((pos / 21 + 1) * RW + pos % 21 + 1) * PS + lp] = a > 0.0f ? a : 0.0f;
                }
            }
        };
        emit(accA, tA);
        if (HASB) emit(accB, tB);
    };
    {
        const int tA = t0 + wv, tB = tA + 8;
        if (tB <= t1) run1(tA, tB, std::true_type{});
        else if (tA <= t1) run1(tA, tA, std::false_type{});
    }
    DNE_PHASE(1, 2);
    float b2[64];
    if (wv < 4) {
        const float *w2 = base + A.L.c2w, *ee = eps + A.L.c2w;
#pragma unroll
        for (int kk = 0; kk < 64; kk++) {
            const int o = (4 * kk + lk) * 32 + nt * 16 + lp;
            float v = sc * ee[o];
            b2[kk] = w2[o] + v;
        }
    }
    __syncthreads();
    DNE_PHASE(1, 3);
    if (wv >= 4) return;
    const int p = min(mt * 16 + lp, 120);
    const int off = ((p / 11) * 2 * RW + (p % 11) * 2) * PS + lk;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 4; kh++) {
#pragma unroll
        for (int kw = 0; kw < 4; kw++) {
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {   // k = (kh*4+kw)*16 + c4*4 + (l>>4)
                const float x = S.a_s[off + (kh * RW + kw) * PS + c4 * 4];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b2[(kh * 4 + kw) * 4 + c4], acc, 0, 0, 0);
            }
        }
    }
    float *o = y2 + (size_t)it.row * 3872;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int pos = mt * 16 + lk * 4 + r;
        if (pos < 121) o[pos * 32 + nt * 16 + lp] = acc[r] + bias2;
    }
    DNE_PHASE(1, 4);
}

template <bool HAS_BN>
__global__ __launch_bounds__(512) void k_conv12t(FwdArgs A, const int *__restrict__ list, int gsize, const uint8_t *__restrict__ stacks,
                                                 float *__restrict__ y1 /*may be null*/, float *__restrict__ y2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char conv12_raw[];
    Conv12Lds &S = *reinterpret_cast<Conv12Lds *>(conv12_raw);
    // no look at the done flags: a finished member still in the list (until the next compaction, at most 15 lock-steps) costs a
    // few idle CUs; a dependent global load in front of everything else costs every lock-step a microsecond
    if (A.tt.n > 0) {
        const Item it = decode_item<true>(A, blockIdx.x >> 2, list, gsize, 1, 0, stacks, nullptr, nullptr);
        conv12t_body<HAS_BN, true>(S, A, it, blockIdx.x & 3, y1, y2);
    } else {
        const Item it = decode_item(A, blockIdx.x >> 2, list, gsize, 1, 0, stacks, nullptr, nullptr);
        conv12t_body<HAS_BN, false>(S, A, it, blockIdx.x & 3, y1, y2);
    }
}

// ------------------------------------------------------------------------- fc (+ out + argmax)
// The HBM-bound kernel: streams the 3872x256 noise slice once per workgroup.  4 waves = the 4 k-slices;
// lane l owns output columns 4l..4l+3 (one 16-byte load per lane per row = 1 KiB per wave-instruction).
// NV activation vectors share the stream:
//   step mode   NV = group size (ES antithetic pair: 2 members with scales +sigma/-sigma; GA: 1)
//   ref mode    SHARED_W: NV = 16 reference frames of one member (same weights)
// Activations are relu(bn2(y2)) formed per 64-row chunk in one VGPR per vector and broadcast with
// v_readlane.  Step mode finishes with bn3 + relu, the 256 x nact output layer and the first-max argmax.
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ------------------------------------------------------------------ output layer (policies.py:329 / 457), the oracle's out_raw_k
// logit[a] = tree over k of p[k] = a3[k] * w[k][a]:  every group of 64 consecutive k is one wavefront, lane = k, and the 64
// products are summed by a lane butterfly with strides 1, 2, 4, 8, 16, 32 -- a balanced binary tree, neighbours first, both lanes
// of a pair holding the same sum at every level (fp addition is commutative); the groups' sums meet in LDS and are combined
// ((S0+S1)+(S2+S3)) (+ ((S4+S5)+(S6+S7)) for the LargeModel's 512 inputs) + bias.  No serial chain, no staging of the weights:
// a thread loads the nact consecutive weights of its own k (base and noise) and that is the layer's whole memory traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// The tree's value in LANE 63.  The two top levels (rows of 16 -> the wave) cross the rows by DPP broadcasts (row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3) instead of two ds_bpermute round trips through the LDS pipe: row 3 ends with (R3 + R2) + (R1 + R0), bit for bit
// the butterfly's (R0 + R1) + (R2 + R3) -- fp addition is commutative at every level.  The other rows hold partial sums: only lane 63 is read.
__device__ __forceinline__ float wave_tree_sum(float v) {
    v = v + dpp_get<0xB1>(v);      // quad_perm [1,0,3,2]: lane ^ 1
    v = v + dpp_get<0x4E>(v);      // quad_perm [2,3,0,1]: lane ^ 2
    v = v + dpp_get<0x141>(v);     // row_half_mirror: the other quad of the 8 (every lane of a quad holds the quad's sum by now)
    v = v + dpp_get<0x140>(v);     // row_mirror: the other half of the row of 16
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3 (+ 0.0f elsewhere)
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
    return v;
}
constexpr int WAVE_TREE_LANE = 63;

constexpr int OUT_NA = 32;         // the action loops are unrolled over the largest action set the engine accepts

// p[s][a] = x[s] * fl(theta[a] + fl(sc[s] * eps[a])) for this thread's k; wb / we point at w[k][0] in the base vector / noise slice.
// The NS members share one base vector and one noise slice (an antithetic pair: scales +sigma / -sigma).
template <int NS>
__device__ __forceinline__ void out_products(float (&p)[NS][OUT_NA], const float (&x)[NS], const float *__restrict__ wb,
                                             const float *__restrict__ we, const float (&sc)[NS], int nact) {
    float th[OUT_NA], ep[OUT_NA];
#pragma unroll
    for (int a = 0; a < OUT_NA; a++) {
        th[a] = a < nact ? wb[a] : 0.0f;
        ep[a] = a < nact ? we[a] : 0.0f;
    }
#pragma unroll
    for (int a = 0; a < OUT_NA; a++)
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) {
            float pv = sc[s2] * ep[a];
            float w = th[a] + pv;
            p[s2][a] = x[s2] * w;
        }
}

// the wavefront's tree sums of p -> red[s][a] (this wave's slot in LDS)
template <int NS>
__device__ __forceinline__ void out_wave_sums(float (&p)[NS][OUT_NA], int nact, float (*red)[OUT_NA], int lane) {
#pragma unroll
    for (int a = 0; a < OUT_NA; a++)
        if (a < nact) {
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) p[s2][a] = wave_tree_sum(p[s2][a]);
        }
    if (lane == WAVE_TREE_LANE) {
#pragma unroll
        for (int a = 0; a < OUT_NA; a++)
            if (a < nact) {
#pragma unroll
                for (int s2 = 0; s2 < NS; s2++) red[s2][a] = p[s2][a];
            }
    }
}

// NOISE = false: the members are plain vectors (GA children written out once per generation, scale 0): the noise rows are not
// streamed at all -- fl(base + fl(0 * eps)) = base -- which halves the bytes of a member-step.
// (a padded register footprint -- at most three / two of its workgroups per CU instead of four, k_fc_duo's FAT idea -- was measured for the
//  GA's k_fc<1>: 1000 members 732 / 812 vs 739 us per lock-step, 500 members 398 / 373 vs 403, no gain over a generation; not kept)
template <int NV, bool SHARED_W, bool HAS_BN, int RB, bool NOISE = true>
__global__ __launch_bounds__(256) void k_fc(FwdArgs A, const int *__restrict__ list, int n_local, int F, int member0,
                                            const float *__restrict__ y2, float *__restrict__ y3,
                                            int32_t *__restrict__ actions, float *__restrict__ logits_out) {
    __shared__ float part[4][NV][256];
    __shared__ float red[4][SHARED_W ? 1 : NV][OUT_NA];
    __shared__ float lg[SHARED_W ? 1 : NV][32];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    // step mode is persistent: a bounded grid walks the active groups, so the kernel occupies only a slice of
    // every CU (it is HBM-bound and needs few waves) and leaves room for the conv / emulator kernels that
    // run concurrently on the other sub-batch streams.  Reference mode launches one block per work item.
    if (!SHARED_W) __builtin_amdgcn_s_setprio(3);   // keep the HBM stream's load issue ahead of co-resident conv / emulator waves
    for (int item = blockIdx.x; item < (SHARED_W ? (int)blockIdx.x + 1 : n_local); item += gridDim.x) {
    int member[NV], row[NV];
    float scale[NV];
    if (SHARED_W) {   // reference mode: blocks of one member stay on one XCD (block b runs on XCD b % 8)
        const int nfg = F / NV, q = blockIdx.x >> 3, x = blockIdx.x & 7;
        const int mloc = (q / nfg) * 8 + x, fg = q % nfg;
        if (mloc >= n_local) return;
#pragma unroll
        for (int v = 0; v < NV; v++) { member[v] = member0 + mloc; row[v] = mloc * F + fg * NV + v; }
    } else {
        const int g = list ? list[item] : item;
#pragma unroll
        for (int v = 0; v < NV; v++) { member[v] = g * NV + v; row[v] = member[v]; }
        if (A.done) {   // finished group still in the list (compaction runs every 32 lock-steps): nothing to compute
            bool all_done = true;
#pragma unroll
            for (int v = 0; v < NV; v++) all_done = all_done && A.done[member[v]] != 0;
            if (all_done) continue;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; v++) scale[v] = A.m_scale[member[v]];
    const int64_t off = A.m_off[member[0]];
    const float *base = A.bases + (size_t)A.m_slot[member[0]] * A.base_stride;
    const float *eps = A.noise + off + L.fcw + lane * 4;
    const float *th = base + L.fcw + lane * 4;

    const int ch = (8 * wv + lane) & 31;   // bn2 channel of this lane's activation rows (968 = 8 mod 32)
    float s2[NV], h2[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        s2[v] = HAS_BN ? A.bn[(size_t)member[v] * 608 + 32 + ch] : 1.0f;
        h2[v] = HAS_BN ? A.bn[(size_t)member[v] * 608 + 64 + ch] : 0.0f;
    }
    float acc[NV][4];
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[v][e] = 0.0f;

    // 968 rows per slice = 121 batches of 8 rows; batch b+1 is in flight while batch b is consumed.
    // Activations arrive in 64-row chunks (one VGPR per vector), the next chunk prefetched a chunk ahead.
    const int kbeg = 968 * wv;
    auto load_x = [&](int c, float (&dst)[NV]) {
        const int nr = c < 15 ? 64 : 8;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            float t = 0.0f;
            if (c < 16 && lane < nr) {
                t = y2[(size_t)row[v] * 3872 + kbeg + 64 * c + lane];
                if (HAS_BN) {
                    t = t * s2[v];
                    t = t + h2[v];
                }
                t = t > 0.0f ? t : 0.0f;
            }
            dst[v] = t;
        }
    };
    // RB rows per batch (968 and 64 are multiples of 2, 4 and 8)
    f4u e_cur[RB], e_nxt[RB];
    f4a t_cur[RB], t_nxt[RB];
    float xv[NV], xn[NV];
    load_x(0, xv);
    load_x(1, xn);
#pragma unroll
    for (int i = 0; i < RB; i++) {
        const size_t ro = (size_t)(kbeg + i) * 256;
        if (NOISE) e_cur[i] = *(const f4u *)(eps + ro);
        t_cur[i] = *(const f4a *)(th + ro);
    }
    constexpr int NB = 968 / RB, BPC = 64 / RB;   // batches per slice, batches per 64-row activation chunk
    float fold[NV][4];                            // the quarter's running left fold over its sub-slices
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int q = 0; q < 4; q++) fold[v][q] = 0.0f;
    int next_sub = FC_SUB0;
    for (int bt = 0; bt < NB; bt++) {
        if (bt + 1 < NB) {
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const size_t ro = (size_t)(kbeg + (bt + 1) * RB + i) * 256;
                if (NOISE) e_nxt[i] = *(const f4u *)(eps + ro);
                t_nxt[i] = *(const f4a *)(th + ro);
            }
        }
        const int li = (bt % BPC) * RB;
#pragma unroll
        for (int i = 0; i < RB; i++) {
            if (SHARED_W) {
                f4a w;
#pragma unroll
                for (int q = 0; q < 4; q++) { float pv = NOISE ? scale[0] * e_cur[i][q] : 0.0f; w[q] = NOISE ? t_cur[i][q] + pv : t_cur[i][q]; }
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const float x = lane_bcast(xv[v], li + i);
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[v][q] = __builtin_fmaf(x, w[q], acc[v][q]);
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const float x = lane_bcast(xv[v], li + i);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float pv = NOISE ? scale[v] * e_cur[i][q] : 0.0f;
                        float w = NOISE ? t_cur[i][q] + pv : t_cur[i][q];
                        acc[v][q] = __builtin_fmaf(x, w, acc[v][q]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; i++) { if (NOISE) e_cur[i] = e_nxt[i]; t_cur[i] = t_nxt[i]; }
        if (bt % BPC == BPC - 1) {   // chunk boundary: rotate the activation registers, prefetch the chunk after next
#pragma unroll
            for (int v = 0; v < NV; v++) xv[v] = xn[v];
            load_x(bt / BPC + 2, xn);
        }
        if ((bt + 1) * RB == next_sub) {   // end of a sub-slice: the chain joins the quarter's running fold and starts again from 0
#pragma unroll
            for (int v = 0; v < NV; v++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    fold[v][q] = next_sub == FC_SUB0 ? acc[v][q] : fold[v][q] + acc[v][q];
                    acc[v][q] = 0.0f;
                }
            next_sub += FC_SUBN;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int q = 0; q < 4; q++) part[wv][v][lane * 4 + q] = fold[v][q];
    __syncthreads();
    float x3[NV];   // relu(bn3(y3)) of column tid, per member: this thread's input k = tid of the output layer
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int j = tid;
        const float s01 = part[0][v][j] + part[1][v][j];
        const float s23 = part[2][v][j] + part[3][v][j];
        float s = s01 + s23;
        float pv = scale[v] * A.noise[off + L.fcb + j];
        const float bias = base[L.fcb + j] + pv;
        s = s + bias;
        y3[(size_t)row[v] * 256 + j] = s;
        float t = s;
        if (!SHARED_W && HAS_BN) {
            t = t * A.bn[(size_t)member[v] * 608 + 96 + j];
            t = t + A.bn[(size_t)member[v] * 608 + 352 + j];
        }
        x3[v] = t > 0.0f ? t : 0.0f;
    }
    if constexpr (!SHARED_W) {
    const int nact = L.nact;
    {
        float p[NV][OUT_NA];
        out_products<NV>(p, x3, base + L.ow + tid * nact, A.noise + off + L.ow + tid * nact, scale, nact);
        out_wave_sums<NV>(p, nact, red[wv], lane);
    }
    __syncthreads();
    if (tid < NV * nact) {
        const int v = tid / nact, a = tid % nact;
        const float s01 = red[0][v][a] + red[1][v][a];
        const float s23 = red[2][v][a] + red[3][v][a];
        const float t = s01 + s23;
        float pv = A.m_scale[member[0] + v] * A.noise[off + L.ob + a];
        const float bias = base[L.ob + a] + pv;
        lg[v][a] = t + bias;
    }
    __syncthreads();
    if (tid < NV) {
        const int v = tid;
        int best = 0;
        for (int a = 1; a < nact; a++)
            if (lg[v][a] > lg[v][best]) best = a;   // tf.argmax: first maximum
        actions[member[0] + v] = best;
        if (logits_out)
            for (int a = 0; a < nact; a++) logits_out[(size_t)(member[0] + v) * nact + a] = lg[v][a];
    }
    __syncthreads();   // LDS is reused by the next group
    }
    }
}


// ------------------------------------------------------- table-ordered fc: units that overlap in the noise table share its rows
// The noise slices of a population overlap: 2500 slices of 1 M floats drawn from a 250 M table cover every table row about
// ten times, and within the fc matrix a (pair, k-slice) unit of 968 rows starts on average 97 rows after the previous one in
// table order.  k_unit_order ranks the units of a window by their table address; k_fc_duo gives each wave the two units of an
// adjacent pair in that order and runs them in table lock-step: unit B starts `gb` row batches after unit A, so that both read
// (nearly) the same table rows in the same load batch -- the second reader finds the lines in L1 / L2 and HBM delivers them once.
// Which units travel together is a schedule, not arithmetic: every unit still walks its own 968 rows in k order from a zero
// accumulator and the partial sums go to y3t[member][slice][256], where k_out combines them ((s0+s1)+(s2+s3)) + bias like
// every other fc variant -- same bits.  Units of finished groups are dropped; an unpartnered unit runs alone.
constexpr int SLICE_FLOATS = 968 * 256;

__global__ __launch_bounds__(256) void k_unit_order(const int64_t *__restrict__ m_off, const int *__restrict__ list, int cnt, int gsize,
                                                    int *__restrict__ order /*[4 * cnt]: unit = 4 * group + slice*/) {
    extern __shared__ long long uo_keys[];
    const int n = 4 * cnt;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int g = list ? list[i >> 2] : i >> 2;
        uo_keys[i] = m_off[(size_t)g * gsize] + (long long)(i & 3) * SLICE_FLOATS;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long k = uo_keys[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
        const long long kj = uo_keys[j];
        rank += (kj < k || (kj == k && j < i)) ? 1 : 0;
    }
    const int g = list ? list[i >> 2] : i >> 2;
    order[rank] = g * 4 + (i & 3);
}

// One side (unit) of a duo: its streams, its activations, its accumulators.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <int NV>
struct DuoSide {
    const float *enext, *tnext;   // wave-uniform: the unit's next row block in the noise table / in the base vector
    const float *xs[NV];
    float scale[NV], s2[NV], h2[NV];
    f32x2 acc[NV][2];             // this lane's 4 columns as two register pairs (v_pk_fma_f32): the chain of the current sub-slice
    f32x2 fold[NV][2];            // the quarter's running left fold over its finished sub-slices
    int nb;                       // row block at which the current sub-slice ends (wave-uniform)
    float xv[NV], xn[NV];         // relu(bn2(y2)) of the current / next 64-row chunk, one row per lane
    int lb;                       // index of the row block computed next (wave-uniform)
    int sl, mem[NV];              // k-slice and members of the unit
    long long key;                // table address of the unit's first row
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
// 16 bytes per lane from (uniform base + per-lane byte offset + OFF); the result lands asynchronously: read it only through
// wait_rows.  Written as asm so that the load is issued exactly here, into the registers of the row just consumed -- the
// compiler's own schedule hoists the loads of a row block to its top, which costs a second register set and a copy per row.
// cache-policy experiments (same-box A/B through DNE_LIB_PATH; the product library leaves both empty): -DDNE_EPS_MOD='" nt"' etc.
#ifndef DNE_EPS_MOD
#define DNE_EPS_MOD ""
#endif
#ifndef DNE_THETA_MOD
#define DNE_THETA_MOD ""
#endif
template <int OFF>
__device__ __forceinline__ void gload4(f32x4 &dst, unsigned voff, const float *sbase) {
    // "+v": the registers of the row just consumed are the destination (its readers come first; nothing is kept alive or copied)
    asm volatile("global_load_dwordx4 %[d], %[vo], %[sb] offset:%[of]" DNE_THETA_MOD : [d] "+v"(dst) : [vo] "v"(voff), [sb] "s"(sbase), [of] "n"(OFF));
}
// the same load, pinned behind the accumulator updates of the row it replaces: without the (untouched) accumulators as
// operands the compiler sinks a whole block's arithmetic below all of its loads and keeps the old rows alive in copies
template <int OFF>
__device__ __forceinline__ void gload4_after(f32x4 &dst, unsigned voff, const float *sbase, f32x2 &a0, f32x2 &a1, f32x2 &a2, f32x2 &a3) {
    asm volatile("global_load_dwordx4 %[d], %[vo], %[sb] offset:%[of]" DNE_EPS_MOD
                 : [d] "+v"(dst), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : [vo] "v"(voff), [sb] "s"(sbase), [of] "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void gload4_after(f32x4 &dst, unsigned voff, const float *sbase, f32x2 &a0, f32x2 &a1) {
    asm volatile("global_load_dwordx4 %[d], %[vo], %[sb] offset:%[of]" DNE_EPS_MOD
                 : [d] "+v"(dst), "+v"(a0), "+v"(a1) : [vo] "v"(voff), [sb] "s"(sbase), [of] "n"(OFF));
}
template <int OFF>   // a base row, pinned behind the accumulator updates of the row it replaces (k_fc_ring)
__device__ __forceinline__ void gload4_theta_after(f32x4 &dst, unsigned voff, const float *sbase, f32x2 &a0, f32x2 &a1, f32x2 &a2, f32x2 &a3) {
    asm volatile("global_load_dwordx4 %[d], %[vo], %[sb] offset:%[of]" DNE_THETA_MOD
                 : [d] "+v"(dst), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : [vo] "v"(voff), [sb] "s"(sbase), [of] "n"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_rows(f32x4 &a, f32x4 &b) {
    asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(a), "+v"(b) : [n] "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_rows(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {   // at most N loads still in flight afterwards
    asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : [n] "n"(N));
}

// SWEEP (round 3): the four waves of a workgroup hold four ADJACENT duos of the table order -- eight units whose slices overlap --
// but started together each wave reads a given table row ~24 row blocks (~100 us) after its neighbour did, long after the line has
// left every cache.  Here the workgroup walks ONE table timeline: the wave with the lowest table address starts, the others join
// when the front reaches their unit's first row (a wave-uniform count of s_barrier, no fence: the rolling loads stay in flight), and
// one s_barrier per row block keeps the four in table lock-step, so that all eight units ask for a table row within the same few
// loads and HBM delivers it once.  A schedule, not arithmetic: same chains, same bits.

// W = rows in flight per stream.  W = 8: 198 registers, two waves per SIMD.  W = 4 (round 4): the same bytes in flight per SIMD from
// twice the waves (<= 128 registers, four waves per SIMD) -- the SQ counters of the W = 8 kernel alone show its waves issuing 17 % of
// their cycles, stalled on instruction dependencies 41 % and parked (timeline start-up, barriers, loads) 41 %: more waves fill those.
// FAT (round 4): the kernel streams as fast from ONE workgroup per CU as from two (alone, 2500 pairs: 1.13 ms on a grid of 256, 1.14 on
// 512; 2.07 on 128 -- the per-CU vector-memory path is what it saturates), but two of its 198-register workgroups fill a CU's register
// file and the other windows' convolutions / renderer / head then wait for a slot instead of overlapping.  FAT touches one high
// accumulation register, which puts the kernel's register footprint past 256 per lane: the hardware then places at most one of its
// workgroups per CU, whatever window it comes from, and a 199-register k_conv12 workgroup always fits beside it.
template <int NV, bool HAS_BN, bool SWEEP = false, int W = 8, bool FAT = false>
__global__ __launch_bounds__(256, FAT ? 1 : W == 8 ? 2 : 4) void k_fc_duo(FwdArgs A, const int *__restrict__ order, int n_units,
                                                const float *__restrict__ y2, float *__restrict__ y3t, int lag) {
    static_assert(W == 8 || W == 4, "sub-slice boundaries are multiples of 8 rows");
    if constexpr (FAT) asm volatile("v_accvgpr_write_b32 a71, %0" : : "v"(0) : "a71");
    __shared__ long long sw_key[2][4];            // SWEEP: the waves' first table addresses / row-block counts, double-buffered by item parity
    __shared__ int sw_len[2][4];
    __shared__ int sw_plan[2][4][4][2];           // SWEEP: [parity][wave][round] -> the duo's units (A, B or -1)
    // W = 4: the quarter's running fold lives in LDS (touched once per sub-slice, every 30 row blocks) instead of 16 registers -- with
    // them the build spills bn2's scale / shift, and every reload drains the rows in flight (s_waitcnt vmcnt(0) inside the row loop)
    constexpr bool FOLD_LDS = W == 4;
    __shared__ f32x4 sw_fold[FOLD_LDS ? 4 : 1][2][NV][FOLD_LDS ? 64 : 1];
    int sw_par = 0;
    constexpr int NBLK = 968 / W, BPC = 64 / W;   // row blocks per unit, per 64-row activation chunk
    const int tid = threadIdx.x, wv = uni(tid >> 6), lane = tid & 63;
    const unsigned voff = lane * 16;
    const Layout &L = A.L;
    {   // wave priority of the HBM stream against co-resident conv / emulator waves (DNE_FC_PRIO, bits 9-10 of `lag`; default 3)
        const int prio = (lag >> 9) & 3;
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    }
    const bool solo = (lag & 256) != 0;   // sparse windows: one unit per wave (twice the waves, nothing to share anyway)
    const int sw_sync = ((lag >> 11) & 7) + 1;   // SWEEP: one s_barrier every sw_sync row blocks of the workgroup's timeline
    int sw_t = 0;                                // SWEEP: the workgroup's time in row blocks (the same sequence in every wave)
#ifdef DNE_PHASE_CLOCK
    bool tk_on = false;   // profiling build: this work item's ticks are stamped (env_synth.h: g_duo_tick)
    int tk_i = 0;
    const int tk_wg = blockIdx.x / 4;
#endif
    auto sw_tick = [&]() {
        sw_t++;
#ifdef DNE_PHASE_CLOCK
        long long tk_b = 0;
        if (tk_on) { __builtin_amdgcn_sched_barrier(0); tk_b = (long long)__builtin_amdgcn_s_memtime(); }
#endif
        if (sw_sync == 1 || sw_t % sw_sync == 0) __builtin_amdgcn_s_barrier();
#ifdef DNE_PHASE_CLOCK
        if (tk_on) {
            const long long tk_a = (long long)__builtin_amdgcn_s_memtime();
            if (lane == 0 && tk_i < DUO_TICK_MAX) { g_duo_tick[tk_wg][wv][tk_i][0] = tk_b; g_duo_tick[tk_wg][wv][tk_i][1] = tk_a; }
            tk_i++;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };
    const int rounds = SWEEP ? ((lag >> 14) & 3) + 1 : 1;   // SWEEP: duos a wave takes one after the other on the workgroup's timeline
    lag &= 255;
    const int n_duos = solo ? n_units : (n_units + 1) >> 1, n_items = (n_duos + 4 * rounds - 1) / (4 * rounds);
    typedef DuoSide<NV> Side;
    // the units of duo d: finished groups dropped, a lone survivor runs as side A; false when there is nothing to do
    auto select = [&](int d, int &uA, int &uB) {
        const bool in_range = d < n_duos;
        uA = in_range ? uni(order[solo ? d : 2 * d]) : -1;
        uB = in_range && !solo && 2 * d + 1 < n_units ? uni(order[2 * d + 1]) : -1;
        if (A.done && in_range) {
            auto finished = [&](int uu) {
                int all_done = 1;
#pragma unroll
                for (int v = 0; v < NV; v++) all_done &= uni(A.done[(uu >> 2) * NV + v]) != 0;
                return all_done != 0;
            };
            if (finished(uA)) uA = -1;
            if (uB >= 0 && finished(uB)) uB = -1;
        }
        if (uA < 0) { uA = uB; uB = -1; }
        return uA >= 0;
    };
    auto unit_key = [&](int uu) { return uni64(A.m_off[(size_t)(uu >> 2) * NV]) + (long long)(uu & 3) * SLICE_FLOATS; };
    auto duo_gb = [&](int uA, int uB) {   // B trails A by gb row blocks: its row (blk - gb) * W + i then sits within W rows of A's row blk * W + i in the table
        if (uB < 0) return NBLK;
        const long long delta = unit_key(uB) - unit_key(uA);
        return delta < 0 ? 0 : (int)min((long long)NBLK, delta / (256 * W) + lag);
    };
    // (work items dealt to the XCDs in contiguous ranges -- one table range per L2 -- measured slower: 273.8 vs 264.5 ms per generation;
    //  a neighbour's rows are read ~100 row blocks apart, long after any cache has dropped them)
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int sw_tail = 0;
        if constexpr (SWEEP) {
            // plan: the wave's duos of this item (round r: duo (item * rounds + r) * 4 + wave), their lengths in row blocks, the table
            // address the wave starts at
            const int par = sw_par;
            sw_par ^= 1;
            int total = 0;
            long long first_key = 0;
            for (int r = 0; r < rounds; r++) {
                int pa, pb;
                const bool ok = select((item * rounds + r) * 4 + wv, pa, pb);
                const int len = !ok ? 0 : pb >= 0 ? NBLK + duo_gb(pa, pb) : NBLK;
                if (ok && total == 0) first_key = unit_key(pa);
                total += len;
                if (lane == 0) { sw_plan[par][wv][r][0] = ok ? pa : -1; sw_plan[par][wv][r][1] = pb; }
            }
            if (lane == 0) { sw_key[par][wv] = first_key; sw_len[par][wv] = total; }
            __syncthreads();
            // the waves' keys ascend with the wave index (table order); a wave starts when the front reaches its first row, unless its
            // predecessor is a whole unit away (nothing to share: no point in waiting)
            int delay = 0, mine = 0, tmax = 0;
            long long prev = 0;
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long kj = sw_key[par][j];
                const int lj = sw_len[par][j];
                if (lj == 0) continue;
                if (any) {
                    const long long gap = (kj - prev) / (256 * W);
                    delay += gap > 0 && gap < NBLK ? (int)gap : 0;
                }
                any = true;
                prev = kj;
                if (j == wv) mine = delay;
                tmax = max(tmax, delay + lj);
            }
            mine = uni(mine); tmax = uni(tmax);
#ifdef DNE_PHASE_CLOCK
            tk_on = blockIdx.x % 4 == 0 && tk_wg < DUO_TICK_WGS && item == (int)(blockIdx.x + gridDim.x);
            tk_i = 0;
            if (tk_on && lane == 0) {
                long long *pl = g_duo_plan[tk_wg][wv];
                pl[0] = mine; pl[1] = total; pl[2] = tmax; pl[3] = 0;
                pl[4] = (long long)__builtin_amdgcn_s_memtime(); pl[6] = (long long)wall_clock64();
            }
#endif
            if (total == 0) {      // a wave without work still takes part in the workgroup's barriers
                for (int i = 0; i < tmax; i++) sw_tick();
#ifdef DNE_PHASE_CLOCK
                if (tk_on && lane == 0) {
                    long long *pl = g_duo_plan[tk_wg][wv];
                    pl[3] = tk_i; pl[5] = (long long)__builtin_amdgcn_s_memtime(); pl[7] = (long long)wall_clock64();
                }
                tk_on = false;
#endif
                continue;
            }
            for (int i = 0; i < mine; i++) sw_tick();
            sw_tail = tmax - mine - total;
        }
        for (int rnd = 0; rnd < rounds; rnd++) {
        int uA, uB;
        if constexpr (SWEEP) {
            uA = uni(sw_plan[(sw_par ^ 1)][wv][rnd][0]);
            uB = uni(sw_plan[(sw_par ^ 1)][wv][rnd][1]);
            if (uA < 0) continue;
        } else {
            if (!select(item * 4 + wv, uA, uB)) continue;
        }
        const bool b_on = uB >= 0;
        if (!b_on) uB = uA;   // valid addresses for the side that is never computed or stored

        Side SA, SB;
        auto init = [&](Side &Z, int uu) {
            const int g = uu >> 2;
            Z.sl = uu & 3;
            const long long off = uni64(A.m_off[(size_t)g * NV]);
            const int slot = uni(A.m_slot[g * NV]);
            Z.key = off + (long long)Z.sl * SLICE_FLOATS;
            Z.enext = A.noise + off + L.fcw + (size_t)Z.sl * SLICE_FLOATS;
            Z.tnext = A.bases + (size_t)slot * A.base_stride + L.fcw + (size_t)Z.sl * SLICE_FLOATS;
            const int ch = (8 * Z.sl + lane) & 31;   // bn2 channel of this lane's activation rows (968 = 8 mod 32)
#pragma unroll
            for (int v = 0; v < NV; v++) {
                Z.mem[v] = g * NV + v;
                Z.scale[v] = A.m_scale[Z.mem[v]];
                Z.s2[v] = HAS_BN ? A.bn[(size_t)Z.mem[v] * 608 + 32 + ch] : 1.0f;
                Z.h2[v] = HAS_BN ? A.bn[(size_t)Z.mem[v] * 608 + 64 + ch] : 0.0f;
                Z.xs[v] = y2 + (size_t)Z.mem[v] * 3872 + 968 * Z.sl;
                Z.acc[v][0] = Z.acc[v][1] = f32x2{0.0f, 0.0f};
                Z.fold[v][0] = Z.fold[v][1] = f32x2{0.0f, 0.0f};
                Z.xv[v] = Z.xn[v] = 0.0f;
            }
            Z.lb = 0;
            Z.nb = FC_SUB0 / W;
        };
        init(SA, uA);
        init(SB, uB);
        const int gb = duo_gb(uA, b_on ? uB : -1);

        // activations: chunk c = rows 64c .. 64c+63 of the unit's slice, one row per lane.  The raw values of chunk c + 1 are
        // requested (asm, like the weight rows) when chunk c starts and turned into relu(bn2(.)) eight row blocks later, long
        // after every load issued before the row loads of those blocks has landed (loads complete in order).
        auto request_x = [&](Side &Z, int c) {
            if (c >= 16) return;
            // derived from voff at the point of use: kept live across the row loop these two offsets are what the W = 4 build spills
            // (and a reload brings an s_waitcnt vmcnt(0) -- every row in flight drained -- into the loop)
            unsigned vo = voff;
            asm volatile("" : "+v"(vo));
            const unsigned xoff = c < 15 ? vo >> 2 : min(vo >> 2, 28u);
#pragma unroll
            for (int v = 0; v < NV; v++)   // xv as an operand: behind take_x's arithmetic, when the old raw values are dead (no copy of a register in flight)
                asm volatile("global_load_dword %[d], %[vo], %[sb]" : [d] "=v"(Z.xn[v]), "+v"(Z.xv[v]) : [vo] "v"(xoff), [sb] "s"(Z.xs[v] + 64 * c));
        };
        auto take_x = [&](Side &Z, int c) {   // xv = chunk c from the raw values in xn
#pragma unroll
            for (int v = 0; v < NV; v++) {
                asm volatile("" : "+v"(Z.xn[v]));   // not before this point of the asm sequence (the value lands asynchronously)
                float t = Z.xn[v];
                if (HAS_BN) {
                    t = t * Z.s2[v];
                    t = t + Z.h2[v];
                }
                t = t > 0.0f ? t : 0.0f;
                Z.xv[v] = (c < 15 || lane < 8) ? t : 0.0f;
            }
        };
        // A rolling window of W rows per stream: row i of a block is consumed and its registers are refilled at once with row i
        // of the next block, so W rows per stream stay in flight and no register is ever copied.  The block after a unit's
        // last one is fetched too and never used: fc bias, bn3 and the output layer follow the fc matrix, and the allocations are
        // padded by a block for small action counts (engine.hip OVERFETCH_FLOATS).
        f32x4 eA[W], tA[W], eB[W], tB[W];
#pragma unroll
        for (int i = 0; i < W; i++) eA[i] = tA[i] = eB[i] = tB[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto refill = [&](Side &Z, f32x4 &e, f32x4 &t, auto ii) {   // row I of the side's next block
            constexpr int I = decltype(ii)::value;
            static_assert(NV == 1 || NV == 2, "accumulator operands");
            if constexpr (NV == 2) gload4_after<(I % 4) * 1024>(e, voff, Z.enext + (I / 4) * 1024, Z.acc[0][0], Z.acc[0][1], Z.acc[1][0], Z.acc[1][1]);
            else gload4_after<(I % 4) * 1024>(e, voff, Z.enext + (I / 4) * 1024, Z.acc[0][0], Z.acc[0][1]);
            gload4<(I % 4) * 1024>(t, voff, Z.tnext + (I / 4) * 1024);
        };
        auto next_block = [&](Side &Z) {
            Z.enext += W * 256;
            Z.tnext += W * 256;
        };
        auto row = [&](Side &Z, const f32x4 &e, const f32x4 &t, int li, const float (&scale)[NV]) {
            const f32x2 elo = {e[0], e[1]}, ehi = {e[2], e[3]}, tlo = {t[0], t[1]}, thi = {t[2], t[3]};
            if constexpr (NV == 2) {
                // an antithetic pair (dne_es_eval: scales +sigma / -sigma exactly): fl(-sigma * eps) = -fl(sigma * eps), so the
                // second member's weight is base - p with the first member's p -- the same two roundings, two multiplies fewer per row
                const f32x2 sc = {scale[0], scale[0]};
                const f32x2 pl = sc * elo, ph = sc * ehi;
                const float x0 = lane_bcast(Z.xv[0], li), x1 = lane_bcast(Z.xv[1], li);
                const f32x2 xx0 = {x0, x0}, xx1 = {x1, x1};
                const f32x2 wl0 = tlo + pl, wh0 = thi + ph, wl1 = tlo - pl, wh1 = thi - ph;
                Z.acc[0][0] = __builtin_elementwise_fma(xx0, wl0, Z.acc[0][0]);
                Z.acc[0][1] = __builtin_elementwise_fma(xx0, wh0, Z.acc[0][1]);
                Z.acc[1][0] = __builtin_elementwise_fma(xx1, wl1, Z.acc[1][0]);
                Z.acc[1][1] = __builtin_elementwise_fma(xx1, wh1, Z.acc[1][1]);
            } else {
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const float x = lane_bcast(Z.xv[v], li);
                    const f32x2 xx = {x, x}, sc = {scale[v], scale[v]};
                    f32x2 pl = sc * elo, ph = sc * ehi;       // base + scale * noise, two roundings, then one fused multiply-add
                    f32x2 wl = tlo + pl, wh = thi + ph;
                    Z.acc[v][0] = __builtin_elementwise_fma(xx, wl, Z.acc[v][0]);
                    Z.acc[v][1] = __builtin_elementwise_fma(xx, wh, Z.acc[v][1]);
                }
            }
        };
        auto end_block = [&](Side &Z) {
            next_block(Z);
            if (Z.lb % BPC == BPC - 1 && Z.lb + 1 < NBLK) {
                take_x(Z, Z.lb / BPC + 1);
                request_x(Z, Z.lb / BPC + 2);
            }
            Z.lb++;
            if (Z.lb == Z.nb) {   // end of a sub-slice (oracle fc_raw): the chain joins the quarter's fold and starts again from 0
                const bool first = Z.nb == FC_SUB0 / W;
                if constexpr (FOLD_LDS) {
                    const int side = &Z == &SB ? 1 : 0;
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        f32x4 a4 = {Z.acc[v][0][0], Z.acc[v][0][1], Z.acc[v][1][0], Z.acc[v][1][1]};
                        if (!first) {
                            const f32x4 f4 = sw_fold[wv][side][v][lane];
                            a4 = f32x4{f4[0] + a4[0], f4[1] + a4[1], f4[2] + a4[2], f4[3] + a4[3]};
                        }
                        sw_fold[wv][side][v][lane] = a4;
                        Z.acc[v][0] = Z.acc[v][1] = f32x2{0.0f, 0.0f};
                    }
                } else {
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) {
                        Z.fold[v][hf] = first ? Z.acc[v][hf] : Z.fold[v][hf] + Z.acc[v][hf];
                        Z.acc[v][hf] = f32x2{0.0f, 0.0f};
                    }
                }
                Z.nb += FC_SUBN / W;
            }
        };
        auto fill = [&](auto sa, auto sb) {   // the first block of the selected sides; everything in flight has landed afterwards
            constexpr bool DA = decltype(sa)::value, DB = decltype(sb)::value;
            auto one = [&](auto ii) {
                constexpr int I = decltype(ii)::value;
                if (DA) refill(SA, eA[I], tA[I], ii);
                if (DB) refill(SB, eB[I], tB[I], ii);
            };
            static_for<W>(one);
            if (DA) next_block(SA);
            if (DB) next_block(SB);
#pragma unroll
            for (int i = 0; i < W; i++) wait_rows<0>(eA[i], tA[i], eB[i], tB[i]);
        };
        auto run = [&](auto da, auto db, int n) {   // n row blocks of the selected sides
            constexpr bool DA = decltype(da)::value, DB = decltype(db)::value;
            // before row i is read, the loads issued after its own are the (W - 1) other rows of every active stream
            constexpr int PENDING = (W - 1) * 2 * ((DA ? 1 : 0) + (DB ? 1 : 0));
            float scA[NV], scB[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) { scA[v] = SA.scale[v]; scB[v] = SB.scale[v]; }
            for (int k = 0; k < n; k++) {
                const int liA = (SA.lb % BPC) * W, liB = (SB.lb % BPC) * W;
                auto one = [&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    wait_rows<PENDING>(eA[I], tA[I], eB[I], tB[I]);
                    if (DA) row(SA, eA[I], tA[I], liA + I, scA);
                    if (DB) row(SB, eB[I], tB[I], liB + I, scB);
                    __builtin_amdgcn_sched_barrier(0);
                    if (DA) refill(SA, eA[I], tA[I], ii);
                    if (DB) refill(SB, eB[I], tB[I], ii);
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_for<W>(one);
                if (DA) end_block(SA);
                if (DB) end_block(SB);
                if constexpr (SWEEP) sw_tick();   // the workgroup's four waves advance one row block at a time
            }
        };
        request_x(SA, 0);
        fill(std::true_type{}, std::false_type{});
        take_x(SA, 0);
        request_x(SA, 1);
        run(std::true_type{}, std::false_type{}, gb);                 // A alone until B's rows come into reach
        if (b_on) {
            request_x(SB, 0);
            fill(std::false_type{}, std::true_type{});
            take_x(SB, 0);
            request_x(SB, 1);
            run(std::true_type{}, std::true_type{}, NBLK - gb);       // table lock-step
            run(std::false_type{}, std::true_type{}, gb);             // B finishes alone
        }
#pragma unroll
        for (int i = 0; i < W; i++) wait_rows<0>(eA[i], tA[i], eB[i], tB[i]);   // the over-fetched block: nothing in flight into dead registers
        auto store = [&](Side &Z) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
                f4a o = {Z.fold[v][0][0], Z.fold[v][0][1], Z.fold[v][1][0], Z.fold[v][1][1]};
                if constexpr (FOLD_LDS) {
                    const f32x4 f4 = sw_fold[wv][&Z == &SB ? 1 : 0][v][lane];
                    o = f4a{f4[0], f4[1], f4[2], f4[3]};
                }
                *(f4a *)(y3t + ((size_t)Z.mem[v] * 4 + Z.sl) * 256 + lane * 4) = o;
            }
        };
        store(SA);
        if (b_on) store(SB);
        }   // rounds
        if constexpr (SWEEP)
            for (int i = 0; i < sw_tail; i++) sw_tick();
#ifdef DNE_PHASE_CLOCK
        if (tk_on && lane == 0) {
            long long *pl = g_duo_plan[tk_wg][wv];
            pl[3] = tk_i; pl[5] = (long long)__builtin_amdgcn_s_memtime(); pl[7] = (long long)wall_clock64();
        }
        tk_on = false;
#endif
    }
}

// ------------------------------------------------------- k_fc_ring (round 5): the workgroup's noise rows through an LDS ring
// What k_fc_duo waits for (profiles/r05_duo_tick_clock.json, r05_pmc_fc_duo_mem.json): a tick of its table
// timeline costs 0.45 us + 0.12 us per streaming unit -- a fixed HBM round trip (every row is consumed exactly one tick after it was
// requested, and the unit at the front of the timeline misses every cache) plus the CU's vector-memory path at ~56 B/clk for the 16 KB
// each unit pulls through it, although the eight units of a workgroup read the SAME 4096-float window of the table within a tick.
//
// Here the window lives in LDS: a ring of RING_SLOTS segments of 2048 floats (one tick of the timeline each; a mirror of slot 0 behind
// the last slot: a block that starts in the last slot runs on into it).  One workgroup = NW compute waves = NW units that follow each
// other in table order, ONE unit per wave (two waves per SIMD hide each other's issue latency; a wave's only loop with loads in flight
// is entered behind a full drain and left into one), plus a LOADER wave that fills the ring by LDS-DMA (global_load_lds_dwordx4: no
// register round trip) RING_PD ticks ahead with three ticks of vmcnt budget of its own -- vmcnt retires in order, so a DMA issued by
// a wave that also streams base rows is forced out one tick later and a tick could never be shorter than the HBM round trip of the
// segment being fetched (the first two forms of this kernel ticked at >= 0.57 us with a single unit streaming).  Every table row
// passes the vector-memory path once per workgroup instead of once per unit (72 KB per full tick instead of 128 KB).
//
// A unit's row starts at an arbitrary float of the table (es.py:67 draws any integer), so the LDS reads cannot be 16-byte reads; lane
// l takes columns l, l+64, l+128, l+192 instead -- consecutive lanes read consecutive banks at any alignment (two ds_read2st64_b32
// per row) -- and the base rows come from a copy of the fc matrix whose rows are stored in that order (k_theta_perm, once per
// evaluation), so that they stay one 16-byte load per lane and row.  Activations arrive finished (relu(bn2(y2)): k_conv12's act2 /
// k_y2_activate), a chunk of 64 rows goes to the wave's own corner of LDS by LDS-DMA eight ticks ahead and a row reads its pair
// (x0, x1) back as one broadcast ds_read2_b32: v_pk_fma takes either half for both lanes (op_sel) -- no v_readlane, and unlike
// scalar loads (s_load shares lgkmcnt with ds_read and returns out of order) these reads keep the counted waits exact.
// Which lane holds which column, which wave fetches what: a schedule.  Every output's chain still runs over k in order from zero, the
// sub-slices fold left, k_out combines the slices (oracle fc_raw) -- same bits.
constexpr int RING_SLOTS = 7, RING_SEG = 2048, RING_PD = 5;   // slots (+ one mirror of slot 0 behind the last), floats per segment, prefetch distance in ticks

__global__ __launch_bounds__(256) void k_theta_perm(const float *__restrict__ fcw /*[rows][256]*/, float *__restrict__ out /*[rows + 16][256]*/, int rows) {
    const int r = blockIdx.x, t = threadIdx.x;   // out[r][4 l + j] = in[r][l + 64 j]; the rows behind the matrix (over-fetched, never used) are zero
    out[(size_t)r * 256 + t] = r < rows ? fcw[(size_t)r * 256 + (t >> 2) + 64 * (t & 3)] : 0.0f;
}

// out[i] = fl(sigma * table[i]) over the whole table (+ its over-fetch padding): k_fc_ring<true>'s source, once per sigma
__global__ __launch_bounds__(256) void k_scale_table(const float *__restrict__ in, float *__restrict__ out, size_t n4, float sigma) {
    const f32x4 *i4 = (const f32x4 *)in;
    f32x4 *o4 = (f32x4 *)out;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = i4[i];
        o4[i] = f32x4{sigma * v[0], sigma * v[1], sigma * v[2], sigma * v[3]};
    }
}

// the same finishing touch for windows whose convolutions ran as other kernels (k_conv1 + k_conv2): y2 <- relu(bn2(y2)) in place
__global__ __launch_bounds__(256) void k_y2_activate(FwdArgs A, const int *__restrict__ list, int gsize, float *__restrict__ y2) {
    const int g = list ? list[blockIdx.x / gsize] : blockIdx.x / gsize, member = g * gsize + blockIdx.x % gsize;
    if (A.done && A.done[member]) return;
    const float *bn = A.bn + (size_t)member * 608;
    float *y = y2 + (size_t)member * 3872;
    for (int i = threadIdx.x; i < 3872; i += 256) {
        float t = y[i] * bn[32 + (i & 31)];
        t = t + bn[64 + (i & 31)];
        y[i] = t > 0.0f ? t : 0.0f;
    }
}

// 16 bytes per lane from global memory straight into LDS at lds_dst + lane * 16 (wave-uniform destination); counted by vmcnt like any
// other vector load, visible to other waves' ds_reads after the issuing wave's counted wait and a barrier
// Cache policy of the ring's DMAs.  A noise row passes an XCD's L2 once per workgroup (its re-use is in LDS now) while the 3.96 MB of
// base rows every unit re-reads barely fit the 4 MB L2, so streaming the noise (-DDNE_RING_EPS_MOD='" nt"') looked right -- and does
// what it should: L2 hit rate 68 -> 79 %, fabric reads 3.9 -> 2.6 GB per 2500-pair launch (profiles/r05_ring_nt_ab.json) -- but the
// launch takes 0.882 instead of 0.873 ms and a generation 227.4 instead of 223.9 ms same-box: the kernel does not wait for L2 misses.
// The plain form stays.
#ifndef DNE_RING_EPS_MOD
#define DNE_RING_EPS_MOD ""
#endif
__device__ __forceinline__ void glds16(const float *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" DNE_RING_EPS_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// row I of a unit's current block: floats vaddr/4 + 256 I + {0, 64, 128, 192} of the ring (the lane's four columns); lands asynchronously
template <int I>
__device__ __forceinline__ void ring_row(f32x2 &lo, f32x2 &hi, unsigned vaddr) {
    asm volatile("ds_read2st64_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read2st64_b32 %1, %2 offset0:%5 offset1:%6"
                 : "=&v"(lo), "=&v"(hi) : "v"(vaddr), "n"(4 * I), "n"(4 * I + 1), "n"(4 * I + 2), "n"(4 * I + 3));   // early-clobber: the second read still needs vaddr
}

// row I's activations (x0, x1): every lane reads the same eight bytes (a broadcast, conflict-free); lands asynchronously
template <int I>
__device__ __forceinline__ void ring_x(f32x2 &x, unsigned vaddr) {   // (member 0's plane, member 1's 64 floats behind it)
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(x) : "v"(vaddr), "n"(I), "n"(64 + I));
}

// At most four waves per SIMD (amdgpu_waves_per_eu: the register allocation is rounded up to 104 per lane): the workgroup's nine take
// 2 / 2 / 2 / 3, a second workgroup of this kernel finds no SIMD for its third wave -- one per CU, k_fc_duo's FAT -- and a
// 199-register k_conv12 wave still fits beside the three.  (A plain launch bound of (576, 1) makes the compiler pad to 129 registers,
// three waves per SIMD: then nothing fits beside them.  An accumulation register named in an asm, k_fc_duo's way, splits the compiler's
// budget 84 / 84 and it spills to AGPRs -- copying registers that loads are still writing.)
// PRE (round 6): every pair of an ES evaluation is perturbed at the same sigma (es.py:412-419), so fl(sigma * eps) is a property of the TABLE: the ring's
// DMAs then stream a copy of the table scaled once per sigma (k_scale_table; ring_noise = that copy) and a row's four v_pk_mul_f32 of twenty packed
// instructions per step disappear -- the same single rounding, made earlier: same bits.  PRE = false: ring_noise = the table itself.
template <bool PRE, int NW>
__global__ __launch_bounds__((NW + 1) * 64) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_fc_ring(FwdArgs A, const int *__restrict__ order, int n_units, const float *__restrict__ y2,
                                                     float *__restrict__ y3t, const float *__restrict__ theta_perm, int flags, const float *__restrict__ ring_noise) {
    constexpr int NV = 2, W = 8, NBLK = 968 / W, BPC = 64 / W, R = RING_SLOTS, PD = RING_PD;
    constexpr int ND = RING_SEG / 256;      // LDS-DMA instructions per segment (1 KB each), all issued by the loader wave
    static_assert(RING_SEG == W * 256, "one segment = one tick of the timeline");
    // ONE shared object, the ring first (LDS-DMA destinations are 16-bit offsets: the eight 8 KB slots end at 64 KB)
    struct Lds {
        float ring[(R + 1) * RING_SEG];
        float xbuf[NW][2][2][64];        // per wave: two chunks of 64 rows of activations, one plane per member of the pair
        long long key[2][NW];            // the waves' units' table addresses (or -1), double-buffered by item parity
    };
    __shared__ __attribute__((aligned(16))) Lds S;
    float (&ring)[(R + 1) * RING_SEG] = S.ring;
    long long (&sw_key)[2][NW] = S.key;
    const int tid = threadIdx.x, wv = uni(tid >> 6), lane = tid & 63;
    const bool loader = wv == NW;
    const unsigned voff = lane * 16, vlane4 = lane * 4;
    const unsigned lds_ring = (unsigned)(size_t)ring, lds_x = (unsigned)(size_t)&S.xbuf[loader ? 0 : wv][0][0][0];
    const Layout &L = A.L;
    {
        const int prio = (flags >> 9) & 3;
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    }
    const int n_items = (n_units + NW - 1) / NW;
    int par = 0;
    DNE_WG_BEGIN;
#ifdef DNE_PHASE_CLOCK
    bool tk_on = false;
    int tk_i = 0;
    const int tk_wg = blockIdx.x / 4;
#endif
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int uu = !loader && item * NW + wv < n_units ? uni(order[item * NW + wv]) : -1;
        if (uu >= 0 && A.done) {   // units of finished pairs are dropped
            int all_done = 1;
#pragma unroll
            for (int v = 0; v < NV; v++) all_done &= uni(A.done[(uu >> 2) * NV + v]) != 0;
            if (all_done) uu = -1;
        }
        const bool ok = uu >= 0;
        // absolute float index of the unit's first weight in the noise table
        const long long key = ok ? uni64(A.m_off[(size_t)(uu >> 2) * NV]) + (long long)L.fcw + (long long)(uu & 3) * SLICE_FLOATS : -1;
        if (lane == 0 && !loader) sw_key[par][wv] = key;
        __syncthreads();
        long long K0 = 0x7fffffffffffffffll, kmax = -1;
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const long long k = sw_key[par][j];
            if (k >= 0) { K0 = min(K0, k); kmax = max(kmax, k); }
        }
        par ^= 1;
        K0 = uni64(K0); kmax = uni64(kmax);
        if (kmax < 0) continue;                       // every unit of this item belongs to a finished pair (the same answer in every wave)
        K0 &= ~3ll;                                   // 16-byte-aligned DMA sources
        const int tmax = (int)((kmax - K0) >> 11) + NBLK;
        const float *seg0 = ring_noise + K0;          // segment 0
        int tau = 0, slot = 0;                        // the workgroup's time in ticks (= segments) and the ring slot of segment tau
        // segment j (clamped to the last one anybody reads) into slot sj: the loader wave's eight 1 KB pieces
        auto dma_seg = [&](int j, int sj) {
            const float *src = seg0 + (size_t)min(j, tmax) * RING_SEG;
            const unsigned dst = lds_ring + (unsigned)(sj * RING_SEG) * 4u;
#pragma unroll
            for (int d = 0; d < ND; d++) glds16(src + 256 * d, voff, dst + 1024 * d);
        };
        if (loader) {
#pragma unroll
            for (int j = 0; j < PD; j++) dma_seg(j, j);
            dma_seg(0, R);   // a block that starts in the last slot runs on into the mirror of slot 0 behind it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int q = ok ? (int)((key - K0) >> 11) : 0;               // the unit's first row sits dlt floats into segment q
        const int dlt = ok ? (int)(key - K0) - q * RING_SEG : 0;
#ifdef DNE_PHASE_CLOCK
        tk_on = !loader && blockIdx.x % 4 == 0 && tk_wg < DUO_TICK_WGS && item == (int)(blockIdx.x + gridDim.x);
        tk_i = 0;
        if (tk_on && lane == 0) {
            long long *pl = g_duo_plan[tk_wg][wv];
            pl[0] = q; pl[1] = ok ? NBLK : 0; pl[2] = tmax; pl[3] = 0;
            pl[4] = (long long)__builtin_amdgcn_s_memtime(); pl[6] = (long long)wall_clock64();
        }
#endif
        // the end of a tick: the barrier.  The loader requests segment tau + PD first (and its mirror copy when it lands in slot 0) and
        // lets the DMAs of this and the two previous ticks stay in flight: what it requested three ticks ago has landed when it arrives
        // at the barrier, every wave can read it from the next tick on -- the tick of its first reader (PD = 5).
        int mir_age = 0;   // ticks since the last mirror copy was requested (the prologue's: everything was waited for)
        auto tick_end = [&]() {
#ifdef DNE_PHASE_CLOCK
            long long tk_b = 0;
            if (tk_on) { __builtin_amdgcn_sched_barrier(0); tk_b = (long long)__builtin_amdgcn_s_memtime(); }
#endif
            if (loader) {
                int sj = slot + PD;
                if (sj >= R) sj -= R;
                dma_seg(tau + PD, sj);
                const bool mir = sj == 0;
                if (mir) { dma_seg(tau + PD, R); mir_age = 0; } else mir_age++;
                // this tick's requests and those of the two ticks before stay in flight (a mirror copy among them: one more segment;
                // R = 7: at most one); what was requested three ticks ago has landed at the barrier
                if (mir_age < 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * ND) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * ND) : "memory");
            }
            __builtin_amdgcn_s_barrier();
#ifdef DNE_PHASE_CLOCK
            if (tk_on) {
                const long long tk_a = (long long)__builtin_amdgcn_s_memtime();
                if (lane == 0 && tk_i < DUO_TICK_MAX) { g_duo_tick[tk_wg][wv][tk_i][0] = tk_b; g_duo_tick[tk_wg][wv][tk_i][1] = tk_a; }
                tk_i++;
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            tau++;
            slot = slot + 1 == R ? 0 : slot + 1;
        };
        auto idle_until = [&](int T) { while (tau < T) tick_end(); };
        idle_until(q);
        if (ok) {
            const int g = uu >> 2, sl = uu & 3;
            const float *tnext = theta_perm + (size_t)sl * SLICE_FLOATS;   // (every member of an ES evaluation has base slot 0: the engine checks)
            const int mem0 = g * NV;
            const float scale0 = A.m_scale[mem0];  // an antithetic pair: +sigma, -sigma exactly
            f32x2 acc[NV][2], fold[NV][2];
#pragma unroll
            for (int v = 0; v < NV; v++) acc[v][0] = acc[v][1] = fold[v][0] = fold[v][1] = f32x2{0.0f, 0.0f};
            // activations (the header above): chunk c of 64 rows, both members' planes, into buffer c & 1 of the wave's corner of LDS
            const float *xg0 = y2 + (size_t)mem0 * 3872 + 968 * sl, *xg1 = xg0 + 3872;
            auto request_x = [&](int c) {   // chunk c = rows 64 c .. 64 c + 63 (the last chunk has eight: the other lanes re-read its last row)
                unsigned vo = vlane4;
                asm volatile("" : "+v"(vo));
                const unsigned xoff = c < 15 ? vo : min(vo, 28u);
                const unsigned dst = lds_x + (unsigned)((c & 1) * 512);
                unsigned keep;   // 4 bytes per lane into the chunk's two planes (no register in between: nothing for the compiler to copy early)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(xoff), "s"(xg0 + 64 * c), "s"(xg1 + 64 * c), "s"(dst), "s"(dst + 256) : "memory");
            };
            request_x(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            f32x4 t[W];
            f32x2 e[4][2];   // the noise values of four consecutive rows (two pairs: one being computed, one on its way), columns (l, l+64) and (l+128, l+192)
#pragma unroll
            for (int i = 0; i < W; i++) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; i++) e[i][0] = e[i][1] = f32x2{0.f, 0.f};
            auto refill = [&](auto ii) {   // row I of the next block of base rows, into the registers of the row just consumed
                constexpr int I = decltype(ii)::value;
                gload4_theta_after<(I % 4) * 1024>(t[I], voff, tnext + (I / 4) * 1024, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
            };
            // the first block of base rows: landed before the loop is entered (the compiler may copy registers at the loop's entry; it
            // must not copy one that a load is still writing)
            static_for<W>(refill);
            tnext += W * 256;
#pragma unroll
            for (int i = 0; i < W; i++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[i]));
            // before the later row of a pair is read, the loads issued after its own are the W - 2 rows of the other pairs (a lower
            // bound: a chunk of activations only makes the wait stricter)
            constexpr int PENDING2 = W - 2;   // ... counted from the later row of a pair
            int nb = FC_SUB0 / W;   // row block at which the current sub-slice ends
            f32x2 xr[4];   // (x0, x1) of four consecutive rows: a row's pair is still needed by its stage B when the read two rows ahead goes out
#pragma unroll
            for (int i = 0; i < 4; i++) xr[i] = f32x2{0.f, 0.f};
            for (int lb = 0; lb < NBLK; lb++) {
                const unsigned va = vlane4 + (lds_ring + (unsigned)(slot * RING_SEG + dlt) * 4u);
                const unsigned vx = lds_x + (unsigned)(((lb >> 3) & 1) * 512 + (lb & 7) * 32);   // this tick's eight rows of the current chunk, member 0's plane (every lane the same address)
                if ((lb & 7) == 0 && lb + 8 < NBLK) request_x((lb >> 3) + 1);   // the next chunk, eight ticks ahead
                auto reads = [&](auto ii) {   // row I's noise values and activations
                    constexpr int I = decltype(ii)::value;
                    ring_row<I>(e[I % 4][0], e[I % 4][1], va);
                    ring_x<I>(xr[I % 4], vx);
                };
                reads(std::integral_constant<int, 0>{});
                reads(std::integral_constant<int, 1>{});
                // Two rows per step: both rows' operands waited for at once, the next pair's LDS reads sent out, then the two rows' arithmetic
                // in one scheduling region -- two independent multiply -> add chains ahead of the fmas instead of one (a wave alone on its
                // SIMD cycle otherwise sits out that latency row after row: 0.17 us per row on the tick clock of the first 8-wave form) --
                // and each row's registers refilled right behind the step that consumed them, as before: a schedule in which the refill
                // ran ahead of or behind its row made the register allocator rotate the base-row registers at the loop's back edge (copies
                // of registers that loads were still writing: a memory fault on the box).
                static_for<W / 2>([&](auto kk) {
                    constexpr int K = decltype(kk)::value, I0 = 2 * K, I1 = 2 * K + 1, S0 = I0 % 4, S1 = I1 % 4;
                    // this pair's LDS reads went out behind the previous step's wait and had that pair's arithmetic to land
                    asm volatile("s_waitcnt vmcnt(%[n]) lgkmcnt(0)" : "+v"(t[I0]), "+v"(t[I1]), "+v"(e[S0][0]), "+v"(e[S0][1]), "+v"(e[S1][0]), "+v"(e[S1][1]), "+v"(xr[S0]), "+v"(xr[S1]) : [n] "n"(PENDING2));
                    if constexpr (I1 + 2 < W) {   // the next pair is on its way while this one is computed
                        reads(std::integral_constant<int, I0 + 2>{});
                        reads(std::integral_constant<int, I1 + 2>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {
                        const f32x2 sc = {scale0, scale0};
                        // fl(-sigma eps) = -fl(sigma eps): the second member's weight is base - p with the first member's p (es.py:415, 419)
                        const f32x2 tl0 = {t[I0][0], t[I0][1]}, th0 = {t[I0][2], t[I0][3]}, tl1 = {t[I1][0], t[I1][1]}, th1 = {t[I1][2], t[I1][3]};
                        f32x2 pl0 = e[S0][0], ph0 = e[S0][1], pl1 = e[S1][0], ph1 = e[S1][1];
                        if constexpr (!PRE) { pl0 = sc * pl0; ph0 = sc * ph0; pl1 = sc * pl1; ph1 = sc * ph1; }
                        const f32x2 a0 = tl0 + pl0, a1 = th0 + ph0, a2 = tl0 - pl0, a3 = th0 - ph0;
                        const f32x2 b0 = tl1 + pl1, b1 = th1 + ph1, b2 = tl1 - pl1, b3 = th1 - ph1;
                        const f32x2 xa0 = {xr[S0][0], xr[S0][0]}, xa1 = {xr[S0][1], xr[S0][1]}, xb0 = {xr[S1][0], xr[S1][0]}, xb1 = {xr[S1][1], xr[S1][1]};
                        acc[0][0] = __builtin_elementwise_fma(xa0, a0, acc[0][0]);
                        acc[0][1] = __builtin_elementwise_fma(xa0, a1, acc[0][1]);
                        acc[1][0] = __builtin_elementwise_fma(xa1, a2, acc[1][0]);
                        acc[1][1] = __builtin_elementwise_fma(xa1, a3, acc[1][1]);
                        acc[0][0] = __builtin_elementwise_fma(xb0, b0, acc[0][0]);
                        acc[0][1] = __builtin_elementwise_fma(xb0, b1, acc[0][1]);
                        acc[1][0] = __builtin_elementwise_fma(xb1, b2, acc[1][0]);
                        acc[1][1] = __builtin_elementwise_fma(xb1, b3, acc[1][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    refill(std::integral_constant<int, I0>{});
                    refill(std::integral_constant<int, I1>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                tnext += W * 256;
                if (lb + 1 == nb) {   // end of a sub-slice (oracle fc_raw): the chain joins the quarter's left fold and starts again from 0
                    const bool first = nb == FC_SUB0 / W;
#pragma unroll
                    for (int v = 0; v < NV; v++)
#pragma unroll
                        for (int hf = 0; hf < 2; hf++) {
                            fold[v][hf] = first ? acc[v][hf] : fold[v][hf] + acc[v][hf];
                            acc[v][hf] = f32x2{0.0f, 0.0f};
                        }
                    nb += FC_SUBN / W;
                }
                tick_end();
            }
#pragma unroll
            for (int i = 0; i < W; i++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[i]));   // the over-fetched block, the last DMAs
#pragma unroll
            for (int v = 0; v < NV; v++) {
                float *o = y3t + ((size_t)(mem0 + v) * 4 + sl) * 256 + lane;
                o[0] = fold[v][0][0]; o[64] = fold[v][0][1]; o[128] = fold[v][1][0]; o[192] = fold[v][1][1];
            }
        }
        idle_until(tmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loader's last DMAs (nobody reads them): not into the next item's ring
#ifdef DNE_PHASE_CLOCK
        if (tk_on && lane == 0) {
            long long *pl = g_duo_plan[tk_wg][wv];
            pl[3] = tk_i; pl[5] = (long long)__builtin_amdgcn_s_memtime(); pl[7] = (long long)wall_clock64();
        }
        tk_on = false;
#endif
    }
    DNE_WG_END(1);
}

// ------------------------------------------------------- fc of the reference pass on the matrix cores
// Virtual batch norm pushes F reference frames through every member's perturbed network (policies.py:399):
// per member a [F x 3872] x [3872 x 256] GEMM with member-unique weights.  16 workgroups per member = 4 k-slices
// x 4 column slabs (kept on one XCD so the member's activations are shared through its L2); wave w owns 16
// columns and all F frames (MT = F/16 accumulators); activations stream through a double-buffered LDS tile of
// 44 k-values, weights are formed per lane as theta + sigma*eps straight into the B operand.  The MFMA's
// k-ordered fmaf chain per slice + the ((s0+s1)+(s2+s3)) + bias combine (k_bn3_partials) are the oracle's order.
template <int MT>
__global__ __launch_bounds__(256, 2) void k_fc_ref(FwdArgs A, int n_local, int member0, int F /* reference frames per member: MT * 16 * frame groups */,
                                                const float *__restrict__ y2, float *__restrict__ y3p /*[n_local][4][F][256]*/) {
    // One workgroup of 4 waves per (member, quarter, group of MT * 16 frames); wave w owns columns 64w .. 64w+63 as four interleaved
    // 16-column MFMA tiles (tile c = columns 64w + 4*lane + c), so a lane's four B operands of a k-row are one 16-byte load.  8-row
    // stages through a double-buffered LDS tile.
    // MT <= 4: a wave's 16 MT accumulator registers and as many for the running fold over the sub-slices (oracle fc_raw) leave two
    // waves per SIMD.  128 reference frames therefore run as TWO frame groups of 64 (MT = 4) per (member, quarter): the pair streams
    // the same weight rows side by side on one XCD (the second reader finds them in L2), the activations are staged once.
    // (Round 3 measured the alternatives at 128 frames in one workgroup: 8 waves x 32 columns 2.7 ms per chunk of 512 members,
    // two 4-wave workgroups per quarter split by columns 3.0 ms, against 1.7 ms for the round-2 kernel without the fold.)
    constexpr int FG = MT * 16, KC = 8, XS = KC + 2, NST = 968 / KC, KK = KC / 4;   // XS = 10: lanes (frame, k) of a half-wave hit 32 distinct banks
    constexpr int LD = (FG * KC + 255) / 256;
    // Stages travel in units of two (one where a sub-slice has an odd stage left): one barrier and one round of loads per unit --
    // with 64 frames per workgroup a single 8-row stage is only 32 MFMAs per wave, too little to hide a barrier behind.
    __shared__ float xs[4][FG * XS];                            // stage s lives in buffer s & 3: the current unit and the next
    __shared__ float bn2[64];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, lp = lane & 15, lk = lane >> 4;
    const Layout &L = A.L;
    const int ngrp = F / FG;                                    // frame groups per member (1, or 2 at 128 frames)
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;          // the workgroups of a member stay on one XCD (block b -> XCD b % 8)
    const int sl = q & 3, fg = (q >> 2) % ngrp, mloc = (q / (4 * ngrp)) * 8 + x;
    if (mloc >= n_local) return;
    const int member = member0 + mloc;
    const float sc = A.m_scale[member];
    const int kbeg = 968 * sl, col0 = 64 * wv + 4 * lp;
    const float *eps = A.noise + A.m_off[member] + L.fcw + (size_t)(kbeg + lk) * 256 + col0;
    const float *th = A.bases + (size_t)A.m_slot[member] * A.base_stride + L.fcw + (size_t)(kbeg + lk) * 256 + col0;
    const float *ysrc = y2 + ((size_t)mloc * F + (size_t)fg * FG) * Y2_PAD_ROW + kbeg;   // padded rows (k_conv2_ref<.., true>)
    if (tid < 64) bn2[tid] = A.bn[(size_t)member * 608 + 32 + tid];   // scale[32] then shift[32]
    // Loads and LDS stores of a unit are unconditional: a unit whose second stage does not exist (the odd stage that ends a
    // sub-slice, the end of the quarter) fetches a clamped stage into a buffer nobody reads.  With the loads under `if (u < n)`
    // the compiler could not pair them with their waits and put s_waitcnt vmcnt(0) between the two stages' loads.
    // Round 4 (the profiling build's phase clock, tools/ref_phase_clock.py): of a unit's 2.4 us the 64 MFMAs took 0.9 -- issuing the
    // next unit's twelve loads took 0.7 and storing it (LDS writes, weights) 0.6: beside the CU's other workgroup streaming MFMAs an
    // instruction outside one's own MFMA stream is issued about once per MFMA.  Now nothing but the barrier sits between two units'
    // MFMAs: the loads run TWO units ahead through register sets the loop body alternates between, so storing the next unit (its
    // LDS buffers and its weights, a second set) depends on nothing the current unit does, and both -- load issue and store -- are
    // dealt in pieces behind the four 8-MFMA chunks of the unit's first stage (one fence per chunk).  The body holds four units:
    // the register set in flight across the back edge is renamed there by the allocator (v_mov_b64 behind a wait), a quarter of
    // the sets instead of half.  Tried and dropped: inline-asm loads into `+v` variables (renamed all the same, without the wait:
    // registers copied before their data arrives), hard-register constraints (copies around every asm), accumulation registers named
    // in the asm text (the compiler uses them itself as soon as the kernel declares any).
    constexpr bool FULL = (FG * KC) % 256 == 0;                 // every thread stages LD activations per stage
    struct Ld { float yr[2][LD]; f4u er[2][KK]; f4a tr[2][KK]; };
    // addresses as (wave-uniform 64-bit base) + (per-lane 32-bit byte offset, fixed for the kernel): the loads take the base from
    // scalar registers and need no vector address arithmetic (16 v_lshl_add_u64 per unit otherwise -- every instruction between
    // two MFMAs costs its issue slot, section 4b of DESIGN.md)
    typedef const __attribute__((address_space(1))) char *gptr;   // global, not generic: an integer cast alone would make these flat loads
    const gptr eps_u = (gptr)uni64((long long)(A.noise + A.m_off[member] + L.fcw + (size_t)kbeg * 256));
    const gptr th_u = (gptr)uni64((long long)(A.bases + (size_t)A.m_slot[member] * A.base_stride + L.fcw + (size_t)kbeg * 256));
    const gptr y_u = (gptr)uni64((long long)ysrc);
    const unsigned woff = (unsigned)(lk * 256 + col0) * 4u;
    unsigned yoff[LD];
#pragma unroll
    for (int j = 0; j < LD; j++) {
        const int e = tid + 256 * j;
        yoff[j] = (unsigned)((e / KC) * Y2_PAD_ROW + e % KC) * 4u;
    }
    auto sbase = [](gptr p) { asm volatile("" : "+s"(p)); return p; };   // the uniform sum stays one scalar value: without this the compiler
                                                                          // hoists (base + lane offset) out of the loop as a 64-bit VGPR pair and adds the rest per load
    auto voff = [](unsigned o) { asm volatile("" : "+v"(o)); return o; };   // ... and the 32-bit offset is widened where it is used (instruction selection works per block)
    auto load_stage = [&](Ld &R, int s0, int u) {
        const int s = s0 + u < NST ? s0 + u : NST - 1;
        const gptr yb = sbase(y_u + (size_t)(s * KC) * 4);
#pragma unroll
        for (int j = 0; j < LD; j++) {
            const int e = tid + 256 * j;
            R.yr[u][j] = FULL || e < FG * KC ? *(const __attribute__((address_space(1))) float *)(yb + voff(yoff[j])) : 0.0f;
        }
        const gptr eb = sbase(eps_u + (size_t)(s * KC) * 1024), tb = sbase(th_u + (size_t)(s * KC) * 1024);
        const unsigned wo = voff(woff);
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {   // 4 rows = 4096 bytes on: past the loads' immediate offset, so a scalar base of its own
            const gptr ebk = kk ? sbase(eb + 4096 * kk) : eb, tbk = kk ? sbase(tb + 4096 * kk) : tb;
            R.er[u][kk] = *(const __attribute__((address_space(1))) f4u *)(ebk + wo);
            R.tr[u][kk] = *(const __attribute__((address_space(1))) f4a *)(tbk + wo);
        }
    };
    auto store_stage = [&](const Ld &R, int s0, int u, float (&wd)[2][KK][4]) {   // activations into their LDS buffer, weights into wd
        const int s = s0 + u < NST ? s0 + u : NST - 1;
#pragma unroll
        for (int j = 0; j < LD; j++) {
            const int e = tid + 256 * j;
            if (FULL || e < FG * KC) {
                const int ch = (kbeg + s * KC + e % KC) & 31;
                float t = R.yr[u][j] * bn2[ch];
                t = t + bn2[32 + ch];
                xs[(s0 + u) & 3][(e / KC) * XS + e % KC] = t > 0.0f ? t : 0.0f;
            }
        }
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        const f32x2v sc2 = {sc, sc};
#pragma unroll
        for (int kk = 0; kk < KK; kk++)
#pragma unroll
            for (int c = 0; c < 4; c += 2) {   // theta + (sigma * eps), two columns per packed instruction (mul, then add: -ffp-contract=off)
                const f32x2v e2 = {R.er[u][kk][c], R.er[u][kk][c + 1]}, t2 = {R.tr[u][kk][c], R.tr[u][kk][c + 1]};
                f32x2v pv, wv2;   // as asm: the compiler splits about half of these into single multiplies and adds
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pv) : "v"(sc2), "v"(e2));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(wv2) : "v"(t2), "v"(pv));
                wd[u][kk][c] = wv2[0];
                wd[u][kk][c + 1] = wv2[1];
            }
    };
    // a unit = two stages, or the odd one that ends a sub-slice (stages 16, 31, 46, ..., 121 end one)
    constexpr int S0 = FC_SUB0 / KC, SN = FC_SUBN / KC;
    auto unit_len = [&](int s) {
        const int end = s < S0 ? S0 : S0 + ((s - S0) / SN + 1) * SN;
        return end - s < 2 ? end - s : 2;
    };
    // MT == 4: the running fold lives in LDS ([16][256] 16-byte words = 64 KB dynamic, lane-contiguous: conflict-free, touched
    // eight times per workgroup) -- as a second register set of the accumulators' size it leaves no room for the second load set
    constexpr bool FOLD_LDS = MT == 4;
    constexpr int FR = FOLD_LDS ? 1 : MT;
    extern __shared__ __attribute__((aligned(16))) float fold_dyn[];
    f32x4 *const fold_s = (f32x4 *)fold_dyn + tid;
    f32x4 acc[MT][4], fold[FR][4];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < FR; m++)
#pragma unroll
        for (int c = 0; c < 4; c++) fold[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    Ld R0, R1;
    float wA[2][KK][4], wB[2][KK][4];
    float aA[2][KK][MT], aB[2][KK][MT];   // a unit's sixteen A operands: requested behind the barrier INSIDE the previous unit, under its second stage
    auto request = [&](float (&a)[2][KK][MT], int s0) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float *xb = xs[(s0 + u) & 3];
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int kk = 0; kk < KK; kk++) a[u][kk][m] = xb[(m * 16 + lp) * XS + 4 * kk + lk];
        }
    };
    int st = 0, s1 = unit_len(0);
    bool first = true;
    load_stage(R0, 0, 0);
    load_stage(R0, 0, 1);
    load_stage(R1, s1, 0);    // unit 1 in flight behind unit 0
    load_stage(R1, s1, 1);
    __syncthreads();          // bn2 visible
    store_stage(R0, 0, 0, wA);
    store_stage(R0, 0, 1, wA);
    __syncthreads();
    request(aA, 0);
    DNE_ACC_DECL;   // profiling build: 0 = first stage's MFMAs with the pieces, 1 = barrier, 2 = next operands requested, 3 = second stage's MFMAs, 4 = fold
    auto step = [&](Ld &Rcur /* in flight: unit s1 */, Ld &Rnext, float (&w)[2][KK][4], float (&wn)[2][KK][4], float (&a)[2][KK][MT],
                    float (&an)[2][KK][MT]) {
        const int n = s1 - st, s2 = s1 + unit_len(s1);
        static_for<2 * KK>([&](auto J) {   // first stage: chunk j = k-group j / 2, position tiles of half j % 2
            constexpr int j = decltype(J)::value, kk = j >> 1, h = j & 1;
#pragma unroll
            for (int m = h * MT / 2; m < (h + 1) * MT / 2; m++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][kk][m], w[0][kk][c], acc[m][c], 0, 0, 0);
            if constexpr (j == 0) load_stage(Rnext, s2, 0);
            else if constexpr (j == 1) load_stage(Rnext, s2, 1);
            else if constexpr (j == 2) store_stage(Rcur, s1, 0, wn);   // into the buffers the unit being computed does not read
            else store_stage(Rcur, s1, 1, wn);
            __builtin_amdgcn_sched_barrier(0);
        });
        DNE_ACC(0);
        __syncthreads();          // the next unit's buffers are complete (its predecessor's readers passed the previous barrier)
        DNE_ACC(1);
        request(an, s1);
        __builtin_amdgcn_sched_barrier(0);
        DNE_ACC(2);
        if (n == 2) {
#pragma unroll
            for (int kk = 0; kk < KK; kk++)
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][kk][m], w[1][kk][c], acc[m][c], 0, 0, 0);
        }
        DNE_ACC(3);
        if (s1 == S0 || (s1 > S0 && (s1 - S0) % SN == 0)) {   // end of a sub-slice: the chains join the quarter's running fold and start again from 0
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if constexpr (FOLD_LDS) {
                        f32x4 *p = fold_s + (m * 4 + c) * 256;
                        if (first) *p = acc[m][c];
                        else *p = *p + acc[m][c];
                    } else {
                        fold[m][c] = first ? acc[m][c] : fold[m][c] + acc[m][c];
                    }
                    acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            first = false;
            DNE_ACC(4);
        }
        st = s1;
        s1 = s2;
    };
    static_assert(S0 % 2 == 0 && SN % 2 == 1 && S0 + 7 * SN == NST && KK == 2 && MT <= 4, "64 units: 8 per sub-slice");
#pragma unroll 1
    for (int it = 0; it < 16; it++) {   // one exit, a whole number of bodies
        step(R1, R0, wA, wB, aA, aB);
        step(R0, R1, wB, wA, aB, aA);
        step(R1, R0, wA, wB, aA, aB);
        step(R0, R1, wB, wA, aB, aA);
    }
    DNE_ACC_STORE_EVERY(5, gridDim.x / 128);
    float *out = y3p + (((size_t)(mloc * 4 + sl) * F) + (size_t)fg * FG) * 256 + col0;
#pragma unroll
    for (int m = 0; m < MT; m++) {
        f32x4 f[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if constexpr (FOLD_LDS) f[c] = fold_s[(m * 4 + c) * 256];
            else f[c] = fold[m][c];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)   // D[row = 4*(l>>4) + r][col = l&15] of tile c -> frame m*16 + 4*lk + r, column col0 + c
            *(f32x4 *)(out + (size_t)(m * 16 + lk * 4 + r) * 256) = f32x4{f[0][r], f[1][r], f[2][r], f[3][r]};
    }
}

// bn3 statistics from the four k-slice partials: y3 = ((p0+p1)+(p2+p3)) + bias per frame, then the batch
// moments over the F frames in frame order (same order as k_bn3_rows); one thread per column.
__global__ __launch_bounds__(256) void k_bn3_partials(FwdArgs A, int member0, int F, const float *__restrict__ y3p) {
    const int mloc = blockIdx.x, member = member0 + mloc, j = threadIdx.x;
    const Layout &L = A.L;
    const float *base = A.bases + (size_t)A.m_slot[member] * A.base_stride;
    const float *eps = A.noise + A.m_off[member];
    const float sc = A.m_scale[member];
    float pbias = sc * eps[L.fcb + j];
    const float bias = base[L.fcb + j] + pbias;
    const float *p = y3p + (size_t)mloc * 4 * F * 256 + j;
    const size_t ss = (size_t)F * 256;
    const float count = (float)F;
    float tot = 0.0f;
    for (int n = 0; n < F; n++) {
        const float s01 = p[(size_t)n * 256] + p[ss + (size_t)n * 256];
        const float s23 = p[2 * ss + (size_t)n * 256] + p[3 * ss + (size_t)n * 256];
        float v = s01 + s23;
        v = v + bias;
        tot = tot + (0.0f + v);
    }
    const float mean = tot / count;
    float totq = 0.0f;
    for (int n = 0; n < F; n++) {
        const float s01 = p[(size_t)n * 256] + p[ss + (size_t)n * 256];
        const float s23 = p[2 * ss + (size_t)n * 256] + p[3 * ss + (size_t)n * 256];
        float v = s01 + s23;
        v = v + bias;
        const float d = v - mean;
        totq = totq + __builtin_fmaf(d, d, 0.0f);
    }
    const float var = totq / count;
    float pb = sc * eps[L.bn3b + j];
    const float beta = base[L.bn3b + j] + pb;
    float pg = sc * eps[L.bn3g + j];
    const float gamma = base[L.bn3g + j] + pg;
    const float inv = 1.0f / sqrtf(var + 1e-3f);
    const float s = inv * gamma;
    const float ms = mean * s;
    A.bn[(size_t)member * 608 + 96 + j] = s;
    A.bn[(size_t)member * 608 + 352 + j] = beta - ms;
    A.bn_mom[(size_t)member * 608 + 96 + j] = mean;
    A.bn_mom[(size_t)member * 608 + 352 + j] = var;
}

struct NoWait {
    __device__ __forceinline__ bool operator()() const { return true; }
};

// ------------------------------------------------------------ fc for the last handful of groups (the speculative tail)
// With one to four groups left the layer is pure latency, and the more CUs pull on it the sooner it is over: 64 workgroups per
// group = (16-column block, quarter), 4 waves each.  A wave takes two sub-slices of the quarter (oracle fc_raw: wave 0 rows 0-247 =
// sub-slices 0 and 1, wave w rows 248 + 240 (w - 1) ...) and holds ALL of its rows in flight at once: lane (rg, cl) loads row
// 4 g + rg of column cl (4 rows x 16 columns per load instruction), perturbs in registers, and runs the two chains of its
// sub-slices, each from 0, around the DPP quad (v_fmac_f32_dpp quad_perm:[j,j,j,j]: row 4 g + j's weight sits in lane j).  The
// four waves work concurrently -- in rounds 1-2 a quarter was one chain that the waves ran in turn -- and the eight sub-slice sums
// meet in LDS for the quarter's left fold.
template <int NV>
struct QuadLds {
    __attribute__((aligned(16))) float xs[NV][968];
    float comb[8][NV][16];
};

template <int NV, bool HAS_BN, bool TT, bool NOISE = true>
__device__ __forceinline__ void fc_quad_body(QuadLds<NV> &S, const FwdArgs &A, const int *__restrict__ list, int item /* position of the group in the window */,
                                             int cg, int sl, const float *__restrict__ y2,
                                             float *__restrict__ y3t /*[member][4 quarters][256]*/) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, rg = lane & 3, cl = lane >> 2;
    const Layout &L = A.L;
    Item first;                                              // the group's first member: the group shares its base vector and noise slice
    first.pos = TT ? item * NV : -1;
    first.member = window_member<TT>(A, list, NV, item * NV);
    int member[NV];
    float scale[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        member[v] = first.member + v;
        scale[v] = TT ? A.tt.scale[first.pos + v] : A.m_scale[member[v]];
    }
    if (A.done) {   // finished group still in the list: nothing to compute
        bool all_done = true;
#pragma unroll
        for (int v = 0; v < NV; v++) all_done = all_done && A.done[member[v]] != 0;
        if (all_done) return;
    }
    const int col = cg * 16 + cl;
    const int kbeg = 968 * sl;
    constexpr int GW = (FC_SUB0 + FC_SUBN) / 4;              // 62: groups of the first wave (sub-slices 0 and 1); the others have 60
    const int na = (wv == 0 ? FC_SUB0 : FC_SUBN) / 4;        // groups of this wave's first sub-slice (32 or 30), 30 in its second
    const int ng = na + FC_SUBN / 4;
    const int g0 = wv == 0 ? 0 : GW + (wv - 1) * (2 * FC_SUBN / 4);   // the wave's first group within the quarter
    const size_t o0 = (size_t)(kbeg + 4 * g0 + rg) * 256 + col;
    const float *eps = item_eps<TT>(A, first) + L.fcw + o0;
    const float *th = item_base<TT>(A, first) + L.fcw + o0;
    // The activation loads are issued first and consumed after the weight loads are in flight: loads return in order, so the
    // barrier below waits for (at most) the first weight rows, not for all of them.
    float yv[NV][4], s2[NV][4], h2[NV][4];
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = tid + 256 * j, ch = (kbeg + i) & 31;
            const bool in = i < 968;
            yv[v][j] = in ? y2[(size_t)member[v] * 3872 + kbeg + i] : 0.0f;
            s2[v][j] = HAS_BN && in ? A.bn[(size_t)member[v] * 608 + 32 + ch] : 1.0f;
            h2[v][j] = HAS_BN && in ? A.bn[(size_t)member[v] * 608 + 64 + ch] : 0.0f;
        }
    __builtin_amdgcn_sched_barrier(0);
    float e[GW], t[GW];
#pragma unroll
    for (int i = 0; i < GW; i++) {
        const int ii = i < ng ? i : ng - 1;                  // wave-uniform clamp: loads 61 and 62 of a 60-group wave are repeats
        e[i] = NOISE ? eps[(size_t)(4 * ii) * 256] : 0.0f;
        t[i] = th[(size_t)(4 * ii) * 256];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = tid + 256 * j;
            if (i < 968) {
                float x = yv[v][j];
                if (HAS_BN) {
                    x = x * s2[v][j];
                    x = x + h2[v][j];
                }
                S.xs[v][i] = x > 0.0f ? x : 0.0f;
            }
        }
    __syncthreads();
    // perturbed weights in place: e[] <- the first member's weight, t[] <- the second's
#pragma unroll
    for (int i = 0; i < GW; i++) {
        const float ee = e[i], tt = t[i];
        float pv = NOISE ? scale[0] * ee : 0.0f;
        e[i] = NOISE ? tt + pv : tt;
        if (NV == 2) {
            float pw = scale[NV - 1] * ee;
            t[i] = tt + pw;
        }
    }
    float acc[NV], ua[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) acc[v] = ua[v] = 0.0f;
    f4a xn[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) xn[v] = *(const f4a *)&S.xs[v][4 * g0];   // rows 4g .. 4g+3 (broadcast), one group ahead
#pragma unroll
    for (int i = 0; i < GW; i++) {
        if (i == FC_SUBN / 4 || i == FC_SUB0 / 4) {          // 30 / 32: the wave's first sub-slice may end here
            if (i == na) {
#pragma unroll
                for (int v = 0; v < NV; v++) { ua[v] = acc[v]; acc[v] = 0.0f; }
            }
        }
        if (i < ng) {
            f4a x4[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) {
                x4[v] = xn[v];
                xn[v] = *(const f4a *)&S.xs[v][4 * (g0 + (i + 1 < GW ? i + 1 : i))];
            }
            // row 4g + j's weight sits in lane j of the quad and enters the fused multiply-add as a DPP operand
            // (v_fmac_f32 = the same single-rounding fma); all four lanes carry the same chain value.  s_nop:
            // the two wait states a DPP read needs after a VALU write of its source register.
            if constexpr (NV == 2) {   // the two members' chains interleaved
                asm("s_nop 1\n\t"
                    "v_fmac_f32_dpp %0, %2, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %3, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %2, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %3, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %2, %6 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %3, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %2, %7 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %3, %11 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                    : "+v"(acc[0]), "+v"(acc[NV - 1])
                    : "v"(e[i]), "v"(t[i]), "v"(x4[0][0]), "v"(x4[0][1]), "v"(x4[0][2]), "v"(x4[0][3]),
                      "v"(x4[NV - 1][0]), "v"(x4[NV - 1][1]), "v"(x4[NV - 1][2]), "v"(x4[NV - 1][3]));
            } else {
                asm("s_nop 1\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                    : "+v"(acc[0])
                    : "v"(e[i]), "v"(x4[0][0]), "v"(x4[0][1]), "v"(x4[0][2]), "v"(x4[0][3]));
            }
        }
    }
    if (rg == 0) {
#pragma unroll
        for (int v = 0; v < NV; v++) { S.comb[2 * wv][v][cl] = ua[v]; S.comb[2 * wv + 1][v][cl] = acc[v]; }
    }
    __syncthreads();
    if (tid < NV * 16) {   // the quarter's left fold over its eight sub-slices
        const int v = tid >> 4, c = tid & 15;
        float f = S.comb[0][v][c];
#pragma unroll
        for (int i = 1; i < 8; i++) f = f + S.comb[i][v][c];
        y3t[((size_t)(first.member + v) * 4 + sl) * 256 + cg * 16 + c] = f;
    }
}

template <int NV, bool HAS_BN, bool NOISE = true>
__global__ __launch_bounds__(256) void k_fc_quad(FwdArgs A, const int *__restrict__ list, const float *__restrict__ y2,
                                                 float *__restrict__ y3t /*[member][4 quarters][256]*/) {
    __shared__ QuadLds<NV> S;
    const int item = blockIdx.x >> 6, cg = (blockIdx.x >> 2) & 15, sl = blockIdx.x & 3;
    if (A.tt.n > 0) fc_quad_body<NV, HAS_BN, true, NOISE>(S, A, list, item, cg, sl, y2, y3t);
    else fc_quad_body<NV, HAS_BN, false, NOISE>(S, A, list, item, cg, sl, y2, y3t);
}

// ------------------------------------------------------------ fc for the tail of a generation (at most ~100 active groups), round 3
// One workgroup of 8 waves per (group, quarter, 64-column block): wave i owns sub-slice i of the quarter (oracle fc_raw: 128 or
// 120 rows, its own chain from 0 -- the 32 chains of an output run concurrently, nothing is handed from wave to wave), and the
// eight sums meet in LDS for the quarter's left fold.  Lane (c4, r) loads 16 bytes = columns 4 c4 .. 4 c4 + 3 of row 4 g + r
// (one instruction = four rows of the block, 1 KiB), sixteen row groups in flight per wave as a rolling window; the chain of a
// column visits the four lanes of a quad in row order with the weight as a DPP operand (v_fmac_f32_dpp quad_perm:[j,j,j,j]), the
// 4 columns x NV members of a lane being 8 independent chains that keep the pipeline full.  Per (group, quarter) the four column
// blocks repeat the activation staging; the noise and base rows are read exactly once.
template <int NV>
struct TailFcLds {
    __attribute__((aligned(16))) float xs[NV][968];
    float comb[8][NV][64];
};

template <int NV, bool HAS_BN, bool TT, bool NOISE>
__device__ __forceinline__ void fc_tail_body(TailFcLds<NV> &S, const FwdArgs &A, const int *__restrict__ list, int item, int sl, int cb,
                                             const float *__restrict__ y2, float *__restrict__ y3t /*[member][4 quarters][256]*/) {
    constexpr int D = 8;                                     // row groups in flight per wave (16: 203 VGPRs instead of 123, one workgroup per CU instead of two -- measured slower at every count)
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r = lane & 3, c4 = lane >> 2;
    const Layout &L = A.L;
    Item first;                                              // the group's first member: the group shares its base vector and noise slice
    first.pos = TT ? item * NV : -1;
    first.member = window_member<TT>(A, list, NV, item * NV);
    int member[NV];
    float scale[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        member[v] = first.member + v;
        scale[v] = TT ? A.tt.scale[first.pos + v] : A.m_scale[member[v]];
    }
    if (A.done) {   // finished group still in the list: nothing to compute
        bool all_done = true;
#pragma unroll
        for (int v = 0; v < NV; v++) all_done = all_done && A.done[member[v]] != 0;
        if (all_done) return;
    }
    DNE_PHASE(2, 0);
    const int kbeg = 968 * sl;
    const int beg = wv == 0 ? 0 : FC_SUB0 + FC_SUBN * (wv - 1);     // this wave's sub-slice within the quarter
    const int ng = (wv == 0 ? FC_SUB0 : FC_SUBN) / 4;               // its 4-row groups: 32 or 30
    const size_t o0 = (size_t)(kbeg + beg + r) * 256 + cb * 64 + c4 * 4;
    const float *ep = item_eps<TT>(A, first) + L.fcw + o0, *tp = item_base<TT>(A, first) + L.fcw + o0;
    // the quarter's activations (two per thread and member): requested first, consumed after the weight rows are in flight
    float yv[NV][2], s2[NV][2], h2[NV][2];
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = tid + 512 * j, ch = (kbeg + i) & 31;
            const bool in = i < 968;
            yv[v][j] = in ? y2[(size_t)member[v] * 3872 + kbeg + i] : 0.0f;
            s2[v][j] = HAS_BN && in ? A.bn[(size_t)member[v] * 608 + 32 + ch] : 1.0f;
            h2[v][j] = HAS_BN && in ? A.bn[(size_t)member[v] * 608 + 64 + ch] : 0.0f;
        }
    __builtin_amdgcn_sched_barrier(0);
    f4u e[D];                                                // NOISE = false (GA children written out: plain rows): never loaded
    f4a t[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        if (NOISE) e[i] = *(const f4u *)(ep + (size_t)i * 1024);
        t[i] = *(const f4a *)(tp + (size_t)i * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = tid + 512 * j;
            if (i < 968) {
                float x = yv[v][j];
                if (HAS_BN) {
                    x = x * s2[v][j];
                    x = x + h2[v][j];
                }
                S.xs[v][i] = x > 0.0f ? x : 0.0f;
            }
        }
    __syncthreads();
    DNE_PHASE(2, 1);
    float acc[NV][4];
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[v][q] = 0.0f;
#pragma unroll
    for (int g = 0; g < FC_SUB0 / 4; g++) {
        if (g < ng) {
            const int slot = g % D;
            f4a x4[NV];
            float w[NV][4];
#pragma unroll
            for (int v = 0; v < NV; v++) {
                x4[v] = *(const f4a *)&S.xs[v][beg + 4 * g];        // rows 4g .. 4g+3 of the sub-slice (broadcast read)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float pv = NOISE ? scale[v] * e[slot][q] : 0.0f;
                    w[v][q] = NOISE ? t[slot][q] + pv : t[slot][q];
                }
            }
            if (g + D < ng) {                                       // this slot's registers take the group D ahead
                if (NOISE) e[slot] = *(const f4u *)(ep + (size_t)(g + D) * 1024);
                t[slot] = *(const f4a *)(tp + (size_t)(g + D) * 1024);
            }
            // row 4g + j's weights sit in lane j of the quad and enter the fused multiply-add as a DPP operand (v_fmac_f32 = the same
            // single-rounding fma); all four lanes of a quad carry the same chain values.  s_nop: the wait states a DPP read needs
            // after a VALU write of its source register.
            if constexpr (NV == 2) {
                asm("s_nop 1\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a10], %[w10], %[x10] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a11], %[w11], %[x10] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a12], %[w12], %[x10] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a13], %[w13], %[x10] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a10], %[w10], %[x11] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a11], %[w11], %[x11] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a12], %[w12], %[x11] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a13], %[w13], %[x11] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a10], %[w10], %[x12] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a11], %[w11], %[x12] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a12], %[w12], %[x12] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a13], %[w13], %[x12] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a10], %[w10], %[x13] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a11], %[w11], %[x13] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a12], %[w12], %[x13] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a13], %[w13], %[x13] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        : [a00] "+v"(acc[0][0]), [a01] "+v"(acc[0][1]), [a02] "+v"(acc[0][2]), [a03] "+v"(acc[0][3]), [a10] "+v"(acc[1][0]), [a11] "+v"(acc[1][1]), [a12] "+v"(acc[1][2]), [a13] "+v"(acc[1][3])
                        : [w00] "v"(w[0][0]), [w01] "v"(w[0][1]), [w02] "v"(w[0][2]), [w03] "v"(w[0][3]), [w10] "v"(w[1][0]), [w11] "v"(w[1][1]), [w12] "v"(w[1][2]), [w13] "v"(w[1][3]), [x00] "v"(x4[0][0]), [x01] "v"(x4[0][1]), [x02] "v"(x4[0][2]), [x03] "v"(x4[0][3]), [x10] "v"(x4[1][0]), [x11] "v"(x4[1][1]), [x12] "v"(x4[1][2]), [x13] "v"(x4[1][3]));
            } else {
                asm("s_nop 1\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x00] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x01] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x02] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a00], %[w00], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a01], %[w01], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a02], %[w02], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        "v_fmac_f32_dpp %[a03], %[w03], %[x03] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                        : [a00] "+v"(acc[0][0]), [a01] "+v"(acc[0][1]), [a02] "+v"(acc[0][2]), [a03] "+v"(acc[0][3])
                        : [w00] "v"(w[0][0]), [w01] "v"(w[0][1]), [w02] "v"(w[0][2]), [w03] "v"(w[0][3]), [x00] "v"(x4[0][0]), [x01] "v"(x4[0][1]), [x02] "v"(x4[0][2]), [x03] "v"(x4[0][3]));
            }
        }
    }
    if (r == 0) {
#pragma unroll
        for (int v = 0; v < NV; v++)
#pragma unroll
            for (int q = 0; q < 4; q++) S.comb[wv][v][c4 * 4 + q] = acc[v][q];
    }
    DNE_PHASE(2, 2);
    __syncthreads();
    DNE_PHASE(2, 3);
    if (tid < NV * 64) {   // the quarter's left fold over its eight sub-slices
        const int v = tid >> 6, col = tid & 63;
        float f = S.comb[0][v][col];
#pragma unroll
        for (int i = 1; i < 8; i++) f = f + S.comb[i][v][col];
        y3t[((size_t)(first.member + v) * 4 + sl) * 256 + cb * 64 + col] = f;
    }
    DNE_PHASE(2, 4);
}

template <int NV, bool HAS_BN, bool NOISE = true>
__global__ __launch_bounds__(512) void k_fc_tail(FwdArgs A, const int *__restrict__ list, const float *__restrict__ y2,
                                                 float *__restrict__ y3t /*[member][4 quarters][256]*/) {
    __shared__ TailFcLds<NV> S;
    const int item = blockIdx.x >> 4, sl = (blockIdx.x >> 2) & 3, cb = blockIdx.x & 3;
    if (A.tt.n > 0) fc_tail_body<NV, HAS_BN, true, NOISE>(S, A, list, item, sl, cb, y2, y3t);
    else fc_tail_body<NV, HAS_BN, false, NOISE>(S, A, list, item, sl, cb, y2, y3t);
}

// Upper part of the tail (more than ~32 groups per window, several windows at once): 4 workgroups per group (one per 64-column
// block), wave = quarter, lane = ONE output column with 2 x 40 rows in flight -- few, lean workgroups: where k_fc_tail's 16
// workgroups of 512 threads per group no longer fit the chip at once this form is faster (round 3 same-box: 312 pairs in 4 windows).
// The wave walks its quarter row by row, so the sub-slices (oracle fc_raw) are a running fold: 8 rows, then 24 batches of 40 --
// the boundaries 128, 248, ..., 968 fall after every third batch.
template <int NV, bool HAS_BN>
__global__ __launch_bounds__(256) void k_fc_cols(FwdArgs A, const int *__restrict__ list, const float *__restrict__ y2,
                                                 float *__restrict__ y3t /*[member][4 quarters][256]*/) {
    __shared__ float xs[4][NV][968];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    const int item = blockIdx.x >> 2, cq = blockIdx.x & 3;
    const int g = list ? list[item] : item;
    int member[NV];
    float scale[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) { member[v] = g * NV + v; scale[v] = A.m_scale[member[v]]; }
    if (A.done) {   // finished group still in the list: nothing to compute
        bool all_done = true;
#pragma unroll
        for (int v = 0; v < NV; v++) all_done = all_done && A.done[member[v]] != 0;
        if (all_done) return;
    }
    const int64_t off = A.m_off[member[0]];
    const float *base = A.bases + (size_t)A.m_slot[member[0]] * A.base_stride;
    const int col = cq * 64 + lane;
    const int kbeg = 968 * wv;
    const float *eps = A.noise + off + L.fcw + (size_t)kbeg * 256 + col;
    const float *th = base + L.fcw + (size_t)kbeg * 256 + col;
    constexpr int RB = 40, R0 = FC_SUB0 - 3 * RB;   // 8 leading rows, then 24 batches of 40
    static_assert(R0 == 8 && FC_SUBN == 3 * RB && R0 + 24 * RB == 968, "batches must end on the sub-slice boundaries");
    float e0[R0], t0[R0], e_cur[RB], t_cur[RB], e_nxt[RB], t_nxt[RB];
#pragma unroll
    for (int i = 0; i < R0; i++) { e0[i] = eps[(size_t)i * 256]; t0[i] = th[(size_t)i * 256]; }
#pragma unroll
    for (int i = 0; i < RB; i++) { e_cur[i] = eps[(size_t)(R0 + i) * 256]; t_cur[i] = th[(size_t)(R0 + i) * 256]; }
    {
        float yv[NV][16], s2[NV], h2[NV];
        const int ch = (kbeg + lane) & 31;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            s2[v] = HAS_BN ? A.bn[(size_t)member[v] * 608 + 32 + ch] : 1.0f;
            h2[v] = HAS_BN ? A.bn[(size_t)member[v] * 608 + 64 + ch] : 0.0f;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int i = lane + 64 * j;
                yv[v][j] = i < 968 ? y2[(size_t)member[v] * 3872 + kbeg + i] : 0.0f;
            }
        }
#pragma unroll
        for (int v = 0; v < NV; v++)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int i = lane + 64 * j;
                if (i < 968) {
                    float t = yv[v][j];
                    if (HAS_BN) {
                        t = t * s2[v];
                        t = t + h2[v];
                    }
                    xs[wv][v][i] = t > 0.0f ? t : 0.0f;
                }
            }
    }
    __syncthreads();
    float acc[NV], fold[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) acc[v] = fold[v] = 0.0f;
#pragma unroll
    for (int i = 0; i < R0; i++) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            float pv = scale[v] * e0[i];
            const float w = t0[i] + pv;
            acc[v] = __builtin_fmaf(xs[wv][v][i], w, acc[v]);
        }
    }
    for (int bt = 0; bt < 24; bt++) {
        if (bt + 1 < 24) {
#pragma unroll
            for (int i = 0; i < RB; i++) {
                e_nxt[i] = eps[(size_t)(R0 + (bt + 1) * RB + i) * 256];
                t_nxt[i] = th[(size_t)(R0 + (bt + 1) * RB + i) * 256];
            }
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
                float pv = scale[v] * e_cur[i];
                const float w = t_cur[i] + pv;
                acc[v] = __builtin_fmaf(xs[wv][v][R0 + bt * RB + i], w, acc[v]);
            }
        }
#pragma unroll
        for (int i = 0; i < RB; i++) { e_cur[i] = e_nxt[i]; t_cur[i] = t_nxt[i]; }
        if (bt % 3 == 2) {   // rows 128, 248, ..., 968 done: the end of a sub-slice
#pragma unroll
            for (int v = 0; v < NV; v++) { fold[v] = bt == 2 ? acc[v] : fold[v] + acc[v]; acc[v] = 0.0f; }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; v++) y3t[((size_t)member[v] * 4 + wv) * 256 + col] = fold[v];
}

// fc slice combine + bias, bn3 + relu, output layer (out_products / out_wave_sums: thread = input k) + first-max argmax from the
// fc partial sums, one workgroup per group.  The members of a group share base vector and noise slice (antithetic pair).
// ------------------------------------------------------- sub-slice fc: one wave per (group, sub-slice chain), nothing shared, nothing waited for
// The mid range (about 100 .. 800 groups alive): too many groups for the latency kernels, too few for one workgroup per group to
// fill 256 CUs (k_fc<1>: 250 members = 250 workgroups, 4 KB in flight per wave).  The numerics contract already cuts every output's
// k-sum into 32 independent chains (4 quarters x 8 sub-slices of 128 / 120 rows, oracle fc_raw), so the chain IS the unit of work
// here: a wave walks the rows of ONE sub-slice (or of `spw` consecutive ones) with whole 1 KB rows per load instruction (lane = 4
// columns) through k_fc_duo's rolling window -- W rows per stream always in flight, each row's registers refilled the moment it is
// consumed, no LDS, no barrier -- and stores the chain's 256 sums to y3s[member][32][256]; the head (FwdArgs::sub_sums) folds them in
// the oracle's order.  250 members = 8000 waves: the whole machine streams.  The block behind a chain's last one is fetched and never
// used: the next sub-slice, or what follows the fc matrix (fc bias / bn3 / output layer, and for small action counts the floats behind the
// member's slice: the bases and noise allocations carry a block of padding for that, engine.hip OVERFETCH_FLOATS).
template <int NV, bool HAS_BN, bool NOISE>
__global__ __launch_bounds__(256) void k_fc_sub(FwdArgs A, const int *__restrict__ list, int n_groups, int spw /* sub-slices per wave: 1, 2, 4, 8 */,
                                                int prio, const float *__restrict__ y2, float *__restrict__ y3s /*[member][32][256]*/) {
    constexpr int W = 8;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const unsigned voff = lane * 16;
    const Layout &L = A.L;
    const int upg = 32 / spw;   // waves per group
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    // a bounded grid walks the work (DNE_FC_SUB_GRID): at 54 registers the whole launch would otherwise sit on every wave slot of the
    // chip and the other window's convolutions / renderer / head -- the links of ITS lock-step chain -- wait for a slot behind it
    for (int w = uni(blockIdx.x * 4 + wv); w < n_groups * upg; w += gridDim.x * 4) {
    const int gi = w / upg, j = w - gi * upg;
    const int g = uni(list ? list[gi] : gi);
    if (A.done) {   // finished group still in the list: nobody reads its sums
        int all_done = 1;
#pragma unroll
        for (int v = 0; v < NV; v++) all_done &= uni(A.done[g * NV + v]) != 0;
        if (all_done) continue;
    }
    const int m0 = g * NV;
    const long long off = uni64(A.m_off[m0]);
    const float *base = A.bases + (size_t)uni(A.m_slot[m0]) * A.base_stride + L.fcw;
    const float *eps = A.noise + off + L.fcw;
    float scale[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) scale[v] = A.m_scale[m0 + v];
    for (int s = 0; s < spw; s++) {
        const int u = j * spw + s, q = u >> 3, i = u & 7;
        const int nrows = i == 0 ? FC_SUB0 : FC_SUBN;
        const int R0 = 968 * q + (i == 0 ? 0 : FC_SUB0 + FC_SUBN * (i - 1));   // first row of the chain within the 3872
        // activations relu(bn2(y2)) of the chain's rows: two chunks of 64, one row per lane (bn2 channel = row mod 32)
        float xa[NV], xb[NV];
        {
            const int ra = R0 + lane, rb = R0 + min(64 + lane, nrows - 1);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                float a = y2[(size_t)(m0 + v) * 3872 + ra], b = y2[(size_t)(m0 + v) * 3872 + rb];
                if (HAS_BN) {
                    const float *bn = A.bn + (size_t)(m0 + v) * 608;
                    a = a * bn[32 + (ra & 31)]; a = a + bn[64 + (ra & 31)];
                    b = b * bn[32 + (rb & 31)]; b = b + bn[64 + (rb & 31)];
                }
                xa[v] = a > 0.0f ? a : 0.0f;
                xb[v] = b > 0.0f ? b : 0.0f;
                // finished BEFORE the first row load is issued: left to itself the compiler finishes them in the middle of the
                // window's first fill and drains the rows already requested (s_waitcnt vmcnt(0)) to do so
                asm volatile("" : "+v"(xa[v]), "+v"(xb[v]));
            }
        }
        const float *tnext = base + (size_t)R0 * 256;   // wave-uniform: the chain's next row block
        const float *enext = eps + (size_t)R0 * 256;
        f32x2 acc[NV][2];
#pragma unroll
        for (int v = 0; v < NV; v++) acc[v][0] = acc[v][1] = f32x2{0.0f, 0.0f};
        f32x4 t[W], e[NOISE ? W : 1];
#pragma unroll
        for (int r = 0; r < W; r++) t[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < (NOISE ? W : 1); r++) e[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int PENDING = (W - 1) * (NOISE ? 2 : 1);
        auto refill = [&](auto ii) {
            constexpr int I = decltype(ii)::value;
            if constexpr (NOISE) {
                if constexpr (NV == 2) gload4_after<(I % 4) * 1024>(e[I], voff, enext + (I / 4) * 1024, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
                else gload4_after<(I % 4) * 1024>(e[I], voff, enext + (I / 4) * 1024, acc[0][0], acc[0][1]);
                gload4<(I % 4) * 1024>(t[I], voff, tnext + (I / 4) * 1024);
            } else {
                if constexpr (NV == 2) gload4_after<(I % 4) * 1024>(t[I], voff, tnext + (I / 4) * 1024, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
                else gload4_after<(I % 4) * 1024>(t[I], voff, tnext + (I / 4) * 1024, acc[0][0], acc[0][1]);
            }
        };
        static_for<W>(refill);
        tnext += W * 256; enext += W * 256;
        const int nb = nrows / W;   // 16 or 15 row blocks
        for (int b = 0; b < nb; b++) {
            const int li = (b & 7) * W;
            float xs[NV];   // rows 0 .. 63 of the chain come from xa, the rest from xb
#pragma unroll
            for (int v = 0; v < NV; v++) xs[v] = b < 8 ? xa[v] : xb[v];
            auto one = [&](auto ii) {
                constexpr int I = decltype(ii)::value;
                wait_rows<PENDING>(t[I], e[NOISE ? I : 0]);
                const f32x2 tlo = {t[I][0], t[I][1]}, thi = {t[I][2], t[I][3]};
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const float x = lane_bcast(xs[v], li + I);
                    const f32x2 xx = {x, x};
                    f32x2 wl = tlo, wh = thi;
                    if constexpr (NOISE) {   // fl(theta + fl(scale * eps)): es.py:412-419's two roundings
                        const f32x2 sc = {scale[v], scale[v]}, elo = {e[I][0], e[I][1]}, ehi = {e[I][2], e[I][3]};
                        const f32x2 pl = sc * elo, ph = sc * ehi;
                        wl = tlo + pl; wh = thi + ph;
                    }
                    acc[v][0] = __builtin_elementwise_fma(xx, wl, acc[v][0]);
                    acc[v][1] = __builtin_elementwise_fma(xx, wh, acc[v][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                refill(ii);
                __builtin_amdgcn_sched_barrier(0);
            };
            static_for<W>(one);
            tnext += W * 256; enext += W * 256;
        }
#pragma unroll
        for (int r = 0; r < W; r++) wait_rows<0>(t[r], e[NOISE ? r : 0]);   // the over-fetched block: nothing in flight into dead registers
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const f4a o = {acc[v][0][0], acc[v][0][1], acc[v][1][0], acc[v][1][1]};
            *(f4a *)(y3s + ((size_t)(m0 + v) * 32 + u) * 256 + lane * 4) = o;
        }
    }
    }   // persistent loop over the launch's waves' worth of work
}

template <int NV, bool HAS_BN>
__global__ __launch_bounds__(256) void k_out(FwdArgs A, const int *__restrict__ list, const float *__restrict__ y3t,
                                            float *__restrict__ y3, int32_t *__restrict__ actions,
                                            float *__restrict__ logits_out) {
    __shared__ float red[4][NV][OUT_NA];
    __shared__ float lg[NV][32];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    const int g = list ? list[blockIdx.x] : blockIdx.x;
    const int nact = L.nact;
    if (A.done) {   // finished group still in the list: its partial sums were not refreshed and nobody reads its action
        bool all_done = true;
#pragma unroll
        for (int v = 0; v < NV; v++) all_done = all_done && A.done[g * NV + v] != 0;
        if (all_done) return;
    }
    DNE_WG_BEGIN;
    const int m0 = g * NV;
    const int64_t off = A.m_off[m0];
    const float *base = A.bases + (size_t)A.m_slot[m0] * A.base_stride;
    const float *eps = A.noise + off;
    float sc[NV], x3[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) sc[v] = A.m_scale[m0 + v];
    const float fb_t = base[L.fcb + tid], fb_e = eps[L.fcb + tid];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int m = m0 + v;
        float t = fc_combine(y3t, m, tid, A.sub_sums != 0);
        float pvb = sc[v] * fb_e;
        const float fb = fb_t + pvb;
        t = t + fb;
        y3[(size_t)m * 256 + tid] = t;
        if (HAS_BN) {
            t = t * A.bn[(size_t)m * 608 + 96 + tid];
            t = t + A.bn[(size_t)m * 608 + 352 + tid];
        }
        x3[v] = t > 0.0f ? t : 0.0f;
    }
    {
        float p[NV][OUT_NA];
        out_products<NV>(p, x3, base + L.ow + tid * nact, eps + L.ow + tid * nact, sc, nact);
        out_wave_sums<NV>(p, nact, red[wv], lane);
    }
    __syncthreads();
    if (tid < NV * nact) {
        const int v = tid / nact, a = tid % nact;
        const float s01 = red[0][v][a] + red[1][v][a];
        const float s23 = red[2][v][a] + red[3][v][a];
        const float t = s01 + s23;
        float pv = A.m_scale[m0 + v] * eps[L.ob + a];
        const float bias = base[L.ob + a] + pv;
        lg[v][a] = t + bias;
    }
    __syncthreads();
    if (tid < NV) {
        const int v = tid, m = m0 + v;
        int best = 0;
        for (int a = 1; a < nact; a++)
            if (lg[v][a] > lg[v][best]) best = a;
        actions[m] = best;
        if (logits_out)
            for (int a = 0; a < nact; a++) logits_out[(size_t)m * nact + a] = lg[v][a];
    }
    DNE_WG_END(2);
}

// Per-member batch-norm scale / shift of a convolution layer from the per-frame moments the convolutions left behind
// (oracle: bn_finish_tiles): S, Q = frame moments summed in frame order; m = S / count; mean = bias + m;
// var = max(Q / count - m*m, 0).  One thread per (member, channel).
template <int C>
__global__ __launch_bounds__(256) void k_bn_finalize(FwdArgs A, int member0, int n_local, int F, const float *__restrict__ fr,
                                                     int npos, int bn_off, int bias_off, int beta_off, int gamma_off) {
    const int i = blockIdx.x * 256 + threadIdx.x, mloc = i / C, c = i % C;
    if (mloc >= n_local) return;
    const int member = member0 + mloc;
    const float *p = fr + (size_t)mloc * F * 2 * C + c;
    float S = 0.0f, Q = 0.0f;
    for (int n = 0; n < F; n++) {
        S = S + p[(size_t)n * 2 * C];
        Q = Q + p[(size_t)n * 2 * C + C];
    }
    const float count = (float)(F * npos);
    const float *base = A.bases + (size_t)A.m_slot[member] * A.base_stride;
    const float *eps = A.noise + A.m_off[member];
    const float sc = A.m_scale[member];
    float pbias = sc * eps[bias_off + c];
    const float bias = base[bias_off + c] + pbias;
    const float m = S / count;
    const float mean = bias + m;
    const float q1 = Q / count;
    const float mm = m * m;
    float var = q1 - mm;
    var = var > 0.0f ? var : 0.0f;
    float pb = sc * eps[beta_off + c];
    const float beta = base[beta_off + c] + pb;
    float pg = sc * eps[gamma_off + c];
    const float gamma = base[gamma_off + c] + pg;
    const float inv = 1.0f / sqrtf(var + 1e-3f);
    const float s = inv * gamma;
    const float ms = mean * s;
    A.bn[(size_t)member * 608 + bn_off + c] = s;
    A.bn[(size_t)member * 608 + bn_off + C + c] = beta - ms;
    A.bn_mom[(size_t)member * 608 + bn_off + c] = mean;
    A.bn_mom[(size_t)member * 608 + bn_off + C + c] = var;
}

// -------------------------------------------------------------- virtual batch norm statistics of the fc layer, generic batch sizes
// tf.contrib.layers.batch_norm(is_training=True, decay=0): batch moments over the F reference frames, biased variance, written
// as per-column scale = gamma * rsqrt(var + 1e-3), shift = beta - mean * scale (tf.nn.batch_normalization); two passes in frame
// order like the oracle's bn_finish.  y: [n_local][F][256] finished fc rows (the reference batch sizes the matrix-core fc does
// not take: k_fc<8, true> wrote them); the convolution layers' moments come out of the convolution epilogues (k_bn_finalize).
__global__ __launch_bounds__(256) void k_bn3_rows(FwdArgs A, int member0, int F, const float *__restrict__ y) {
    const int mloc = blockIdx.x, member = member0 + mloc, j = threadIdx.x;
    const Layout &L = A.L;
    const float *ym = y + (size_t)mloc * F * 256 + j;
    const float count = (float)F;
    float tot = 0.0f;
    for (int n = 0; n < F; n++) tot = tot + (0.0f + ym[(size_t)n * 256]);
    const float mean = tot / count;
    float totq = 0.0f;
    for (int n = 0; n < F; n++) {
        const float d = ym[(size_t)n * 256] - mean;
        totq = totq + __builtin_fmaf(d, d, 0.0f);
    }
    const float var = totq / count;
    const float *base = A.bases + (size_t)A.m_slot[member] * A.base_stride;
    const float *eps = A.noise + A.m_off[member];
    const float sc = A.m_scale[member];
    float pb = sc * eps[L.bn3b + j];
    const float beta = base[L.bn3b + j] + pb;
    float pg = sc * eps[L.bn3g + j];
    const float gamma = base[L.bn3g + j] + pg;
    const float inv = 1.0f / sqrtf(var + 1e-3f);
    const float s = inv * gamma;
    const float ms = mean * s;
    A.bn[(size_t)member * 608 + 96 + j] = s;
    A.bn[(size_t)member * 608 + 352 + j] = beta - ms;
    A.bn_mom[(size_t)member * 608 + 96 + j] = mean;
    A.bn_mom[(size_t)member * 608 + 352 + j] = var;
}

}  // namespace dne

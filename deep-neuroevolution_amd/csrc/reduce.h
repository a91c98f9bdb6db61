// reduce.h -- on-device reduction of episode returns into the ES update / GA selection, plus the
// stand-alone perturbation kernel and the GA genome rebuild.
//
// Replaces es_distributed/es.py:70-85 (compute_centered_ranks), :115-122,291-296 (batched_weighted_sum,
// g /= returns.size), :298 + optimizers.py:10-17,29-32,45-50 (L2 + SGD/Adam step), es.py:412-419
// (theta +- sigma*eps), ga.py:145 (truncation selection), ga.py:256-264 + policies.py:42-44 +
// tf_util.py:122-130 (seed chain -> normc(noise[s0]) + sigma * sum noise[s_k]), nses.py:12-20 (BC distance).
// All HBM-bound streaming kernels; same operation order as the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dne {

// es.py:70-85.  rank_i = #{x_j < x_i} + #{j < i : x_j == x_i}  (argsort ties by flat index)
__global__ __launch_bounds__(256) void k_centered_ranks(const float *__restrict__ x, int n, float *__restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float xi = x[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
        const float xj = x[j];
        rank += (xj < xi) || (xj == xi && j < i);
    }
    float r = (float)rank;
    r = r / (float)(n - 1);
    y[i] = r - 0.5f;
}

// es.py:292  proc_returns_n2[:, 0] - proc_returns_n2[:, 1]
__global__ void k_pair_weights(const float *__restrict__ proc, int n, float *__restrict__ w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = proc[2 * i] - proc[2 * i + 1];
}

// nses.py:226-228  (rew_ranks + proc) / 2.0
__global__ void k_average2(const float *__restrict__ a, const float *__restrict__ b, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float s = a[i] + b[i]; out[i] = s / 2.0f; }
}

// The per-pair record that travels between GPUs: what a worker's Result carries per antithetic pair
// (es.py:18-23: noise_inds_n, returns_n2, lengths_n2, signreturns_n2), 32 bytes.
struct PairRecord {
    int64_t noise_idx;
    float ret[2];
    int32_t len[2];
    float sign[2];
};
static_assert(sizeof(PairRecord) == 32, "wire record is 32 bytes");

// this rank's shard of the last evaluation -> `per` wire records (the tail past n_local is zero)
__global__ void k_records_pack(const int64_t *__restrict__ m_off, const float *__restrict__ ret, const float *__restrict__ sign,
                               const int32_t *__restrict__ len, int n_local, int per, PairRecord *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= per) return;
    PairRecord r{};
    if (j < n_local) {
        r.noise_idx = m_off[2 * j];
        r.ret[0] = ret[2 * j]; r.ret[1] = ret[2 * j + 1];
        r.len[0] = len[2 * j]; r.len[1] = len[2 * j + 1];
        r.sign[0] = sign[2 * j]; r.sign[1] = sign[2 * j + 1];
    }
    out[j] = r;
}

// gathered [world][per] records -> arrays in global pair order (pair i was evaluated by rank i % world as its (i / world)-th)
__global__ void k_records_unpack(const PairRecord *__restrict__ in, int n_global, int world, int per, int64_t *__restrict__ idx,
                                 float *__restrict__ ret, float *__restrict__ sign, int32_t *__restrict__ len,
                                 PairRecord *__restrict__ ordered) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_global) return;
    const PairRecord r = in[(size_t)(i % world) * per + i / world];
    idx[i] = r.noise_idx;
    ret[2 * i] = r.ret[0]; ret[2 * i + 1] = r.ret[1];
    sign[2 * i] = r.sign[0]; sign[2 * i + 1] = r.sign[1];
    len[2 * i] = r.len[0]; len[2 * i + 1] = r.len[1];
    if (ordered) ordered[i] = r;
}

// es.py:291-296: g[p] = (sum_i w_i * noise[idx_i + p]) / denom as an i-ordered fmaf chain per parameter.
// Every noise slice is read exactly once: N * 4P bytes of coalesced HBM gathers.
__global__ __launch_bounds__(256) void k_weighted_sum(const float *__restrict__ noise, const int64_t *__restrict__ idx,
                                                      const float *__restrict__ w, int n, int P, float denom,
                                                      float *__restrict__ g) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float acc = 0.0f;
#pragma unroll 8
    for (int i = 0; i < n; i++) acc = __builtin_fmaf(w[i], noise[idx[i] + p], acc);
    g[p] = acc / denom;
}

// es.py:298 globalg = -g + l2coeff*theta, then optimizers.py:45-50 (Adam) with float32 array arithmetic
// (numpy 1.12 value-based casting).  Per-block partial sums of step^2 and theta^2 in double
// give optimizers.py:14 ratio = |step| / |theta|.
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ theta, float *__restrict__ m, float *__restrict__ v,
                                              const float *__restrict__ g, int P, float l2, float neg_a, float b1,
                                              float ob1, float b2, float ob2, float eps, double *__restrict__ partial) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    double ss = 0.0, tt = 0.0;
    if (p < P) {
        const float th = theta[p];
        float ng = -g[p];
        float l = l2 * th;
        const float gg = ng + l;
        float m1 = b1 * m[p];
        float m2 = ob1 * gg;
        const float mm = m1 + m2;
        float g2 = gg * gg;
        float v1 = b2 * v[p];
        float v2 = ob2 * g2;
        const float vv = v1 + v2;
        float num = neg_a * mm;
        float den = sqrtf(vv);
        den = den + eps;
        const float step = num / den;
        m[p] = mm;
        v[p] = vv;
        theta[p] = th + step;
        ss = (double)step * step;
        tt = (double)th * th;
    }
    __shared__ double rs[256], rt[256];
    rs[threadIdx.x] = ss; rt[threadIdx.x] = tt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { rs[threadIdx.x] += rs[threadIdx.x + s]; rt[threadIdx.x] += rt[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = rs[0]; partial[2 * blockIdx.x + 1] = rt[0]; }
}

// optimizers.py:29-32  v = momentum*v + (1-momentum)*g ; step = -stepsize*v
__global__ __launch_bounds__(256) void k_sgd(float *__restrict__ theta, float *__restrict__ v, const float *__restrict__ g,
                                             int P, float l2, float mo, float om, float nlr, double *__restrict__ partial) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    double ss = 0.0, tt = 0.0;
    if (p < P) {
        const float th = theta[p];
        float ng = -g[p];
        float l = l2 * th;
        const float gg = ng + l;
        float v1 = mo * v[p];
        float v2 = om * gg;
        const float vv = v1 + v2;
        const float step = nlr * vv;
        v[p] = vv;
        theta[p] = th + step;
        ss = (double)step * step;
        tt = (double)th * th;
    }
    __shared__ double rs[256], rt[256];
    rs[threadIdx.x] = ss; rt[threadIdx.x] = tt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { rs[threadIdx.x] += rs[threadIdx.x + s]; rt[threadIdx.x] += rt[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = rs[0]; partial[2 * blockIdx.x + 1] = rt[0]; }
}

// es.py:413-419: v = sigma * noise[idx:idx+P]; out[2i] = theta + v; out[2i+1] = theta - v
__global__ __launch_bounds__(256) void k_materialize(const float *__restrict__ theta, const float *__restrict__ noise,
                                                     const int64_t *__restrict__ idx, int P, float sigma,
                                                     float *__restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int i = blockIdx.y;
    const float v = sigma * noise[idx[i] + p];
    const float t = theta[p];
    out[(size_t)(2 * i) * P + p] = t + v;
    out[(size_t)(2 * i + 1) * P + p] = t - v;
}

// ga.py:256: v = noise.get(seeds[0], P)   |   ga.py:262-263: v += noise_stdev * noise.get(seed, P)
__global__ __launch_bounds__(256) void k_copy_noise(const float *__restrict__ noise, int64_t off, int P, float *__restrict__ dst) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) dst[p] = noise[off + p];
}
__global__ __launch_bounds__(256) void k_axpy_noise(const float *__restrict__ noise, int64_t off, int P, float sigma,
                                                    const float *__restrict__ src, float *__restrict__ dst) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) { float v = sigma * noise[off + p]; dst[p] = src[p] + v; }
}
// ga.py:262-263 for a whole chain in one pass: dst[p] = (...((src[p] + sigma*noise[s_1 + p]) + sigma*noise[s_2 + p])...),
// the same additions in the same order as one k_axpy_noise per seed, but theta stays in a register and the n noise
// slices stream through once (4 (n + 2) P bytes instead of 12 n P).  Eight slices are in flight per thread.
// pw (may be null): one mutation power per seed (gpu_implementation/neuroevolution/models/base.py:141-149, where a genome is
// ((idx0,), (idx1, power1), ...)); null: the same sigma for every seed (ga.py:262-263)
__global__ __launch_bounds__(256) void k_chain_sum(const float *__restrict__ noise, const int64_t *__restrict__ offs,
                                                   const float *__restrict__ pw, int n, int P, float sigma,
                                                   const float *__restrict__ src, float *__restrict__ dst) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float v = src[p];
    int s = 0;
    for (; s + 8 <= n; s += 8) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = noise[offs[s + j] + p];
#pragma unroll
        for (int j = 0; j < 8; j++) { float t = (pw ? pw[s + j] : sigma) * e[j]; v = v + t; }
    }
    for (; s < n; s++) { float t = (pw ? pw[s] : sigma) * noise[offs[s] + p]; v = v + t; }
    dst[p] = v;
}

// gpu_implementation base.py:128-129: theta = noise.get(idx, num_params).copy() * self.scale_by (element-wise, float32)
__global__ __launch_bounds__(256) void k_copy_noise_scaled_batch(const float *__restrict__ noise, const int64_t *__restrict__ offs,
                                                                 const int32_t *__restrict__ slots, size_t stride, int P,
                                                                 const float *__restrict__ scale_by, float *__restrict__ bases) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) bases[(size_t)slots[blockIdx.y] * stride + p] = noise[offs[blockIdx.y] + p] * scale_by[p];
}

// tf_util.py:122-130 _normalize on a [K][C] view: out *= std / sqrt(square(out).sum(axis=0)); the axis-0
// sum is sequential in k (numpy adds row by row).  One thread per column; bias tensors are zeroed.
// (the column sum is a dependent chain, but its operands are not: 16 rows are loaded ahead of the adds, K % 16 == 0)
__device__ __forceinline__ void normc_column(float *__restrict__ w, int K, int C, int c, float std) {
    float ss = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = k0 + j < K ? w[(size_t)(k0 + j) * C + c] : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (k0 + j < K) { float sq = x[j] * x[j]; ss = ss + sq; }
    }
    const float rt = sqrtf(ss);
    const float sc = std / rt;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = k0 + j < K ? w[(size_t)(k0 + j) * C + c] : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (k0 + j < K) w[(size_t)(k0 + j) * C + c] = x[j] * sc;
    }
}
__global__ __launch_bounds__(64) void k_normc(float *__restrict__ w, int K, int C, float std) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    normc_column(w, K, C, c, std);
}
// batched forms for a whole generation of fresh genomes (generation 0 of the GA: every child is its own
// normc(noise[s0])): blockIdx.y = genome, written into base slot slots[genome]
__global__ __launch_bounds__(256) void k_copy_noise_batch(const float *__restrict__ noise, const int64_t *__restrict__ offs,
                                                          const int32_t *__restrict__ slots, size_t stride, int P,
                                                          float *__restrict__ bases) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) bases[(size_t)slots[blockIdx.y] * stride + p] = noise[offs[blockIdx.y] + p];
}
__global__ __launch_bounds__(64) void k_normc_batch(float *__restrict__ bases, const int32_t *__restrict__ slots, size_t stride,
                                                    int off, int K, int C, float std) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    normc_column(bases + (size_t)slots[blockIdx.y] * stride + off, K, C, c, std);
}
__global__ void k_zero_batch(float *__restrict__ bases, const int32_t *__restrict__ slots, size_t stride, int off, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bases[(size_t)slots[blockIdx.y] * stride + off + i] = 0.0f;
}
__global__ void k_zero(float *__restrict__ w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = 0.0f;
}

// ga.py:145: position of each candidate in the order (-return, arrival index); the first T are kept.
__global__ __launch_bounds__(256) void k_ga_select(const float *__restrict__ r, int m, int t, int32_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float ri = r[i];
    int pos = 0;
    for (int j = 0; j < m; j++) pos += (r[j] > ri) || (r[j] == ri && j < i);
    if (pos < t) out[pos] = i;
}

// nses.py:12-20 integer part: A = sum_{i < min(n,m)} |x_i - y_i|^2, B = sum over the longer trajectory's
// remaining rows against the shorter one's last row.  One workgroup per archive entry (exact int64 sums).
__global__ __launch_bounds__(256) void k_bc_sqdist(const uint8_t *__restrict__ archive, const int64_t *__restrict__ arow0,
                                                   const int32_t *__restrict__ alen, const uint8_t *__restrict__ bc,
                                                   int m, int dim, long long *__restrict__ out /*[narch][2]*/) {
    const int a = blockIdx.x;
    const uint8_t *x = archive + arow0[a] * dim;
    const int n = alen[a];
    const int lo = n < m ? n : m, hi = n < m ? m : n;
    long long sa = 0, sb = 0;
    for (long long e = threadIdx.x; e < (long long)hi * dim; e += 256) {
        const int i = (int)(e / dim), d = (int)(e % dim);
        const int xi = i < n ? i : n - 1, yi = i < m ? i : m - 1;
        const int df = (int)x[(size_t)xi * dim + d] - (int)bc[(size_t)yi * dim + d];
        if (i < lo) sa += df * df; else sb += df * df;
    }
    __shared__ long long ra[256], rb[256];
    ra[threadIdx.x] = sa; rb[threadIdx.x] = sb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { ra[threadIdx.x] += ra[threadIdx.x + s]; rb[threadIdx.x] += rb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * a] = ra[0]; out[2 * a + 1] = rb[0]; }
}

// batched form over the engine-resident trajectories: block (a, i) = archive entry a vs member i
__global__ __launch_bounds__(256) void k_bc_sqdist_batch(const uint8_t *__restrict__ archive, const int64_t *__restrict__ arow0,
                                                         const int32_t *__restrict__ alen, const uint8_t *__restrict__ bc,
                                                         const int32_t *__restrict__ lens, int bc_stride_rows, int narch,
                                                         long long *__restrict__ out /*[n][narch][2]*/) {
    const int a = blockIdx.x, i = blockIdx.y, dim = 128;
    const uint8_t *x = archive + arow0[a] * dim;
    const uint8_t *y = bc + (size_t)i * bc_stride_rows * dim;
    const int n = alen[a], m = lens[i];
    const int lo = n < m ? n : m, hi = n < m ? m : n;
    long long sa = 0, sb = 0;
    for (int e = threadIdx.x; e < hi * (dim / 4); e += 256) {   // 4 bytes per thread-step
        const int r = e / (dim / 4), d4 = e % (dim / 4);
        const int xi = r < n ? r : n - 1, yi = r < m ? r : m - 1;
        const uint32_t xv = *(const uint32_t *)(x + (size_t)xi * dim + 4 * d4);
        const uint32_t yv = *(const uint32_t *)(y + (size_t)yi * dim + 4 * d4);
        int s = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int df = (int)((xv >> (8 * b)) & 255u) - (int)((yv >> (8 * b)) & 255u);
            s += df * df;
        }
        if (r < lo) sa += s; else sb += s;
    }
    __shared__ long long ra[256], rb[256];
    ra[threadIdx.x] = sa; rb[threadIdx.x] = sb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { ra[threadIdx.x] += ra[threadIdx.x + s]; rb[threadIdx.x] += rb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[((size_t)i * narch + a) * 2] = ra[0]; out[((size_t)i * narch + a) * 2 + 1] = rb[0]; }
}


// GA children written out once per generation: child = parent + scale * noise[off ..] (base.py:141-142 compute_mutation; the same
// two roundings the forward kernels apply on the fly).  grid (ceil(P / 256), children).
__global__ __launch_bounds__(256) void k_materialize_children(const float *__restrict__ noise, float *__restrict__ bases, size_t stride, int P,
                                                              const int32_t *__restrict__ pslot, const int64_t *__restrict__ off,
                                                              const float *__restrict__ scale, const int32_t *__restrict__ cslot) {
    const int p = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (p >= P) return;
    float pv = scale[j] * noise[off[j] + p];
    bases[(size_t)cslot[j] * stride + p] = bases[(size_t)pslot[j] * stride + p] + pv;
}

}  // namespace dne

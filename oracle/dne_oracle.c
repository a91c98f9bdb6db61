/*
 * dne_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See dne_oracle.h for scope, pinning status and the rule on who may use it.
 *
 * Numerics contract (what "bit-exact" means for the HIP engine):
 *   - every dot product of the convolutions and the fc is an fp32 fmaf chain in the stated k order, starting at 0
 *   - fc: 4 quarters of 968 rows, each the left fold of 8 sub-slice chains (fc_raw), combined ((q0+q1)+(q2+q3)) + bias
 *   - the output layer sums its K products by a fixed binary tree (out_raw_k)
 *   - everything else is one IEEE fp32 operation per written operator
 *   built with -ffp-contract=off so the compiler never fuses or splits
 */
#include "dne_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ layout */

void orc_layout_make(int kind, int nact, orc_layout *L) {
    int o = 0;
    memset(L, 0, sizeof(*L));
    L->kind = kind;
    L->nact = nact;
    if (kind == ORC_KIND_ES) { /* policies.py:319-330, tf.contrib.layers variable order */
        L->c1w = o; o += 8 * 8 * 4 * 16;
        L->c1b = o; o += 16;
        L->bn1b = o; o += 16;
        L->bn1g = o; o += 16;
        L->c2w = o; o += 4 * 4 * 16 * 32;
        L->c2b = o; o += 32;
        L->bn2b = o; o += 32;
        L->bn2g = o; o += 32;
        L->fcw = o; o += 3872 * 256;
        L->fcb = o; o += 256;
        L->bn3b = o; o += 256;
        L->bn3g = o; o += 256;
        L->ow = o; o += 256 * nact;
        L->ob = o; o += nact;
    } else if (kind == ORC_KIND_GA_LARGE) { /* models/dqn.py:39-47: variables in creation order (models/base.py:35-41, 157-167) */
        L->c1w = o; o += 8 * 8 * 4 * 32;
        L->c1b = o; o += 32;
        L->c2w = o; o += 4 * 4 * 32 * 64;
        L->c2b = o; o += 64;
        L->c3w = o; o += 3 * 3 * 64 * 64;
        L->c3b = o; o += 64;
        L->fcw = o; o += 7744 * 512;
        L->fcb = o; o += 512;
        L->ow = o; o += 512 * nact;
        L->ob = o; o += nact;
        L->bn1b = L->bn1g = L->bn2b = L->bn2g = L->bn3b = L->bn3g = -1;
    } else { /* policies.py:449-459 via tf_util.py:133-162 */
        L->c1w = o; o += 4096;
        L->c1b = o; o += 16;
        L->c2w = o; o += 8192;
        L->c2b = o; o += 32;
        L->fcw = o; o += 3872 * 256;
        L->fcb = o; o += 256;
        L->ow = o; o += 256 * nact;
        L->ob = o; o += nact;
        L->bn1b = L->bn1g = L->bn2b = L->bn2g = L->bn3b = L->bn3g = -1;
    }
    L->P = o;
}

int orc_num_params(int kind, int nact) {
    orc_layout L;
    orc_layout_make(kind, nact, &L);
    return L.P;
}

/* ----------------------------------------------------------- perturbation */

/* es.py:413 v = noise_stdev * noise.get(idx, P)   (float32 array * python float -> float32)
 * es.py:415 / 419  params + v ; params - v */
void orc_perturb(const float *theta, const float *noise, int64_t idx, float sigma, int sign, int P,
                 float *out) {
    const float s = sign >= 0 ? sigma : -sigma;
    for (int p = 0; p < P; p++) {
        float v = s * noise[idx + p];
        out[p] = theta[p] + v;
    }
}

/* ------------------------------------------------------------- the network */

static float OB_LUT[256]; /* atari_wrappers.py:183-186: float32(u8) / 255.0 */
static int ob_lut_ready = 0;
static void ob_lut_init(void) {
    if (ob_lut_ready) return;
    for (int i = 0; i < 256; i++) OB_LUT[i] = (float)i / 255.0f;
    ob_lut_ready = 1;
}

/* conv1: 8x8 stride 4, SAME (pad 2/2), HWIO weights, NHWC input. policies.py:321 / 451 */
static void conv1_raw_acc(const float *w, const float *b, const uint8_t *ob, float *y1, float *raw /* may be NULL: pre-bias sums */) {
    ob_lut_init();
    for (int oy = 0; oy < 21; oy++)
        for (int ox = 0; ox < 21; ox++) {
            float acc[16];
            for (int c = 0; c < 16; c++) acc[c] = 0.0f;
            for (int kh = 0; kh < 8; kh++) {
                int iy = oy * 4 - 2 + kh;
                if (iy < 0 || iy >= 84) continue;
                for (int kw = 0; kw < 8; kw++) {
                    int ix = ox * 4 - 2 + kw;
                    if (ix < 0 || ix >= 84) continue;
                    for (int ci = 0; ci < 4; ci++) {
                        float x = OB_LUT[ob[(iy * 84 + ix) * 4 + ci]];
                        const float *wk = w + ((kh * 8 + kw) * 4 + ci) * 16;
                        for (int co = 0; co < 16; co++) acc[co] = fmaf(x, wk[co], acc[co]);
                    }
                }
            }
            float *o = y1 + (oy * 21 + ox) * 16;
            for (int co = 0; co < 16; co++) o[co] = acc[co] + b[co];
            if (raw)
                for (int co = 0; co < 16; co++) raw[(oy * 21 + ox) * 16 + co] = acc[co];
        }
}
static void conv1_raw(const float *w, const float *b, const uint8_t *ob, float *y1) { conv1_raw_acc(w, b, ob, y1, NULL); }

/* tf.nn.batch_normalization(x, mean, var, beta, gamma, eps) followed by relu:
 *   inv = rsqrt(var + eps) * gamma ; y = x * inv + (beta - mean * inv)       [external: TF semantics]
 * scale/shift are precomputed per channel (orc_bn_finish); GA passes bn == NULL -> plain relu */
static inline float bn_relu(float x, const float *scale, const float *shift, int c) {
    float t = x;
    if (scale) {
        t = x * scale[c];
        t = t + shift[c];
    }
    return t > 0.0f ? t : 0.0f;
}

/* conv2: 4x4 stride 2, SAME (pad 1 top/left, 2 bottom/right). policies.py:323 / 452 */
static void conv2_raw_acc(const float *w, const float *b, const float *a1 /*[21][21][16]*/, float *y2, float *raw) {
    for (int oy = 0; oy < 11; oy++)
        for (int ox = 0; ox < 11; ox++) {
            float acc[32];
            for (int c = 0; c < 32; c++) acc[c] = 0.0f;
            for (int kh = 0; kh < 4; kh++) {
                int iy = oy * 2 - 1 + kh;
                if (iy < 0 || iy >= 21) continue;
                for (int kw = 0; kw < 4; kw++) {
                    int ix = ox * 2 - 1 + kw;
                    if (ix < 0 || ix >= 21) continue;
                    for (int ci = 0; ci < 16; ci++) {
                        float x = a1[(iy * 21 + ix) * 16 + ci];
                        const float *wk = w + ((kh * 4 + kw) * 16 + ci) * 32;
                        for (int co = 0; co < 32; co++) acc[co] = fmaf(x, wk[co], acc[co]);
                    }
                }
            }
            float *o = y2 + (oy * 11 + ox) * 32;
            for (int co = 0; co < 32; co++) o[co] = acc[co] + b[co];
            if (raw)
                for (int co = 0; co < 32; co++) raw[(oy * 11 + ox) * 32 + co] = acc[co];
        }
}
static void conv2_raw(const float *w, const float *b, const float *a1, float *y2) { conv2_raw_acc(w, b, a1, y2, NULL); }

/* fc 3872 -> 256 (policies.py:327 / 455).  TensorFlow's summation order is unknowable; the order defined
 * here (round 3) gives a GPU 32 independent chains per output instead of 4:
 *   the 3872 inputs are 4 quarters of 968 rows; a quarter is 8 sub-slices of 128, 120, 120, 120, 120, 120, 120, 120 rows
 *   (every boundary a multiple of 8 rows: whole 4-row matrix-core steps and whole 8-row streaming blocks);
 *   sub-slice sum u_i = fmaf chain over its rows in order, from 0;
 *   quarter q = ((((((u0 + u1) + u2) + u3) + u4) + u5) + u6) + u7   (left fold: a kernel that walks a quarter needs one running sum);
 *   y = ((q0 + q1) + (q2 + q3)) + bias.
 * (Rounds 1-2: each quarter was one chain of 968.) */
const int ORC_FC_SUB[9] = {0, 128, 248, 368, 488, 608, 728, 848, 968};   /* sub-slice boundaries within a quarter */
static void fc_raw(const float *w, const float *b, const float *a2, float *y3) {
    static __thread float quarter[4][256];
    float u[256];
    for (int q = 0; q < 4; q++) {
        float *Q = quarter[q];
        for (int i = 0; i < 8; i++) {
            for (int j = 0; j < 256; j++) u[j] = 0.0f;
            for (int k = q * 968 + ORC_FC_SUB[i]; k < q * 968 + ORC_FC_SUB[i + 1]; k++) {
                float x = a2[k];
                const float *wk = w + (size_t)k * 256;
                for (int j = 0; j < 256; j++) u[j] = fmaf(x, wk[j], u[j]);
            }
            if (i == 0)
                for (int j = 0; j < 256; j++) Q[j] = u[j];
            else
                for (int j = 0; j < 256; j++) Q[j] = Q[j] + u[j];
        }
    }
    for (int j = 0; j < 256; j++) {
        float s01 = quarter[0][j] + quarter[1][j];
        float s23 = quarter[2][j] + quarter[3][j];
        float t = s01 + s23;
        y3[j] = t + b[j];
    }
}

/* Sum of 64 values by a balanced binary tree: neighbours first, (x0+x1), (x2+x3), ..., then pairs of pairs, ... -- what 64
 * lanes of a wavefront produce with a butterfly of strides 1, 2, 4, 8, 16, 32 (fp addition is commutative, so both lanes of a
 * pair hold the same sum at every level). */
static float tree64(const float *x) {
    float t[64];
    for (int i = 0; i < 64; i++) t[i] = x[i];
    for (int n = 64; n > 1; n /= 2)
        for (int i = 0; i < n / 2; i++) t[i] = t[2 * i] + t[2 * i + 1];
    return t[0];
}

/* out K -> nact (policies.py:329 / 457; K = 256, LargeModel 512).  TensorFlow's summation order is unknowable;
 * the order defined here (round 3) is the one a GPU forms without a serial chain: the K products p[k] = a3[k] * w[k][a] (one
 * rounding each), every group of 64 consecutive k summed by tree64, the groups combined pairwise in order
 * ((S0+S1)+(S2+S3)) [+ ((S4+S5)+(S6+S7)) for K = 512], + bias.  (Rounds 1-2: one fmaf chain over k.) */
static void out_raw_k(const float *w, const float *b, const float *a3, int K, int nact, float *logits) {
    for (int a = 0; a < nact; a++) {
        float p[512], S[8];
        for (int k = 0; k < K; k++) p[k] = a3[k] * w[k * nact + a];
        for (int g = 0; g < K / 64; g++) S[g] = tree64(p + 64 * g);
        float s01 = S[0] + S[1], s23 = S[2] + S[3];
        float t = s01 + s23;
        if (K == 512) {
            float s45 = S[4] + S[5], s67 = S[6] + S[7];
            float u = s45 + s67;
            t = t + u;
        }
        logits[a] = t + b[a];
    }
}
static void out_raw(const float *w, const float *b, const float *a3, int nact, float *logits) { out_raw_k(w, b, a3, 256, nact, logits); }

/* tf.argmax: index of the first maximum */
static int argmax_first(const float *x, int n) {
    int best = 0;
    for (int i = 1; i < n; i++)
        if (x[i] > x[best]) best = i;
    return best;
}

void orc_forward_debug(const orc_layout *L, const float *th, const float *bn, const uint8_t *ob,
                       float *y1, float *y2, float *y3, float *logits) {
    static __thread float a1[7056], a2[3872], a3[256];
    const float *s1 = bn ? bn : NULL, *h1 = bn ? bn + 16 : NULL;
    const float *s2 = bn ? bn + 32 : NULL, *h2 = bn ? bn + 64 : NULL;
    const float *s3 = bn ? bn + 96 : NULL, *h3 = bn ? bn + 352 : NULL;
    conv1_raw(th + L->c1w, th + L->c1b, ob, y1);
    for (int i = 0; i < 7056; i++) a1[i] = bn_relu(y1[i], s1, h1, i & 15);
    conv2_raw(th + L->c2w, th + L->c2b, a1, y2);
    for (int i = 0; i < 3872; i++) a2[i] = bn_relu(y2[i], s2, h2, i & 31);
    fc_raw(th + L->fcw, th + L->fcb, a2, y3);
    for (int i = 0; i < 256; i++) a3[i] = bn_relu(y3[i], s3, h3, i);
    out_raw(th + L->ow, th + L->ob, a3, L->nact, logits);
}

/* ---- LargeModel (GPU tree).  models/base.py:50-78: extract_image_patches (depth order kh, kw, ci; SAME padding: the
 * pad before is floor(total / 2)) x weights reshaped [k*k*cin, cout], + bias; models/dqn.py:39-47 puts relu on every layer
 * but the last.  One fmaf chain per output in (kh, kw, ci) order, taps outside the image skipped (they would add x = 0). */
static void conv_same(const float *w, const float *b, const float *in, int hin, int cin, int k, int stride, int hout, int cout,
                      float *out) {
    int total = (hout - 1) * stride + k - hin;
    int pad = total > 0 ? total / 2 : 0;
    for (int oy = 0; oy < hout; oy++)
        for (int ox = 0; ox < hout; ox++) {
            float acc[64];
            for (int c = 0; c < cout; c++) acc[c] = 0.0f;
            for (int kh = 0; kh < k; kh++) {
                int iy = oy * stride - pad + kh;
                if (iy < 0 || iy >= hin) continue;
                for (int kw = 0; kw < k; kw++) {
                    int ix = ox * stride - pad + kw;
                    if (ix < 0 || ix >= hin) continue;
                    for (int ci = 0; ci < cin; ci++) {
                        float x = in[(iy * hin + ix) * cin + ci];
                        const float *wk = w + ((size_t)(kh * k + kw) * cin + ci) * cout;
                        for (int co = 0; co < cout; co++) acc[co] = fmaf(x, wk[co], acc[co]);
                    }
                }
            }
            float *o = out + (size_t)(oy * hout + ox) * cout;
            for (int co = 0; co < cout; co++) o[co] = acc[co] + b[co];
        }
}

void orc_forward_large_debug(const orc_layout *L, const float *th, const uint8_t *ob, float *y1, float *y2, float *y3, float *y4,
                             float *logits) {
    static __thread float x0[84 * 84 * 4], a1[21 * 21 * 32], a2[11 * 11 * 64], a3[11 * 11 * 64], a4[512];
    static __thread float part[4][512];
    ob_lut_init();
    for (int i = 0; i < 84 * 84 * 4; i++) x0[i] = OB_LUT[ob[i]];
    conv_same(th + L->c1w, th + L->c1b, x0, 84, 4, 8, 4, 21, 32, y1);
    for (int i = 0; i < 21 * 21 * 32; i++) a1[i] = y1[i] > 0.0f ? y1[i] : 0.0f;
    conv_same(th + L->c2w, th + L->c2b, a1, 21, 32, 4, 2, 11, 64, y2);
    for (int i = 0; i < 11 * 11 * 64; i++) a2[i] = y2[i] > 0.0f ? y2[i] : 0.0f;
    conv_same(th + L->c3w, th + L->c3b, a2, 11, 64, 3, 1, 11, 64, y3);
    for (int i = 0; i < 11 * 11 * 64; i++) a3[i] = y3[i] > 0.0f ? y3[i] : 0.0f;
    const float *w = th + L->fcw;
    for (int s = 0; s < 4; s++) { /* 4 k-slices of 1936 rows, ((s0+s1)+(s2+s3)) + bias */
        float *acc = part[s];
        for (int j = 0; j < 512; j++) acc[j] = 0.0f;
        for (int kk = s * 1936; kk < (s + 1) * 1936; kk++) {
            float x = a3[kk];
            const float *wk = w + (size_t)kk * 512;
            for (int j = 0; j < 512; j++) acc[j] = fmaf(x, wk[j], acc[j]);
        }
    }
    for (int j = 0; j < 512; j++) {
        float s01 = part[0][j] + part[1][j];
        float s23 = part[2][j] + part[3][j];
        float t = s01 + s23;
        y4[j] = t + th[L->fcb + j];
        a4[j] = y4[j] > 0.0f ? y4[j] : 0.0f;
    }
    out_raw_k(th + L->ow, th + L->ob, a4, 512, L->nact, logits);
}

int orc_act(const orc_layout *L, const float *th, const float *bn, const uint8_t *ob, float *logits) {
    static __thread float y1[21 * 21 * 32], y2[11 * 11 * 64], y3[11 * 11 * 64];
    float lg[32];
    if (L->kind == ORC_KIND_GA_LARGE) {
        static __thread float y4[512];
        orc_forward_large_debug(L, th, ob, y1, y2, y3, y4, lg);
        if (logits) memcpy(logits, lg, sizeof(float) * L->nact);
        return argmax_first(lg, L->nact);
    }
    orc_forward_debug(L, th, bn, ob, y1, y2, y3, lg);
    if (logits) memcpy(logits, lg, sizeof(float) * L->nact);
    return argmax_first(lg, L->nact);
}

/* batch moments, tf.nn.moments semantics (biased variance) [external], with the
 * summation order fixed as: per frame a sequential sum over positions, then a
 * sequential sum over frames.  y: [nref][npos][C].  decay=0 => moving stats := batch stats. */
static void bn_finish(const float *y, int nref, int npos, int C, const float *beta, const float *gamma,
                      float *scale, float *shift, float *mean_out, float *var_out) {
    const float count = (float)(nref * npos);
    for (int c = 0; c < C; c++) {
        float tot = 0.0f;
        for (int n = 0; n < nref; n++) {
            float s = 0.0f;
            const float *yn = y + (size_t)n * npos * C + c;
            for (int p = 0; p < npos; p++) s = s + yn[(size_t)p * C];
            tot = tot + s;
        }
        float mean = tot / count;
        float totq = 0.0f;
        for (int n = 0; n < nref; n++) {
            float q = 0.0f;
            const float *yn = y + (size_t)n * npos * C + c;
            for (int p = 0; p < npos; p++) {
                float d = yn[(size_t)p * C] - mean;
                q = fmaf(d, d, q);
            }
            totq = totq + q;
        }
        float var = totq / count;
        float inv = 1.0f / sqrtf(var + 1e-3f);
        float sc = inv * gamma[c];
        float ms = mean * sc;
        scale[c] = sc;
        shift[c] = beta[c] - ms;
        if (mean_out) { mean_out[c] = mean; var_out[c] = var; }
    }
}

/* Batch moments of a convolution layer, in one pass over the pre-bias sums a = y - bias (the bias is the shift of
 * tf.nn.moments' sufficient statistics [external: TF computes shifted sums and mean = shift + m_ss / count,
 * var = v_ss / count - (m_ss / count)^2]; TF's own shift and summation order are unknowable).
 * The summation order is a fixed tree over the 16-position tiles of the matrix-core convolution, so that the GPU
 * forms the sums in the convolution's epilogue instead of re-reading the activations:
 *   tile t = positions 16t .. 16t+15 (positions >= npos count as exact zeros);
 *   row group g = 4 consecutive positions:  s_g = ((a0 + a1) + a2) + a3,  q_g = fma(a3,a3, fma(a2,a2, fma(a1,a1, a0*a0)));
 *   tile:   T = (s_0 + s_1) + (s_2 + s_3)                                  (same for q);
 *   frame:  ngroups interleaved (conv1: 4, tile t in group t % 4) or blocked (conv2: 2, tile t in group t / 4) tile
 *           groups, each summed sequentially in tile order from 0; conv1: Fr = (W0 + W1) + (W2 + W3), conv2: Fr = W0 + W1;
 *   batch:  S = sum over frames in order, from 0.
 * raw: [nref][npos][C] pre-bias sums. */
static void bn_finish_tiles(const float *raw, int nref, int npos, int C, int interleaved, const float *bias,
                            const float *beta, const float *gamma, float *scale, float *shift, float *mean_out, float *var_out) {
    const int ntile = (npos + 15) / 16, ngroups = interleaved ? 4 : 2;
    const float count = (float)(nref * npos);
    for (int c = 0; c < C; c++) {
        float S = 0.0f, Q = 0.0f;
        for (int n = 0; n < nref; n++) {
            const float *yn = raw + (size_t)n * npos * C + c;
            float Ws[4] = {0.0f, 0.0f, 0.0f, 0.0f}, Wq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int t = 0; t < ntile; t++) {
                float sg[4], qg[4];
                for (int g = 0; g < 4; g++) {
                    float a[4];
                    for (int r = 0; r < 4; r++) {
                        int p = t * 16 + g * 4 + r;
                        a[r] = p < npos ? yn[(size_t)p * C] : 0.0f;
                    }
                    float s01 = a[0] + a[1];
                    float s012 = s01 + a[2];
                    sg[g] = s012 + a[3];
                    float q = a[0] * a[0];
                    q = fmaf(a[1], a[1], q);
                    q = fmaf(a[2], a[2], q);
                    qg[g] = fmaf(a[3], a[3], q);
                }
                float s_lo = sg[0] + sg[1], s_hi = sg[2] + sg[3];
                float q_lo = qg[0] + qg[1], q_hi = qg[2] + qg[3];
                float Ts = s_lo + s_hi, Tq = q_lo + q_hi;
                int grp = interleaved ? (t & 3) : (t >> 2);
                Ws[grp] = Ws[grp] + Ts;
                Wq[grp] = Wq[grp] + Tq;
            }
            float Fs, Fq;
            if (ngroups == 4) {
                float a = Ws[0] + Ws[1], b = Ws[2] + Ws[3];
                Fs = a + b;
                float cq = Wq[0] + Wq[1], dq = Wq[2] + Wq[3];
                Fq = cq + dq;
            } else {
                Fs = Ws[0] + Ws[1];
                Fq = Wq[0] + Wq[1];
            }
            S = S + Fs;
            Q = Q + Fq;
        }
        float m = S / count;
        float mean = bias[c] + m;
        float q1 = Q / count;
        float mm = m * m;
        float var = q1 - mm;
        var = var > 0.0f ? var : 0.0f;
        float inv = 1.0f / sqrtf(var + 1e-3f);
        float sc = inv * gamma[c];
        float ms = mean * sc;
        scale[c] = sc;
        shift[c] = beta[c] - ms;
        if (mean_out) { mean_out[c] = mean; var_out[c] = var; }
    }
}

/* mom (may be NULL): the batch moments themselves, mean / variance in bn's layout -- what batch_norm(decay=0)
 * leaves in moving_mean / moving_variance (policies.py:322-328) */
void orc_es_ref_pass_moments(const orc_layout *L, const float *th, const uint8_t *ref, int nref, float *bn, float *mom) {
    float *y1 = (float *)malloc(sizeof(float) * (size_t)nref * 7056);
    float *y2 = (float *)malloc(sizeof(float) * (size_t)nref * 3872);
    float *y3 = (float *)malloc(sizeof(float) * (size_t)nref * 256);
    float *raw = (float *)malloc(sizeof(float) * (size_t)nref * 7056);
    float *a = (float *)malloc(sizeof(float) * 7056);
    for (int n = 0; n < nref; n++)
        conv1_raw_acc(th + L->c1w, th + L->c1b, ref + (size_t)n * ORC_OB_BYTES, y1 + (size_t)n * 7056, raw + (size_t)n * 7056);
    bn_finish_tiles(raw, nref, 441, 16, 1, th + L->c1b, th + L->bn1b, th + L->bn1g, bn, bn + 16, mom, mom ? mom + 16 : NULL);
    for (int n = 0; n < nref; n++) {
        for (int i = 0; i < 7056; i++) a[i] = bn_relu(y1[(size_t)n * 7056 + i], bn, bn + 16, i & 15);
        conv2_raw_acc(th + L->c2w, th + L->c2b, a, y2 + (size_t)n * 3872, raw + (size_t)n * 3872);
    }
    bn_finish_tiles(raw, nref, 121, 32, 0, th + L->c2b, th + L->bn2b, th + L->bn2g, bn + 32, bn + 64, mom ? mom + 32 : NULL, mom ? mom + 64 : NULL);
    for (int n = 0; n < nref; n++) {
        for (int i = 0; i < 3872; i++) a[i] = bn_relu(y2[(size_t)n * 3872 + i], bn + 32, bn + 64, i & 31);
        fc_raw(th + L->fcw, th + L->fcb, a, y3 + (size_t)n * 256);
    }
    bn_finish(y3, nref, 1, 256, th + L->bn3b, th + L->bn3g, bn + 96, bn + 352, mom ? mom + 96 : NULL, mom ? mom + 352 : NULL);   /* fc: 128 values per column, two passes */
    free(y1); free(y2); free(y3); free(a); free(raw);
}

void orc_es_ref_pass(const orc_layout *L, const float *th, const uint8_t *ref, int nref, float *bn) {
    orc_es_ref_pass_moments(L, th, ref, nref, bn, NULL);
}

/* ----------------------------------------------------- SynthAtari (fixture) */
/* RAM map -- keep in sync with DESIGN.md "SynthAtari" */
enum {
    R_FC0 = 0, R_FC1 = 1, R_RNG = 2, R_PX = 6, R_PROW = 7, R_LIVES = 8, R_GO = 9, R_TEMP = 10,
    R_COOL = 11, R_OFF = 12, R_VIS = 16, R_IGLOO = 20, R_LEVEL = 21, R_SCORE = 22, R_FREEZE = 25,
    R_HZX = 26, R_HZA = 30, R_DIR = 34, R_LASTA = 38, R_TICK = 39
};
static const int8_t ACT_DX[18] = {0, 0, 0, 1, -1, 0, 1, -1, 1, -1, 0, 1, -1, 0, 1, -1, 1, -1};
static const int8_t ACT_DY[18] = {0, 0, -1, 0, 0, 1, -1, -1, 1, 1, -1, 0, 0, 1, -1, -1, 1, 1};
static const int8_t ACT_FIRE[18] = {0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1};

static const uint8_t PALETTE[16][3] = {
    {0, 0, 0},       {170, 170, 170}, {45, 50, 184},   {24, 26, 167},  {214, 214, 214}, {0, 28, 136},
    {236, 236, 236}, {84, 138, 210},  {198, 108, 58},  {181, 83, 40},  {192, 192, 192}, {252, 252, 84},
    {92, 186, 92},   {74, 74, 74},    {252, 144, 144}, {0, 44, 160}};

void orc_palette(uint8_t *rgb) { memcpy(rgb, PALETTE, 48); }

static uint32_t ram_rng_next(uint8_t *ram) {
    uint32_t s = (uint32_t)ram[R_RNG] | ((uint32_t)ram[R_RNG + 1] << 8) | ((uint32_t)ram[R_RNG + 2] << 16) |
                 ((uint32_t)ram[R_RNG + 3] << 24);
    s = s * 1664525u + 1013904223u;
    ram[R_RNG] = (uint8_t)s; ram[R_RNG + 1] = (uint8_t)(s >> 8);
    ram[R_RNG + 2] = (uint8_t)(s >> 16); ram[R_RNG + 3] = (uint8_t)(s >> 24);
    return s >> 16;
}

void orc_raw_reset(uint8_t *ram, uint32_t seed) {
    memset(ram, 0, ORC_RAM);
    uint32_t s = seed ^ 0x9E3779B9u;
    ram[R_RNG] = (uint8_t)s; ram[R_RNG + 1] = (uint8_t)(s >> 8);
    ram[R_RNG + 2] = (uint8_t)(s >> 16); ram[R_RNG + 3] = (uint8_t)(s >> 24);
    ram[R_PX] = 76;
    ram[R_LIVES] = 3;
    ram[R_TEMP] = 45;
    for (int r = 0; r < 4; r++) {
        ram[R_OFF + r] = (uint8_t)(ram_rng_next(ram) % 160u);
        ram[R_DIR + r] = (uint8_t)(r & 1);
    }
    for (int r = 0; r < 4; r++) ram[R_HZX + r] = (uint8_t)(ram_rng_next(ram) % 160u);
}

static int on_floe(const uint8_t *ram, int px, int r) {
    int rel = (px + 4 + 160 - ram[R_OFF + r]) % 160;
    return (rel % 40) < 32;
}

int orc_raw_frame(uint8_t *ram, int a) {
    if (ram[R_GO]) return 0;
    int fc = (ram[R_FC0] | (ram[R_FC1] << 8));
    fc = (fc + 1) & 0xffff;
    ram[R_FC0] = (uint8_t)fc; ram[R_FC1] = (uint8_t)(fc >> 8);
    ram[R_LASTA] = (uint8_t)a;
    int rew = 0;
    int level = ram[R_LEVEL];
    int speed = level >= 3 ? 2 : 1;
    for (int r = 0; r < 4; r++) {
        int off = ram[R_OFF + r];
        off = ram[R_DIR + r] ? (off + 160 - speed) % 160 : (off + speed) % 160;
        ram[R_OFF + r] = (uint8_t)off;
    }
    for (int r = 0; r < 4; r++) {
        if (ram[R_HZA + r]) {
            int x = ram[R_HZX + r];
            x = ram[R_DIR + r] ? (x + 1) % 160 : (x + 159) % 160;
            ram[R_HZX + r] = (uint8_t)x;
        } else if ((fc & 63) == 16 * r) {
            uint32_t v = ram_rng_next(ram);
            if ((v & 3u) == 0u) {
                ram[R_HZA + r] = 1;
                ram[R_HZX + r] = ram[R_DIR + r] ? 0 : 159;
            }
        }
    }
    int dx = ACT_DX[a], dy = ACT_DY[a], fire = ACT_FIRE[a];
    int died = 0;
    if (ram[R_FREEZE] > 0) {
        ram[R_FREEZE]--;
    } else {
        int px = ram[R_PX], prow = ram[R_PROW];
        if (prow > 0) { /* carried by the floe; the screen wraps on the ice rows */
            px += ram[R_DIR + prow - 1] ? -speed : speed;
            px = (px + 2 * dx + 160) % 160;
        } else {
            px += 2 * dx;
            if (px < 8) px = 8;
            if (px > 144) px = 144;
        }
        if (ram[R_COOL] > 0) {
            ram[R_COOL]--;
        } else if (dy != 0) {
            int tgt = prow + dy;
            if (tgt < 0) {
                if (ram[R_IGLOO] >= 16 && px >= 104) {
                    rew += 10 * ram[R_TEMP] + 100;
                    if (level < 255) level++;
                    ram[R_LEVEL] = (uint8_t)level;
                    ram[R_IGLOO] = 0;
                    for (int r = 0; r < 4; r++) { ram[R_VIS + r] = 0; ram[R_HZA + r] = 0; }
                    ram[R_TEMP] = 45;
                    ram[R_TICK] = 0;
                    prow = 0;
                    px = 76;
                    ram[R_FREEZE] = 64;
                }
            } else if (tgt <= 4) {
                prow = tgt;
                ram[R_COOL] = 12;
                if (prow == 0) { if (px < 8) px = 8; if (px > 144) px = 144; }
                if (prow > 0) {
                    int r = prow - 1;
                    if (on_floe(ram, px, r)) {
                        if (!ram[R_VIS + r]) {
                            ram[R_VIS + r] = 1;
                            rew += 10;
                            if (ram[R_IGLOO] < 16) ram[R_IGLOO]++;
                            if (ram[R_VIS] && ram[R_VIS + 1] && ram[R_VIS + 2] && ram[R_VIS + 3])
                                for (int q = 0; q < 4; q++) ram[R_VIS + q] = 0;
                        }
                    } else {
                        died = 1;
                    }
                }
            }
        } else if (fire && prow > 0) {
            if (ram[R_IGLOO] > 0) {
                ram[R_DIR + prow - 1] ^= 1;
                ram[R_IGLOO]--;
                ram[R_COOL] = 12;
            }
        }
        if (!died && prow > 0) {
            int r = prow - 1;
            if (!on_floe(ram, px, r)) died = 1;
            if (!died && ram[R_HZA + r]) {
                int d = px + 4 - (int)ram[R_HZX + r];
                if (d < 0) d = -d;
                if (d < 8) died = 1;
            }
        }
        ram[R_PX] = (uint8_t)px;
        ram[R_PROW] = (uint8_t)prow;
    }
    ram[R_TICK]++;
    if (ram[R_TICK] >= 48) {
        ram[R_TICK] = 0;
        if (ram[R_TEMP] > 0) ram[R_TEMP]--;
        if (ram[R_TEMP] == 0) died = 1;
    }
    if (died) {
        if (ram[R_LIVES] == 0) ram[R_GO] = 1;
        else ram[R_LIVES]--;
        ram[R_PROW] = 0;
        ram[R_PX] = 76;
        ram[R_FREEZE] = 128; /* death sequence, about two seconds as in the real game */
        ram[R_COOL] = 0;
        for (int r = 0; r < 4; r++) ram[R_HZA + r] = 0;
        if (ram[R_TEMP] == 0) ram[R_TEMP] = 45;
    }
    if (rew) {
        uint32_t sc = ram[R_SCORE] | (ram[R_SCORE + 1] << 8) | ((uint32_t)ram[R_SCORE + 2] << 16);
        sc = (sc + (uint32_t)(rew / 10)) & 0xffffffu;
        ram[R_SCORE] = (uint8_t)sc; ram[R_SCORE + 1] = (uint8_t)(sc >> 8); ram[R_SCORE + 2] = (uint8_t)(sc >> 16);
    }
    return rew;
}

static uint8_t render_pixel(const uint8_t *ram, int x, int y) {
    /* player on top */
    int prow = ram[R_PROW], px = ram[R_PX];
    int fc = ram[R_FC0] | (ram[R_FC1] << 8);
    int py0 = prow == 0 ? 62 : 80 + 32 * (prow - 1);
    int blink = ram[R_FREEZE] > 0 && (fc & 4);
    if (!blink && ((x - px + 160) % 160) < 8 && y >= py0 && y < py0 + 16) return 8;
    if (y < 8) return 0;
    if (y < 16) {
        if (x >= 8 && x < 8 + 2 * ram[R_TEMP]) return 11;
        for (int i = 0; i < ram[R_LIVES]; i++)
            if (x >= 120 + 10 * i && x < 126 + 10 * i) return 12;
        return 1;
    }
    if (y < 20) {
        if (x >= 8 && x < 136 && ((x - 8) % 8) < 6) {
            int b = (x - 8) / 8;
            int sc = ram[R_SCORE] | (ram[R_SCORE + 1] << 8);
            if ((sc >> b) & 1) return 14;
        }
        return 1;
    }
    if (y < 64) {
        if (x >= 112 && x < 144 && y >= 40) {
            int ig = ram[R_IGLOO];
            if (ig >= 16 && x >= 124 && x < 132 && y >= 52) return 13;
            int bx = (x - 112) / 8, by = (63 - y) / 6;
            if (by * 4 + bx < ig) return 10;
        }
        return (ram[R_LEVEL] & 1) ? 3 : 2;
    }
    if (y < 80) return 4;
    if (y < 208) {
        int r = (y - 80) / 32, yo = (y - 80) % 32;
        if (yo >= 4 && yo < 12 && ram[R_HZA + r]) {
            int d = x - (int)ram[R_HZX + r];
            if (d < 0) d = -d;
            if (d < 6) return 9;
        }
        if (yo >= 16 && yo < 28) {
            int rel = (x + 160 - ram[R_OFF + r]) % 160;
            if ((rel % 40) < 32) return ram[R_VIS + r] ? 7 : 6;
        }
        return ((yo >> 3) & 1) ? 15 : 5;
    }
    return 0;
}

void orc_raw_render(const uint8_t *ram, uint8_t *screen) {
    for (int y = 0; y < 210; y++)
        for (int x = 0; x < 160; x++) screen[y * 160 + x] = render_pixel(ram, x, y);
}

/* ------------------------------------------ WarpFrame (PIL BILINEAR resize) */
/* Pillow Resample.c precompute_coeffs + ImagingResampleHorizontal/Vertical_32bpc
 * for mode "F" (third-party, absent from the reference tree; restated from the
 * published algorithm and checked against the PIL installed here):
 *   support = scale (downscaling), taps k[x] = triangle((x+xmin-center+0.5)/scale) normalised,
 *   double accumulation, float32 intermediate image, horizontal pass first. */
#define KH_SIZE 5 /* ceil(160/84)*2+1 */
#define KV_SIZE 7 /* ceil(210/84)*2+1 */
static double KH[84 * KH_SIZE], KV[84 * KV_SIZE];
static int BH[84 * 2], BV[84 * 2];
static int resize_ready = 0;

static void precompute_coeffs(int inSize, int outSize, int ksize, int *bounds, double *kk) {
    double scale = (double)((float)inSize - 0.0f) / outSize;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    double support = 1.0 * filterscale;
    for (int xx = 0; xx < outSize; xx++) {
        double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        double *k = kk + xx * ksize;
        int x;
        for (x = 0; x < xmax; x++) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            double w = t < 1.0 ? 1.0 - t : 0.0;
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0.0;
        bounds[xx * 2] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
}

static void resize_init(void) {
    if (resize_ready) return;
    precompute_coeffs(160, 84, KH_SIZE, BH, KH);
    precompute_coeffs(210, 84, KV_SIZE, BV, KV);
    resize_ready = 1;
}

void orc_resize_tables(double *kh, int *bh, double *kv, int *bv) {
    resize_init();
    memcpy(kh, KH, sizeof(KH)); memcpy(bh, BH, sizeof(BH));
    memcpy(kv, KV, sizeof(KV)); memcpy(bv, BV, sizeof(BV));
}

/* gray float frame [210][160] -> u8 [84][84] */
static void resize_gray(const float *gray, uint8_t *out) {
    static __thread float tmp[210 * 84];
    resize_init();
    for (int yy = 0; yy < 210; yy++)
        for (int xx = 0; xx < 84; xx++) {
            int xmin = BH[xx * 2], xmax = BH[xx * 2 + 1];
            const double *k = KH + xx * KH_SIZE;
            double ss = 0.0;
            for (int x = 0; x < xmax; x++) ss += (double)gray[yy * 160 + x + xmin] * k[x];
            tmp[yy * 84 + xx] = (float)ss;
        }
    for (int yy = 0; yy < 84; yy++) {
        int ymin = BV[yy * 2], ymax = BV[yy * 2 + 1];
        const double *k = KV + yy * KV_SIZE;
        for (int xx = 0; xx < 84; xx++) {
            double ss = 0.0;
            for (int y = 0; y < ymax; y++) ss += (double)tmp[(y + ymin) * 84 + xx] * k[y];
            float f = (float)ss;
            out[yy * 84 + xx] = (uint8_t)f; /* np.array(img, dtype=np.uint8): truncation */
        }
    }
}

/* atari_wrappers.py:139  frame = dot(obs.astype(f32), [0.299, 0.587, 0.114] f32)
 * op order fixed as (r*0.299 + g*0.587) + b*0.114, one fp32 rounding per operator */
static inline float gray_of(uint8_t r, uint8_t g, uint8_t b) {
    float t0 = (float)r * 0.299f;
    float t1 = (float)g * 0.587f;
    float t2 = (float)b * 0.114f;
    float s = t0 + t1;
    return s + t2;
}

void orc_warp_rgb(const uint8_t *rgb, uint8_t *out) {
    static __thread float gray[210 * 160];
    for (int i = 0; i < 210 * 160; i++) gray[i] = gray_of(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
    resize_gray(gray, out);
}

/* MaxAndSkipEnv max over the last two raw frames (atari_wrappers.py:105) + WarpFrame */
static void observe(const uint8_t *ram_prev, const uint8_t *ram_cur, uint8_t *out) {
    static __thread uint8_t sa[210 * 160], sb[210 * 160];
    static __thread float gray[210 * 160];
    orc_raw_render(ram_prev, sa);
    orc_raw_render(ram_cur, sb);
    for (int i = 0; i < 210 * 160; i++) {
        const uint8_t *ca = PALETTE[sa[i]], *cb = PALETTE[sb[i]];
        uint8_t r = ca[0] > cb[0] ? ca[0] : cb[0];
        uint8_t g = ca[1] > cb[1] ? ca[1] : cb[1];
        uint8_t b = ca[2] > cb[2] ? ca[2] : cb[2];
        gray[i] = gray_of(r, g, b);
    }
    resize_gray(gray, out);
}

static void skip_step(orc_wenv *e, int action, int *total, int *done) {
    /* atari_wrappers.py:95-107 */
    *total = 0;
    *done = 0;
    for (int i = 0; i < 4; i++) {
        memcpy(e->ram_prev, e->ram_cur, ORC_RAM);
        *total += orc_raw_frame(e->ram_cur, action);
        if (e->ram_cur[R_GO]) { *done = 1; break; }
    }
}

void orc_wenv_reset(orc_wenv *e, uint32_t seed) {
    /* atari_wrappers.py:18-31: raw reset + noops in [1,30] of action 0 (count fixed by the seed,
     * SURVEY 8d), :40-48 FIRE skipped step then action 2 skipped step, :167-171 fill the stack */
    orc_raw_reset(e->ram_cur, seed);
    memcpy(e->ram_prev, e->ram_cur, ORC_RAM);
    int noops = 1 + (int)(seed % 30u);
    for (int i = 0; i < noops; i++) {
        memcpy(e->ram_prev, e->ram_cur, ORC_RAM);
        orc_raw_frame(e->ram_cur, 0);
    }
    int tot, done;
    skip_step(e, 1, &tot, &done);
    skip_step(e, 2, &tot, &done);
    e->done = 0;
    uint8_t frame[84 * 84];
    observe(e->ram_prev, e->ram_cur, frame);
    for (int i = 0; i < 84 * 84; i++)
        for (int c = 0; c < 4; c++) e->stack[i * 4 + c] = frame[i];
}

void orc_wenv_step(orc_wenv *e, int action, float *reward, int *done) {
    int tot, d;
    skip_step(e, action, &tot, &d);
    uint8_t frame[84 * 84];
    observe(e->ram_prev, e->ram_cur, frame);
    for (int i = 0; i < 84 * 84; i++) { /* FrameStack: drop oldest, append newest (atari_wrappers.py:173-180) */
        uint8_t *s = e->stack + i * 4;
        s[0] = s[1]; s[1] = s[2]; s[2] = s[3]; s[3] = frame[i];
    }
    e->done = d;
    *reward = (float)tot;
    *done = d;
}

/* ----------------------------------------------------------------- rollout */

void orc_rollout(const orc_layout *L, const float *theta, const uint8_t *ref, int nref, uint32_t env_seed,
                 int tslimit, float *ret, float *signret, int32_t *len, uint8_t *bc, int32_t *actions_out) {
    orc_wenv e;
    float bn[ORC_BN_FLOATS];
    const float *bnp = NULL;
    if (tslimit > ORC_ENV_MAX_EPISODE_STEPS) tslimit = ORC_ENV_MAX_EPISODE_STEPS; /* policies.py:383-385 */
    orc_wenv_reset(&e, env_seed); /* policies.py:398 */
    if (L->kind == ORC_KIND_ES) { /* policies.py:399: reference batch through the perturbed net */
        orc_es_ref_pass(L, theta, ref, nref, bn);
        bnp = bn;
    }
    float r = 0.0f, s = 0.0f;
    int t = 0;
    for (int i = 0; i < tslimit; i++) {
        int a = orc_act(L, theta, bnp, e.stack, NULL);
        float rew;
        int done;
        orc_wenv_step(&e, a, &rew, &done);
        if (actions_out) actions_out[t] = a;
        if (bc && L->kind == ORC_KIND_ES) memcpy(bc + (size_t)t * ORC_RAM, e.ram_cur, ORC_RAM); /* policies.py:410,418 */
        r += rew;                                   /* es.py:425 rews.sum() */
        s += (rew > 0.0f) - (rew < 0.0f);           /* es.py:423 np.sign(rews).sum() */
        t++;
        if (done) break;
    }
    if (bc && L->kind != ORC_KIND_ES) memcpy(bc, e.ram_cur, ORC_RAM); /* policies.py:510 */
    *ret = r; *signret = s; *len = t;
}

void orc_es_eval(const orc_layout *L, const float *theta, const float *noise, const int64_t *idx, int n,
                 float sigma, int tslimit, const uint8_t *ref, int nref, const uint32_t *env_seed,
                 float *returns_n2, float *signreturns_n2, int32_t *lengths_n2) {
    float *th = (float *)malloc(sizeof(float) * L->P);
    for (int i = 0; i < n; i++)
        for (int s = 0; s < 2; s++) {
            orc_perturb(theta, noise, idx[i], sigma, s == 0 ? 1 : -1, L->P, th);
            orc_rollout(L, th, ref, nref, env_seed[2 * i + s], tslimit, &returns_n2[2 * i + s],
                        &signreturns_n2[2 * i + s], &lengths_n2[2 * i + s], NULL, NULL);
        }
    free(th);
}

/* ------------------------------------------------------------------ reduce */

void orc_centered_ranks(const float *x, int n, float *y) {
    /* es.py:70-85; argsort ties resolved by flat index */
    const float denom = (float)(n - 1);
    for (int i = 0; i < n; i++) {
        int rank = 0;
        for (int j = 0; j < n; j++) rank += (x[j] < x[i]) || (x[j] == x[i] && j < i);
        float r = (float)rank;
        r = r / denom;
        y[i] = r - 0.5f;
    }
}

void orc_weighted_sum(const float *noise, const int64_t *idx, const float *w, int N, int P, float denom,
                      float *g) {
    /* es.py:291-296; order: i = 0..N-1 fmaf chain per parameter, then g /= returns_n2.size */
    for (int p = 0; p < P; p++) g[p] = 0.0f;
    for (int i = 0; i < N; i++) {
        const float *e = noise + idx[i];
        const float wi = w[i];
        for (int p = 0; p < P; p++) g[p] = fmaf(wi, e[p], g[p]);
    }
    for (int p = 0; p < P; p++) g[p] = g[p] / denom;
}

double orc_adam_update(float *theta, float *m, float *v, const float *g, int P, float l2coeff, int t,
                       double stepsize, double beta1, double beta2, double epsilon) {
    /* es.py:298 globalg = -g + l2coeff*theta ; optimizers.py:45-50 with every array op in float32
     * (numpy 1.12 value-based casting) ; optimizers.py:10-17 */
    double a = stepsize * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
    const float na = (float)(-a), b1 = (float)beta1, b2 = (float)beta2;
    const float ob1 = (float)(1.0 - beta1), ob2 = (float)(1.0 - beta2), eps = (float)epsilon;
    double ns = 0.0, nt = 0.0;
    for (int p = 0; p < P; p++) {
        float ng = -g[p];
        float l2 = l2coeff * theta[p];
        float gg = ng + l2;
        float m1 = b1 * m[p];
        float m2 = ob1 * gg;
        float mm = m1 + m2;
        float g2 = gg * gg;
        float v1 = b2 * v[p];
        float v2 = ob2 * g2;
        float vv = v1 + v2;
        float num = na * mm;
        float den = sqrtf(vv);
        den = den + eps;
        float step = num / den;
        nt += (double)theta[p] * theta[p];
        ns += (double)step * step;
        m[p] = mm;
        v[p] = vv;
        theta[p] = theta[p] + step;
    }
    return sqrt(ns) / sqrt(nt);
}

double orc_sgd_update(float *theta, float *v, const float *g, int P, float l2coeff, double stepsize,
                      double momentum) {
    /* optimizers.py:29-32 */
    const float mo = (float)momentum, om = (float)(1.0 - momentum), nlr = (float)(-stepsize);
    double ns = 0.0, nt = 0.0;
    for (int p = 0; p < P; p++) {
        float ng = -g[p];
        float l2 = l2coeff * theta[p];
        float gg = ng + l2;
        float v1 = mo * v[p];
        float v2 = om * gg;
        float vv = v1 + v2;
        float step = nlr * vv;
        nt += (double)theta[p] * theta[p];
        ns += (double)step * step;
        v[p] = vv;
        theta[p] = theta[p] + step;
    }
    return sqrt(ns) / sqrt(nt);
}

/* ---------------------------------------------------------------------- GA */

static void normc_tensor(float *w, int K, int C, float std) {
    /* tf_util.py:122-130: out *= std / sqrt(square(out).sum(axis=0)); numpy's axis-0 reduction adds
     * row by row, i.e. sequentially in k */
    for (int c = 0; c < C; c++) {
        float ss = 0.0f;
        for (int k = 0; k < K; k++) {
            float x = w[(size_t)k * C + c];
            float sq = x * x;
            ss = ss + sq;
        }
        float rt = sqrtf(ss);
        float sc = std / rt;
        for (int k = 0; k < K; k++) w[(size_t)k * C + c] = w[(size_t)k * C + c] * sc;
    }
}

void orc_ga_normc(const orc_layout *L, float *th) {
    /* policies.py:42-44 reinitialize(): weights column-normalised, biases zeroed
     * (tf_util.py:137,143,152,158); std 1.0 except out (ac_init_std = 0.1, policies.py:434,457) */
    normc_tensor(th + L->c1w, 256, 16, 1.0f);
    memset(th + L->c1b, 0, sizeof(float) * 16);
    normc_tensor(th + L->c2w, 256, 32, 1.0f);
    memset(th + L->c2b, 0, sizeof(float) * 32);
    normc_tensor(th + L->fcw, 3872, 256, 1.0f);
    memset(th + L->fcb, 0, sizeof(float) * 256);
    normc_tensor(th + L->ow, 256, L->nact, 0.1f);
    memset(th + L->ob, 0, sizeof(float) * L->nact);
}

void orc_ga_rebuild(const orc_layout *L, const float *noise, const int64_t *seeds, int nseeds, float sigma,
                    float *th) {
    /* ga.py:256-264 */
    memcpy(th, noise + seeds[0], sizeof(float) * L->P);
    orc_ga_normc(L, th);
    for (int s = 1; s < nseeds; s++) {
        const float *e = noise + seeds[s];
        for (int p = 0; p < L->P; p++) {
            float v = sigma * e[p];
            th[p] = th[p] + v;
        }
    }
}

void orc_ga_select(const float *returns, int M, int T, int32_t *out_idx) {
    /* ga.py:145 top-T by return; order defined as (-return, arrival index) */
    uint8_t *used = (uint8_t *)calloc(M, 1);
    for (int t = 0; t < T; t++) {
        int best = -1;
        for (int i = 0; i < M; i++) {
            if (used[i]) continue;
            if (best < 0 || returns[i] > returns[best]) best = i;
        }
        used[best] = 1;
        out_idx[t] = best;
    }
    free(used);
}

/* ----------------------------------------------------------------- novelty */

static double sqdist_rows(const uint8_t *a, const uint8_t *b, int dim) {
    int64_t s = 0;
    for (int i = 0; i < dim; i++) {
        int d = (int)a[i] - (int)b[i];
        s += d * d;
    }
    return (double)s;
}

double orc_bc_distance(const uint8_t *x, int n, const uint8_t *y, int m, int dim) {
    /* nses.py:12-20: the shorter trajectory is padded with its last row; float64.
     * integer sums are exact, so a = sqrt(A), b = sqrt(B), result sqrt(a^2 + b^2) follows the
     * reference's three roundings */
    const uint8_t *lng = n > m ? x : y, *sht = n > m ? y : x;
    int nl = n > m ? n : m, ns = n > m ? m : n;
    double A = 0.0, B = 0.0;
    for (int i = 0; i < ns; i++) A += sqdist_rows(lng + (size_t)i * dim, sht + (size_t)i * dim, dim);
    for (int i = ns; i < nl; i++) B += sqdist_rows(lng + (size_t)i * dim, sht + (size_t)(ns - 1) * dim, dim);
    double a = sqrt(A), b = sqrt(B);
    return sqrt(a * a + b * b);
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

double orc_novelty(const uint8_t *const *archive, const int *archive_len, int narchive, const uint8_t *bc,
                   int bc_len, int dim, int k) {
    /* nses.py:22-32 mean of the k smallest distances */
    double *d = (double *)malloc(sizeof(double) * narchive);
    for (int i = 0; i < narchive; i++) d[i] = orc_bc_distance(archive[i], archive_len[i], bc, bc_len, dim);
    qsort(d, narchive, sizeof(double), cmp_double);
    int kk = k < narchive ? k : narchive;
    double s = 0.0;
    for (int i = 0; i < kk; i++) s += d[i];
    free(d);
    return s / kk;
}

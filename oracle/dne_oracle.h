/*
 * dne_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the hot path of uber-research/deep-neuroevolution's
 * es_distributed/ CPU redis-worker path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product
 * (deep-neuroevolution_amd/) never links, imports or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the upstream repository root).
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   pinned by vectors generated from the real reference code run in the build
 *   container (tests/golden/make_golden.py imports es_distributed/es.py,
 *   optimizers.py, atari_wrappers.py, nses.py):  noise table + sample_index,
 *   compute_centered_ranks, batched_weighted_sum, Adam/SGD, wrap_deepmind
 *   (noop/fire/max-and-skip/warp/stack over the SynthAtari fixture, real PIL),
 *   novelty distance.
 *   PARITY UNPINNED: the TensorFlow-0.12.1 forward numerics (conv2d SAME,
 *   matmul, contrib.layers.batch_norm, argmax) and the ALE emulator are absent
 *   third-party dependencies; they are restated from their documented
 *   semantics (cross-checked against torch-CPU within 1e-4, not bit-pinned).
 */
#ifndef DNE_ORACLE_H
#define DNE_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_KIND_ES 0 /* ESAtariPolicy  policies.py:305-429 */
#define ORC_KIND_GA 1 /* GAAtariPolicy  policies.py:433-513 */
#define ORC_KIND_GA_LARGE 2 /* LargeModel of the GPU tree: gpu_implementation/neuroevolution/models/dqn.py:39-47 over models/base.py:50-95 */

#define ORC_OB_BYTES (84 * 84 * 4)
#define ORC_RAM 128
#define ORC_BN_FLOATS 608 /* scale[16] shift[16] scale[32] shift[32] scale[256] shift[256] */
#define ORC_ENV_MAX_EPISODE_STEPS 400000 /* gym 0.9.4 NoFrameskip-v4 TimeLimit, in raw frames */

/* flat-theta layout: creation order of trainable variables
 * (policies.py:21-24, tf_util.py:224-246) */
typedef struct {
    int kind, nact, P;
    int c1w, c1b, bn1b, bn1g;
    int c2w, c2b, bn2b, bn2g;
    int fcw, fcb, bn3b, bn3g;
    int ow, ob;
    int c3w, c3b; /* LargeModel only (else 0) */
} orc_layout;

void orc_layout_make(int kind, int nact, orc_layout *L);
int orc_num_params(int kind, int nact);

/* A1  es.py:412-419   out = theta + sign * fl32(sigma * noise[idx:idx+P]) */
void orc_perturb(const float *theta, const float *noise, int64_t idx, float sigma, int sign, int P, float *out);

/* A3  policies.py:319-330 (is_ref=True): batch moments of the reference batch -> per-channel scale/shift */
void orc_es_ref_pass(const orc_layout *L, const float *theta, const uint8_t *ref, int nref, float *bn);
void orc_es_ref_pass_moments(const orc_layout *L, const float *theta, const uint8_t *ref, int nref, float *bn, float *mom /*608 or NULL*/);
/* A3/A4  single-observation act: returns argmax action; logits (nact floats) optional */
int orc_act(const orc_layout *L, const float *theta, const float *bn, const uint8_t *ob, float *logits);
/* LargeModel: conv 32 8x8/4, conv 64 4x4/2, conv 64 3x3/1 (all SAME, + bias, relu), fc 512, out.  Same numerics contract; the fc
 * is 4 k-slices of 1936 rows.  Intermediates are raw (pre-relu, bias added): y1[21*21*32], y2[11*11*64], y3[11*11*64], y4[512]. */
void orc_forward_large_debug(const orc_layout *L, const float *theta, const uint8_t *ob, float *y1, float *y2, float *y3, float *y4,
                             float *logits);
/* intermediate activations for kernel-level parity tests (raw = pre-BN/pre-ReLU) */
void orc_forward_debug(const orc_layout *L, const float *theta, const float *bn, const uint8_t *ob,
                       float *y1_raw /*7056*/, float *y2_raw /*3872*/, float *y3_raw /*256*/, float *logits);

/* SynthAtari raw environment (fixture standing in for ALE) */
void orc_raw_reset(uint8_t *ram, uint32_t seed);
int orc_raw_frame(uint8_t *ram, int action); /* returns integer reward */
void orc_raw_render(const uint8_t *ram, uint8_t *screen /*210*160 palette idx*/);
void orc_palette(uint8_t *rgb /*16*3*/);

/* A5  atari_wrappers.py:204-222 wrap_deepmind over SynthAtari */
typedef struct {
    uint8_t ram_prev[ORC_RAM];
    uint8_t ram_cur[ORC_RAM];
    uint8_t stack[ORC_OB_BYTES]; /* [84][84][4], oldest channel first */
    int done;
} orc_wenv;
void orc_wenv_reset(orc_wenv *e, uint32_t seed);
void orc_wenv_step(orc_wenv *e, int action, float *reward, int *done);
/* WarpFrame only (atari_wrappers.py:138-142) on an RGB frame, for checking against PIL */
void orc_warp_rgb(const uint8_t *rgb /*210*160*3*/, uint8_t *out /*84*84*/);
void orc_resize_tables(double *kh /*84*5*/, int *bh /*84*2*/, double *kv /*84*7*/, int *bv /*84*2*/);

/* A2  policies.py:378-429 (ES) / 473-513 (GA): one episode.
 * bc: ES -> RAM per step [len][128] (may be NULL); GA -> final RAM [128] (may be NULL)
 * actions_out: optional trace [tslimit] */
void orc_rollout(const orc_layout *L, const float *theta, const uint8_t *ref, int nref, uint32_t env_seed,
                 int tslimit, float *ret, float *signret, int32_t *len, uint8_t *bc, int32_t *actions_out);

/* A2+A7  es.py:411-426: n antithetic pairs evaluated one at a time */
void orc_es_eval(const orc_layout *L, const float *theta, const float *noise, const int64_t *idx, int n,
                 float sigma, int tslimit, const uint8_t *ref, int nref, const uint32_t *env_seed /*2n*/,
                 float *returns_n2, float *signreturns_n2, int32_t *lengths_n2);

/* A8  es.py:70-85, ties broken by flat index (stable) */
void orc_centered_ranks(const float *x, int n, float *y);
/* A9  es.py:115-122,291-296: g = (sum_i w_i * noise[idx_i:idx_i+P]) / (2N); i-ordered fmaf chain */
void orc_weighted_sum(const float *noise, const int64_t *idx, const float *w, int N, int P, float denom, float *g);
/* A10 es.py:298 + optimizers.py:10-17,45-50 / 29-32.  Returns update ratio. */
double orc_adam_update(float *theta, float *m, float *v, const float *g, int P, float l2coeff, int t,
                       double stepsize, double beta1, double beta2, double epsilon);
double orc_sgd_update(float *theta, float *v, const float *g, int P, float l2coeff, double stepsize,
                      double momentum);

/* A11 ga.py:251-264; policies.py:42-44; tf_util.py:122-130 */
void orc_ga_normc(const orc_layout *L, float *theta);
void orc_ga_rebuild(const orc_layout *L, const float *noise, const int64_t *seeds, int nseeds, float sigma,
                    float *theta);
/* A12 ga.py:136-149 with the deterministic order (-return, arrival index) */
void orc_ga_select(const float *returns, int M, int T, int32_t *out_idx);

/* A13 nses.py:12-32 */
double orc_bc_distance(const uint8_t *x, int n, const uint8_t *y, int m, int dim);
double orc_novelty(const uint8_t *const *archive, const int *archive_len, int narchive, const uint8_t *bc,
                   int bc_len, int dim, int k);

#ifdef __cplusplus
}
#endif
#endif

/*
 * dne_hip.h -- C ABI of libdne_hip.so, the MI355X (gfx950) ES/GA rollout-and-aggregate engine.
 *
 * Drop-in boundary for the hot path of uber-research/deep-neuroevolution's es_distributed/ CPU path.
 * The reference has no FFI on that path (its seams are Python call signatures); each entry
 * point below names the reference code it replaces (paths relative to the upstream repository root).
 * The reference-side binding a maintainer adds is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions: every call returns 0 on success, <0 on error (dne_last_error gives the text); the caller
 * owns all host buffers, the engine owns all device memory; plain pointers and sizes only (no torch
 * types); a handle is bound to one HIP device, one host thread per handle; no callbacks.
 * "member" = one episode slot (one perturbed policy + one environment).  For ES, members 2i and 2i+1
 * are the antithetic pair of noise index i (theta + sigma*eps, theta - sigma*eps).
 */
#ifndef DNE_HIP_H
#define DNE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNE_KIND_ES 0 /* ESAtariPolicy  es_distributed/policies.py:305-429 */
#define DNE_KIND_GA 1 /* GAAtariPolicy  es_distributed/policies.py:433-513 */
#define DNE_KIND_GA_LARGE 2 /* LargeModel of the GPU tree (conv 32/64/64, fc 512): gpu_implementation/neuroevolution/models/dqn.py:39-47.
                               GA entry points only, genomes with per-seed powers (dne_ga_set_init_scale + dne_ga_eval_powers) */
#define DNE_OB_BYTES (84 * 84 * 4)
#define DNE_RAM_BYTES 128
#define DNE_BN_FLOATS 608

#define DNE_PROC_CENTERED_RANK 0      /* es.py:281-282 */
#define DNE_PROC_SIGN 1               /* es.py:283-284 */
#define DNE_PROC_CENTERED_SIGN_RANK 2 /* es.py:285-286 */

#define DNE_OPT_ADAM 0 /* optimizers.py:35-50 */
#define DNE_OPT_SGD 1  /* optimizers.py:23-32 */

typedef struct dne_handle dne_handle;

typedef struct {
    int32_t device_id;
    int32_t policy_kind;  /* DNE_KIND_* */
    int32_t n_actions;    /* env.action_space.n (18 for Frostbite) */
    int32_t max_members;  /* episode slots evaluated concurrently (ES: 2 * pairs per call) */
    int32_t ref_count;    /* size of the virtual-batch-norm reference batch (es.py:160-162: 128); multiple of 8 */
    int32_t ref_chunk;    /* members per reference-pass chunk (bounds scratch memory); 0 = default */
    int32_t record_bc;    /* 1: keep behaviour characterisations (ES: RAM per step, GA: final RAM) */
    int32_t bc_max_steps; /* ES BC capacity in steps per member (<= timestep limit) */
    int32_t profile_events; /* 1: bracket hot kernels with HIP events (dne_get_profile) */
    int32_t bc_final_only;  /* ES with record_bc: keep only each member's final RAM ([members][128], what es_modified.py's
                               dumps use: bc_vec[-1], es_modified.py:176,197) instead of the whole trajectory */
    int32_t reserved[6];
} dne_config;

typedef struct { /* filled by dne_get_profile; times from HIP events on the engine's stream */
    double eval_ms;       /* wall of the last dne_*_eval call (events) */
    double fc_ms;         /* sum over launches of the streaming fc+act kernel in the last eval */
    int64_t fc_launches;
    int64_t fc_group_steps; /* sum over launches of groups processed (ES: pairs, GA: members) */
    int64_t env_steps;    /* sum of episode lengths of the last eval */
    double conv_ms, env_ms, ref_ms; /* other stages of the last eval (0 if not profiled) */
    double reduce_ms;     /* last dne_weighted_sum / dne_es_update aggregate kernel */
    double materialize_ms;/* last dne_materialize kernel */
    /* the streaming kernel proper (k_fc), without the small-count tail path (k_fc_cols + k_out) */
    double fc_full_ms;
    double fc_full_launches;
    double fc_full_units; /* env-steps (member-steps actually taken) processed by those launches */
    double fc_full_kind;  /* which kernel those launches were: 5 = k_fc_ring (the workgroup's noise rows through an LDS ring), 4 = k_fc_sub (one wave per sub-slice chain), 3 = k_fc_duo (table-ordered units), 2 = k_fc2 (two pairs per work item), 1 = k_fc */
    double fc_full_union_ms; /* time during which at least one of those launches was running (windows run them concurrently) */
    double reserved[1];
} dne_profile;

/* ---- lifecycle ------------------------------------------------------------------------------------ */
int dne_create(const dne_config *cfg, dne_handle **out);
void dne_destroy(dne_handle *h);
const char *dne_last_error(dne_handle *h); /* h may be NULL: error of the last failed dne_create */
int dne_num_params(int policy_kind, int n_actions); /* Policy.num_params, policies.py:22 */
int dne_get_profile(dne_handle *h, dne_profile *out);
/* Debugging aid with no reference counterpart: every device buffer of the handle lies between two poisoned 4 KiB zones;
 * returns the number of damaged zones (0 = no kernel wrote outside its buffer), <0 on a HIP error.  Also run by
 * dne_destroy.  Environment knobs: DNE_REDZONE=0 (off), DNE_TRACE=1 (stage breadcrumbs on stderr), DNE_DEBUG_SYNC=1
 * (synchronise and check after every launch set of an evaluation, naming the lock-step that failed). */
int dne_check_redzones(dne_handle *h);

/* ---- SharedNoiseTable (es.py:51-67) as one device buffer --------------------------------------------- */
int dne_noise_upload(dne_handle *h, const float *host, size_t count);          /* es.py:57-60 */
/* the same in pieces, so the table can be uploaded while it is being sampled (es.py:59 draws 250M numbers): */
int dne_noise_alloc(dne_handle *h, size_t count);
int dne_noise_write(dne_handle *h, size_t offset, const float *host, size_t count);
int dne_noise_get(dne_handle *h, int64_t idx, int dim, float *out_host);        /* es.py:63-64 get(i, dim) */

/* ---- flat parameters (tf_util.py:224-246 SetFromFlat/GetFlat; policies.py:99-103) -------------------- */
int dne_set_theta(dne_handle *h, int slot, const float *theta, size_t n); /* slot 0 = the ES parent theta */
int dne_get_theta(dne_handle *h, int slot, float *out, size_t n);
int dne_set_ref_batch(dne_handle *h, const uint8_t *ref /*[ref_count][84][84][4]*/, int count); /* policies.py:332-335 */

/* A1 es.py:412-419: out[i][0] = theta + sigma*noise[idx_i:], out[i][1] = theta - sigma*noise[idx_i:]
 * written to an engine-owned device buffer; out_host (n*2*P floats) may be NULL */
int dne_materialize(dne_handle *h, const int64_t *noise_idx, int n, float sigma, float *out_host);

/* ---- batched environment (wrap_deepmind over the device-resident stepper; atari_wrappers.py:204-222) -
 * shape follows the reference GPU tree's native op ABI (gym_tensorflow/tf_env.cpp:115-318) */
int dne_env_reset(dne_handle *h, int n, const uint32_t *seeds);
int dne_env_step(dne_handle *h, int n, const int32_t *actions, float *reward, int32_t *done);
int dne_env_observation(dne_handle *h, int n, uint8_t *out /*[n][84][84][4]*/);
int dne_env_ram(dne_handle *h, int n, uint8_t *out /*[n][128]*/);
int dne_env_set_observation(dne_handle *h, int n, const uint8_t *obs /*[n][84][84][4]*/);
/* emulator state injection (ALE's cloneState/restoreState is the nearest reference notion; used by the renderer parity
 * tests): RAM before / after the last raw frame of n members; the observation becomes 4 copies of the warped max frame,
 * as after a FrameStack reset (atari_wrappers.py:171-176) */
int dne_env_set_ram(dne_handle *h, int n, const uint8_t *ram_prev /*[n][128]*/, const uint8_t *ram_cur /*[n][128]*/);

/* ---- policy forward on explicit members (policies.py:319-330,374-375 / 449-459,469-470) ---------------
 * member i uses theta_i = base[base_slot[i]] + scale[i] * noise[noise_off[i]:]   (ES: scale = +-sigma) */
int dne_set_members(dne_handle *h, int n, const int32_t *base_slot, const int64_t *noise_off, const float *scale);
int dne_ref_pass(dne_handle *h, int n);                                   /* policies.py:399 (ES only) */
int dne_get_bn(dne_handle *h, int n, float *out /*[n][608] scale,shift per layer*/);
/* the batch moments behind them: what batch_norm(decay=0) leaves in moving_mean / moving_variance (policies.py:322-328),
 * i.e. the moving_mean / moving_variance datasets of a snapshot (policies.py:49-57); [n][608] mean,variance per layer */
int dne_get_bn_moments(dne_handle *h, int n, float *out);
int dne_act(dne_handle *h, int n, int32_t *actions, float *logits /*[n][n_actions] or NULL*/);
int dne_debug_activations(dne_handle *h, int member, float *y1 /*7056*/, float *y2 /*3872*/, float *y3 /*256*/);
/* LargeModel engines (DNE_KIND_GA_LARGE): raw outputs of conv1 [441*32], conv2 / conv3 [121*64] and the fc [512] of one member after
   dne_act -- kernel-level parity against the oracle's orc_forward_large_debug (models/dqn.py:39-47). */
int dne_debug_activations_large(dne_handle *h, int member, float *y1, float *y2, float *y3, float *y4);

/* ---- A1-A7: whole-batch evaluation ------------------------------------------------------------------ */
/* es.py:411-426 for n pairs at once: returns_n2/signreturns_n2/lengths_n2 are [n][2] like Result (es.py:18-23).
 * env_seed[2n]: per-episode environment seed (noop count = 1 + seed % 30).  bc (may be NULL, needs record_bc):
 * [2n][bc_max_steps][128] RAM trajectories (policies.py:410,418) */
int dne_es_eval(dne_handle *h, const int64_t *noise_idx, int n, float sigma, int tslimit,
                const uint32_t *env_seed, float *returns_n2, float *signreturns_n2, int32_t *lengths_n2,
                uint8_t *bc);
/* generic: members set by dne_set_members, one episode each (eval episodes es.py:388-405 use scale 0) */
int dne_eval_members(dne_handle *h, int n, int tslimit, const uint32_t *env_seed, float *returns,
                     float *signreturns, int32_t *lengths, uint8_t *bc);
/* ga.py:251-271 for n children: chains in CSR form; theta = normc(noise[s0]) + sigma*sum noise[s_k].
 * bc (may be NULL): final RAM [n][128] (policies.py:510) */
int dne_ga_eval(dne_handle *h, const int32_t *chain_offsets /*n+1*/, const int64_t *seeds, int n, float sigma,
                int tslimit, const uint32_t *env_seed, float *returns, float *signreturns, int32_t *lengths,
                uint8_t *bc);
/* ga.py:151-158 / 256-264: rebuild one genome into base slot `slot` (and optionally copy it out) */
int dne_ga_rebuild(dne_handle *h, int slot, const int64_t *seeds, int nseeds, float sigma, float *out_host);

/* The gpu tree's genome form (gpu_implementation/neuroevolution/models/base.py:118-149, ga.py:161-166): seeds =
 * ((idx0,), (idx1, power1), ...), theta = noise[idx0] * scale_by + sum_k power_k * noise[idx_k]; scale_by = the
 * per-parameter initial scale (base.py:190-201).  powers[] runs parallel to seeds[]; the root's entry is ignored. */
int dne_ga_set_init_scale(dne_handle *h, const float *scale_by, size_t n);
int dne_ga_rebuild_powers(dne_handle *h, int slot, const int64_t *seeds, const float *powers, int nseeds, float *out_host);
int dne_ga_eval_powers(dne_handle *h, const int32_t *chain_offsets /*n+1*/, const int64_t *seeds, const float *powers, int n,
                       int tslimit, const uint32_t *env_seed, float *returns, float *signreturns, int32_t *lengths,
                       uint8_t *bc);

/* ---- A8-A10: on-device reduce ------------------------------------------------------------------------ */
int dne_centered_ranks(dne_handle *h, const float *x, int n, float *out);           /* es.py:70-85, stable ties */
/* es.py:291-296: g = (sum_i w[i] * noise[idx[i]:idx[i]+P]) / denom, kept on device; g_host may be NULL */
int dne_weighted_sum(dne_handle *h, const int64_t *idx, const float *w, int n, float denom, float *g_host);
/* es.py:298 + optimizers.py: theta(slot 0) <- theta + step(-g + l2coeff*theta) using the device g */
int dne_optimizer_step(dne_handle *h, int opt_kind, float l2coeff, double stepsize, double beta1_or_momentum,
                       double beta2, double epsilon, double *update_ratio);
int dne_optimizer_reset(dne_handle *h); /* zero m, v, t (a fresh Adam/SGD, optimizers.py:24-27,36-43) */
/* nses.py:95-117,283-284 keeps one optimizer per meta-population member: swap its state in and out
 * (Adam: m, v, t ; SGD: v in `v`, m ignored).  Pointers may be NULL to skip a field. */
int dne_optimizer_get_state(dne_handle *h, float *m, float *v, int32_t *t);
int dne_optimizer_set_state(dne_handle *h, const float *m, const float *v, int32_t t);
/* es.py:281-298 in one call: process returns (proc_mode), aggregate, optimizer step */
int dne_es_update(dne_handle *h, const int64_t *idx, const float *returns_n2, const float *signreturns_n2,
                  int n, int proc_mode, int opt_kind, float l2coeff, double stepsize, double beta1_or_momentum,
                  double beta2, double epsilon, double *update_ratio);

/* ---- (e) the exchange step between GPUs: the master's Result collection es.py:226-277 for co-located GPU workers ----
 * A pair's wire record is 32 bytes, little-endian: int64 noise_idx, float ret[2], int32 len[2], float signret[2]
 * (the per-pair fields of Result, es.py:18-23).  Pair i of the N-pair population is evaluated by rank i % nranks.
 * dne_comm_unique_id (rank 0) + dne_comm_init (every rank, same 128 bytes) build an RCCL communicator over xGMI;
 * librccl.so.1 is opened on demand, single-GPU use never touches it. */
int dne_comm_unique_id(void *out128);
int dne_comm_init(dne_handle *h, int rank, int nranks, const void *unique_id128);
/* ncclCommInitRank blocks until every rank arrives: a caller that bounds it runs dne_comm_init on a second host thread and, on
 * time-out, calls dne_comm_abort from the first.  The late dne_comm_init then returns DNE_COMM_DROPPED (-2) WITHOUT touching the
 * handle (no error text: dne_last_error is not meaningful for that code); the handle must not be destroyed while it is in flight. */
#define DNE_COMM_DROPPED (-2)
/* what the communicator itself reports (ncclCommUserRank / ncclCommCount); without a communicator: rank 0 of 1, *is_rccl = 0.
 * The launcher side of gpu_implementation/neuroevolution/concurrent_worker.py:129-142 (one worker per visible device) asks
 * dne_device_count for the number of HIP devices this process can see. */
int dne_comm_info(dne_handle *h, int *rank, int *nranks, int *is_rccl);
int dne_device_count(int *count);
/* a second engine of the same process and device (another workload) takes part in the owner's communicator */
int dne_comm_share(dne_handle *h, dne_handle *owner);
/* leave RCCL for good (the ranks agreed on another carrier, or an initialisation hangs on another thread) */
int dne_comm_abort(dne_handle *h);
/* sum (op 0) / max (op 1) of n <= 64 doubles over all ranks, then a device synchronise; n = 0: barrier */
int dne_comm_allreduce(dne_handle *h, double *inout, int n, int op);
/* generic all-gather of `bytes` host bytes per rank (the GA's 32-byte child records: parent index, fresh seed, return,
 * length -- the per-child content of a GA Result, ga.py:266-271); recv holds nranks * bytes in rank order */
int dne_comm_allgather(dne_handle *h, const void *send, size_t bytes, void *recv);
/* all-gather of the records of the n_local pairs this rank evaluated in its last dne_es_eval, taken from the device
 * accumulators; the gathered set stays on the device in global pair order.  records_out: [n_global] records or NULL */
int dne_allgather_results(dne_handle *h, int n_local, int n_global, void *records_out);
/* test hook: un-shard a [nranks][ceil(n_global / nranks)] all-gather result (host copy) into global pair order on the device */
int dne_debug_unshard(dne_handle *h, const void *gathered, int n_global, int nranks, void *ordered_out);
/* other transports (the redis Result path, gloo in the CPU tests): this rank's shard out / the gathered set in */
int dne_records_pack(dne_handle *h, int n_local, void *records_out /*[n_local]*/);
int dne_records_set(dne_handle *h, const void *records /*[n_global], global pair order*/, int n_global);
/* es.py:281-298 on the gathered device-resident records (identical on every rank -> bit-identical theta) */
int dne_es_update_gathered(dne_handle *h, int proc_mode, int opt_kind, float l2coeff, double stepsize,
                           double beta1_or_momentum, double beta2, double epsilon, double *update_ratio);

/* ga.py:145: indices of the top-T returns, ordered by (-return, arrival index) */
int dne_ga_select(dne_handle *h, const float *returns, int m, int t, int32_t *out_idx);

/* ---- A13 nses.py:12-32: novelty of one behaviour characterisation against an archive -------------------
 * The archive (MasterClient.add_to_novelty_archive / get_archive, dist.py:93-98) is append-only, so it can live on the
 * device: dne_archive_append uploads one entry, and the two novelty calls score against the resident archive when their
 * `archive` argument is NULL.  A non-NULL archive replaces the resident one (one-shot form). */
int dne_archive_append(dne_handle *h, const uint8_t *bc /*[bc_len][dim]*/, int bc_len, int dim);
int dne_archive_clear(dne_handle *h);
int dne_archive_size(dne_handle *h);
int dne_novelty(dne_handle *h, const uint8_t *archive /*concatenated rows*/, const int32_t *archive_len,
                int narchive, const uint8_t *bc, int bc_len, int dim, int k, double *out);
/* nses.py:381-382 for a whole evaluated batch: novelty of each of the n members' RAM trajectories recorded by
 * the last dne_es_eval / dne_eval_members (record_bc engines; member i has lengths[i] rows of 128 bytes, which
 * never leave the device) against the archive */
int dne_novelty_batch(dne_handle *h, const uint8_t *archive, const int32_t *archive_len, int narchive, int n,
                      const int32_t *lengths, int k, double *out);

#ifdef __cplusplus
}
#endif
#endif

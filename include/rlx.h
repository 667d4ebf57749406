/*
 * rlx.h -- C ABI of the MI355X-native rollout + PPO/GRPO update hot path (librlx_hip.so).
 *
 * The reference (RLinf, /root/reference) has NO native/FFI layer: its extension surface is the
 * Python registries in rlinf/algorithms/registry.py:30-124 and rlinf/models/__init__.py:31-53.
 * Each entry point below therefore cites the reference Python function whose arithmetic it
 * replaces; the Python host in rlinf_amd/ binds them with ctypes and re-exposes them behind the
 * reference's own registry names (INTEGRATION.md shows the stub a maintainer would add).
 *
 * Conventions (all entry points):
 *   - plain C: device pointers + sizes, no torch types.  Pointers are caller-owned DEVICE memory
 *     (tensor.data_ptr()), contiguous in the layout stated per argument.  `bool` tensors are passed
 *     as uint8_t (torch stores one byte per bool).
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream),
 *     re-entrant, no hidden device allocation: scratch is a caller-provided workspace whose size
 *     comes from the matching *_workspace_bytes() query.
 *   - return 0 on success or a negative errno-style code; rlx_last_error() gives a thread-local
 *     message.  No exception ever crosses the boundary.
 *   - "time-chunk" layout: the reference's embodied buffers are [n_chunk(+1), B, C(, ...)]
 *     (rlinf/data/schema/embodied_trajectory_builder.py:46-67); time step t = k*C + c lives at
 *     ((k*B + b)*C + c).  The kernels index that layout directly, so the reference's
 *     transpose/reshape pre- and post-processing (rlinf/algorithms/utils.py:67-131,155-174)
 *     needs no copies.
 */
#ifndef RLX_H
#define RLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLX_VERSION 112 /* 0.1.2: bumped whenever an argument struct or a signature changes (round 6: rlx_xgmi_self_timing, the
                           * exchange inside rlx_xgmi_clip_adamw_step under sync_words; round 5: rlx_categorical_sample's
                           * softmax_lanes, rlx_adamw_params.sync_words; round 4's struct growth had left it at 100) */

typedef void* rlx_stream_t; /* hipStream_t */

enum rlx_status {
    RLX_OK = 0,
    RLX_EINVAL = -22,  /* bad argument (NULL pointer, non-positive size, unsupported combination) */
    RLX_ENOSPC = -28,  /* workspace too small */
    RLX_EHIP = -5,     /* a HIP runtime call failed; see rlx_last_error() */
    RLX_ENOSYS = -38   /* variant not compiled in */
};

int rlx_version(void);
/* 1: a development build (-DRLX_DEV_VARIANTS): the refuted / experimental kernel variants are compiled in and the RLX_* variant
 * environment switches are read; 0: the product build -- the measured-best path only, no variant switch is read, entry points asked
 * for a variant that is not compiled return RLX_ENOSYS. */
int rlx_dev_variants(void);
const char* rlx_last_error(void);

/* sizeof() of the argument structs below as THIS library was compiled, in declaration order (gae_params, ppo_loss_params,
 * gather_field, adamw_group, adamw_params, mlp_layout, value_job, rollout_step, ppo_step_args, decoupled_loss_params,
 * token_rows, token_loss_params, copy_segment): a binding written in another language checks its mirrors against these at
 * load time instead of discovering a stale layout through a wrong result.  Writes min(n, 13) entries, returns 13. */
int rlx_abi_struct_sizes(size_t* sizes, int n);

/* Number of compute units / wave size of the current device (plumbing for launch heuristics). */
int rlx_device_info(int* num_cu, int* wave_size);

/* ------------------------------------------------------------------------------------------
 * a9  done_prefix_mask   <- compute_loss_mask, rlinf/utils/metric_utils.py:516-537
 *   dones      [n_chunk+1, B, C] u8 (bool)
 *   loss_mask  [n_chunk,   B, C] u8 (bool)   mask[t,b] = no done at flat rows (C-1) .. (C-1)+t
 *   mask_sum   [B] i64                       per-env count of valid steps (the reference returns
 *                                            this broadcast as a stride-0 view of [1,B,1])
 * Integer work: bit-exact.
 * ------------------------------------------------------------------------------------------ */
int rlx_done_prefix_mask(const uint8_t* dones, uint8_t* loss_mask, int64_t* mask_sum,
                         int n_chunk, int batch, int chunk, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a10-a12  gae_scan   <- compute_gae_advantages_and_returns, rlinf/algorithms/advantages.py:24-86
 *                        (+ safe_normalize, rlinf/algorithms/utils.py:397-404, and the
 *                         [n,B,C]<->[T,B] pre/post-processing, utils.py:67-131,155-174)
 *   rewards     [n_chunk,   B, C] f32
 *   values      [n_chunk+1, B, C] f32, or NULL => critic-free (gamma = lambda = 1, delta = r)
 *   dones       [n_chunk+1, B, C] u8
 *   loss_mask   [n_chunk,   B, C] u8 or NULL (only selects the elements the moments are taken over)
 *   advantages  [n_chunk,   B, C] f32 out
 *   returns     [n_chunk,   B, C] f32 out
 * gamma_lambda is passed separately because the reference forms gamma*gae_lambda in double before
 * rounding it to f32 (advantages.py:76).
 * variant: 0 = auto; otherwise (1 | nseg << 8) with nseg in {1,2,4,8,16} time segments per 64-env
 * group.  nseg == 1 is a pure streaming scan whose un-normalised outputs are bit-identical to the
 * reference's CPU loop; nseg > 1 is the segmented scan (per-wave affine maps combined through LDS),
 * equal to it within f32 rounding.  (Wider per-lane vectors were measured and compiled out.)
 * ------------------------------------------------------------------------------------------ */
typedef struct rlx_gae_params {
    float gamma;
    float gamma_lambda;        /* (float)((double)gamma * (double)gae_lambda) */
    int32_t normalize_advantages;
    int32_t normalize_returns;
    float norm_eps;            /* 1e-5 in the reference */
    int32_t variant;
} rlx_gae_params;

size_t rlx_gae_workspace_bytes(int n_chunk, int batch, int chunk);
int rlx_gae_scan(const float* rewards, const float* values, const uint8_t* dones,
                 const uint8_t* loss_mask, float* advantages, float* returns,
                 void* workspace, size_t workspace_bytes,
                 int n_chunk, int batch, int chunk, const rlx_gae_params* params, rlx_stream_t stream);

/* a12 stand-alone: x <- (x - mean(x[mask])) / (std_unbiased(x[mask]) + eps); no-op if nothing selected. */
size_t rlx_standardize_workspace_bytes(size_t n);
int rlx_masked_standardize(float* x, const uint8_t* mask, size_t n, float eps,
                           void* workspace, size_t workspace_bytes, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a12b  global advantage statistics (pipeline mode, SURVEY.md 8f item 2: "global advantage stats")
 *   masked_stats         <- rlinf/utils/distributed.py:942-954: stats[0..2] (f64, device) = count, sum, sum of squares of x[mask]
 *                           (mask NULL = all); accumulate != 0 adds to what stats already holds (several batches of one rank)
 *   normalize_from_stats <- rlinf/utils/distributed.py:957-965: out = (x - mean) * rsqrt(max(var, 0) + 1e-5) in f64 -> f32,
 *                           count clamped to >= 1, variance uncorrected -- the stats are summed across ranks in between
 *                           (EnvWorker.send_rollout_trajectories_pipeline, rlinf/workers/env/env_worker.py:1548-1572)
 * ------------------------------------------------------------------------------------------ */
size_t rlx_masked_stats_workspace_bytes(int64_t n);
int rlx_masked_stats(const float* x, const uint8_t* mask, int64_t n, double* stats, int accumulate, void* workspace,
                     size_t workspace_bytes, rlx_stream_t stream);
int rlx_normalize_from_stats(const float* x, const double* stats, float* out, int64_t n, rlx_stream_t stream);
/* a12c  masked_normalization (async PPO learner, rlinf/utils/distributed.py:866-937 with its defaults: f64, biased
 *       variance, eps outside the root): out = ((mask ? x : 0) - mean) / (sqrt(var) + eps) from stats = (count, sum, sumsq)
 *       of x[mask] -- rlx_masked_stats output, summed over ranks in between (the reference all-reduces the three sums). */
int rlx_masked_normalize(const float* x, const uint8_t* mask, const double* stats, double eps, float* out, int64_t n,
                         rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a13  grpo_group_adv  <- calculate_scores, rlinf/algorithms/utils.py:134-152
 *                         + compute_grpo_advantages, rlinf/algorithms/advantages.py:89-121
 *   rewards [n_chunk,B,C] f32, dones [n_chunk+1,B,C] u8, loss_mask [n_chunk,B,C] u8 (required)
 *   scores  [B] f32 out  (return of each env's FIRST episode segment)
 *   advantages [n_chunk,B,C] f32 out = ((s - mean_g)/(std_g_unbiased + eps))[b] * mask[t,b]
 *   groups are `group_size` consecutive envs; batch % group_size must be 0.
 * ------------------------------------------------------------------------------------------ */
int rlx_grpo_group_adv(const float* rewards, const uint8_t* dones, const uint8_t* loss_mask,
                       float* scores, float* advantages, int n_chunk, int batch, int chunk,
                       int group_size, float eps, rlx_stream_t stream);
/* the group normalisation + broadcast alone (compute_grpo_advantages, advantages.py:107-121) from given scores */
int rlx_grpo_from_scores(const float* scores, const uint8_t* loss_mask, float* advantages, int n_chunk,
                         int batch, int chunk, int group_size, float eps, rlx_stream_t stream);
/* the score scan alone (calculate_scores, rlinf/algorithms/utils.py:134-152): scores [B] f32 out */
int rlx_episode_scores(const float* rewards, const uint8_t* dones, float* scores, int n_chunk, int batch,
                       int chunk, rlx_stream_t stream);

/* reward filter  <- EmbodiedFSDPActor._process_received_rollout_batch, rlinf/workers/actor/embodied_fsdp_actor_worker.py:235-281
 *   (same code in preprocess_embodied_batch, rlinf/utils/utils.py:801-832)
 *   rewards [n_chunk, B, C] f32, loss_mask [n_chunk, B, C] u8 or NULL.  For every group of `group_size` consecutive envs:
 *   keep = lower <= mean_over_group( sum_t,c rewards * loss_mask ) <= upper
 *   out_mask = keep & loss_mask  [n_chunk, B, C]   (loss_mask given)   or   keep  [n_chunk, B, 1]  (loss_mask NULL) */
int rlx_reward_filter_mask(const float* rewards, const uint8_t* loss_mask, uint8_t* out_mask, int n_chunk, int batch,
                           int chunk, int group_size, float lower, float upper, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a18-a21  ppo_loss  <- preprocess_loss_inputs (rlinf/algorithms/utils.py:280-376),
 *                       compute_ppo_actor_loss (rlinf/algorithms/losses.py:170-312),
 *                       compute_ppo_critic_loss (losses.py:315-380), huber_loss (utils.py:20-23),
 *                       masked_mean / masked_mean_ratio (rlinf/utils/utils.py:323-356),
 *                       compute_critic_explained_variance_stats (rlinf/utils/metric_utils.py:232-258)
 * One thread owns one ADVANTAGE element e (n_adv of them = numel of advantages / loss_mask /
 * values / prev_values / returns) and the `raw_per_adv` consecutive per-dimension log-probs under
 * it, which form `sub_per_adv` loss elements of raw_per_adv/sub_per_adv summed log-probs each:
 *     logprob_type  action_level: raw = A, sub = 1      (sum over action_dim, one ratio per chunk)
 *                   token_level : raw = A, sub = A      (one ratio per dim, advantage broadcast)
 *                   chunk_level : raw = C*A, sub = 1    (one ratio per env-step)
 *   logprobs, old_logprobs  [n_adv * raw_per_adv] f32
 *   advantages              [n_adv] f32
 *   values, prev_values, returns [n_adv] f32, all NULL when has_critic == 0 (registry name "actor")
 *   loss_mask               [n_adv] u8 or NULL;  loss_mask_sum [n_adv] i64 or NULL
 * Forward writes
 *   g_logp  [n_adv * sub_per_adv] f32   d(loss)/d(summed log-prob) up to the global scale out[16]
 *   g_value [n_adv] f32                 d(loss)/d(value)           up to the global scale out[17]
 *   out     [RLX_PPO_OUT_FLOATS] f32    see enum rlx_ppo_out; everything stays on the device
 * Backward expands them: d_logprobs[e*raw + j] = grad_out * out[16] * g_logp[...],
 *                        d_values[e] = grad_out * out[17] * g_value[e]   (grad_out: device scalar).
 * ------------------------------------------------------------------------------------------ */
enum rlx_ppo_out {
    RLX_PPO_LOSS = 0,            /* actor (0 under critic_warmup) + critic */
    RLX_PPO_POLICY_LOSS = 1,     /* actor/policy_loss        */
    RLX_PPO_POLICY_LOSS_ABS = 2, /* actor/policy_loss_abs    */
    RLX_PPO_RATIO = 3,           /* actor/ratio              */
    RLX_PPO_RATIO_ABS = 4,       /* actor/ratio_abs          */
    RLX_PPO_CLIPPED_RATIO = 5,   /* actor/clipped_ratio      */
    RLX_PPO_DUAL_CLIPPED_RATIO = 6, /* actor/dual_cliped_ratio */
    RLX_PPO_APPROX_KL = 7,       /* actor/approx_kl          */
    RLX_PPO_CLIP_FRACTION = 8,   /* actor/clip_fraction      */
    RLX_PPO_VALUE_LOSS = 9,      /* critic/value_loss        */
    RLX_PPO_VALUE_CLIP_RATIO = 10, /* critic/value_clip_ratio */
    RLX_PPO_EV_COUNT = 11,       /* explained-variance sufficient statistics (metric_utils.py:252-258) */
    RLX_PPO_EV_RETURNS_SUM = 12,
    RLX_PPO_EV_RETURNS_SQ_SUM = 13,
    RLX_PPO_EV_ERRORS_SUM = 14,
    RLX_PPO_EV_ERRORS_SQ_SUM = 15,
    RLX_PPO_ACTOR_GRAD_SCALE = 16,
    RLX_PPO_CRITIC_GRAD_SCALE = 17,
    RLX_PPO_OUT_FLOATS = 20
};

typedef struct rlx_ppo_loss_params {
    float ratio_lo, ratio_hi;      /* (float)(1.0 - clip_ratio_low), (float)(1.0 + clip_ratio_high): formed in
                                      double on the host exactly like the reference's python scalars */
    float clip_ratio_c;            /* used when use_dual_clip */
    float clip_log_ratio_min, clip_log_ratio_max;
    float value_clip, huber_delta;
    int32_t use_dual_clip, use_clip_log_ratio_min, use_clip_log_ratio_max;
    int32_t has_critic, critic_warmup;
    int32_t max_episode_steps;     /* > 0 together with loss_mask and loss_mask_sum selects the
                                      masked_mean_ratio aggregation (losses.py:219-227) */
    int32_t raw_per_adv, sub_per_adv;
    int32_t metric_unbroadcast;    /* the ratio metrics (ratio, ratio_abs, clipped, dual-clipped) divide by the UN-broadcast mask
                                      count: the reference only expands the mask for 3-D ratios (losses.py:288-290), so a 2-D
                                      ratio [bsz, C] against a [bsz, 1] mask -- reward_type chunk_level with logprob_type
                                      action_level, C > 1 -- keeps the [bsz] count */
} rlx_ppo_loss_params;

size_t rlx_ppo_loss_workspace_bytes(int64_t n_adv);
int rlx_ppo_loss_fwd(const float* logprobs, const float* old_logprobs, const float* advantages,
                     const float* values, const float* prev_values, const float* returns,
                     const uint8_t* loss_mask, const int64_t* loss_mask_sum, int64_t n_adv,
                     const rlx_ppo_loss_params* params, float* g_logp, float* g_value, float* out,
                     void* workspace, size_t workspace_bytes, rlx_stream_t stream);
int rlx_ppo_loss_bwd(const float* g_logp, const float* g_value, const float* out, const float* grad_out,
                     float* d_logprobs, float* d_values, int64_t n_adv, int raw_per_adv, int sub_per_adv,
                     rlx_stream_t stream);

/* a22 entropy bonus of the Gaussian MLP policy  <- EmbodiedFSDPActor.train_micro_batch, rlinf/workers/actor/embodied_fsdp_actor_worker.py:679-690
 *   (+ reshape_entropy rlinf/utils/utils.py:384-408, masked_mean :323-330).  The entropy of Normal(mu, exp(logstd)) is
 *   state-independent, so after rlx_ppo_loss_fwd / rlx_ppo_step have written out_row:
 *     ent = elem_scale * sum_a (0.5 + 0.5 log 2pi + logstd_a)   (0 when a loss mask is present and empty: out_row[18] == 0)
 *     out_row[RLX_PPO_LOSS] -= entropy_bonus * ent;  out_row[19] = ent ("actor/entropy_loss");
 *     grad_logstd[a] -= entropy_bonus * grad_scale * elem_scale        (grad_scale = 1 / gradient_accumulation) */
int rlx_gaussian_entropy_bonus(const float* logstd, int n_act, float* grad_logstd, float* out_row, float entropy_bonus,
                               float grad_scale, int has_mask, float elem_scale, rlx_stream_t stream);
/* The same behind a decoupled rlx_ppo_step: grad_logstd is in SUM form there (a later multiplication by *actor_scale, a device
 * float -- that step's out[RLX_PPO_ACTOR_GRAD_SCALE] -- finishes it), so the bonus term is divided by *actor_scale here. */
int rlx_gaussian_entropy_bonus_deferred(const float* logstd, int n_act, float* grad_logstd, float* out_row, float entropy_bonus,
                                        float grad_scale, int has_mask, float elem_scale, const float* actor_scale,
                                        rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a16  shuffle_gather  <- process_nested_dict_for_train, rlinf/utils/nested_dict_process.py:272-285
 *   for every field f:  dst_f[i, :] = src_f[index[i], :]   (rows of row_bytes_f bytes; the caller has
 *   already dropped the last time row of dones/terminations/truncations/prev_values by passing a
 *   shorter view).  Up to RLX_GATHER_MAX_FIELDS fields per launch.  Bit-exact by construction.
 * ------------------------------------------------------------------------------------------ */
#define RLX_GATHER_MAX_FIELDS 16
typedef struct rlx_gather_field {
    const void* src;
    void* dst;
    int64_t row_bytes;
} rlx_gather_field;
int rlx_gather_rows(const rlx_gather_field* fields, int n_fields, const int64_t* index, int64_t n_rows,
                    rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a23  clip_adamw_step  <- FSDPModelManager.optimizer_step, rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:429-463
 *        = torch.nn.utils.clip_grad_norm_ (strategy/fsdp.py:363-369) + torch.optim.AdamW.step
 *          with the two parameter groups of build_optimizer (fsdp_model_manager.py:501-590)
 * Flat f32 buffers of n elements: params, grads, exp_avg, exp_avg_sq.  `groups` partitions [0, n)
 * into ranges with their own learning rate (names containing "value_head" use value_lr).
 *   total_norm = ||grads||_2;  grads *= min(1, max_norm / (total_norm + 1e-6))  (always applied)
 *   if total_norm is non-finite the AdamW update is skipped (step count is the caller's to keep)
 *   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2)*g*g;
 *   p -= (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * stats[0] = total_norm (pre-clip), stats[1] = 1 if the step was applied else 0.  Device-side.
 * grad_partials > 1: grads holds that many stacked partial buffers [grad_partials][n] which are summed
 * first (split-K weight-gradient slabs); the sum is what gets clipped and applied.
 * ------------------------------------------------------------------------------------------ */
#define RLX_ADAMW_MAX_GROUPS 8
struct rlx_mlp_layout; /* below */
typedef struct rlx_adamw_group {
    int64_t begin, end; /* element range */
    double lr;          /* double like torch's python scalars: the kernel forms lr / bias_correction1 and 1 - lr * wd in double */
} rlx_adamw_group;
typedef struct rlx_adamw_params {
    double beta1, beta2, eps, weight_decay; /* doubles: 1 - beta ** step is formed in double from the exact python values */
    float max_grad_norm;     /* <= 0: no clipping */
    int32_t step;            /* 1-based step count of THIS update */
    int32_t n_groups;
    int32_t grad_partials;   /* >= 1 */
    float grad_scale;        /* applied to the summed gradient before the norm (e.g. 1/world_size) */
    rlx_adamw_group groups[RLX_ADAMW_MAX_GROUPS];
    /* optional (both or neither): keep the fragment-tile weight image (rlx_mlp_pack_tiles) in step with the parameters --
     * every updated weight is also written to its tile slot(s), which saves the re-pack launch before the next forward */
    const struct rlx_mlp_layout* tile_layout;
    float* tiles;
    int32_t tiles_bf16;      /* the image is the bf16 one (rlx_mlp_pack_tiles_bf16) */
    /* optional: scales that only exist on the device when the slab sum runs (the decoupled rlx_ppo_step: 1 / count of the
     * behaviour mask, left in its metric row).  The grad_partials slabs are deferred_groups consecutive groups (one per
     * micro-batch); elements of the two ranges are multiplied by deferred_scale[g * deferred_stride] as group g is summed.
     * NULL: none.  Applied wherever THIS rank's slabs are collapsed: rlx_clip_adamw_step (one rank), the staging launch of
     * rlx_xgmi_clip_adamw_step, rlx_sum_slabs_deferred in front of an RCCL all-reduce -- every rank scales its own gradient by
     * its own loss denominator before the mean over ranks, like the reference's per-rank loss.backward(). */
    const float* deferred_scale;
    int32_t deferred_stride;
    int32_t deferred_groups;
    int64_t deferred_range[2][2]; /* [begin, end) x 2; an empty range is begin == end */
    /* optional (round 5): rlx_adamw_sync_words(n) 64-bit words of device memory, 16-byte aligned, ZEROED ONCE by the caller and
     * from then on owned by the library (one buffer per parameter set; calls that share it must be stream-ordered).  With it
     * rlx_clip_adamw_step runs slab sum + norm + clip + AdamW as ONE launch -- the blocks exchange their norm partials through
     * these words instead of through a launch boundary -- whenever the plan allows it (n % 4 == 0, 16-byte aligned buffers,
     * <= 512 blocks of 2048 parameters, all resident); same element arithmetic (the norm's f64 partials are formed over different
     * blocks, so its last bit may differ from the two-launch form's).  The launch needs ALL its workgroups resident at once; the
     * library takes the bound against the compute units the launch stream may use (hipExtStreamGetCUMask) and falls back to two
     * launches when they cannot hold the plan, but it cannot see what ANOTHER process or stream holds.  Should a poll expire for
     * that reason (2 s) the step is skipped as a whole and word 1 of the buffer is set and STAYS set: every later call on these
     * words reports a NaN norm and skips as a whole too (no block ever applies an update another block skipped), until the
     * caller -- who sees the non-finite norm, reads word 1 and finds it set -- passes NULL from then on (what rlinf_amd.ops.
     * check_adamw_sync makes the learners do) or zeroes the buffer to try again.  NULL: two launches. */
    uint64_t* sync_words;
} rlx_adamw_params;
size_t rlx_adamw_workspace_bytes(int64_t n);
size_t rlx_adamw_sync_words(int64_t n);
/* out[i] = sum_k grads[k][i] (k < slabs): collapse the split-K slabs before a data-parallel all-reduce. */
int rlx_sum_slabs(const float* grads, int64_t n, int slabs, float* out, rlx_stream_t stream);
/* ... with p->deferred_scale applied group by group (only the deferred_* fields of p are read). */
int rlx_sum_slabs_deferred(const float* grads, int64_t n, int slabs, float* out, const struct rlx_adamw_params* p,
                           rlx_stream_t stream);

/* step_state: NULL (use p->step) or device int32[2] = {steps applied, pending flag}; the kernels then keep the
 * step count on the device (t = state[0] + 1, advanced only when the update was applied), which makes a
 * captured hipGraph of the update loop replayable.  Initialise it to {0, 0}. */
int rlx_clip_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const rlx_adamw_params* p, float* stats, int32_t* step_state, void* workspace,
                        size_t workspace_bytes, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a6  bootstrap_rewards  <- EnvWorker.compute_bootstrap_rewards, rlinf/workers/env/env_worker.py:718-758
 *   rewards [B, C] f32 (in place): rewards[b, C-1] += gamma * bootstrap_values[b, 0] where flags[b, C-1]
 *   (flags = dones for bootstrap_type "always", truncations for "standard").
 * ------------------------------------------------------------------------------------------ */
int rlx_bootstrap_rewards(float* rewards, const uint8_t* flags, const float* bootstrap_values, int batch, int chunk,
                          int value_stride, float gamma, rlx_stream_t stream);

/* a7  store_env_rows  <- EmbodiedTrajectoryBuilder.append_step_result (env fields),
 *     rlinf/data/schema/embodied_trajectory_builder.py:72-93: one env step's [B, C] outputs into the buffer rows
 *     rewards[t], terminations / truncations / dones[t+1], with dones = terminations | truncations.  n = B * C. */
int rlx_store_env_rows(const float* rewards, const uint8_t* terminations, const uint8_t* truncations, float* reward_row,
                       uint8_t* done_row, uint8_t* termination_row, uint8_t* truncation_row, int64_t n, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a1-a5, a17  MLP policy  <- MLPPolicy, rlinf/models/embodiment/mlp_policy/mlp_policy.py:91-107 (modules),
 *     :238-293 (_sample_actions/_generate_actions), :295-320 (predict_action_batch), :202-236 (default_forward),
 *     ValueHead rlinf/models/embodiment/modules/value_head.py:17-66, Normal.log_prob/entropy (torch).
 * PPO configuration only: tanh MLPs obs->256->256->256, state-independent log-std, no tanh squashing.
 * All parameters live in ONE flat f32 buffer in the reference's named_parameters() order; `rlx_mlp_layout`
 * holds the element offsets.  Dense layers run on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32
 * products, f32 accumulate); heads, sampling, log-prob and entropy are fused epilogues.
 *   net 0 = value head (value_head.mlp.{0,2,4,6}), net 1 = actor (backbone.{0,2,4} + actor_mean)
 * ------------------------------------------------------------------------------------------ */
typedef struct rlx_mlp_layout {
    int32_t obs_dim;      /* D (42) */
    int32_t act_dim;      /* num_action_chunks * action_dim (8) */
    int32_t val_dim;      /* value head outputs (num_action_chunks, 1) */
    int32_t hidden;       /* must be 256 */
    int64_t n_params;     /* total elements of the flat buffer */
    int64_t off_logstd;   /* actor_logstd [1, act_dim] */
    int64_t off_w[2][4];  /* [net][layer] weight, row-major [out, in] */
    int64_t off_b[2][4];  /* [net][layer] bias, or -1 (value head's last layer has none) */
} rlx_mlp_layout;

/* Derived weight images (zero-padded first layers, transposed hidden layers) the kernels consume;
 * rebuild after every optimizer step.  packed: rlx_mlp_packed_bytes() bytes. */
size_t rlx_mlp_packed_bytes(const rlx_mlp_layout* layout);
int rlx_mlp_pack(const float* params, const rlx_mlp_layout* layout, float* packed, rlx_stream_t stream);

/* Rollout: obs-preprocess is the identity on `states` (mlp_policy.py:122-124); one fused launch computes
 *   mean, value, action = eps * exp(logstd) + mean (eps == NULL: eval mode, action = mean),
 *   logprob = Normal(mean, std).log_prob(action) per dimension.
 *   states [M, D], eps [M, act_dim] -> action [M, act_dim], logprob [M, act_dim], value [M, val_dim] */
int rlx_mlp_rollout(const float* params, const float* packed, const rlx_mlp_layout* layout, const float* states,
                    const float* eps, int64_t m, float* action, float* logprob, float* value, rlx_stream_t stream);

/* Value head only (get_bootstrap_values, rlinf/workers/rollout/hf/huggingface_worker.py:612-627): value [M, val_dim]. */
int rlx_mlp_value(const float* params, const float* packed, const rlx_mlp_layout* layout, const float* states, int64_t m,
                  float* value, rlx_stream_t stream);

/* Training forward on stored (states, action): logprob, entropy [M, act_dim], value [M, val_dim]; keeps what
 * backward needs: mean [M, act_dim] and the six hidden activations acts [2][3][M][256]. */
int rlx_mlp_train_fwd(const float* params, const float* packed, const rlx_mlp_layout* layout, const float* states,
                      const float* action, int64_t m, float* logprob, float* entropy, float* value, float* mean,
                      float* acts, rlx_stream_t stream);

/* Training backward: from d_logprob, d_value (and optionally d_entropy) to parameter gradients.
 *   grads: [slabs][n_params] f32 split-K slabs, every element written (summed by rlx_clip_adamw_step).
 *   workspace: rlx_mlp_bwd_workspace_bytes(layout, m) bytes. */
int rlx_mlp_bwd_slabs(int64_t m);
size_t rlx_mlp_bwd_workspace_bytes(const rlx_mlp_layout* layout, int64_t m);
int rlx_mlp_train_bwd(const float* params, const float* packed, const rlx_mlp_layout* layout, const float* states,
                      const float* action, const float* mean, const float* acts, const float* d_logprob,
                      const float* d_entropy, const float* d_value, int64_t m, float* grads, int slabs,
                      void* workspace, size_t workspace_bytes, rlx_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Fused hot launches (what the workers of rlinf_amd/ call every step; the entry points above stay for
 * integrations that need the stages separately).
 *
 * rlx_mlp_rollout_step: ONE launch per rollout step.
 *   policy job   <- MLPPolicy.predict_action_batch, mlp_policy.py:295-320 (obs-preprocess :122-124, forward
 *                   :238-254, sample / log-prob / value :256-293): states [m, D], eps [m, act_dim] (NULL = eval)
 *                   -> action, logprob [m, act_dim], value [m, val_dim], written in place (trajectory-buffer rows);
 *                   states_copy (optional) receives the rows too (forward_inputs.states).
 *   value jobs   <- MultiStepRolloutWorker.get_bootstrap_values, huggingface_worker.py:612-627, optionally fused
 *                   with EnvWorker.compute_bootstrap_rewards, env_worker.py:718-758:
 *                   values [m, val_dim] (optional) = V(states); rewards [m, chunk] (optional, in place):
 *                   rewards[b, chunk-1] += gamma * V(states)[b, 0] where flags[b, chunk-1].
 *   Needs obs_dim <= 64, act_dim, val_dim <= 16.
 *
 * rlx_mlp_pack_tiles: the weight image the fused launches stream at precision "32" -- an OPAQUE buffer of
 *   rlx_mlp_tiles_bytes_for(layout, 0) bytes (rlx_mlp_tiles_bytes(): large enough for any format).  By default: every weight as three
 *   bf16 planes hi | mid | lo (x = hi + mid + lo exactly), each plane the 16 (out) x 32 (in) k-step-major tile image of the bf16
 *   variant -- the f32 launches run six of the nine plane products on the bf16 matrix pipe (csrc/ppo_step_f32x.hip: f32-accurate at
 *   6 / 16 of the f32 MFMA time).  With RLX_F32_EXACT_MFMA=1 in the environment (read once per process): 16 x 16 f32 tiles in MFMA
 *   fragment order for the exact-f32-MFMA launches.  First layer zero-padded to 64 inputs; the hidden layers also transposed, for
 *   the backward-data GEMMs.  Rebuild after every change of params (rlx_ppo_step does so itself, into its workspace, when it is
 *   given no image), or let rlx_clip_adamw_step keep it in step (rlx_adamw_params.tiles).
 * ------------------------------------------------------------------------------------------ */
size_t rlx_mlp_tiles_bytes(const rlx_mlp_layout* layout);
size_t rlx_mlp_tiles_bytes_for(const rlx_mlp_layout* layout, int32_t bf16); /* exact size of the f32 (0) / bf16 (1) image */
int rlx_mlp_pack_tiles(const float* params, const rlx_mlp_layout* layout, float* tiles, rlx_stream_t stream);
/* bf16 variant ("PPO bf16": bf16 operands of the dense layers, f32 accumulate / master weights / everything else):
 * 16 (out) x 32 (in) bf16 tiles of 1 KiB; rlx_mlp_tiles_bytes_for(layout, 1) bytes. */
int rlx_mlp_pack_tiles_bf16(const float* params, const rlx_mlp_layout* layout, void* tiles, rlx_stream_t stream);

typedef struct rlx_value_job {
    const float* states;
    int64_t m;
    float* values;
    float* rewards;
    const uint8_t* flags;
    int32_t chunk;
    float gamma;
    /* optional fused store of the env step's outputs (replaces rlx_store_env_rows + the in-place fold above): when
     * env_rewards != NULL the job writes, for its rows,  termination_row / truncation_row / done_row (= term | trunc) and
     * rewards[b, c] = env_rewards[b, c] (+ gamma * V(states)[b, 0] on c = chunk-1 where the flag is set; the flag is
     * done for flag_is_truncation == 0 (bootstrap_type "always"), truncation otherwise).  `flags` is then ignored. */
    const float* env_rewards;
    const uint8_t* env_terminations;
    const uint8_t* env_truncations;
    uint8_t* done_row;
    uint8_t* termination_row;
    uint8_t* truncation_row;
    int32_t flag_is_truncation;
} rlx_value_job;
typedef struct rlx_rollout_step {
    const float* params;
    const float* tiles;   /* fragment-tile weight image: rlx_mlp_pack_tiles() after every change of params */
    const rlx_mlp_layout* layout;
    const float* states;
    const float* eps;
    int64_t m;
    float* action;
    float* logprob;
    float* value;
    float* states_copy;
    int32_t n_value_jobs; /* 0..2 */
    rlx_value_job value_jobs[2];
    int32_t bf16;         /* 0: precision "32" (f32-accurate products: bf16 planes on the bf16 matrix pipe, or the exact f32 MFMA with
                             RLX_F32_EXACT_MFMA=1), tiles from rlx_mlp_pack_tiles; 1: bf16 MFMA operands (f32 accumulate), tiles
                             from rlx_mlp_pack_tiles_bf16 */
} rlx_rollout_step;
int rlx_mlp_rollout_step(const rlx_rollout_step* step, rlx_stream_t stream);

/* rlx_ppo_step: forward + loss + backward of one micro-batch in two launches
 *   <- EmbodiedFSDPActor.train_micro_batch, rlinf/workers/actor/embodied_fsdp_actor_worker.py:591-700
 *      = MLPPolicy.default_forward (mlp_policy.py:202-236) + policy_loss "actor_critic" | "actor"
 *        (algorithms/registry.py:77-92, utils.py:280-376, losses.py:170-380) + loss.backward().
 *   states [m, D], action / old_logprobs [m, act_dim], advantages / prev_values / returns / loss_mask /
 *   loss_mask_sum: [m * act_dim / raw_per_adv] elements (the RAW minibatch views; preprocess_loss_inputs is fused).
 *   grad_out = d(total loss)/d(this micro-batch's loss) = 1 / gradient_accumulation.
 *   grads [slabs][n_params]: split-K gradient slabs, every element written (rlx_clip_adamw_step sums them);
 *   slabs must equal rlx_ppo_step_slabs(layout, m).  out: the rlx_ppo_out metric row (device).
 *   Needs obs_dim <= 64, act_dim, val_dim <= 16 and (has_critic) act_dim / raw_per_adv == val_dim. */
struct rlx_decoupled_loss_params; /* a19b, below */
typedef struct rlx_ppo_step_args {
    const float* params;
    const rlx_mlp_layout* layout;
    const rlx_ppo_loss_params* loss;
    const float* states;
    const float* action;
    const float* old_logprobs;
    const float* advantages;
    const float* prev_values;
    const float* returns;
    const uint8_t* loss_mask;
    const int64_t* loss_mask_sum;
    int64_t m;
    float grad_out;
    float* grads;
    int32_t slabs;
    float* out;
    void* workspace;
    size_t workspace_bytes;
    const float* tiles; /* optional: an up-to-date fragment-tile image of params (rlx_mlp_pack_tiles[_bf16], or -- f32 only --
                           kept fresh by rlx_clip_adamw_step); NULL = rlx_ppo_step packs one into its workspace first */
    int32_t bf16;       /* 1: bf16 MFMA operands; activations / gradients travel to the weight-gradient kernel as bf16 */
    /* Optional: the DECOUPLED (asynchronous PPO) actor loss of a19b inside the fused step
     *   <- AsyncPPOEmbodiedFSDPActor.run_training, rlinf/workers/actor/async_ppo_fsdp_worker.py:374-467
     *      (policy_loss "decoupled_actor_critic": losses.py:27-167, :383-393).  NULL: the classic loss above.
     *   decoupled->ppo is ignored (`loss` is used); proximal_logprobs (RLX_PROX_GIVEN) / versions (RLX_PROX_FROM_VERSIONS, else
     *   optional): [m, act_dim] like old_logprobs.  current_version_dev: optional device float read when the launch EXECUTES
     *   (instead of decoupled->current_version), so that a captured hipGraph can be replayed under the next policy version.
     *   `out` is then the rlx_dppo_out row.  The denominator of the actor loss -- the count of the behaviour mask -- depends on
     *   every row's forward, so the gradients of the ACTOR network (actor_logstd, backbone.*, actor_mean.*) leave in SUM form:
     *   multiply them by out[RLX_PPO_ACTOR_GRAD_SCALE] (rlx_clip_adamw_step does: rlx_adamw_params.deferred_scale).  The value
     *   network's gradients are final as always.  The row-split bf16 launch (RLX_FUSED_ROWS) has no decoupled form: the
     *   column-split one runs. */
    const struct rlx_decoupled_loss_params* decoupled;
    const float* proximal_logprobs;
    const float* versions;
    const float* current_version_dev;
} rlx_ppo_step_args;
int rlx_ppo_step_slabs(const rlx_mlp_layout* layout, int64_t m);                   /* bf16 == 0 */
/* the split-K slab count depends on the operand precision (the f32 launch is bound by the matrix pipe and wants two
   workgroups per CU, the bf16 one by slab bytes and wants fewer, larger slabs): grads must hold exactly this many */
int rlx_ppo_step_slabs_for(const rlx_mlp_layout* layout, int64_t m, int32_t bf16);
size_t rlx_ppo_step_workspace_bytes(const rlx_mlp_layout* layout, int64_t m);
int rlx_ppo_step(const rlx_ppo_step_args* args, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a19b  decoupled_loss  <- compute_decoupled_ppo_actor_loss rlinf/algorithms/losses.py:27-167 and
 *        compute_decoupled_ppo_actor_critic_loss :383-393 (registry name "decoupled_actor_critic"), after
 *        preprocess_loss_inputs incl. its proximal_logprobs / versions branches (rlinf/algorithms/utils.py:310-352).
 * Same element geometry, critic, workspace protocol and BACKWARD as a18-a21 (use rlx_ppo_loss_bwd with this `out`).
 *   proximal_logprobs [n_adv * raw_per_adv] f32   RLX_PROX_GIVEN
 *   versions          [n_adv * raw_per_adv] f32   behaviour-policy version per raw entry (the first entry of each loss
 *                                                 element's slice is used); needed for RLX_PROX_FROM_VERSIONS, optional
 *                                                 otherwise (feeds out[RLX_DPPO_AVERAGE_VERSION])
 * ------------------------------------------------------------------------------------------ */
enum rlx_proximal_mode {
    RLX_PROX_GIVEN = 0,          /* proximal_logprobs supplied */
    RLX_PROX_IS_OLD = 1,         /* proximal = old_logprobs (neither proximal_logprobs nor versions/current_version) */
    RLX_PROX_FROM_VERSIONS = 2   /* old + clamp((v_theta-1-v_b)/(v_theta-v_b), 0, 1) * (logprobs - old), losses.py:72-89 */
};
enum rlx_dppo_out {             /* slots 0-8 of the RLX_PPO_OUT_FLOATS row; 9-18 are the same as rlx_ppo_out */
    RLX_DPPO_LOSS = 0,
    RLX_DPPO_POLICY_LOSS = 1,
    RLX_DPPO_PROXIMAL_RATIO = 2,
    RLX_DPPO_CLIPPED_PROXIMAL_RATIO = 3,
    RLX_DPPO_DUAL_CLIP_FRACTION = 4,
    RLX_DPPO_BEHAV_CLIP_FRACTION = 5,
    RLX_DPPO_PROXIMAL_APPROX_KL = 6,
    RLX_DPPO_BEHAV_APPROX_KL = 7,
    RLX_DPPO_CLIP_FRACTION = 8,
    RLX_DPPO_AVERAGE_VERSION = 19
};
typedef struct rlx_decoupled_loss_params {
    rlx_ppo_loss_params ppo;       /* clip_log_ratio_* are not part of this loss and ignored */
    int32_t proximal_mode;         /* rlx_proximal_mode */
    float current_version;
    int32_t use_behave_threshold;
    float behave_weight_threshold;
} rlx_decoupled_loss_params;
size_t rlx_decoupled_loss_workspace_bytes(int64_t n_adv);
int rlx_decoupled_loss_fwd(const float* logprobs, const float* old_logprobs, const float* proximal_logprobs,
                           const float* versions, const float* advantages, const float* values,
                           const float* prev_values, const float* returns, const uint8_t* loss_mask,
                           const int64_t* loss_mask_sum, int64_t n_adv, const rlx_decoupled_loss_params* params,
                           float* g_logp, float* g_value, float* out, void* workspace, size_t workspace_bytes,
                           rlx_stream_t stream);

/* ==========================================================================================
 * Token tier: the reasoning (LLM) learner's per-token path over vocabulary logits.
 * ========================================================================================== */

/* ------------------------------------------------------------------------------------------
 * t1  token_logprob  <- compute_logprobs_from_logits (op_type="torch") rlinf/utils/utils.py:454-492,
 *                       compute_entropy_from_logits rlinf/utils/utils.py:495-512,
 *                       logits.div_(temperature) rlinf/workers/actor/fsdp_actor_worker.py:476-498
 *   logits   n_tokens rows of `vocab` elements (f32 or bf16); row i starts at element
 *            (i / rows_per_seq) * seq_stride + (i % rows_per_seq) * row_stride, so the reference's
 *            response slice logits[:, -resp-1:-1, :] is addressed in place (no reshape copy).
 *   labels   [n_tokens] i64; -100 (torch's ignore_index) gives logprob -0.0, any other out-of-range label NaN
 *            (torch raises there; a kernel cannot).
 *   logprob  [n_tokens] f32 = x[label] - logsumexp(x),  x = logits / temperature
 *   entropy  [n_tokens] f32 = logsumexp(x) - sum softmax(x) * x      (optional, NULL skips it)
 *   lse      [n_tokens] f32 logsumexp(x), kept for the backward pass
 * One pass over the logits (online softmax): algorithmic bytes = n_tokens * vocab * sizeof(dtype).  All arithmetic
 * in f32; temperature != 1 divides in the logits' dtype (bf16 logits are re-rounded to bf16, as div_ does);
 * round_outputs rounds logprob to the logits dtype the way torch's cross_entropy in that dtype does.
 *
 * token_logprob_bwd: d_logits[i,v] = ( d_logprob[i] * (1[v==label_i] - p_iv)
 *                                    - d_entropy[i] * p_iv * (log p_iv + H_i) ) / temperature
 * written in the logits dtype with its own strides; d_logits may alias logits (in-place).  Rows whose
 * d_logprob and d_entropy are both zero are written as zeros without reading their logits.
 * ------------------------------------------------------------------------------------------ */
enum rlx_dtype { RLX_DTYPE_F32 = 0, RLX_DTYPE_BF16 = 1, RLX_DTYPE_F16 = 2,
                 RLX_DTYPE_RAW8 = 3, RLX_DTYPE_RAW16 = 4, RLX_DTYPE_RAW32 = 5, RLX_DTYPE_RAW64 = 6 /* integer / bool tensors by width */ };
typedef struct rlx_token_rows {
    int64_t n_tokens;
    int32_t vocab;
    int32_t dtype;          /* rlx_dtype of logits and d_logits */
    int64_t rows_per_seq;   /* >= 1; n_tokens for a flat contiguous matrix */
    int64_t seq_stride;     /* elements */
    int64_t row_stride;     /* elements, >= vocab */
    float temperature;      /* > 0; 1.0 = no scaling */
    int32_t round_outputs;  /* 1: logprob rounded to `dtype` (no-op for f32) */
} rlx_token_rows;
int rlx_token_logprob_fwd(const void* logits, const int64_t* labels, const rlx_token_rows* rows, float* logprob,
                          float* entropy, float* lse, rlx_stream_t stream);
/* Packed / variable-length sequences (FSDPActor.forward_batch with runner.enable_dynamic_batch_size or actor.model.
 * variable_seq_lengths, rlinf/workers/actor/fsdp_actor_worker.py:450-505; unpack_fsdp_logprobs / unpack_sequences,
 * rlinf/hybrid_engines/fsdp/utils.py:858-1010): `logits` are the rows of ONE packed stream (the valid windows of the batch's
 * sequences back to back), labels[t] = the stream's token t + 1 (eos behind the last).  The unpack is fused into the stores:
 *   logprob[lp_dst[t]]  = log softmax(x_t)[labels[t]]     (the reference shifts the packed log-probs right by one and scatters
 *   entropy[ent_dst[t]] = H(softmax(x_t))                  segment i to columns [idx_start_i, idx_end_i) of row i, then keeps the
 * last response_len columns: lp_dst / ent_dst are that map, flat indices into the [bsz, response_len] outputs, -1 = dropped --
 * note the entropy is NOT shifted, as in the reference).  Outputs must be zero-filled by the caller (pad_val 0).  A row with no
 * destination is not read at all (the prompt tokens: only their last one feeds a response log-prob); lse [n_tokens] is kept per
 * packed row for the backward.  rlx_token_logprob_bwd_packed is rlx_token_logprob_bwd with the same maps: d_logprob / d_entropy
 * and the forward's `entropy` are the UNPACKED [bsz, response_len] tensors, row t takes the gradient of the element it fed
 * (a row that fed none is written as zeros without being read). */
int rlx_token_logprob_fwd_packed(const void* logits, const int64_t* labels, const rlx_token_rows* rows, const int32_t* lp_dst,
                                 const int32_t* ent_dst, float* logprob, float* entropy, float* lse, rlx_stream_t stream);
int rlx_token_logprob_bwd_packed(const void* logits, const int64_t* labels, const rlx_token_rows* rows, const float* lse,
                                 const float* entropy, const int32_t* lp_dst, const int32_t* ent_dst, const float* d_logprob,
                                 const float* d_entropy, void* d_logits, int64_t d_seq_stride, int64_t d_row_stride,
                                 rlx_stream_t stream);
int rlx_token_logprob_bwd(const void* logits, const int64_t* labels, const rlx_token_rows* rows, const float* lse,
                          const float* entropy, const float* d_logprob, const float* d_entropy, void* d_logits,
                          int64_t d_seq_stride, int64_t d_row_stride, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * t2  token_loss  <- the micro-batch loss of FSDPActor.training_step, rlinf/workers/actor/fsdp_actor_worker.py:694-781:
 *        compute_ppo_actor_loss(loss_agg_func=..., fast_path_zero_loss_mask=True) rlinf/algorithms/losses.py:170-312
 *        - entropy_bonus * agg(entropy) + kl_beta * agg(kl_penalty(ref_logprobs, logprobs, type))
 *      with agg = get_loss_agg_func(loss_agg) rlinf/utils/utils.py:337-381 and kl_penalty rlinf/algorithms/utils.py:26-64.
 *   logprobs / old_logprobs / advantages / ref_logprobs / entropy [bsz, seq] f32, loss_mask [bsz, seq] u8 (NULL = all on)
 *   out          [RLX_TOK_OUT_FLOATS] f32 (enum below)
 *   g_logp       [bsz, seq] f32  d(sum-form loss)/d logprob per token (policy + kl_beta * kl), unscaled
 *   g_entropy    [bsz, seq] f32  -entropy_bonus * mask (only when use_entropy and entropy_bonus != 0, else may be NULL)
 *   row_weight   [bsz] f32       the aggregation's per-sequence weight (1/count, 1/bsz or 1/(bsz*count_row))
 * token_loss_bwd: d_logprobs = grad_out * row_weight[row] * g_logp (same for d_entropy), grad_out a device scalar
 * (the learner's 1/gradient_accumulation rides in it).
 * fast_path_zero_loss_mask reproduces the reference's check of the FIRST sequence only (losses.py:206): when
 * loss_mask[0] is all zero the policy term and its metrics are zero and out[RLX_TOK_POLICY_ON] = 0.
 * ------------------------------------------------------------------------------------------ */
enum rlx_loss_agg { RLX_AGG_TOKEN_MEAN = 0, RLX_AGG_SEQ_MEAN_TOKEN_SUM = 1, RLX_AGG_SEQ_MEAN_TOKEN_MEAN = 2 };
enum rlx_kl_type { RLX_KL_NONE = 0, RLX_KL_K1 = 1, RLX_KL_ABS = 2, RLX_KL_K2 = 3, RLX_KL_K3 = 4 };
enum rlx_tok_out {
    RLX_TOK_LOSS = 0,          /* policy - entropy_bonus*entropy_loss + kl_beta*kl_loss ("actor/final_loss") */
    RLX_TOK_POLICY_LOSS,
    RLX_TOK_POLICY_LOSS_ABS,
    RLX_TOK_RATIO,
    RLX_TOK_RATIO_ABS,
    RLX_TOK_CLIPPED_RATIO,
    RLX_TOK_DUAL_CLIPPED_RATIO,
    RLX_TOK_APPROX_KL,
    RLX_TOK_CLIP_FRACTION,
    RLX_TOK_ENTROPY_LOSS,
    RLX_TOK_KL_LOSS,
    RLX_TOK_TOKEN_NUM,         /* loss_mask.count_nonzero() */
    RLX_TOK_POLICY_ON,         /* 0 when the zero-loss-mask fast path fired */
    RLX_TOK_OUT_FLOATS = 16
};
typedef struct rlx_token_loss_params {
    rlx_ppo_loss_params ppo;   /* ratio_lo/hi, dual clip, log-ratio clamps, critic_warmup; critic fields ignored */
    int32_t loss_agg;          /* rlx_loss_agg */
    int32_t fast_path_zero_loss_mask;
    int32_t kl_type;           /* rlx_kl_type; RLX_KL_NONE or ref_logprobs == NULL skips the term */
    float kl_beta;
    int32_t use_entropy;       /* entropy != NULL and the learner's calculate_entropy */
    float entropy_bonus;       /* 0 = entropy is only reported */
} rlx_token_loss_params;
size_t rlx_token_loss_workspace_bytes(int64_t bsz, int64_t seq);
int rlx_token_loss_fwd(const float* logprobs, const float* old_logprobs, const float* advantages,
                       const float* ref_logprobs, const float* entropy, const uint8_t* loss_mask, int64_t bsz,
                       int64_t seq, const rlx_token_loss_params* params, float* g_logp, float* g_entropy,
                       float* row_weight, float* out, void* workspace, size_t workspace_bytes, rlx_stream_t stream);
int rlx_token_loss_bwd(const float* g_logp, const float* g_entropy, const float* row_weight, const float* grad_out,
                       float* d_logprobs, float* d_entropy, int64_t bsz, int64_t seq, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * t3  grpo_seq_adv  <- preprocess_reasoning_advantages_inputs (adv_type="grpo") rlinf/algorithms/utils.py:177-262
 *                      + compute_grpo_advantages rlinf/algorithms/advantages.py:89-121
 *                      + postprocess_reasoning_advantages_outputs rlinf/algorithms/utils.py:265-277
 *   rewards [bsz] f32, loss_mask [bsz, seq] u8 -> advantages [bsz, seq] f32 in the sequence-major layout
 *   (the reference transposes to [seq, bsz], broadcasts, and transposes back with a copy).
 * ------------------------------------------------------------------------------------------ */
int rlx_grpo_seq_adv(const float* rewards, const uint8_t* loss_mask, float* advantages, int64_t bsz, int64_t seq,
                     int group_size, float eps, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * t3b  reinpp_seq_adv  <- adv_type "reinpp" on reasoning batches: preprocess_reasoning_advantages_inputs
 *                         rlinf/algorithms/utils.py:218-219,245-251 + compute_reinpp_advantages rlinf/algorithms/advantages.py:300-364
 *                         (use_reinpp_baseline False) + postprocess_reasoning_advantages_outputs rlinf/algorithms/utils.py:265-277
 *   rewards [bsz] f32, loss_mask [bsz, seq] u8, logprob / ref_logprob [bsz, seq] f32 (read only when kl_beta > 0; the
 *   per-token reward is then -kl_beta * kl_penalty(logprob, ref_logprob, kl_type)) -> advantages [bsz, seq] f32 =
 *   (return-to-go - masked mean) * rsqrt(max(masked variance, 1e-8)), every position (not only masked ones), like the
 *   reference.  The scalar reward of sequence b sits at seq-1 minus the index of the first True of sequence bsz-1-b's mask
 *   (the reference flips the batch axis when it looks for the last valid token) -- seq-1 for masks that start with True.
 * ------------------------------------------------------------------------------------------ */
size_t rlx_reinpp_workspace_bytes(int64_t bsz);
int rlx_reinpp_seq_adv(const float* rewards, const uint8_t* loss_mask, const float* logprob, const float* ref_logprob,
                       int kl_type, float kl_beta, float* advantages, int64_t bsz, int64_t seq, void* workspace,
                       size_t workspace_bytes, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * t5  gae_seq  <- adv_type "gae" on reasoning batches: preprocess_reasoning_advantages_inputs rlinf/algorithms/utils.py:177-262
 *                 + compute_gae_advantages_and_returns rlinf/algorithms/advantages.py:24-86
 *                 + postprocess_reasoning_advantages_outputs rlinf/algorithms/utils.py:265-277
 *   values [bsz, seq] f32, rewards [bsz] f32 (the scalar reward sits on the LAST position of the row, as the reference
 *   places it) -> advantages, returns [bsz, seq] f32, un-normalised; gamma_lambda = (float)((double)gamma * gae_lambda).
 *   Normalisation is rlx_masked_standardize over the [bsz, seq] buffer with the loss mask (layout-agnostic).
 *   Rows of 16-byte aligned length (seq % 4 == 0, 16-byte aligned base pointers, seq <= 32768) run in registers, one wave per
 *   2048 tokens (csrc/gae_seq.hip); other shapes are staged through LDS.  `workspace` (rlx_gae_seq_workspace_bytes) is only read
 *   by a development variant of the long-row kernel; NULL is accepted otherwise.
 * ------------------------------------------------------------------------------------------ */
size_t rlx_gae_seq_workspace_bytes(int64_t bsz, int64_t seq);
int rlx_gae_seq(const float* values, const float* rewards, float* advantages, float* returns, int64_t bsz, int64_t seq,
                float gamma, float gamma_lambda, void* workspace, size_t workspace_bytes, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * t4  categorical_sample (K2)  <- OpenVLA-OFT's discrete action head: _discrete_prediction's sampling branch and
 *       _compute_logprobs_and_entropy, rlinf/models/embodiment/openvla_oft/official/openvla_oft_action_model.py:258-287,363-414
 *   logits   n rows of K <= 1024 action-bin logits (the n_action_bins window of the vocabulary; rows addressed like t1,
 *            so the window is read in place out of the model's [B, seq, V] output)
 *   noise    [n, K] dense, same dtype: the Exp(1) draw q of torch.multinomial(num_samples=1), which returns
 *            argmax(softmax(x) / q); NULL = do_sample False (argmax of the raw logits, no temperature / top-k)
 *   x = logits / temperature (in the tensor's dtype), then transformers' TopKLogitsWarper (scores below the k-th
 *   largest -> -inf; top_k <= 0 or >= K disables it)
 *   tokens   [n] i64 bin index in [0, K);  logprob [n] f32 = log softmax(x)[token] (optional; rounded to the dtype when
 *            rows->round_outputs);  actions [n] f32 = bin_centers[clamp(K - token - 1, 0, n_centers - 1)] (optional)
 *   Exactness of `tokens`: BIT-EXACT against the reference's CPU path (torch.multinomial over torch.softmax on the host).  The
 *   kernel replays ATen's vectorised last-dim softmax operation for operation -- Sleef's expf_u10, the row sum accumulated per SIMD
 *   lane over W-element chunks and folded by a W-lane butterfly, p = e * (1 / sum), then p / q and the first-index argmax --
 *   so no tie allowance is needed (tests/test_gpu_token_path.py: torch.equal).  `softmax_lanes` = W: 16 on AVX-512 builds of
 *   torch (0 selects it; the committed golden files were written on such a host), 8 on AVX2 hosts -- the order of the f32
 *   additions in the row sum is the one thing of the reference's result that depends on the host it ran on.
 * ------------------------------------------------------------------------------------------ */
int rlx_categorical_sample(const void* logits, const rlx_token_rows* rows, const void* noise, int top_k, int softmax_lanes,
                           const float* bin_centers, int n_centers, int64_t* tokens, float* logprob, float* actions,
                           rlx_stream_t stream);

/* ==========================================================================================
 * Weight-sync tier: the sparse weight patch of the actor -> rollout push.
 *   w1 patch_scan / w2 patch_emit  <- GPUSnapshotPatchBuilder.create_patch, rlinf/hybrid_engines/weight_syncer/patch_syncer.py:648-774
 *                                     (to(snapshot dtype) -> ne -> nonzero -> gather -> snapshot[rows, cols] = values)
 *                                     + PatchBuilder.delta_encode :290-327
 *   w3 patch_apply                 <- the per-tensor body of PatchWeightSyncer.apply :1040-1137 + PatchBuilder.delta_decode :329-370
 * One tensor per call, seen through its 2-D COO view (as_coo_2d_view :60-95): n_elems = rows * cols, row-major.
 *   scan : ONE read of value (value_dtype) and snapshot (snapshot_dtype; equal, or f32 -> bf16 / f16 when the receiver
 *          holds a narrower copy); writes nnz[0] (device i64) = number of elements with value.to(dtype) != snapshot under
 *          torch.ne (NaN != NaN, +0 == -0); keeps a bit mask and block offsets in `workspace`.
 *   emit : after the host has read nnz (the reference synchronises on nonzero() at the same point) and allocated
 *          out_rows / out_cols [nnz] i64 and out_values [nnz] of the snapshot dtype: fills them in nonzero() order --
 *          absolute indices, or PatchBuilder.delta_encode's deltas -- updates the snapshot in place and raises
 *          maxima[0] / maxima[1] (device u64, zeroed by the caller) to the largest emitted row / column code, which is what
 *          downscale_nonnegative_indices :35-57 needs to pick uint8 / int32 / int64.
 *   apply: target[r, c] = value for every entry, indices given in any of the three index dtypes (code 0 u8, 1 i32, 2 i64),
 *          absolute or delta-encoded.
 * Integer / byte work: bit-exact.
 * ========================================================================================== */
size_t rlx_patch_workspace_bytes(int64_t n_elems);
int rlx_patch_scan(const void* value, int value_dtype, const void* snapshot, int snapshot_dtype, int64_t n_elems,
                   void* workspace, size_t workspace_bytes, int64_t* nnz, rlx_stream_t stream);
int rlx_patch_emit(const void* value, int value_dtype, void* snapshot, int snapshot_dtype, int64_t n_elems, int64_t cols,
                   int delta_encoding, const void* workspace, int64_t nnz /* what scan reported: sizes of the outputs */,
                   int64_t* out_rows, int64_t* out_cols, void* out_values, uint64_t* maxima, rlx_stream_t stream);
size_t rlx_patch_apply_workspace_bytes(int64_t nnz);
int rlx_patch_apply(void* target, int dtype, int64_t target_rows, int64_t target_cols, const void* rows,
                    int rows_index_dtype, const void* cols, int cols_index_dtype, int delta_encoded, const void* values,
                    int64_t nnz, void* workspace, size_t workspace_bytes, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * w4  copy_segments  <- the bucket weight syncer's per-parameter casts and copies:
 *       sender    iter_named_tensor_buckets, rlinf/hybrid_engines/weight_syncer/bucket_syncer.py:110-121
 *                 (tensor.to(device=bucket_device, dtype=transport_dtype) per parameter)
 *       receiver  BucketWeightSyncer.apply :296-323 (load_state_dict: param.copy_(received) per parameter)
 * One launch moves every tensor of a bucket between its own storage and one flat transport buffer.
 *   table   n_segments x (src, dst, n elements, src dtype, dst dtype); dtype pairs: equal dtypes (raw copy by width) and
 *           any pair of f32 / bf16 / f16 (c10's conversions: round to nearest even, canonical NaN).  Integer / bool / f64
 *           tensors never change dtype on this path (the reference only re-types floating tensors, :199-201).
 *   plan    HOST function: validates the table and fills first_chunk (RLX_COPY_CHUNK elements per workgroup);
 *           the caller then places the table in device memory.
 *   run     asynchronous on `stream`; src / dst ranges of different segments must not overlap.
 * Byte work: bit-exact against torch's .to(dtype).
 * ------------------------------------------------------------------------------------------ */
#define RLX_COPY_CHUNK 4096
typedef struct rlx_copy_segment {
    const void* src;
    void* dst;
    int64_t n;                    /* elements */
    int32_t src_dtype, dst_dtype; /* enum rlx_dtype */
    int64_t first_chunk;          /* filled by rlx_copy_segments_plan */
} rlx_copy_segment;
int rlx_copy_segments_plan(rlx_copy_segment* table_host, int32_t n_segments, int64_t* total_chunks);
int rlx_copy_segments(const rlx_copy_segment* table_dev, int32_t n_segments, int64_t total_chunks, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * f3  patch-stream codec  <- NVCompCompressor._compress_tensor / _decompress_tensor,
 *      rlinf/hybrid_engines/weight_syncer/compressor.py:148-199 (the rows / cols / value-byte fields of a WeightPatch travel as
 *      compressed byte tensors inside a CompressedWeightPatch, patch_syncer.py:205-250).
 * nvCOMP is NVIDIA-only and its container is not public: the payload is this build's own "RLXZ" v1 format (csrc/zplane_codec.hip:
 * byte planes of 4096-element blocks, each stored as nothing / two-level zero masks + the nonzero bytes / raw, whichever is
 * smallest -- what delta-encoded COO indices and the exponent bytes of changed weights actually consist of).  Lossless, byte-exact.
 *   bound_bytes      worst-case size of a compressed stream (allocate `out` that large)
 *   compress         in: n_elems elements of elem_size (1, 2, 4, 8) bytes, 16-byte aligned; *out_bytes (device u64) = stream length
 *   parse_header     the first 24 bytes of a stream (HOST copy) -> n_elems, elem_size, total stream length
 *   decompress       *status (device int, zeroed by the caller) != 0: the stream is corrupt / truncated (nothing out of bounds
 *                    is ever read or written)
 * ------------------------------------------------------------------------------------------ */
size_t rlx_zplane_bound_bytes(int64_t n_elems, int elem_size);
size_t rlx_zplane_workspace_bytes(int64_t n_elems, int elem_size);
int rlx_zplane_compress(const void* in, int64_t n_elems, int elem_size, void* out, size_t out_capacity, uint64_t* out_bytes,
                        void* workspace, size_t workspace_bytes, rlx_stream_t stream);
int rlx_zplane_parse_header(const void* header_host, int64_t* n_elems, int* elem_size, uint64_t* total_bytes);
int rlx_zplane_decompress(const void* in, size_t in_bytes, void* out, int64_t n_elems, int elem_size, int* status, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a15  rollout_metrics  <- compute_rollout_metrics, rlinf/utils/metric_utils.py:422-506
 *   masked mean / min / max of rewards, advantages, returns: `count` (<= 3) f32 arrays of sizes[k] elements and ONE optional
 *   mask of mask_elems bytes (element e covers sizes[k] / mask_elems consecutive elements of array k: [.., 1] against [.., C]).
 *   out f64[count][4] = {sum, count, -min, max} on the device: (sum, count) reduce with SUM over ranks, (-min, max) with MAX;
 *   an empty selection leaves count 0 and -inf / -inf (the caller reports NaN like the reference, :458-460).
 * ------------------------------------------------------------------------------------------ */
size_t rlx_rollout_metrics_workspace_bytes(void);
int rlx_rollout_metrics(const float* const* arrays, const int64_t* sizes, int count, const uint8_t* mask, int64_t mask_elems,
                        double* out, void* workspace, size_t workspace_bytes, rlx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * e  data-parallel gradient all-reduce over xGMI  <- FSDP NO_SHARD gradient synchronisation,
 *      rlinf/hybrid_engines/fsdp/strategy/fsdp.py:480-496 (sync_grad / no_sync around backward), the averaged gradient that
 *      FSDPModelManager.optimizer_step then clips and applies (fsdp_model_manager.py:429-463).
 * One process per GPU.  Every rank owns a fine-grained device buffer (two staging slots of n_max floats, two reduced-shard areas,
 * norm partials and two flag sets), exported with hipIpcGetMemHandle and mapped by every peer.  An all-reduce is a short chain of
 * launches with no host involvement and no RCCL call -- so the whole optimizer step stays one hipGraph-capturable kernel chain.
 * Two forms (rlx_xgmi_configure `algo`; the default is direct for world <= 3, rs + ag from 4 ranks on):
 *   direct   stage    sum this rank's split-K gradient slabs into its staging slot (seq + 1) & 1
 *            reduce   publish "slot complete" to every peer's flag array, wait for every peer's flag, then read all W staged
 *                     gradients over xGMI and add them IN RANK ORDER (every rank forms bit-identical sums), scale by grad_scale,
 *                     write the reduced gradient locally with the squared-norm partials the AdamW launch needs
 *            clip + AdamW                                                        [n floats per link, one hand-shake]
 *   rs + ag  stage    as above
 *            reduce-scatter   hand-shake 0, then rank r reads shard r of every rank's staged gradient, adds in rank order,
 *                     scales, writes the shard + its norm partials into its own exported shard area
 *            gather + clip + AdamW   hand-shake 1, then every rank reads shard q and its norm partials from rank q inside the
 *                     AdamW launch (same reduction tree on every rank -> the same norm and bit-identical parameters)
 *                                                                                [2 n / W floats per link, two hand-shakes]
 * The hand-shake runs inside the consuming launch (wait_mode 0: every block polls) or as a one-wave launch of its own in front
 * of it (wait_mode 1: ranks that share a device must not hold it with a device-wide spin).
 * The sequence number lives on the device and is advanced by the launch that consumes the result (graph replay needs no
 * host bookkeeping).  A wait that exceeds timeout_ms sets the status word (rlx_xgmi_status) instead of hanging the GPU, and the
 * AdamW launch of such an all-reduce SKIPS its update (stats[1] = 0): garbage sums are never applied.
 * 1.15 MB x 7 peers at ~50 GB/s per xGMI link is ~25 us of wire time; RCCL's ring needs 2 (W - 1) latency-bound steps.
 * ------------------------------------------------------------------------------------------ */
#define RLX_XGMI_MAX_RANKS 8
#define RLX_XGMI_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t) */
typedef struct rlx_xgmi_comm rlx_xgmi_comm;
/* create: allocates this rank's buffer; `handle_out` (RLX_XGMI_HANDLE_BYTES) is what the peers need -- exchange the handles
 * out of band (torch.distributed all_gather_object / the reference's Worker.send) and call connect with all of them
 * (world x RLX_XGMI_HANDLE_BYTES, own slot ignored).  mem_kind: 0 fine-grained (default), 1 uncached, 2 plain hipMalloc. */
int rlx_xgmi_create(int rank, int world, int64_t n_max, int timeout_ms, int mem_kind, rlx_xgmi_comm** comm, void* handle_out);
int rlx_xgmi_connect(rlx_xgmi_comm* comm, const void* all_handles);
/* Same-process emulation of a W-rank group (tests: every "rank" on its own stream of ONE device, no IPC): wires W
 * communicators created with rank 0 .. W-1 to each other's buffers directly. */
int rlx_xgmi_connect_local(rlx_xgmi_comm* const* comms, int world);
/* TIMING TOOL (bench.py `scaling_model`): every peer of `comm` (created as rank r of W) is this rank's OWN buffer, so the chain of
 * launches a rank of a W-GPU job runs per optimizer step -- stage, hand-shake, reduce(-scatter), hand-shake, gather + clip + AdamW --
 * executes on one device with local-memory reads in place of xGMI reads and hand-shakes that are satisfied at once.  The reduced
 * values are NOT a valid all-reduce (every shard is this rank's shard); use scratch parameter / moment buffers. */
int rlx_xgmi_connect_self(rlx_xgmi_comm* comm);
/* How a self-connected communicator plays the ONE-LAUNCH exchange (rlx_xgmi_clip_adamw_step with rlx_adamw_params.sync_words):
 * 0 (default) EXACT -- a block plays its own W - 1 contributors / the owner that answers it: with W a power of two the result is
 * the single-GPU step's bit for bit (parity tests); 1 TIMING -- the rank's blocks b, b + nblk / W, ... play ranks 0, 1, ...'s
 * copies of owned block b, so an owner's contributions come from, and its answers go to, other workgroups running concurrently
 * (a rank's real communication structure on one device; values are sums of different blocks' gradients: scratch buffers only;
 * needs nblk % W == 0, else the exact form runs). */
int rlx_xgmi_self_timing(rlx_xgmi_comm* comm, int on);
/* algo: 0 direct, 1 reduce-scatter + all-gather; wait_mode: 0 inline, 1 own launch; timeout_ms > 0: new bound of every peer
 * wait.  -1 (0 for the timeout) keeps the current value.  Every rank of a group must configure the same algo. */
int rlx_xgmi_configure(rlx_xgmi_comm* comm, int algo, int wait_mode, int timeout_ms);
int rlx_xgmi_destroy(rlx_xgmi_comm* comm);
/* 0 = ok, 1 = a peer wait timed out since the last call (cleared by the call; synchronises the stream's device first). */
int rlx_xgmi_status(rlx_xgmi_comm* comm);
/* Asynchronous form for a loop that reads its metrics late: *dst (device f32) = the status word as it stands when the launch runs on
 * `stream`, i.e. behind the exchanges queued so far -- it travels to the host with the step's metric vector; nothing is cleared and
 * the host is not blocked (rlx_xgmi_status above is a blocking read on the null stream that also clears the word). */
int rlx_xgmi_status_snapshot(rlx_xgmi_comm* comm, float* dst, rlx_stream_t stream);
/* out[i] = scale * sum_r in_r[i]; `in` is this rank's contribution ([slabs][n], summed first), out a local buffer. */
int rlx_xgmi_allreduce_f32(rlx_xgmi_comm* comm, const float* in, int slabs, float* out, int64_t n, float scale,
                           void* workspace, size_t workspace_bytes /* rlx_adamw_workspace_bytes(n) */, rlx_stream_t stream);
/* rlx_clip_adamw_step with the all-reduce in front: grads [p->grad_partials][n] are this rank's slabs, grad_flat [n]
 * receives the reduced gradient (scaled by p->grad_scale = 1 / world for the data-parallel mean), then clip + AdamW. */
int rlx_xgmi_clip_adamw_step(rlx_xgmi_comm* comm, float* params, const float* grads, float* grad_flat, float* exp_avg,
                             float* exp_avg_sq, int64_t n, const rlx_adamw_params* p, float* stats, int32_t* step_state,
                             void* workspace, size_t workspace_bytes, rlx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RLX_H */

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

#define RLX_VERSION 100 /* 0.1.0 */

typedef void* rlx_stream_t; /* hipStream_t */

enum rlx_status {
    RLX_OK = 0,
    RLX_EINVAL = -22,  /* bad argument (NULL pointer, non-positive size, unsupported combination) */
    RLX_ENOSPC = -28,  /* workspace too small */
    RLX_EHIP = -5,     /* a HIP runtime call failed; see rlx_last_error() */
    RLX_ENOSYS = -38   /* variant not compiled in */
};

int rlx_version(void);
const char* rlx_last_error(void);

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

#ifdef __cplusplus
}
#endif
#endif /* RLX_H */

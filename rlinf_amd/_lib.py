"""ctypes binding of librlx_hip.so (the C ABI declared in include/rlx.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises, and
every op that needs the GPU raises when no HIP device is visible.  Build with
``python -m rlinf_amd.csrc.build`` (or ``__graft_entry__.build()``).
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# RLX_LIB_TAG=<tag>: load librlx_hip_<tag>.so, a variant build of the same sources (see csrc/build.py); dev sweeps only
_TAG = os.environ.get("RLX_LIB_TAG", "")
LIB_PATH = os.path.join(_HERE, "librlx_hip" + ("_" + _TAG if _TAG else "") + ".so")


class RlxError(RuntimeError):
    pass


class GaeParams(Structure):
    _fields_ = [
        ("gamma", c_float),
        ("gamma_lambda", c_float),
        ("normalize_advantages", c_int32),
        ("normalize_returns", c_int32),
        ("norm_eps", c_float),
        ("variant", c_int32),
    ]


class PpoLossParams(Structure):
    _fields_ = [
        ("ratio_lo", c_float), ("ratio_hi", c_float), ("clip_ratio_c", c_float),
        ("clip_log_ratio_min", c_float), ("clip_log_ratio_max", c_float),
        ("value_clip", c_float), ("huber_delta", c_float),
        ("use_dual_clip", c_int32), ("use_clip_log_ratio_min", c_int32), ("use_clip_log_ratio_max", c_int32),
        ("has_critic", c_int32), ("critic_warmup", c_int32), ("max_episode_steps", c_int32),
        ("raw_per_adv", c_int32), ("sub_per_adv", c_int32), ("metric_unbroadcast", c_int32),
    ]


class GatherField(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("row_bytes", c_int64)]


GATHER_MAX_FIELDS = 16
ADAMW_MAX_GROUPS = 8


class AdamwGroup(Structure):
    _fields_ = [("begin", c_int64), ("end", c_int64), ("lr", c_double)]


class CopySegment(Structure):  # rlx_copy_segment
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("n", c_int64), ("src_dtype", c_int32), ("dst_dtype", c_int32),
                ("first_chunk", c_int64)]


class MlpLayout(Structure):
    _fields_ = [
        ("obs_dim", c_int32), ("act_dim", c_int32), ("val_dim", c_int32), ("hidden", c_int32),
        ("n_params", c_int64), ("off_logstd", c_int64),
        ("off_w", (c_int64 * 4) * 2), ("off_b", (c_int64 * 4) * 2),
    ]


class AdamwParams(Structure):
    _fields_ = [
        ("beta1", c_double), ("beta2", c_double), ("eps", c_double), ("weight_decay", c_double),
        ("max_grad_norm", c_float), ("step", c_int32), ("n_groups", c_int32), ("grad_partials", c_int32),
        ("grad_scale", c_float), ("groups", AdamwGroup * ADAMW_MAX_GROUPS),
        ("tile_layout", POINTER(MlpLayout)), ("tiles", c_void_p), ("tiles_bf16", c_int32),
        ("deferred_scale", c_void_p), ("deferred_stride", c_int32), ("deferred_groups", c_int32),
        ("deferred_range", (c_int64 * 2) * 2), ("sync_words", c_void_p),
    ]


class ValueJob(Structure):
    _fields_ = [("states", c_void_p), ("m", c_int64), ("values", c_void_p), ("rewards", c_void_p), ("flags", c_void_p),
                ("chunk", c_int32), ("gamma", c_float), ("env_rewards", c_void_p), ("env_terminations", c_void_p),
                ("env_truncations", c_void_p), ("done_row", c_void_p), ("termination_row", c_void_p),
                ("truncation_row", c_void_p), ("flag_is_truncation", c_int32)]


class RolloutStep(Structure):
    _fields_ = [("params", c_void_p), ("tiles", c_void_p), ("layout", POINTER(MlpLayout)), ("states", c_void_p), ("eps", c_void_p),
                ("m", c_int64), ("action", c_void_p), ("logprob", c_void_p), ("value", c_void_p),
                ("states_copy", c_void_p), ("n_value_jobs", c_int32), ("value_jobs", ValueJob * 2), ("bf16", c_int32)]


class PpoStepArgs(Structure):
    _fields_ = [("params", c_void_p), ("layout", POINTER(MlpLayout)), ("loss", POINTER(PpoLossParams)),
                ("states", c_void_p), ("action", c_void_p), ("old_logprobs", c_void_p), ("advantages", c_void_p),
                ("prev_values", c_void_p), ("returns", c_void_p), ("loss_mask", c_void_p), ("loss_mask_sum", c_void_p),
                ("m", c_int64), ("grad_out", c_float), ("grads", c_void_p), ("slabs", c_int32), ("out", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("tiles", c_void_p), ("bf16", c_int32),
                ("decoupled", c_void_p), ("proximal_logprobs", c_void_p), ("versions", c_void_p), ("current_version_dev", c_void_p)]


PPO_OUT_FLOATS = 20
PPO_ACTOR_GRAD_SCALE, PPO_CRITIC_GRAD_SCALE = 16, 17  # rlx_ppo_out / rlx_dppo_out: 1 / the loss denominators
PPO_OUT_NAMES = {
    "loss": 0, "actor/policy_loss": 1, "actor/policy_loss_abs": 2, "actor/ratio": 3, "actor/ratio_abs": 4,
    "actor/clipped_ratio": 5, "actor/dual_cliped_ratio": 6, "actor/approx_kl": 7, "actor/clip_fraction": 8,
    "critic/value_loss": 9, "critic/value_clip_ratio": 10, "ev/count": 11, "ev/returns_sum": 12,
    "ev/returns_sq_sum": 13, "ev/errors_sum": 14, "ev/errors_sq_sum": 15, "actor/entropy_loss": 19,
}


class DecoupledLossParams(Structure):  # include/rlx.h: rlx_decoupled_loss_params
    _fields_ = [("ppo", PpoLossParams), ("proximal_mode", c_int32), ("current_version", c_float),
                ("use_behave_threshold", c_int32), ("behave_weight_threshold", c_float)]


PROX_GIVEN, PROX_IS_OLD, PROX_FROM_VERSIONS = 0, 1, 2
DPPO_OUT_NAMES = {
    "loss": 0, "actor/policy_loss": 1, "actor/proximal_ratio": 2, "actor/clipped_proximal_ratio": 3,
    "actor/dual_clip_fraction": 4, "actor/behav_clip_fraction": 5, "actor/proximal_approx_kl": 6,
    "actor/behav_approx_kl": 7, "actor/clip_fraction": 8, "critic/value_loss": 9, "critic/value_clip_ratio": 10,
    "ev/count": 11, "ev/returns_sum": 12, "ev/returns_sq_sum": 13, "ev/errors_sum": 14, "ev/errors_sq_sum": 15,
    "mask_count": 18, "actor/average_version": 19,
}


class TokenRows(Structure):  # include/rlx.h: rlx_token_rows
    _fields_ = [("n_tokens", c_int64), ("vocab", c_int32), ("dtype", c_int32), ("rows_per_seq", c_int64),
                ("seq_stride", c_int64), ("row_stride", c_int64), ("temperature", c_float), ("round_outputs", c_int32)]


class TokenLossParams(Structure):  # include/rlx.h: rlx_token_loss_params
    _fields_ = [("ppo", PpoLossParams), ("loss_agg", c_int32), ("fast_path_zero_loss_mask", c_int32),
                ("kl_type", c_int32), ("kl_beta", c_float), ("use_entropy", c_int32), ("entropy_bonus", c_float)]


DTYPE_F32, DTYPE_BF16, DTYPE_F16, DTYPE_RAW8, DTYPE_RAW16, DTYPE_RAW32, DTYPE_RAW64 = range(7)
LOSS_AGG = {"token-mean": 0, "seq-mean-token-sum": 1, "seq-mean-token-mean": 2}
KL_TYPE = {None: 0, "kl": 1, "k1": 1, "abs": 2, "mse": 3, "k2": 3, "low_var_kl": 4, "k3": 4}
TOK_OUT_FLOATS = 16
TOK_OUT_NAMES = {"loss": 0, "actor/policy_loss": 1, "actor/policy_loss_abs": 2, "actor/ratio": 3, "actor/ratio_abs": 4,
                 "actor/clipped_ratio": 5, "actor/dual_cliped_ratio": 6, "actor/approx_kl": 7, "actor/clip_fraction": 8,
                 "actor/entropy_loss": 9, "actor/kl_loss": 10, "actor/token_num": 11, "policy_on": 12}

# name -> (restype, argtypes); kept in one table so tests can check it against include/rlx.h
PROTOTYPES = {
    "rlx_version": (c_int, []),
    "rlx_dev_variants": (c_int, []),
    "rlx_last_error": (c_char_p, []),
    "rlx_abi_struct_sizes": (c_int, [c_void_p, c_int]),
    "rlx_device_info": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "rlx_done_prefix_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rlx_gae_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rlx_gae_scan": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                             c_int, c_int, c_int, POINTER(GaeParams), c_void_p]),
    "rlx_standardize_workspace_bytes": (c_size_t, [c_size_t]),
    "rlx_masked_standardize": (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_void_p, c_size_t, c_void_p]),
    "rlx_masked_stats_workspace_bytes": (c_size_t, [c_int64]),
    "rlx_masked_stats": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "rlx_normalize_from_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rlx_grpo_group_adv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_float, c_void_p]),
    "rlx_grpo_from_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rlx_episode_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rlx_gaussian_entropy_bonus": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_int, c_float, c_void_p]),
    "rlx_gaussian_entropy_bonus_deferred": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_int, c_float, c_void_p,
                                                    c_void_p]),
    "rlx_reward_filter_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "rlx_ppo_loss_workspace_bytes": (c_size_t, [c_int64]),
    "rlx_ppo_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 POINTER(PpoLossParams), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rlx_ppo_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                 c_void_p]),
    "rlx_mlp_packed_bytes": (c_size_t, [POINTER(MlpLayout)]),
    "rlx_mlp_pack": (c_int, [c_void_p, POINTER(MlpLayout), c_void_p, c_void_p]),
    "rlx_mlp_rollout": (c_int, [c_void_p, c_void_p, POINTER(MlpLayout), c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "rlx_mlp_value": (c_int, [c_void_p, c_void_p, POINTER(MlpLayout), c_void_p, c_int64, c_void_p, c_void_p]),
    "rlx_mlp_train_fwd": (c_int, [c_void_p, c_void_p, POINTER(MlpLayout), c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlx_mlp_bwd_slabs": (c_int, [c_int64]),
    "rlx_mlp_bwd_workspace_bytes": (c_size_t, [POINTER(MlpLayout), c_int64]),
    "rlx_mlp_train_bwd": (c_int, [c_void_p, c_void_p, POINTER(MlpLayout), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "rlx_gather_rows": (c_int, [POINTER(GatherField), c_int, c_void_p, c_int64, c_void_p]),
    "rlx_adamw_workspace_bytes": (c_size_t, [c_int64]),
    "rlx_adamw_sync_words": (c_size_t, [c_int64]),
    "rlx_sum_slabs": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "rlx_sum_slabs_deferred": (c_int, [c_void_p, c_int64, c_int, c_void_p, POINTER(AdamwParams), c_void_p]),
    "rlx_clip_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, POINTER(AdamwParams), c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "rlx_bootstrap_rewards": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "rlx_store_env_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rlx_mlp_tiles_bytes": (c_size_t, [POINTER(MlpLayout)]),
    "rlx_mlp_tiles_bytes_for": (c_size_t, [POINTER(MlpLayout), c_int32]),
    "rlx_mlp_pack_tiles": (c_int, [c_void_p, POINTER(MlpLayout), c_void_p, c_void_p]),
    "rlx_mlp_pack_tiles_bf16": (c_int, [c_void_p, POINTER(MlpLayout), c_void_p, c_void_p]),
    "rlx_mlp_rollout_step": (c_int, [POINTER(RolloutStep), c_void_p]),
    "rlx_ppo_step_slabs": (c_int, [POINTER(MlpLayout), c_int64]),
    "rlx_ppo_step_slabs_for": (c_int, [POINTER(MlpLayout), c_int64, c_int32]),
    "rlx_ppo_step_workspace_bytes": (c_size_t, [POINTER(MlpLayout), c_int64]),
    "rlx_ppo_step": (c_int, [POINTER(PpoStepArgs), c_void_p]),
    "rlx_decoupled_loss_workspace_bytes": (c_size_t, [c_int64]),
    "rlx_decoupled_loss_fwd": (c_int, [c_void_p] * 10 + [c_int64, POINTER(DecoupledLossParams), c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "rlx_token_logprob_fwd": (c_int, [c_void_p, c_void_p, POINTER(TokenRows), c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlx_token_logprob_fwd_packed": (c_int, [c_void_p, c_void_p, POINTER(TokenRows), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "rlx_token_logprob_bwd": (c_int, [c_void_p, c_void_p, POINTER(TokenRows), c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_int64, c_void_p]),
    "rlx_token_logprob_bwd_packed": (c_int, [c_void_p, c_void_p, POINTER(TokenRows), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "rlx_token_loss_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "rlx_token_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                   POINTER(TokenLossParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "rlx_token_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "rlx_categorical_sample": (c_int, [c_void_p, POINTER(TokenRows), c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "rlx_patch_workspace_bytes": (c_size_t, [c_int64]),
    "rlx_patch_scan": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "rlx_patch_emit": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlx_patch_apply_workspace_bytes": (c_size_t, [c_int64]),
    "rlx_patch_apply": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int64,
                                c_void_p, c_size_t, c_void_p]),
    "rlx_zplane_bound_bytes": (c_size_t, [c_int64, c_int]),
    "rlx_zplane_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "rlx_zplane_compress": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rlx_zplane_parse_header": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int), POINTER(ctypes.c_uint64)]),
    "rlx_zplane_decompress": (c_int, [c_void_p, c_size_t, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "rlx_copy_segments_plan": (c_int, [POINTER(CopySegment), c_int32, POINTER(c_int64)]),
    "rlx_copy_segments": (c_int, [c_void_p, c_int32, c_int64, c_void_p]),
    "rlx_reinpp_workspace_bytes": (c_size_t, [c_int64]),
    "rlx_reinpp_seq_adv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_int64, c_int64, c_void_p,
                                   c_size_t, c_void_p]),
    "rlx_masked_normalize": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_double, c_void_p, c_int64, c_void_p]),
    "rlx_gae_seq_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "rlx_gae_seq": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_void_p, c_size_t,
                            c_void_p]),
    "rlx_grpo_seq_adv": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_float, c_void_p]),
    "rlx_rollout_metrics_workspace_bytes": (c_size_t, []),
    "rlx_rollout_metrics": (c_int, [POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p, c_int64, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "rlx_xgmi_create": (c_int, [c_int, c_int, c_int64, c_int, c_int, POINTER(c_void_p), c_void_p]),
    "rlx_xgmi_connect": (c_int, [c_void_p, c_void_p]),
    "rlx_xgmi_connect_local": (c_int, [POINTER(c_void_p), c_int]),
    "rlx_xgmi_connect_self": (c_int, [c_void_p]),
    "rlx_xgmi_self_timing": (c_int, [c_void_p, c_int]),
    "rlx_xgmi_configure": (c_int, [c_void_p, c_int, c_int, c_int]),
    "rlx_xgmi_destroy": (c_int, [c_void_p]),
    "rlx_xgmi_status": (c_int, [c_void_p]),
    "rlx_xgmi_status_snapshot": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rlx_xgmi_allreduce_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_float, c_void_p, c_size_t, c_void_p]),
    "rlx_xgmi_clip_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         POINTER(AdamwParams), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}
XGMI_HANDLE_BYTES, XGMI_MAX_RANKS = 64, 8

_lib = None


def load() -> ctypes.CDLL:
    """Load librlx_hip.so and bind every prototype.  Raises RlxError loudly when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RlxError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m rlinf_amd.csrc.build` "
            "(needs hipcc); rlinf_amd has no CPU fallback by design.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise RlxError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RlxError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.rlx_version() < 112:
        raise RlxError("librlx_hip.so is older than this Python package; rebuild it")
    mirrors = (GaeParams, PpoLossParams, GatherField, AdamwGroup, AdamwParams, MlpLayout, ValueJob, RolloutStep, PpoStepArgs,
               DecoupledLossParams, TokenRows, TokenLossParams, CopySegment)  # the order rlx_abi_struct_sizes documents
    sizes = (c_size_t * len(mirrors))()
    if lib.rlx_abi_struct_sizes(ctypes.cast(sizes, c_void_p), len(mirrors)) != len(mirrors):
        raise RlxError("librlx_hip.so and rlinf_amd/_lib.py disagree on the number of argument structs; rebuild the library")
    for cls, size in zip(mirrors, sizes):
        if ctypes.sizeof(cls) != size:
            raise RlxError(f"{cls.__name__}: the ctypes mirror is {ctypes.sizeof(cls)} bytes, the library's struct {size}; "
                           "rebuild librlx_hip.so (python -m rlinf_amd.csrc.build)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().rlx_last_error()
        raise RlxError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")

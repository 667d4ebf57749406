"""Process hook for a real RLinf install: ``RLINF_EXT_MODULE=rlinf_amd.ext``.

RLinf imports the module named by that environment variable in every worker process and calls its
``register()`` once (rlinf/scheduler/cluster/utils.py:81-110, invoked from Worker._env_setup_before_init,
rlinf/scheduler/worker/worker.py:348-394).  ``register()`` re-registers the hot-path names in RLinf's own
registries -- the last registration wins (rlinf/algorithms/registry.py:33-53) -- and the ``mlp_policy`` builder in its model
registry (rlinf/models/__init__.py:31-53, ``force=True``), so the unmodified learner
(rlinf/workers/actor/embodied_fsdp_actor_worker.py:117-127,286-321,591-700) and rollout worker
(rlinf/workers/rollout/hf/huggingface_worker.py:137-151,500-530) call the HIP kernels.

Because RLinf's ``policy_loss`` runs preprocess_loss_inputs BEFORE the registered callee and ``.item()``s
every metric after it, the callees registered here take the already-shaped tensors (sub_per_adv == raw
== 1 per element) and return plain tensors, exactly what that caller expects.
"""

from __future__ import annotations


def register() -> None:
    import torch
    from rlinf.algorithms import registry as ref_registry  # the REAL package; ImportError is logged by RLinf

    from rlinf_amd import ops
    from rlinf_amd._lib import DPPO_OUT_NAMES, PPO_OUT_NAMES, TOK_OUT_NAMES
    from rlinf_amd.algorithms import advantages as adv
    from rlinf_amd.algorithms import utils as u
    from rlinf_amd.algorithms.losses import (_ACTOR_KEYS, _CRITIC_KEYS, _DECOUPLED_KEYS, _EV_MAP, _TOKEN_KEYS_ZERO,
                                             _token_fused)

    ref_registry.register_advantage("gae")(adv.compute_gae_advantages_and_returns)
    ref_registry.register_advantage("grpo")(adv.compute_grpo_advantages)
    ref_registry.register_advantage("reinpp")(adv.compute_reinpp_advantages)  # reasoning: [1, B] rewards, [L, B] tensors

    def _reasoning_actor(kw):
        """task_type='reasoning': [bsz, seq] token tensors and the learner's loss_agg_func (fsdp_actor_worker.py:736-750)."""
        loss, m = _token_fused(kw)
        out = m.device_vector
        on = bool(out[TOK_OUT_NAMES["policy_on"]].item())  # the reference synchronises on the same check (losses.py:206)
        keys = _ACTOR_KEYS if on else _TOKEN_KEYS_ZERO
        zero = out.new_zeros(())
        return loss, {k: (out[TOK_OUT_NAMES[k]] if (on and k in TOK_OUT_NAMES) else zero) for k in keys}

    def _shaped_loss(has_critic):
        def fn(logprobs, old_logprobs, advantages, clip_ratio_low, clip_ratio_high, **kw):
            if not has_critic and kw.get("task_type", "embodied") != "embodied":
                return _reasoning_actor(dict(kw, logprobs=logprobs, old_logprobs=old_logprobs, advantages=advantages,
                                             clip_ratio_low=clip_ratio_low, clip_ratio_high=clip_ratio_high))
            dev = u.compute_device(logprobs)
            per_adv = logprobs.numel() // max(advantages.numel(), 1)  # token_level keeps the action dim
            st = lambda t: None if t is None else u.stage(t, dev).contiguous()  # noqa: E731
            loss, out = ops.ppo_loss(
                st(logprobs).reshape(advantages.numel(), per_adv), st(old_logprobs).reshape(advantages.numel(), per_adv),
                st(advantages), logprob_type="token_level" if per_adv > 1 else "action_level", action_dim=per_adv,
                clip_ratio_low=clip_ratio_low, clip_ratio_high=clip_ratio_high,
                values=st(kw.get("values")) if has_critic else None,
                prev_values=st(kw.get("prev_values")) if has_critic else None,
                returns=st(kw.get("returns")) if has_critic else None, value_clip=kw.get("value_clip"),
                huber_delta=kw.get("huber_delta"), loss_mask=st(kw.get("loss_mask")),
                loss_mask_sum=st(kw.get("loss_mask_sum")), max_episode_steps=kw.get("max_episode_steps"),
                clip_ratio_c=kw.get("clip_ratio_c"), clip_log_ratio_min=kw.get("clip_log_ratio_min"),
                clip_log_ratio_max=kw.get("clip_log_ratio_max"), critic_warmup=bool(kw.get("critic_warmup", False)),
                has_critic=has_critic)
            keys = list(_ACTOR_KEYS) + (list(_CRITIC_KEYS) if has_critic else [])
            metrics = {k: out[PPO_OUT_NAMES[k]] for k in keys}
            if has_critic:
                metrics.update({v: out[PPO_OUT_NAMES[k]] for k, v in _EV_MAP.items()})
            return loss, metrics

        return fn

    def _shaped_decoupled(logprobs, old_logprobs, advantages, clip_ratio_low, clip_ratio_high, **kw):
        """decoupled_actor_critic on the tensors preprocess_loss_inputs has already shaped (losses.py:27-167,383-393)."""
        dev = u.compute_device(logprobs)
        n_adv = advantages.numel()
        per_adv = logprobs.numel() // max(n_adv, 1)
        st = lambda t: None if t is None else u.stage(t, dev).contiguous()  # noqa: E731
        rs = lambda t: None if t is None else st(t).reshape(n_adv, per_adv)  # noqa: E731
        versions, cur = kw.get("versions"), kw.get("current_version")
        if versions is not None and versions.numel() != logprobs.numel():
            versions = versions.expand(logprobs.shape)
        loss, out = ops.ppo_loss(
            rs(logprobs), rs(old_logprobs), st(advantages), logprob_type="token_level" if per_adv > 1 else "action_level",
            action_dim=per_adv, clip_ratio_low=clip_ratio_low, clip_ratio_high=clip_ratio_high, values=st(kw.get("values")),
            prev_values=st(kw.get("prev_values")), returns=st(kw.get("returns")), value_clip=kw.get("value_clip"),
            huber_delta=kw.get("huber_delta"), loss_mask=st(kw.get("loss_mask")), loss_mask_sum=st(kw.get("loss_mask_sum")),
            max_episode_steps=kw.get("max_episode_steps"), clip_ratio_c=kw.get("clip_ratio_c"),
            critic_warmup=bool(kw.get("critic_warmup", False)), has_critic=True,
            decoupled=dict(proximal_logprobs=rs(kw.get("proximal_logprobs")), versions=rs(versions), current_version=cur,
                           behave_weight_threshold=kw.get("behave_weight_threshold")))
        metrics = {k: out[DPPO_OUT_NAMES[k]] for k in list(_DECOUPLED_KEYS) + list(_CRITIC_KEYS)}
        metrics.update({v: out[DPPO_OUT_NAMES[k]] for k, v in _EV_MAP.items()})
        lm = kw.get("loss_mask")
        if versions is not None and cur is not None and (lm is None or lm.shape == logprobs.shape) \
                and float(out[DPPO_OUT_NAMES["mask_count"]] if lm is not None else 1.0) > 0:
            metrics["actor/average_version"] = out[DPPO_OUT_NAMES["actor/average_version"]]
            metrics["actor/current_version"] = out.new_tensor(float(cur))
        return loss, metrics

    # the model leg (rlinf/models/__init__.py:31-53): cfg.actor.model.model_type == "mlp_policy" now builds the HIP module in
    # the learner (embodied_fsdp_actor_worker.py:117-127) and in the rollout worker (huggingface_worker.py:137-151)
    from rlinf import models as ref_models

    from rlinf_amd.models.embodiment.mlp_policy_module import build_reference_named_mlp_policy

    ref_models.register_model("mlp_policy", build_reference_named_mlp_policy, category="embodied", force=True)

    ref_registry.register_policy_loss("actor_critic")(_shaped_loss(True))
    ref_registry.register_policy_loss("actor")(_shaped_loss(False))
    ref_registry.register_policy_loss("decoupled_actor_critic")(_shaped_decoupled)
    if not torch.cuda.is_available():
        raise RuntimeError("rlinf_amd.ext registered HIP kernels but no HIP device is visible")

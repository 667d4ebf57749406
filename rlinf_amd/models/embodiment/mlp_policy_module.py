"""The HIP MLP policy as the module a REAL RLinf install gets from its own model registry.

``rlinf_amd.ext.register()`` hands ``rlinf.models.register_model("mlp_policy", builder, category="embodied", force=True)``
(rlinf/models/__init__.py:31-53) a builder that returns this class.  The reference's learner and rollout worker then do to it
what they do to their own ``MLPPolicy`` (rlinf/models/embodiment/mlp_policy/mlp_policy.py):

* ``FSDPModelManager.setup_model_and_optimizer`` walks ``named_parameters()`` and sorts names containing ``value_head`` into the
  critic learning-rate group (rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:274-306,501-590), torch's AdamW and
  ``clip_grad_norm_`` update the parameters in place, ``state_dict()`` feeds checkpoints and the weight syncers -- so the
  parameters here are ordinary ``nn.Parameter``s under the reference's names, shapes and ``named_parameters()`` order
  (``actor_logstd``, ``value_head.mlp.{0,2,4,6}.*``, ``backbone.{0,2,4}.*``, ``actor_mean.*``), NOT the flat buffer the
  fused learner of this package owns;
* ``train_micro_batch`` calls ``model(forward_inputs=..., compute_logprobs=..., compute_entropy=..., compute_values=...,
  use_cache=False)`` and back-propagates ``policy_loss`` through the result (embodied_fsdp_actor_worker.py:626-700): forward
  and backward are the HIP launches ``rlx_mlp_train_fwd`` / ``rlx_mlp_train_bwd`` behind one autograd node whose inputs are
  the named parameters;
* the rollout worker calls ``predict_action_batch(env_obs=..., mode=..., return_obs=...)`` under ``no_grad``
  (workers/rollout/hf/huggingface_worker.py:500-530), ``load_state_dict`` / the weight syncer's ``apply`` and then
  ``set_global_step`` (:629-675): the step is the one-launch MFMA kernel ``rlx_mlp_rollout_step`` on a weight image that is
  re-packed whenever a parameter's storage or version changed.

Whatever the host framework did to the parameters (moved them, re-flattened them, wrote through ``.data``), the kernels read a
private flat f32 copy gathered from them with one multi-tensor copy, so nothing here depends on aliasing.
"""

from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from .mlp_policy import HIDDEN, MLPPolicy


class _NamedParamsTrainFn(torch.autograd.Function):
    """default_forward as ONE autograd node over the named parameters (mlp_policy.py:202-236)."""

    @staticmethod
    def forward(ctx, module, states, action, *params):
        core = module._refresh(params)
        logprob, entropy, value, mean, acts = ops.mlp_train_fwd(core.flat.data, core.packed(), core.layout, states, action)
        ctx.module, ctx.key = module, module._key
        ctx.save_for_backward(states, action, mean, acts, *params)
        return logprob, entropy, value

    @staticmethod
    def backward(ctx, d_logprob, d_entropy, d_value):
        states, action, mean, acts, *params = ctx.saved_tensors
        module = ctx.module
        core = module._refresh(params) if module._key != ctx.key else module._core  # another forward ran in between
        lay, M, dev = core.layout, states.shape[0], states.device
        zeros = lambda t, w: torch.zeros((M, w), dtype=torch.float32, device=dev) if t is None else t.contiguous()  # noqa: E731
        slabs = ops.mlp_train_bwd(core.flat.data, core.packed(), lay, states, action, mean, acts, zeros(d_logprob, lay.act_dim),
                                  None if d_entropy is None else d_entropy.contiguous(), zeros(d_value, lay.val_dim))
        flat_grad = ops.sum_slabs(slabs)
        grads = []
        for (name, shp), need in zip(core.shapes.items(), ctx.needs_input_grad[3:]):
            o = core.offsets[name]
            grads.append(flat_grad[o:o + math.prod(shp)].view(shp) if need else None)
        return (None, None, None, *grads)


class ReferenceNamedMLPPolicy(nn.Module):
    """Constructor arguments, parameter names, initialisation (same RNG stream), method names and return structures of the
    reference's MLPPolicy; PPO / GRPO configurations (with or without a value head, no Q head)."""

    def __init__(self, obs_dim, action_dim, num_action_chunks, add_value_head, add_q_head, q_head_type="default",
                 value_granularity="action_level", critic_obs_dim=None, compute_dtype=torch.float32):
        super().__init__()
        # the kernels' private flat copy + layout + weight images; built first so that its initialisers draw from the RNG in
        # the reference's order (MLPPolicy.reset_parameters); NOT a submodule: its buffer must not show up as a parameter
        core = MLPPolicy(obs_dim, action_dim, num_action_chunks, add_value_head, add_q_head, q_head_type, value_granularity,
                         critic_obs_dim, compute_dtype=compute_dtype)
        core.flat.requires_grad_(False)
        object.__setattr__(self, "_core", core)
        self.obs_dim, self.action_dim, self.num_action_chunks = core.obs_dim, core.action_dim, core.num_action_chunks
        self.critic_obs_dim = critic_obs_dim or obs_dim
        self.value_granularity, self.compute_dtype = value_granularity, compute_dtype
        self.independent_std, self.final_tanh, self.action_scale = True, False, None
        self.torch_compile_enabled, self.cuda_graph_manager = False, None
        act = self.num_action_chunks * self.action_dim
        with torch.random.fork_rng(devices=[]):  # the holders' default initialisers must not advance the caller's RNG
            if add_value_head:
                self.value_head = nn.Module()
                self.value_head.mlp = nn.Sequential(nn.Linear(obs_dim, HIDDEN), nn.Tanh(), nn.Linear(HIDDEN, HIDDEN), nn.Tanh(),
                                                    nn.Linear(HIDDEN, HIDDEN), nn.Tanh(),
                                                    nn.Linear(HIDDEN, core.value_dim, bias=False))
            self.backbone = nn.Sequential(nn.Linear(obs_dim, HIDDEN), nn.Tanh(), nn.Linear(HIDDEN, HIDDEN), nn.Tanh(),
                                          nn.Linear(HIDDEN, HIDDEN), nn.Tanh())
            self.actor_mean = nn.Linear(HIDDEN, act)
        self.actor_logstd = nn.Parameter(torch.empty(1, act))
        names = [n for n, _ in self.named_parameters()]
        if names != list(core.shapes):
            raise RuntimeError(f"parameter names {names} differ from the reference's {list(core.shapes)}")
        with torch.no_grad():
            for name, p in self.named_parameters():
                p.copy_(core.view(name))
        self._key = None

    # ---- the kernels' view of the parameters ----------------------------------------------------------------------
    def _named(self):
        return [p for _, p in self.named_parameters()]

    @staticmethod
    def _local(p: torch.Tensor) -> torch.Tensor:
        t = p.detach()
        if hasattr(t, "full_tensor"):  # FSDP2 hands out DTensors; the kernels need this rank's whole copy
            t = t.full_tensor()
        return t

    def _refresh(self, params=None) -> MLPPolicy:
        """Gather the named parameters into the kernels' flat buffer when any of them moved or changed since the last launch."""
        core = self._core
        params = self._named() if params is None else list(params)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self._key or self._dirty:
            src = [self._local(p).to(torch.float32) for p in params]
            if src[0].device != core.flat.device:
                core._apply(lambda t: t.to(src[0].device))
            torch._foreach_copy_([core.view(name) for name in core.shapes], src)
            core.mark_updated()
            self._key, self._dirty = key, False
        return core

    _dirty = True

    def mark_updated(self):
        """Parameters were written through a path that bumps no version counter (``p.data`` views): re-gather next launch."""
        self._dirty = True

    def set_global_step(self, global_step):
        """The reference calls this after every weight apply (huggingface_worker.py:670-672) and before every training step
        (embodied_fsdp_actor_worker.py:707-708): the natural place to invalidate the weight image."""
        self.global_step = global_step
        self.mark_updated()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self.mark_updated()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        dev = self.actor_logstd.device  # the kernels' copy follows the DEVICE only: it stays f32 whatever dtype cast ``fn`` applies
        if self._core.flat.device != dev:
            self._core._apply(lambda t: t.to(dev))
        self.mark_updated()
        return out

    # ---- reference API ----------------------------------------------------------------------------------------------
    def preprocess_env_obs(self, env_obs):
        return {"states": env_obs["states"].to(self.actor_logstd.device)}  # mlp_policy.py:122-124

    def forward(self, forward_type=None, **kwargs):
        kind = getattr(forward_type, "value", forward_type)
        if kind in (None, "default"):
            return self.default_forward(**kwargs)
        raise NotImplementedError  # base_policy.py:52-56; the SAC / CrossQ / IQL forwards belong to the Q-head variants

    def default_forward(self, forward_inputs, compute_logprobs=True, compute_entropy=True, compute_values=True, **kwargs):
        dev = self.actor_logstd.device
        states = forward_inputs["states"].to(dev, torch.float32).contiguous()
        action = forward_inputs["action"].to(dev, torch.float32).contiguous().reshape(states.shape[0], -1)
        if compute_values and not self._core.has_value_head:
            raise NotImplementedError  # mlp_policy.py:230-235
        logprob, entropy, value = _NamedParamsTrainFn.apply(self, states, action, *self._named())
        out = {}
        if compute_logprobs:
            out["logprobs"] = logprob
        if compute_entropy:
            out["entropy"] = entropy
        if compute_values:
            out["values"] = value
        return out

    @torch.no_grad()
    def predict_action_batch(self, env_obs, calculate_logprobs=True, calculate_values=True, return_obs=True, mode="train",
                             eps: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None, **kwargs):
        return self._refresh().predict_action_batch(env_obs, calculate_logprobs, calculate_values, return_obs, mode, eps=eps,
                                                    generator=generator, **kwargs)

    # The reference's rollout worker turns these on from its config (huggingface_worker.py:179-189,879-897).  A rollout step is
    # ONE hand-written launch already: there is nothing for a tracing compiler or a graph capture to remove.
    def enable_torch_compile(self, mode: str = "max-autotune-no-cudagraphs"):
        self.torch_compile_enabled = True

    def capture_cuda_graph(self, train_batch_size: int, eval_batch_size: int):
        return None

    def release_cuda_graph(self):
        return None

    def is_cuda_graph_enabled(self) -> bool:
        return False

    def reference_state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.state_dict().items())


def build_reference_named_mlp_policy(cfg, torch_dtype):
    """``ModelBuilder(cfg, torch_dtype) -> nn.Module`` for rlinf.models.register_model; reads the keys the reference's own
    builder reads (rlinf/models/embodiment/mlp_policy/__init__.py:19-75).  The Q-head (SAC / CrossQ) and IQL variants are not
    on the PPO / GRPO path this package replaces: for those the reference's own builder is called, unchanged."""
    get = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
    if get("add_q_head", False) or get("iql_config", None) is not None:
        from rlinf.models.embodiment.mlp_policy import get_model as reference_builder

        return reference_builder(cfg, torch_dtype)
    return ReferenceNamedMLPPolicy(
        obs_dim=int(get("obs_dim")), action_dim=int(get("action_dim")), num_action_chunks=int(get("num_action_chunks", 1)),
        add_value_head=bool(get("add_value_head", True)), add_q_head=False, q_head_type=get("q_head_type", "default"),
        value_granularity=get("value_granularity", "action_level"), critic_obs_dim=get("critic_obs_dim", None),
        compute_dtype=torch.bfloat16 if torch_dtype == torch.bfloat16 else torch.float32)

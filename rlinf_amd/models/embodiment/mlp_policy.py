"""MLPPolicy on HIP kernels -- same constructor, method names and return structures as the reference's
rlinf/models/embodiment/mlp_policy/mlp_policy.py (the PPO / GRPO configurations: with or without a value head, no Q head).

All parameters live in ONE flat f32 buffer (``self.flat``) laid out in the reference's
``named_parameters()`` order, so (a) ``state_dict`` round-trips with the reference's names and shapes,
(b) the optimizer kernel walks one buffer, (c) data-parallel gradient all-reduce is a single RCCL call.
Forward passes are the fused kernels of csrc/mlp_policy.hip; ``default_forward`` is an autograd node so
``loss.backward()`` works like upstream, and ``PolicyTrainStep`` (workers/actor) bypasses autograd for
the hot loop.
"""

from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from ..._lib import MlpLayout, RlxError

HIDDEN = 256


def _reference_shapes(obs_dim: int, action_dim: int, num_action_chunks: int, value_dim: int):
    act = num_action_chunks * action_dim
    return OrderedDict([  # construction order of the reference: own params, value head, backbone, actor_mean
        ("actor_logstd", (1, act)),
        ("value_head.mlp.0.weight", (HIDDEN, obs_dim)), ("value_head.mlp.0.bias", (HIDDEN,)),
        ("value_head.mlp.2.weight", (HIDDEN, HIDDEN)), ("value_head.mlp.2.bias", (HIDDEN,)),
        ("value_head.mlp.4.weight", (HIDDEN, HIDDEN)), ("value_head.mlp.4.bias", (HIDDEN,)),
        ("value_head.mlp.6.weight", (value_dim, HIDDEN)),
        ("backbone.0.weight", (HIDDEN, obs_dim)), ("backbone.0.bias", (HIDDEN,)),
        ("backbone.2.weight", (HIDDEN, HIDDEN)), ("backbone.2.bias", (HIDDEN,)),
        ("backbone.4.weight", (HIDDEN, HIDDEN)), ("backbone.4.bias", (HIDDEN,)),
        ("actor_mean.weight", (act, HIDDEN)), ("actor_mean.bias", (act,)),
    ])


class _MlpTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, policy, states, action):
        logprob, entropy, value, mean, acts = ops.mlp_train_fwd(flat, policy.packed(), policy.layout, states, action)
        ctx.policy = policy
        ctx.save_for_backward(flat, states, action, mean, acts)
        return logprob, entropy, value

    @staticmethod
    def backward(ctx, d_logprob, d_entropy, d_value):
        flat, states, action, mean, acts = ctx.saved_tensors
        pol = ctx.policy
        M = states.shape[0]
        zl = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=flat.device) if t is None else t  # noqa: E731
        grads = ops.mlp_train_bwd(flat, pol.packed(), pol.layout, states, action, mean, acts,
                                  zl(d_logprob, (M, pol.layout.act_dim)).contiguous(),
                                  None if d_entropy is None else d_entropy.contiguous(),
                                  zl(d_value, (M, pol.layout.val_dim)).contiguous())
        return ops.sum_slabs(grads), None, None, None


class MLPPolicy(nn.Module):
    def __init__(self, obs_dim, action_dim, num_action_chunks, add_value_head, add_q_head, q_head_type="default",
                 value_granularity="action_level", critic_obs_dim=None, compute_dtype=torch.float32):
        super().__init__()
        # operand precision of the 256-wide dense layers on the fused launches: float32 (exact-f32 MFMA) or bfloat16
        # (bf16 MFMA operands, f32 accumulate; master weights and everything else stay float32) -- the reference's
        # ``precision`` / amp_autocast switch
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"compute_dtype must be float32 or bfloat16, got {compute_dtype}")
        self.compute_dtype = compute_dtype
        if add_q_head:
            raise NotImplementedError("the Q-head (SAC) variant of the MLP policy is outside the PPO / GRPO hot path")
        # add_value_head False (mlp_policy.py:42-66: the value-free policy of actor-only PPO / GRPO configurations): the reference
        # builds no value head, predict_action_batch returns zeros as prev_values (:283-286) and default_forward(compute_values=True)
        # raises (:230-235).  The fused launches are built around the (value net, policy net) pair, so the flat buffer keeps a
        # PHANTOM value net behind the real parameters: all zeros (its outputs are exactly 0: tanh(0) = 0 through every layer),
        # absent from shapes / state_dict / named_views / the optimizer's groups, never trained (an actor-only loss gives it zero
        # gradients and AdamW holds no range over it).
        self.has_value_head = bool(add_value_head)
        self.obs_dim, self.action_dim, self.num_action_chunks = int(obs_dim), int(action_dim), int(num_action_chunks)
        self.value_granularity = value_granularity
        self.value_dim = 1 if value_granularity == "chunk_level" else self.num_action_chunks
        self.independent_std, self.final_tanh, self.action_scale = True, False, None
        full = _reference_shapes(self.obs_dim, self.action_dim, self.num_action_chunks, self.value_dim)
        self.shapes = OrderedDict((k, v) for k, v in full.items() if self.has_value_head or not k.startswith("value_head."))
        # Every tensor starts on a 16-byte boundary of the flat buffer (the kernels stream the weight matrices with 16-byte
        # accesses): an odd action_dim -- the reference's LIBERO MLP configurations use 7 -- leaves up to three unowned floats
        # behind actor_logstd / actor_mean.bias.  Unowned elements are zero, get zero gradients (the learner zero-fills its
        # slabs once; no kernel writes them) and therefore stay zero under AdamW; state_dict() exposes the owned views only.
        self.offsets, off = OrderedDict(), 0
        for name, shp in self.shapes.items():
            off = (off + 3) // 4 * 4
            self.offsets[name] = off
            off += math.prod(shp)
        self.n_exposed = off  # END of the last exposed tensor (the span; == exposed_numel when every size is a multiple of 4)
        self.exposed_numel = sum(math.prod(shp) for shp in self.shapes.values())  # what the reference's named_parameters() hold
        self._phantom_offsets = OrderedDict()
        for name, shp in full.items():
            if name not in self.shapes:
                off = (off + 3) // 4 * 4  # (the kernels want 16-byte aligned weight matrices)
                self._phantom_offsets[name] = off
                off += math.prod(shp)
        self.n_params = off if self.has_value_head else (off + 3) // 4 * 4  # (whole float4s: the slab kernels' vector paths)
        self.flat = nn.Parameter(torch.empty(self.n_params, dtype=torch.float32))
        self.layout = self._make_layout()
        self._packed = None
        self._packed_version = -1
        self.reset_parameters()

    # ---- layout / parameter plumbing ---------------------------------------------------------------------
    def _make_layout(self) -> MlpLayout:
        lay = MlpLayout()
        lay.obs_dim, lay.act_dim = self.obs_dim, self.num_action_chunks * self.action_dim
        lay.val_dim, lay.hidden, lay.n_params = self.value_dim, HIDDEN, self.n_params
        lay.off_logstd = self.offsets["actor_logstd"]
        nets = (("value_head.mlp.0", "value_head.mlp.2", "value_head.mlp.4", "value_head.mlp.6"),
                ("backbone.0", "backbone.2", "backbone.4", "actor_mean"))
        where = {**self.offsets, **self._phantom_offsets}
        for y, names in enumerate(nets):
            for l, nm in enumerate(names):
                lay.off_w[y][l] = where[nm + ".weight"]
                lay.off_b[y][l] = where.get(nm + ".bias", -1)
        return lay

    def view(self, name: str) -> torch.Tensor:
        o = self.offsets[name]
        shp = self.shapes[name]
        return self.flat.data[o:o + math.prod(shp)].view(shp)

    def group_ranges(self, lr: float, value_lr: float, train_value_head: bool = True):
        """Contiguous (begin, end, lr) ranges: names containing 'value_head' use value_lr
        (rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:533-560).  ``train_value_head=False`` leaves the value head out:
        under an actor-only loss the reference never computes values (embodied_fsdp_actor_worker.py:620-630), their
        gradients stay None and torch's AdamW does not touch such parameters -- no decay, no moments."""
        out = []
        for name, shp in self.shapes.items():
            if not train_value_head and "value_head" in name:
                continue
            b = self.offsets[name]
            e = b + math.prod(shp)
            r = value_lr if "value_head" in name else lr
            if out and out[-1][2] == r and 0 <= b - out[-1][1] < 4:  # adjacent, or across an alignment gap (zeros: see __init__)
                out[-1] = (out[-1][0], e, r)
            else:
                out.append((b, e, r))
        return out

    @torch.no_grad()
    def reset_parameters(self):
        """Same initialisers, in the same construction order (hence the same RNG stream), as the reference:
        ValueHead kaiming-normal(fan_out, tanh)/N(0, 0.02) (value_head.py:52-64), orthogonal(sqrt 2) backbone,
        orthogonal(0.01 sqrt 2) actor_mean, logstd = -0.5 (mlp_policy.py:91-105, modules/utils.py:20-23)."""
        D, act = self.obs_dim, self.num_action_chunks * self.action_dim
        self.flat.data.zero_()  # (the phantom value net of a value-free policy stays zero)
        vh = [nn.Linear(D, HIDDEN), nn.Linear(HIDDEN, HIDDEN), nn.Linear(HIDDEN, HIDDEN),
              nn.Linear(HIDDEN, self.value_dim, bias=False)] if self.has_value_head else []  # (not built: no RNG draws either)
        for i, m in enumerate(vh):
            if i == 3:
                nn.init.normal_(m.weight, mean=0.0, std=0.02)
            else:
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="tanh")
                nn.init.zeros_(m.bias)
        bb = []
        for fan_in in (D, HIDDEN, HIDDEN):
            m = nn.Linear(fan_in, HIDDEN)
            nn.init.orthogonal_(m.weight, math.sqrt(2))
            nn.init.constant_(m.bias, 0.0)
            bb.append(m)
        am = nn.Linear(HIDDEN, act)
        nn.init.orthogonal_(am.weight, 0.01 * math.sqrt(2))
        nn.init.constant_(am.bias, 0.0)
        sd = {"actor_logstd": torch.ones(1, act) * -0.5}
        for i, m in zip((0, 2, 4, 6), vh):
            sd[f"value_head.mlp.{i}.weight"] = m.weight
            if m.bias is not None:
                sd[f"value_head.mlp.{i}.bias"] = m.bias
        for i, m in zip((0, 2, 4), bb):
            sd[f"backbone.{i}.weight"], sd[f"backbone.{i}.bias"] = m.weight, m.bias
        sd["actor_mean.weight"], sd["actor_mean.bias"] = am.weight, am.bias
        self.load_reference_state_dict(sd)

    @torch.no_grad()
    def load_reference_state_dict(self, sd):
        missing = [k for k in self.shapes if k not in sd]
        extra = [k for k in sd if k not in self.shapes]
        if missing or extra:
            raise RuntimeError(f"state dict mismatch: missing {missing}, unexpected {extra}")
        for name, shp in self.shapes.items():
            t = sd[name]
            if tuple(t.shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {name}: {tuple(t.shape)} vs {tuple(shp)}")
            self.view(name).copy_(t.detach().to(torch.float32))
        self.mark_updated()

    def exposed_flat(self) -> torch.Tensor:
        """The owned elements in named_parameters() order, as ONE vector (a copy): what ``torch.cat([p.reshape(-1) for p in
        reference_model.parameters()])`` is for the reference's module."""
        return torch.cat([self.view(name).detach().reshape(-1) for name in self.shapes])

    def reference_state_dict(self):
        return OrderedDict((name, self.view(name).detach().clone()) for name in self.shapes)

    def named_views(self):
        """Reference-named VIEWS into the flat buffer (what nn.Module.state_dict() hands out in the reference: detached
        tensors aliasing the parameters).  Writing through them changes the weights -- call mark_updated() afterwards."""
        return OrderedDict((name, self.view(name).detach()) for name in self.shapes)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        """The reference's names and shapes, aliasing ``flat`` like torch's own state_dict aliases parameters: checkpoints,
        weight-sync buckets and patches written for the reference's MLPPolicy apply to this one unchanged."""
        out = OrderedDict() if destination is None else destination
        for name, t in self.named_views().items():
            out[prefix + name] = t
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Accepts the reference's key set (or the single-key {'flat': ...} form older checkpoints of this package used)."""
        if set(state_dict.keys()) == {"flat"}:
            with torch.no_grad():
                src = state_dict["flat"].to(self.flat.device, torch.float32).reshape(-1)
                if src.numel() not in (self.n_params, self.n_exposed):
                    raise RuntimeError(f"size mismatch for flat: {src.numel()} elements, this policy holds {self.n_exposed}")
                self.flat.data[:src.numel()].copy_(src)
            self.mark_updated()
            return torch.nn.modules.module._IncompatibleKeys([], [])
        if not strict:
            state_dict = {k: state_dict.get(k, self.view(k)) for k in self.shapes}
        self.load_reference_state_dict(state_dict)
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def mark_updated(self, tiles_fresh: bool = False):
        """Call after the flat parameters change (optimizer step, weight sync): the derived weight images are rebuilt
        lazily.  ``tiles_fresh``: the optimizer kernel already wrote the new weights into the tile image."""
        self._packed_version = -1
        if not tiles_fresh:
            self._tiles_version = -1

    def tiles(self) -> torch.Tensor:
        """Fragment-tile weight image for the fused launches (ops.mlp_rollout_step / ops.ppo_step); rebuilt lazily, always
        into the same buffer (captured hipGraphs keep pointing at it)."""
        ver = self.flat._version
        t = getattr(self, "_tiles", None)
        if t is None or t.device != self.flat.device or getattr(self, "_tiles_version", -1) != ver:
            self._tiles = ops.mlp_pack_tiles(self.flat.data, self.layout, t if (t is not None and t.device == self.flat.device) else None,
                                             bf16=self.compute_dtype == torch.bfloat16)
            self._tiles_version = ver
        return self._tiles

    def packed(self) -> torch.Tensor:
        ver = self.flat._version
        if self._packed is None or self._packed.device != self.flat.device or self._packed_version != ver:
            self._packed = ops.mlp_pack(self.flat.data, self.layout, self._packed if
                                        (self._packed is not None and self._packed.device == self.flat.device) else None)
            self._packed_version = ver
        return self._packed

    # ---- reference API ---------------------------------------------------------------------------------------
    def preprocess_env_obs(self, env_obs):
        return {"states": env_obs["states"].to(self.flat.device)}  # mlp_policy.py:122-124

    def forward(self, forward_type=None, **kwargs):
        return self.default_forward(**kwargs)

    def default_forward(self, forward_inputs, compute_logprobs=True, compute_entropy=True, compute_values=True, **kwargs):
        states = forward_inputs["states"].to(self.flat.device, torch.float32).contiguous()
        action = forward_inputs["action"].to(self.flat.device, torch.float32).contiguous()
        action = action.reshape(states.shape[0], -1)
        if compute_values and not self.has_value_head:
            raise NotImplementedError  # mlp_policy.py:230-235
        logprob, entropy, value = _MlpTrainFn.apply(self.flat, self, states, action)
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
        """mlp_policy.py:295-320.  ``eps`` injects the N(0,1) draw (parity tests); otherwise it is drawn on device."""
        states = self.preprocess_env_obs(env_obs)["states"].to(torch.float32).contiguous()
        M = states.shape[0]
        if mode == "train":
            if eps is None:
                eps = torch.randn((M, self.layout.act_dim), dtype=torch.float32, device=states.device, generator=generator)
            eps = eps.to(states.device, torch.float32).contiguous()
        elif mode == "eval":
            eps = None
        else:
            raise NotImplementedError(f"{mode=}")
        action, logprob, value = ops.mlp_rollout_step(self.flat.data, self.tiles(), self.layout, states, eps)
        if not calculate_values or not self.has_value_head:
            value = torch.zeros_like(logprob[..., :1])  # mlp_policy.py:283-286
        chunk_actions = action.reshape(-1, self.num_action_chunks, self.action_dim)
        forward_inputs = {"action": action, "model_action": action}
        if return_obs:
            forward_inputs["states"] = states
        return chunk_actions, {"prev_logprobs": logprob, "prev_values": value, "forward_inputs": forward_inputs}

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._packed = None
        return out

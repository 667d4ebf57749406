"""CPU restatement of the reference's rollout + PPO/GRPO arithmetic.  TEST INFRASTRUCTURE ONLY.

Every function below restates, with stock torch CPU ops in the same evaluation order (so the fp32
rounding is the reference's), one function of RLinf's hot path and cites the file:line it follows
(paths relative to /root/reference).  It exists so that parity tests and the ``cpu_baseline`` leg of
``bench.py`` can run on the GPU box, where the reference tree does not exist.  It is pinned against
the real reference by ``tests/test_oracle_vs_reference.py`` (in the build container) and against the
committed fixtures in ``tests/golden/`` (everywhere).

Nothing under ``rlinf_amd/`` imports this file.
"""

from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)
LOG_SQRT_2PI = math.log(math.sqrt(2.0 * math.pi))  # torch.distributions.Normal.log_prob's constant


# --------------------------------------------------------------------------------------------
# a9  loss mask            rlinf/utils/metric_utils.py:516-537
# --------------------------------------------------------------------------------------------
def loss_mask_from_dones(dones: torch.Tensor):
    """dones bool [n+1, B, C]  ->  (mask bool [n, B, C], mask_sum int64 [n, B, C] (stride-0 view)).

    A step is valid iff no done flag is set at or before its own row, counting only the last
    n*C+1 rows of the time-flattened [(n+1)*C, B] done matrix (metric_utils.py:520-527)."""
    n_plus_1, bsz, chunk = dones.shape
    n = n_plus_1 - 1
    flat = dones.transpose(1, 2).reshape(-1, bsz)[-(n * chunk + 1):]
    valid = (flat.cumsum(dim=0) == 0)[:-1]
    mask = valid.reshape(n, chunk, bsz).transpose(1, 2)
    count = mask.sum(dim=(0, 2), keepdim=True).expand_as(mask)
    return mask, count


# --------------------------------------------------------------------------------------------
# a10  [n,B,C] <-> [T,B]    rlinf/algorithms/utils.py:67-131,155-174
# --------------------------------------------------------------------------------------------
def flatten_embodied_inputs(rewards, dones, values=None, loss_mask=None, loss_mask_sum=None,
                            reward_type="action_level", want_values=True):
    if reward_type == "chunk_level":  # utils.py:80-89
        rewards = rewards.sum(dim=-1, keepdim=True)
        dones = dones.max(dim=-1, keepdim=True)[0]
        if loss_mask is not None:
            loss_mask = loss_mask.max(dim=-1, keepdim=True)[0]
        if loss_mask_sum is not None:
            loss_mask_sum = loss_mask_sum.max(dim=-1, keepdim=True)[0]
    n, bsz, chunk = rewards.shape
    steps = n * chunk
    rewards_tb = rewards.transpose(1, 2).reshape(steps, bsz)
    mask_tb = None if loss_mask is None else loss_mask.transpose(1, 2).reshape(steps, bsz)
    dones_tb = dones.transpose(1, 2).reshape((n + 1) * chunk, bsz)[-(steps + 1):]
    values_tb = None
    if want_values and values is not None:
        values_tb = values.transpose(1, 2).reshape((n + 1) * chunk, bsz)[: steps + 1]
    return dict(rewards=rewards_tb, dones=dones_tb, values=values_tb, loss_mask=mask_tb,
                loss_mask_sum=loss_mask_sum, n=n, bsz=bsz, chunk=chunk, steps=steps)


def unflatten_embodied_output(x_tb: torch.Tensor, n: int, chunk: int):
    return x_tb.reshape(n, chunk, -1).transpose(1, 2)  # utils.py:166-172


# --------------------------------------------------------------------------------------------
# a12  safe_normalize       rlinf/algorithms/utils.py:397-404
# --------------------------------------------------------------------------------------------
def masked_standardize(x: torch.Tensor, mask: Optional[torch.Tensor], eps: float = 1e-5):
    picked = x[mask]  # mask None -> x[None]: adds a leading dim, selects everything
    if len(picked) > 0:
        x = (x - picked.mean()) / (picked.std() + eps)  # std unbiased; eps added to std
    return x


# --------------------------------------------------------------------------------------------
# a11  GAE                  rlinf/algorithms/advantages.py:24-86
# --------------------------------------------------------------------------------------------
def gae_tb(rewards, dones, values=None, gamma=1.0, gae_lambda=1.0, normalize_advantages=True,
           normalize_returns=False, loss_mask=None):
    """rewards [T,B] f32, values [T+1,B] f32 or None, dones [T+1,B] bool -> (adv, ret) [T,B]."""
    steps = rewards.shape[0]
    ret = torch.zeros_like(rewards)
    no_critic = values is None
    if no_critic:
        gamma, gae_lambda = 1, 1
    acc = 0  # python int, exactly like the reference (advantages.py:59)
    for t in range(steps - 1, -1, -1):
        alive = ~dones[t + 1]
        if no_critic:
            delta = rewards[t]
        else:
            delta = rewards[t] + gamma * values[t + 1] * alive - values[t]
        acc = delta + gamma * gae_lambda * alive * acc
        ret[t] = acc if no_critic else acc + values[t]
    adv = ret if no_critic else ret - values[:-1]
    if normalize_advantages:
        adv = masked_standardize(adv, loss_mask)
    if normalize_returns:
        ret = masked_standardize(ret, loss_mask)
    return adv, ret


# --------------------------------------------------------------------------------------------
# a13  scores + GRPO        rlinf/algorithms/utils.py:134-152, advantages.py:89-121
# --------------------------------------------------------------------------------------------
def first_episode_scores(rewards_tb, dones_tb):
    steps, bsz = rewards_tb.shape
    score = torch.zeros(bsz)  # CPU, default dtype -- as the reference allocates it (utils.py:139)
    for t in range(steps - 1, -1, -1):
        score = score * ~dones_tb[t + 1]
        score += rewards_tb[t]
    return score


def grpo_tb(scores, loss_mask_tb, group_size: int, eps: float = 1e-6):
    grouped = scores.view(-1, group_size)
    mean = grouped.mean(dim=-1, keepdim=True).expand_as(grouped)
    std = grouped.std(dim=-1, keepdim=True).expand_as(grouped)
    centred = (grouped - mean) / (std + eps)
    adv = (torch.zeros_like(loss_mask_tb) + centred.view(1, -1)) * loss_mask_tb
    return adv


# --------------------------------------------------------------------------------------------
# a14  dispatcher           rlinf/algorithms/registry.py:95-124 (embodied branch)
# --------------------------------------------------------------------------------------------
def embodied_adv_and_returns(*, adv_type, rewards, dones, values=None, loss_mask=None,
                             loss_mask_sum=None, gamma=1.0, gae_lambda=1.0, group_size=8,
                             reward_type="action_level", normalize_advantages=True,
                             normalize_returns=False):
    f = flatten_embodied_inputs(rewards, dones, values, loss_mask, loss_mask_sum, reward_type,
                                want_values=(adv_type == "gae"))
    if adv_type == "gae":
        adv, ret = gae_tb(f["rewards"], f["dones"], f["values"], gamma, gae_lambda,
                          normalize_advantages, normalize_returns, f["loss_mask"])
    elif adv_type == "grpo":
        scores = first_episode_scores(f["rewards"], f["dones"])
        adv, ret = grpo_tb(scores, f["loss_mask"], group_size), None
    else:
        raise ValueError(adv_type)
    out = {"advantages": unflatten_embodied_output(adv, f["n"], f["chunk"])}
    if ret is not None:
        out["returns"] = unflatten_embodied_output(ret, f["n"], f["chunk"])
    return out


# --------------------------------------------------------------------------------------------
# a6  bootstrap rewards     rlinf/workers/env/env_worker.py:718-758
# --------------------------------------------------------------------------------------------
def bootstrap_rewards(rewards, dones_or_trunc, bootstrap_values, gamma):
    """rewards [B,C]; flags [B,C] bool; bootstrap_values [B,1]: r[:, -1] += gamma*V where flag[:, -1]."""
    out = rewards.clone()
    last = dones_or_trunc[:, -1]
    out[last, -1] += gamma * bootstrap_values[last, 0]
    return out


# --------------------------------------------------------------------------------------------
# a21  aggregations         rlinf/utils/utils.py:323-356
# --------------------------------------------------------------------------------------------
def masked_mean(values, mask):
    if mask is None:
        return values.mean()
    if (~mask).all():
        return (values * mask).sum()
    return (values * mask).sum() / mask.sum()


def masked_mean_ratio(values, mask, ratio):
    return (values / ratio * mask).mean()


def huber(err, delta):  # rlinf/algorithms/utils.py:20-23
    return torch.where(err.abs() < delta, 0.5 * err ** 2, delta * (err.abs() - 0.5 * delta))


# --------------------------------------------------------------------------------------------
# a18  loss-input shaping   rlinf/algorithms/utils.py:280-376
# --------------------------------------------------------------------------------------------
def _pad_dims(t, target_shape):
    if t is None:
        return None
    if t.shape != target_shape:
        while t.dim() < len(target_shape):
            t = t.unsqueeze(-1)
    return t


def shape_loss_inputs(logprobs, old_logprobs, advantages, logprob_type, action_dim,
                      loss_mask=None, loss_mask_sum=None, values=None, prev_values=None,
                      returns=None, reward_type="action_level"):
    if reward_type == "chunk_level":
        advantages = advantages.flatten()
        loss_mask = None if loss_mask is None else loss_mask.flatten()
        loss_mask_sum = None if loss_mask_sum is None else loss_mask_sum.flatten()
        values = None if values is None else values.flatten()
        prev_values = None if prev_values is None else prev_values.flatten()
        returns = None if returns is None else returns.flatten()
    bsz = logprobs.shape[0]
    if logprob_type == "token_level":
        logprobs = logprobs.reshape(bsz, -1, action_dim)
        old_logprobs = old_logprobs.reshape(bsz, -1, action_dim)
        advantages = advantages.unsqueeze(-1)
        loss_mask = None if loss_mask is None else loss_mask.unsqueeze(-1)
        loss_mask_sum = None if loss_mask_sum is None else loss_mask_sum.unsqueeze(-1)
    elif logprob_type == "action_level":
        logprobs = logprobs.reshape(bsz, -1, action_dim).sum(dim=-1)
        old_logprobs = old_logprobs.reshape(bsz, -1, action_dim).sum(dim=-1)
    elif logprob_type == "chunk_level":
        logprobs = logprobs.reshape(bsz, -1, action_dim).sum(dim=[1, 2])
        old_logprobs = old_logprobs.reshape(bsz, -1, action_dim).sum(dim=[1, 2])
    shp = logprobs.shape
    return dict(logprobs=logprobs, old_logprobs=old_logprobs, advantages=_pad_dims(advantages, shp),
                loss_mask=_pad_dims(loss_mask, shp), loss_mask_sum=_pad_dims(loss_mask_sum, shp),
                values=_pad_dims(values, shp), prev_values=_pad_dims(prev_values, shp),
                returns=_pad_dims(returns, shp))


# --------------------------------------------------------------------------------------------
# a19  PPO actor loss       rlinf/algorithms/losses.py:170-312
# --------------------------------------------------------------------------------------------
def ppo_actor_loss(logprobs, old_logprobs, advantages, clip_ratio_low, clip_ratio_high,
                   loss_mask=None, clip_ratio_c=None, max_episode_steps=None, loss_mask_sum=None,
                   critic_warmup=False, clip_log_ratio_min=None, clip_log_ratio_max=None):
    agg, agg_ratio = masked_mean, None
    if max_episode_steps is not None and loss_mask_sum is not None and loss_mask is not None:
        agg_ratio = (loss_mask_sum * 1.0) / max_episode_steps
        agg = masked_mean_ratio
    if loss_mask is None:
        loss_mask = torch.ones_like(logprobs).bool()
    n_valid = loss_mask.count_nonzero() or 1
    log_ratio = logprobs - old_logprobs
    if clip_log_ratio_min is not None:
        log_ratio = torch.clamp(log_ratio, min=clip_log_ratio_min)
    if clip_log_ratio_max is not None:
        log_ratio = torch.clamp(log_ratio, max=clip_log_ratio_max)
    ratio = torch.where(loss_mask, torch.exp(log_ratio), 0)
    kl_terms = torch.where(loss_mask, log_ratio.detach(), 0.0)
    clipped = torch.clamp(ratio, 1.0 - clip_ratio_low, 1.0 + clip_ratio_high)
    surr1 = -advantages * ratio
    surr2 = -advantages * clipped
    is_clipped = surr1.detach() < surr2.detach()
    per_elem = torch.max(surr1, surr2)
    if clip_ratio_c is not None:
        surr3 = torch.sign(advantages) * clip_ratio_c * advantages
        is_dual = surr3.detach() < per_elem.detach()
        per_elem = torch.min(per_elem, surr3)
    else:
        is_dual = torch.zeros_like(is_clipped)

    def _agg(v):
        return agg(v, loss_mask) if agg_ratio is None else agg(v, loss_mask, agg_ratio)

    loss_abs = _agg(per_elem.abs())
    loss = _agg(per_elem)
    is_dual = (is_dual * loss_mask).bool()
    clip_fraction = (is_clipped * loss_mask).sum() / float(n_valid)
    approx_kl = -torch.sum(kl_terms) / float(n_valid)
    dual_ratio = torch.where(is_dual, ratio, 0)
    if critic_warmup:
        loss = torch.tensor(0.0)
    m = loss_mask
    if ratio.dim() > 2 and loss_mask.shape[-1] == 1 and ratio.shape[-1] > 1:
        m = loss_mask.expand_as(ratio)
    metrics = {
        "actor/policy_loss": loss.detach(),
        "actor/policy_loss_abs": loss_abs.detach(),
        "actor/ratio": masked_mean(ratio.detach(), m),
        "actor/ratio_abs": masked_mean((ratio - 1).abs().detach(), m),
        "actor/clipped_ratio": masked_mean(clipped.detach(), m),
        "actor/dual_cliped_ratio": masked_mean(dual_ratio.detach(), m),
        "actor/approx_kl": approx_kl.detach(),
        "actor/clip_fraction": clip_fraction.detach(),
    }
    return loss, metrics


EV_KEYS = ("count", "returns_sum", "returns_sq_sum", "errors_sum", "errors_sq_sum")


# --------------------------------------------------------------------------------------------
# a20  PPO critic loss      rlinf/algorithms/losses.py:315-380, metric_utils.py:232-258
# --------------------------------------------------------------------------------------------
def ppo_critic_loss(values, returns, prev_values, value_clip, huber_delta, loss_mask=None,
                    max_episode_steps=None, loss_mask_sum=None):
    agg_ratio = None
    if max_episode_steps is not None and loss_mask_sum is not None and loss_mask is not None:
        agg_ratio = (loss_mask_sum * 1.0) / max_episode_steps
    v_clip = prev_values + (values - prev_values).clamp(-value_clip, value_clip)
    per_elem = torch.max(huber(returns - values, huber_delta), huber(returns - v_clip, huber_delta))
    if agg_ratio is None:
        loss = masked_mean(per_elem, loss_mask)
    else:
        loss = masked_mean_ratio(per_elem, loss_mask, agg_ratio)
    clip_ratio = ((v_clip - prev_values).abs() > value_clip).float().mean()
    r = returns.detach().float()
    v = values.detach().float()
    if loss_mask is not None:
        mk = torch.broadcast_to(loss_mask.bool(), r.shape)
        r, v = r[mk], v[mk]
    else:
        r, v = r.reshape(-1), v.reshape(-1)
    err = r - v
    ev = dict(count=torch.tensor(float(r.numel())), returns_sum=r.sum(), returns_sq_sum=(r * r).sum(),
              errors_sum=err.sum(), errors_sq_sum=(err * err).sum())
    metrics = {"critic/value_loss": loss.detach(), "critic/value_clip_ratio": clip_ratio.detach()}
    metrics.update({f"ev/{k}": ev[k] for k in EV_KEYS})
    return loss, metrics


def ppo_actor_critic_loss(**kw):
    """registry name "actor_critic" (losses.py:397-425) after preprocess_loss_inputs."""
    actor_kw = {k: kw[k] for k in ("logprobs", "old_logprobs", "advantages", "clip_ratio_low",
                                   "clip_ratio_high", "loss_mask", "clip_ratio_c", "max_episode_steps",
                                   "loss_mask_sum", "critic_warmup", "clip_log_ratio_min",
                                   "clip_log_ratio_max") if k in kw}
    critic_kw = {k: kw[k] for k in ("values", "returns", "prev_values", "value_clip", "huber_delta",
                                    "loss_mask", "max_episode_steps", "loss_mask_sum") if k in kw}
    la, ma = ppo_actor_loss(**actor_kw)
    lc, mc = ppo_critic_loss(**critic_kw)
    ma.update(mc)
    return la + lc, ma


def explained_variance(stats):  # metric_utils.py:261-290
    n = float(stats["count"])
    if n < 2:
        return float("nan")
    rs, rss = float(stats["returns_sum"]), float(stats["returns_sq_sum"])
    es, ess = float(stats["errors_sum"]), float(stats["errors_sq_sum"])
    rc = torch.tensor(rss, dtype=torch.float32) - torch.tensor(rs, dtype=torch.float32) ** 2 / n
    if torch.isnan(rc) or rc == 0:
        return float("nan")
    ec = torch.tensor(ess, dtype=torch.float32) - torch.tensor(es, dtype=torch.float32) ** 2 / n
    return float(1 - ec / rc)


# --------------------------------------------------------------------------------------------
# a2/a3/a17  MLP policy     rlinf/models/embodiment/mlp_policy/mlp_policy.py:91-107,202-293
#            value head     rlinf/models/embodiment/modules/value_head.py:17-66
#            init           rlinf/models/embodiment/modules/utils.py:20-23
# --------------------------------------------------------------------------------------------
class OracleMLPPolicy(nn.Module):
    """Same parameter names/shapes as the reference MLPPolicy (PPO configuration: state-independent
    log-std, no tanh squashing, value head 3x256 tanh with bias-free last layer)."""

    def __init__(self, obs_dim=42, action_dim=8, num_action_chunks=1, hidden=256, add_value_head=True):
        super().__init__()
        self.obs_dim, self.action_dim, self.num_action_chunks = obs_dim, action_dim, num_action_chunks
        self.add_value_head = add_value_head  # False: mlp_policy.py:56-62 builds no value head (value-free PPO / GRPO)

        def ortho(layer, gain):
            nn.init.orthogonal_(layer.weight, gain)
            nn.init.constant_(layer.bias, 0.0)
            return layer

        # construction order == the reference's (value head first, mlp_policy.py:56-105), so
        # named_parameters() order and a shared torch seed line up tensor by tensor.
        if add_value_head:
            vh = nn.Module()
            vh.mlp = nn.Sequential(
                nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, num_action_chunks, bias=False))
            for m in vh.mlp:
                if isinstance(m, nn.Linear):
                    if m is vh.mlp[-1]:
                        nn.init.normal_(m.weight, mean=0.0, std=0.02)
                    else:
                        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="tanh")
                        nn.init.zeros_(m.bias)
            self.value_head = vh
        g = math.sqrt(2.0)
        self.backbone = nn.Sequential(
            ortho(nn.Linear(obs_dim, hidden), g), nn.Tanh(),
            ortho(nn.Linear(hidden, hidden), g), nn.Tanh(),
            ortho(nn.Linear(hidden, hidden), g), nn.Tanh())
        self.actor_mean = ortho(nn.Linear(hidden, num_action_chunks * action_dim), 0.01 * g)
        self.actor_logstd = nn.Parameter(torch.ones(1, num_action_chunks * action_dim) * -0.5)

    def _mean_logstd(self, states):
        mean = self.actor_mean(self.backbone(states))
        return mean, self.actor_logstd.expand_as(mean)

    @torch.no_grad()
    def act(self, states, eps=None, mode="train"):
        """mlp_policy.py:256-320 with the N(0,1) draw injected: action = eps*std + mean (the order
        torch.normal(mean_tensor, std_tensor) uses on CPU), eval mode acts with the mean."""
        mean, logstd = self._mean_logstd(states)
        std = torch.exp(logstd)
        action = mean.clone() if mode == "eval" else eps * std + mean
        # torch.distributions.Normal.log_prob: -(x-mu)^2/(2 var) - log(scale) - log(sqrt(2 pi))
        logp = -((action - mean) ** 2) / (2 * std ** 2) - std.log() - LOG_SQRT_2PI
        value = self.value_head.mlp(states) if self.add_value_head else torch.zeros_like(logp[..., :1])  # mlp_policy.py:283-286
        return action, logp, value

    def evaluate(self, states, action):
        """mlp_policy.py:202-236: per-dim log-prob of a stored action, entropy, value (with grad)."""
        mean, logstd = self._mean_logstd(states)
        std = torch.exp(logstd)
        logp = -((action - mean) ** 2) / (2 * std ** 2) - std.log() - LOG_SQRT_2PI
        ent = 0.5 + HALF_LOG_2PI + torch.log(std)
        if not self.add_value_head:  # (the reference raises when asked for values, :230-235; its actor-only callers do not ask)
            return dict(logprobs=logp, entropy=ent)
        return dict(logprobs=logp, entropy=ent, values=self.value_head.mlp(states))


# --------------------------------------------------------------------------------------------
# a16  flatten + shuffle    rlinf/utils/nested_dict_process.py:272-285, :96-118
# --------------------------------------------------------------------------------------------
_T_PLUS_1_KEYS = ("dones", "terminations", "truncations", "prev_values")


def flatten_and_shuffle(batch: dict, perm: torch.Tensor) -> dict:
    out = {}
    for k, v in batch.items():
        if k in _T_PLUS_1_KEYS:
            v = v[:-1]
        if v is None:
            out[k] = None
        elif isinstance(v, torch.Tensor):
            out[k] = v.reshape(-1, *v.shape[2:])[perm]
        elif isinstance(v, dict):
            out[k] = flatten_and_shuffle(v, perm)
    return out


def chunk_batch(batch: dict, parts: int) -> list:
    out = [dict() for _ in range(parts)]
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            pieces = [c.contiguous() for c in torch.chunk(v, parts, dim=0)]
        elif isinstance(v, dict):
            pieces = chunk_batch(v, parts)
        else:
            pieces = [None] * parts
        for i in range(parts):
            out[i][k] = pieces[i]
    return out


# --------------------------------------------------------------------------------------------
# a22/a23  one optimizer step   rlinf/workers/actor/embodied_fsdp_actor_worker.py:591-700,
#          rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:429-463,501-590
# --------------------------------------------------------------------------------------------
def build_adamw(policy: nn.Module, lr=3e-4, value_lr=3e-4, betas=(0.9, 0.999), eps=1e-8, wd=0.01):
    actor, critic = [], []
    for name, p in policy.named_parameters():
        (critic if "value_head" in name else actor).append(p)
    return torch.optim.AdamW([{"params": actor, "lr": lr, "betas": betas},
                              {"params": critic, "lr": value_lr, "betas": betas}], eps=eps,
                             weight_decay=wd)


def amp(enabled: bool):
    """The learner's ``amp_context`` (fsdp_model_manager.py:122-142: torch.amp.autocast(dtype=bf16) around the model forward
    ONLY, embodied_fsdp_actor_worker.py:624-632).  A fresh context per forward: autocast caches its bf16 weight copies per
    context, and a copy made under no_grad (a rollout forward) would cut the training forward off from the parameters."""
    return torch.autocast("cpu", dtype=torch.bfloat16, enabled=bool(enabled), cache_enabled=False)


def reshape_entropy(entropy, entropy_type: str, action_dim: int, batch_size: int):
    """rlinf/utils/utils.py:384-408: action_level sums over action_dim ([bsz, C]), chunk_level over the whole row ([bsz]); any
    other name (token_level) leaves the per-dimension tensor as it is."""
    if entropy_type == "action_level":
        return entropy.reshape(batch_size, -1, action_dim).sum(dim=-1)
    if entropy_type == "chunk_level":
        return entropy.sum(dim=-1)
    return entropy


def ppo_minibatch_step(policy, opt, mb: dict, *, clip_low=0.2, clip_high=0.2, value_clip=1.0,
                       huber_delta=10.0, entropy_bonus=0.0, clip_grad=0.5, action_dim=8,
                       logprob_type="action_level", critic_warmup=False, max_episode_steps=None, autocast=False,
                       entropy_type="action_level", reward_type="action_level", loss_type="actor_critic"):
    """forward -> actor_critic (or, loss_type "actor": value-free PPO / GRPO) loss -> backward -> clip_grad_norm_ -> AdamW
    (skipped if norm non-finite).
    ``loss_mask_sum`` + ``max_episode_steps`` (both present when auto_reset is off) switch the aggregation to
    masked_mean_ratio, as train_micro_batch's loss_kwargs do (embodied_fsdp_actor_worker.py:641-662, losses.py:219-227)."""
    opt.zero_grad()
    with amp(autocast):
        out = policy.evaluate(mb["states"], mb["action"])
    if autocast:  # the loss asserts f32 inputs (losses.py:232-240); log-probs come out f32 by type promotion, values are cast
        out = {k: v.float() for k, v in out.items()}
    critic = loss_type == "actor_critic"
    shaped = shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], logprob_type,
                               action_dim, loss_mask=mb.get("loss_mask"), loss_mask_sum=mb.get("loss_mask_sum"),
                               values=out["values"] if critic else None, prev_values=mb["prev_values"] if critic else None,
                               returns=mb.get("returns") if critic else None, reward_type=reward_type)
    if critic:
        loss, metrics = ppo_actor_critic_loss(clip_ratio_low=clip_low, clip_ratio_high=clip_high, critic_warmup=critic_warmup,
                                              value_clip=value_clip, huber_delta=huber_delta, max_episode_steps=max_episode_steps,
                                              **shaped)
    else:  # registry name "actor" (losses.py:170-312 alone): the value head gets no gradient, i.e. no update (grad None)
        loss, metrics = ppo_actor_loss(shaped["logprobs"], shaped["old_logprobs"], shaped["advantages"], clip_low, clip_high,
                                       loss_mask=shaped["loss_mask"], max_episode_steps=max_episode_steps,
                                       loss_mask_sum=shaped["loss_mask_sum"])
    if entropy_bonus > 0 and not critic_warmup:  # embodied_fsdp_actor_worker.py:680
        ent = reshape_entropy(out["entropy"], entropy_type, action_dim, out["logprobs"].shape[0])
        ent_loss = masked_mean(ent, shaped["loss_mask"])
        loss = loss - entropy_bonus * ent_loss
        metrics["actor/entropy_loss"] = ent_loss.detach()
    loss.backward()
    if critic_warmup:  # build_optimizer(enable_critic_warmup=True) froze everything but the value head
        for n, p in policy.named_parameters():  # (fsdp_model_manager.py:523-531): no gradient, no norm share, no update
            if "value_head" not in n:
                p.grad = None
    gnorm = torch.nn.utils.clip_grad_norm_(policy.parameters(), clip_grad)
    if torch.isfinite(gnorm):
        opt.step()
    metrics["actor/grad_norm"] = gnorm.detach()
    metrics["actor/total_loss"] = loss.detach()
    return metrics


def restart_optimizer(opt):
    """What the step that ends the critic warm-up does (fsdp_model_manager.py:451-459): a NEW AdamW over all parameters,
    i.e. every moment and step count back to zero.  Clearing the state of the existing one is the same thing (torch
    re-initialises a parameter's state lazily when it is empty)."""
    opt.state.clear()


# --------------------------------------------------------------------------------------------
# a19b  decoupled PPO actor loss   rlinf/algorithms/losses.py:27-167 (+ :383-393 with the critic)
# --------------------------------------------------------------------------------------------
def decoupled_actor_loss(logprobs, old_logprobs, clip_ratio_low, clip_ratio_high, advantages, proximal_logprobs=None,
                         versions=None, current_version=None, loss_mask=None, clip_ratio_c=None, max_episode_steps=None,
                         loss_mask_sum=None, critic_warmup=False, behave_weight_threshold=None):
    if loss_mask is None:
        loss_mask = torch.ones_like(logprobs).bool()
    agg_ratio = None
    if max_episode_steps is not None and loss_mask_sum is not None and loss_mask is not None:
        agg_ratio = (loss_mask_sum * 1.0) / max_episode_steps
    if proximal_logprobs is None:
        if versions is None or current_version is None:
            proximal_logprobs = old_logprobs.detach()
        else:  # interpolate the proximal policy between behaviour and current by version distance (:72-89)
            v_b = versions.float()
            v_t = float(current_version)
            diff, gap = v_t - v_b, (v_t - 1.0) - v_b
            alpha = torch.where((diff > 0) & (versions >= 0), gap / diff, torch.zeros_like(v_b))
            while alpha.dim() < logprobs.dim():
                alpha = alpha.unsqueeze(-1)
            alpha = torch.clamp(alpha, 0.0, 1.0)
            proximal_logprobs = (old_logprobs + alpha * (logprobs - old_logprobs)).detach()
    n_valid = loss_mask.count_nonzero() or 1
    prox_ratio = torch.where(loss_mask, torch.exp(logprobs - proximal_logprobs), 0.0)
    clipped = torch.clamp(prox_ratio, 1.0 - clip_ratio_low, 1.0 + clip_ratio_high)
    surr1 = -advantages * prox_ratio
    surr2 = -advantages * clipped
    per_elem = torch.max(surr1, surr2)
    if clip_ratio_c is not None:
        assert clip_ratio_c > 1.0, clip_ratio_c
        surr3 = torch.sign(advantages) * clip_ratio_c * advantages
        is_dual = surr3.detach() < per_elem.detach()
        per_elem = torch.min(per_elem, surr3)
    else:
        is_dual = torch.zeros_like(per_elem, dtype=torch.bool)
    behav_weight = torch.exp(proximal_logprobs - old_logprobs)
    behav_mask = ((behav_weight <= behave_weight_threshold).logical_and(loss_mask)
                  if behave_weight_threshold is not None else loss_mask)
    n_behav = behav_mask.count_nonzero() or 1
    weighted = per_elem * behav_weight
    loss = masked_mean(weighted, behav_mask) if agg_ratio is None else masked_mean_ratio(weighted, behav_mask, agg_ratio)
    if critic_warmup:
        loss = torch.tensor(0.0)
    with torch.no_grad():
        clip_fraction = (surr1 < surr2).logical_and(loss_mask).count_nonzero() / n_valid
        dual_clip_fraction = is_dual.logical_and(loss_mask).count_nonzero() / n_valid
        prox_kl = -torch.where(loss_mask, logprobs - proximal_logprobs, 0.0).sum() / n_valid
        behav_kl = -torch.where(behav_mask, proximal_logprobs - old_logprobs, 0.0).sum() / n_behav
        behav_clip_fraction = 1.0 - (n_behav / n_valid)
    metrics = {
        "actor/policy_loss": loss.detach(),
        "actor/proximal_ratio": masked_mean(prox_ratio.detach(), loss_mask),
        "actor/clipped_proximal_ratio": masked_mean(clipped.detach(), loss_mask),
        "actor/clip_fraction": clip_fraction,
        "actor/dual_clip_fraction": dual_clip_fraction,
        "actor/behav_clip_fraction": behav_clip_fraction,
        "actor/proximal_approx_kl": prox_kl,
        "actor/behav_approx_kl": behav_kl,
    }
    if versions is not None and current_version is not None and versions.shape == loss_mask.shape and loss_mask.any():
        metrics["actor/average_version"] = versions[loss_mask].float().mean()
        metrics["actor/current_version"] = torch.tensor(float(current_version))
    return loss, metrics


def shape_decoupled_inputs(proximal_logprobs, versions, logprob_type, action_dim, bsz, target_shape):
    """The proximal / versions part of preprocess_loss_inputs (rlinf/algorithms/utils.py:310-352)."""
    if logprob_type == "token_level":
        if proximal_logprobs is not None:
            proximal_logprobs = proximal_logprobs.reshape(bsz, -1, action_dim)
        if versions is not None:
            versions = versions.reshape(bsz, -1, action_dim)
    elif logprob_type == "action_level":
        if proximal_logprobs is not None:
            proximal_logprobs = proximal_logprobs.reshape(bsz, -1, action_dim).sum(dim=-1)
        if versions is not None:
            versions = versions.reshape(bsz, -1, action_dim)[..., 0]
    elif logprob_type == "chunk_level":
        if proximal_logprobs is not None:
            proximal_logprobs = proximal_logprobs.reshape(bsz, -1, action_dim).sum(dim=[1, 2])
        if versions is not None:
            versions = versions.reshape(bsz, -1, action_dim)[:, 0, 0]
    return proximal_logprobs, _pad_dims(versions, target_shape)


def decoupled_actor_critic_loss(*, proximal_logprobs=None, versions=None, current_version=None,
                                behave_weight_threshold=None, **kw):
    actor_kw = {k: kw[k] for k in ("logprobs", "old_logprobs", "advantages", "clip_ratio_low", "clip_ratio_high")}
    for k in ("loss_mask", "clip_ratio_c", "max_episode_steps", "loss_mask_sum", "critic_warmup"):
        if k in kw:
            actor_kw[k] = kw[k]
    a_loss, a_m = decoupled_actor_loss(proximal_logprobs=proximal_logprobs, versions=versions,
                                       current_version=current_version,
                                       behave_weight_threshold=behave_weight_threshold, **actor_kw)
    c_loss, c_m = ppo_critic_loss(kw["values"], kw["returns"], kw["prev_values"], kw["value_clip"], kw["huber_delta"],
                                  loss_mask=kw.get("loss_mask"), max_episode_steps=kw.get("max_episode_steps"),
                                  loss_mask_sum=kw.get("loss_mask_sum"))
    return a_loss + c_loss, {**a_m, **c_m}


# --------------------------------------------------------------------------------------------
# a8  rollout-epoch fold     rlinf/utils/nested_dict_process.py:251-269
# --------------------------------------------------------------------------------------------
def fold_rollout_epochs(nested: dict, rollout_epoch: int) -> dict:
    """[rollout_epoch * n, bsz, ...] -> [n, rollout_epoch * bsz, ...] for every tensor (recursively)."""
    out = {}
    for key, value in nested.items():
        if isinstance(value, torch.Tensor):
            v = value.reshape(rollout_epoch, -1, *value.shape[1:]).transpose(0, 1)
            out[key] = v.reshape(v.shape[0], -1, *v.shape[3:])
        elif isinstance(value, dict):
            out[key] = fold_rollout_epochs(value, rollout_epoch)
    return out


# --------------------------------------------------------------------------------------------
# a12b  advantage normalisation from sufficient statistics   rlinf/utils/distributed.py:942-965
# --------------------------------------------------------------------------------------------
def masked_normalization(x, mask=None, eps=1e-5):
    """rlinf/utils/distributed.py:866-937 with the defaults the async learner uses (dim=None, biased variance,
    high_precision, one process): masked-out elements enter as ZERO, statistics in float64, eps outside the root."""
    x = x.to(torch.float64).clone()
    if mask is None:
        factor = torch.tensor(float(x.numel()), dtype=torch.float64)
    else:
        mask = mask.to(torch.float64)
        assert mask.shape == x.shape, (mask.shape, x.shape)
        x = x * mask
        factor = mask.sum()
    mean = x.sum() / factor
    var = x.square().sum() / factor - mean ** 2
    return ((x - mean) / (var.sqrt() + eps)).float()


def masked_stats(x, mask=None):
    x = x.to(dtype=torch.float64)
    x = x[mask.bool()] if mask is not None else x.reshape(-1)
    return torch.tensor([x.numel(), x.sum(), x.square().sum()], dtype=torch.float64)


def normalize_from_stats(x, stats):
    stats = stats.to(dtype=torch.float64)
    count = stats[0].clamp_min(1.0)
    mean = stats[1] / count
    var = stats[2] / count - mean.square()
    return ((x.to(dtype=torch.float64) - mean) * torch.rsqrt(var.clamp_min(0.0) + 1e-5)).float()

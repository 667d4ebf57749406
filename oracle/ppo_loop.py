"""CPU oracle of ONE embodied PPO iteration (rollout -> advantages -> update), sync mode.  TEST INFRASTRUCTURE.

Restates the control flow of the reference's hot loops with the oracle arithmetic of ppo_oracle.py:
  rollout        EnvWorker._run_interact_once, rlinf/workers/env/env_worker.py:1058-1306 (row alignment A.1),
                 MultiStepRolloutWorker.generate_one_epoch, rlinf/workers/rollout/hf/huggingface_worker.py:677-800,
                 bootstrap on auto-reset (env_worker.py:718-758, huggingface_worker.py:612-627)
  advantages     EmbodiedFSDPActor.compute_advantages_and_returns, workers/actor/embodied_fsdp_actor_worker.py:286-321
  update         EmbodiedFSDPActor.run_training, :483-589 (one randperm per call, re-chunked every epoch)
The simulator is replaced by the same pre-generated tensors the product's synthetic env uses.
Used by tests/ (end-to-end parity at small sizes) and by bench.py's cpu_baseline leg; never by rlinf_amd/.
"""

from __future__ import annotations

import time

import torch

from . import ppo_oracle as O


def synthetic_env_tensors(seed: int, T: int, B: int, obs_dim: int = 42, max_episode_steps: int = 50,
                          mode: str = "periodic", p_done: float = 0.02):
    """obs [T+1,B,D] ~ N(0,1), final_obs [T,B,D] ~ N(0,1), rewards [T,B] ~ U(0,1),
    dones [T+1,B] (row 0 False): 'periodic' = truncation every max_episode_steps (auto-reset), or Bernoulli(p)."""
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T + 1, B, obs_dim, generator=g)
    final_obs = torch.randn(T, B, obs_dim, generator=g)
    rewards = torch.rand(T, B, generator=g)
    if mode == "periodic":
        dones = torch.zeros(T + 1, B, dtype=torch.bool)
        for t in range(1, T + 1):
            if t % max_episode_steps == 0:
                dones[t] = True
    else:
        dones = torch.rand(T + 1, B, generator=g) < p_done
        dones[0] = False
    return dict(obs=obs, final_obs=final_obs, rewards=rewards, dones=dones)


def rollout(policy: O.OracleMLPPolicy, env: dict, eps: torch.Tensor, gamma: float, auto_reset: bool = True, autocast: bool = False):
    """-> batch dict in the reference's buffer shapes ([n, B, ...] / [n + 1, B, ...], n chunk steps of C = num_action_chunks env
    steps each).  ``env`` is indexed by ENV step; a chunk step (ManiSkill's chunk_step, maniskill_env.py:327-372) plays C of
    them: rewards side by side [B, C], the done flag of every env that finished anywhere inside the chunk raised in the LAST
    column, auto-reset -- and therefore the bootstrap value of the true final observation (env_worker.py:718-758: first value
    column, added to the last reward column) -- at the chunk's end only.  ``autocast``: the policy forwards run under bf16
    autocast (`precision: bf16`), everything else stays f32."""
    C = policy.num_action_chunks
    T_env, B = env["rewards"].shape
    assert T_env % C == 0
    T = T_env // C
    A = policy.action_dim * C
    states = torch.empty(T, B, policy.obs_dim)
    action = torch.empty(T, B, A)
    logp = torch.empty(T, B, A)
    values = torch.empty(T + 1, B, C)
    rewards = torch.empty(T, B, C)
    dones = torch.zeros(T + 1, B, C, dtype=torch.bool)
    obs = env["obs"][0]
    for t in range(T):
        with O.amp(autocast):
            a, lp, v = policy.act(obs, eps=eps[t], mode="train")
        states[t], action[t], logp[t], values[t] = obs, a, lp, v
        r = env["rewards"][t * C:(t + 1) * C].transpose(0, 1).clone()           # [B, C]
        d = torch.zeros(B, C, dtype=torch.bool)
        d[:, -1] = env["dones"][t * C + 1:(t + 1) * C + 1].any(dim=0)
        if auto_reset and bool(d.any()) and policy.add_value_head:  # value-free policy: get_bootstrap_values is None (huggingface_worker.py:617-620)
            with O.amp(autocast):
                vf = policy.value_head.mlp(env["final_obs"][(t + 1) * C - 1]).detach()[:, :1].float()
            r = O.bootstrap_rewards(r, d, vf, gamma)
        rewards[t], dones[t + 1] = r, d
        obs = env["obs"][(t + 1) * C]
    with O.amp(autocast):
        values[T] = policy.value_head.mlp(obs).detach() if policy.add_value_head else 0.0  # (zeros: mlp_policy.py:283-286)
    return dict(rewards=rewards, dones=dones, prev_values=values, prev_logprobs=logp,
                forward_inputs=dict(states=states, action=action))


def rollout_epochs(policy, env: dict, eps: torch.Tensor, gamma: float, rollout_epoch: int, auto_reset: bool = True,
                   autocast: bool = False):
    """``rollout_epoch`` back-to-back epochs of T steps each (EnvWorker._run_interact_once's outer loop,
    env_worker.py:1074: every epoch opens with its own all-False bootstrap row and closes with its own value row),
    stacked on the time axis as the trajectory builder does and folded into the batch axis by
    process_nested_dict_for_adv (embodied_fsdp_actor_worker.py:208-216).  env holds rollout_epoch * T steps; epoch e
    plays steps [e*T, (e+1)*T)."""
    T = env["rewards"].shape[0] // rollout_epoch
    parts = []
    for e in range(rollout_epoch):
        sl = dict(obs=env["obs"][e * T:(e + 1) * T + 1], final_obs=env["final_obs"][e * T:(e + 1) * T],
                  rewards=env["rewards"][e * T:(e + 1) * T], dones=env["dones"][e * T:(e + 1) * T + 1])
        parts.append(rollout(policy, sl, eps[e * T:(e + 1) * T], gamma, auto_reset, autocast))
    stacked = {k: (dict((kk, torch.cat([p[k][kk] for p in parts], 0)) for kk in parts[0][k]) if isinstance(parts[0][k], dict)
                   else torch.cat([p[k] for p in parts], 0)) for k in parts[0]}
    return O.fold_rollout_epochs(stacked, rollout_epoch)


def advantages(batch: dict, gamma: float, gae_lambda: float, auto_reset: bool = True, adv_type: str = "gae",
               reward_type: str = "action_level", group_size: int = 1):
    lm = lms = None
    if not auto_reset:  # embodied_fsdp_actor_worker.py:219-233
        lm, lms = O.loss_mask_from_dones(batch["dones"])
        if reward_type == "chunk_level":  # :228-230
            lm, lms = lm.any(dim=-1, keepdim=True), lms[..., -1:]
    out = O.embodied_adv_and_returns(adv_type=adv_type, rewards=batch["rewards"], dones=batch["dones"],
                                     values=batch["prev_values"] if adv_type == "gae" else None, gamma=gamma, gae_lambda=gae_lambda,
                                     loss_mask=lm, loss_mask_sum=lms, reward_type=reward_type, group_size=group_size)
    batch = dict(batch)
    batch.update(advantages=out["advantages"].contiguous())
    if "returns" in out:
        batch.update(returns=out["returns"].contiguous())
    if lm is not None:
        batch.update(loss_mask=lm.contiguous(), loss_mask_sum=lms.contiguous())
    return batch


def pipeline_advantages(batch: dict, gamma: float, gae_lambda: float, auto_reset: bool = True):
    """runner.use_training_pipeline: the env side computes un-normalised GAE per stage batch and normalises with the
    (count, sum, sumsq) statistics summed over the batches that feed one learner rank
    (EnvWorker.compute_advantages_and_returns / send_rollout_trajectories_pipeline, env_worker.py:1469-1572).  GAE and the
    loss mask are per-column, so one pass over the rank's whole buffer equals the per-stage passes."""
    lm = lms = None
    if not auto_reset:
        lm, lms = O.loss_mask_from_dones(batch["dones"])
    out = O.embodied_adv_and_returns(adv_type="gae", rewards=batch["rewards"], dones=batch["dones"],
                                     values=batch["prev_values"], gamma=gamma, gae_lambda=gae_lambda, loss_mask=lm,
                                     loss_mask_sum=lms, normalize_advantages=False)
    adv = out["advantages"].contiguous()
    batch = dict(batch)
    batch.update(advantages=O.normalize_from_stats(adv, O.masked_stats(adv, lm)), returns=out["returns"].contiguous())
    if lm is not None:
        batch.update(loss_mask=lm.contiguous(), loss_mask_sum=lms.contiguous())
    return batch


def pipeline_permutation(T: int, B: int, stage_num: int, generator: torch.Generator) -> torch.Tensor:
    """Row order in which the learner sees one rank's [T, B] buffer in pipeline mode: every stage's [T, B/stage_num] block
    is flattened and shuffled on its own with the rank's STATEFUL generator (pack_pipeline_micro_batches,
    env_worker.py:1519-1537; seeded once in _init_pipeline_params :355-361), stage after stage.  Returned as indices
    into the rank's flattened [T*B] buffer."""
    n = B // stage_num
    parts = []
    for s in range(stage_num):
        local = torch.randperm(T * n, generator=generator)
        parts.append((local // n) * B + s * n + (local % n))
    return torch.cat(parts)


def update(policy, opt, batch: dict, *, seed: int, global_batch: int, update_epoch: int, clip_low=0.2, clip_high=0.2,
           value_clip=1.0, huber_delta=10.0, clip_grad=0.5, max_steps: int | None = None, entropy_bonus: float = 0.0,
           perm: torch.Tensor | None = None, critic_warmup_steps: int = 0, steps_done: int = 0, max_episode_steps=None,
           autocast: bool = False, entropy_type: str = "action_level", flat: dict | None = None,
           logprob_type: str = "action_level", reward_type: str = "action_level", loss_type: str = "actor_critic"):
    """``flat``: already flattened + shuffled rows (the pipeline learner with rollout_epoch > 1 concatenates per-epoch
    shuffles); otherwise ``batch`` is flattened with ``perm`` (default: the seeded randperm of run_training)."""
    if flat is None:
        T, B = batch["prev_logprobs"].shape[:2]
        if perm is None:
            perm = torch.randperm(T * B, generator=torch.Generator().manual_seed(seed))
        flat = O.flatten_and_shuffle(batch, perm)
    n_rows = flat["prev_logprobs"].shape[0]
    n_mb = n_rows // global_batch
    assert n_rows % global_batch == 0
    metrics, steps = [], 0
    for _ in range(update_epoch):
        for mb in O.chunk_batch(flat, n_mb):
            m = O.ppo_minibatch_step(
                policy, opt, dict(states=mb["forward_inputs"]["states"], action=mb["forward_inputs"]["action"],
                                  prev_logprobs=mb["prev_logprobs"], advantages=mb["advantages"],
                                  prev_values=mb["prev_values"], returns=mb.get("returns"), loss_mask=mb.get("loss_mask"),
                                  loss_mask_sum=mb.get("loss_mask_sum")),
                clip_low=clip_low, clip_high=clip_high, value_clip=value_clip, huber_delta=huber_delta,
                clip_grad=clip_grad, action_dim=policy.action_dim, entropy_bonus=entropy_bonus, max_episode_steps=max_episode_steps,
                autocast=autocast, entropy_type=entropy_type, logprob_type=logprob_type, reward_type=reward_type, loss_type=loss_type,
                critic_warmup=(steps_done + steps) < critic_warmup_steps)  # optimizer_steps < critic_warmup_steps (:664)
            metrics.append(m)
            steps += 1
            if critic_warmup_steps > 0 and steps_done + steps == critic_warmup_steps:
                O.restart_optimizer(opt)
            if max_steps is not None and steps >= max_steps:
                return metrics
    return metrics


def async_update(policy, opt, batch: dict, *, seed: int, global_batch: int, micro_batch: int, update_epoch: int, version: int,
                 clip_low=0.2, clip_high=0.2, clip_ratio_c=3.0, value_clip=1.0, huber_delta=10.0, clip_grad=0.5,
                 entropy_bonus: float = 0.0, behave_weight_threshold=None, normalize_advantages: bool = True,
                 logprob_type: str = "action_level", max_episode_steps=None):
    """AsyncPPOEmbodiedFSDPActor.run_training (rlinf/workers/actor/async_ppo_fsdp_worker.py:274-497), one rank: shuffle the
    flattened batch (randperm seeded actor.seed + rank), masked_normalization of the advantages over all of it, fixed global
    batches cut into micro-batches, decoupled actor-critic loss with current_version = version + 1 and dual clip, entropy
    bonus, loss / gradient_accumulation, clip_grad_norm_ + AdamW per global batch.  -> per-micro-batch metrics, grad norms."""
    T, B = batch["prev_logprobs"].shape[:2]
    perm = torch.randperm(T * B, generator=torch.Generator().manual_seed(seed))
    flat = O.flatten_and_shuffle(batch, perm)
    if normalize_advantages:
        flat["advantages"] = O.masked_normalization(flat["advantages"], flat.get("loss_mask"))
    accum = global_batch // micro_batch
    n_global = (T * B) // global_batch
    assert (T * B) % global_batch == 0 and global_batch % micro_batch == 0
    A = policy.action_dim
    metrics, norms = [], []
    for _ in range(update_epoch):
        for gb in O.chunk_batch(flat, n_global):
            opt.zero_grad()
            for mb in O.chunk_batch(gb, accum):
                out = policy.evaluate(mb["forward_inputs"]["states"], mb["forward_inputs"]["action"])
                bsz = out["logprobs"].shape[0]
                shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], logprob_type, A,
                                             loss_mask=mb.get("loss_mask"), loss_mask_sum=mb.get("loss_mask_sum"),
                                             values=out["values"], prev_values=mb["prev_values"], returns=mb["returns"])
                prox, ver = O.shape_decoupled_inputs(mb.get("proximal_logprobs"), mb.get("versions"), logprob_type, A, bsz,
                                                     shaped["logprobs"].shape)
                loss, m = O.decoupled_actor_critic_loss(
                    proximal_logprobs=prox, versions=ver, current_version=version + 1,
                    behave_weight_threshold=behave_weight_threshold, clip_ratio_low=clip_low, clip_ratio_high=clip_high,
                    clip_ratio_c=clip_ratio_c, value_clip=value_clip, huber_delta=huber_delta,
                    max_episode_steps=max_episode_steps, critic_warmup=False, **shaped)
                m = dict(m)
                ent_loss = torch.tensor(0.0)
                if entropy_bonus > 0:
                    ent = out["entropy"].reshape(bsz, -1, A).sum(dim=-1)
                    ent_loss = O.masked_mean(ent, shaped["loss_mask"])
                    loss = loss - entropy_bonus * ent_loss
                loss = loss / accum
                loss.backward()
                m["actor/entropy_loss"] = float(ent_loss.detach())
                m["actor/total_loss"] = float(loss.detach())
                metrics.append(m)
            gn = torch.nn.utils.clip_grad_norm_(policy.parameters(), clip_grad)
            if torch.isfinite(gn):
                opt.step()
            norms.append(float(gn))
    return metrics, norms


def iteration(policy, opt, env, eps, *, gamma, gae_lambda, seed, global_batch, update_epoch, auto_reset=True,
              max_update_steps=None, timings=None, rollout_epoch: int = 1, entropy_bonus: float = 0.0,
              pipeline: dict | None = None, critic_warmup_steps: int = 0, steps_done: int = 0, max_episode_steps=None,
              autocast: bool = False, entropy_type: str = "action_level", adv_type: str = "gae", loss_type: str = "actor_critic",
              reward_type: str = "action_level", logprob_type: str = "action_level", group_size: int = 1):
    """``pipeline`` = dict(stage_num=..., generator=<the rank's stateful shuffle generator>) selects
    runner.use_training_pipeline's data path: global-statistics normalisation, per-stage shuffles, and -- every
    micro-batch being available at once here -- the epoch-major schedule PipelineEmbodiedFSDPActor.run_training reduces
    to (fsdp_actor_worker_pipeline.py:84-160: fixed global batches, epoch 1 in arrival order, then each again)."""
    t0 = time.perf_counter()
    if pipeline is not None and rollout_epoch > 1:
        # every rollout epoch is its own learner batch (env_worker.py:1324-1330 sends per epoch): all epochs are rolled out with
        # the iteration's FROZEN weights, each gets its own statistics normalisation and stage shuffles (the generator keeps
        # running), the learner makes its first pass epoch by epoch and then revisits the stored global batches oldest first
        # (fsdp_actor_worker_pipeline.py:84-160) -- one epoch-major sweep over the concatenated rows
        T = env["rewards"].shape[0] // rollout_epoch
        flats, batches = [], []
        for e in range(rollout_epoch):
            sl = dict(obs=env["obs"][e * T:(e + 1) * T + 1], final_obs=env["final_obs"][e * T:(e + 1) * T],
                      rewards=env["rewards"][e * T:(e + 1) * T], dones=env["dones"][e * T:(e + 1) * T + 1])
            batches.append(rollout(policy, sl, eps[e * T:(e + 1) * T], gamma, auto_reset, autocast))
        for e in range(rollout_epoch):
            batches[e] = pipeline_advantages(batches[e], gamma, gae_lambda, auto_reset)
            T_, B_ = batches[e]["prev_logprobs"].shape[:2]
            flats.append(O.flatten_and_shuffle(batches[e], pipeline_permutation(T_, B_, pipeline["stage_num"], pipeline["generator"])))
        flat = {k: (dict((kk, torch.cat([f[k][kk] for f in flats], 0)) for kk in flats[0][k]) if isinstance(flats[0][k], dict)
                    else torch.cat([f[k] for f in flats], 0)) for k in flats[0]}
        metrics = update(policy, opt, None, seed=seed, global_batch=global_batch, update_epoch=update_epoch,
                         max_steps=max_update_steps, entropy_bonus=entropy_bonus, critic_warmup_steps=critic_warmup_steps,
                         steps_done=steps_done, max_episode_steps=max_episode_steps, autocast=autocast, entropy_type=entropy_type,
                         flat=flat)
        return batches, metrics
    batch = (rollout(policy, env, eps, gamma, auto_reset, autocast) if rollout_epoch == 1
             else rollout_epochs(policy, env, eps, gamma, rollout_epoch, auto_reset, autocast))
    t1 = time.perf_counter()
    perm = None
    if pipeline is None:
        batch = advantages(batch, gamma, gae_lambda, auto_reset, adv_type=adv_type, reward_type=reward_type, group_size=group_size)
    else:
        batch = pipeline_advantages(batch, gamma, gae_lambda, auto_reset)
        T_, B_ = batch["prev_logprobs"].shape[:2]
        perm = pipeline_permutation(T_, B_, pipeline["stage_num"], pipeline["generator"])
    t2 = time.perf_counter()
    metrics = update(policy, opt, batch, seed=seed, global_batch=global_batch, update_epoch=update_epoch,
                     max_steps=max_update_steps, entropy_bonus=entropy_bonus, perm=perm,
                     critic_warmup_steps=critic_warmup_steps, steps_done=steps_done, max_episode_steps=max_episode_steps,
                     autocast=autocast, entropy_type=entropy_type, logprob_type=logprob_type, reward_type=reward_type,
                     loss_type=loss_type)
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(rollout=t1 - t0, advantages=t2 - t1, update=t3 - t2, update_steps=len(metrics))
    return batch, metrics

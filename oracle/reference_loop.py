"""One embodied PPO iteration on the CPU with THE REFERENCE'S OWN FUNCTIONS (TEST / BASELINE INFRASTRUCTURE): its MLPPolicy
(predict_action_batch / default_forward, rlinf/models/embodiment/mlp_policy/mlp_policy.py), its calculate_adv_and_returns and
policy_loss with their built-in callees (rlinf/algorithms/registry.py:77-124), torch's clip_grad_norm_ and AdamW in the two
parameter groups FSDPModelManager.build_optimizer makes (fsdp_model_manager.py:501-590) -- loaded from /root/reference or the
staged oracle/_ref copy by oracle/reference_loader.py.  Only the control flow around them is restated (the reference's own loop
lives in Ray worker classes that cannot be imported here): rollout row alignment env_worker.py:1058-1306, bootstrap on
auto-reset :718-758, one randperm per run_training and minibatch chunking embodied_fsdp_actor_worker.py:483-589 -- the same
flow oracle/ppo_loop.py restates, which tests/test_reference_learner_loop.py pins against the reference's run_training.
bench.py's cpu_baseline (kind "reference") times this; nothing in rlinf_amd/ imports it."""

from __future__ import annotations

import time

import torch

from . import reference_loader as RL


def build(obs_dim: int, act_dim: int, lr: float = 3e-4, value_lr: float = 3e-4, seed: int = 1234):
    ref = RL.load()
    torch.manual_seed(seed)
    pol = ref.mlp_policy.MLPPolicy(obs_dim, act_dim, 1, True, False)
    opt = torch.optim.AdamW(
        [{"params": [p for n, p in pol.named_parameters() if "value_head" not in n], "lr": lr, "betas": (0.9, 0.999)},
         {"params": [p for n, p in pol.named_parameters() if "value_head" in n], "lr": value_lr, "betas": (0.9, 0.999)}],
        eps=1e-8, weight_decay=0.01)
    return ref, pol, opt


@torch.no_grad()
def rollout(ref, pol, env: dict, gamma: float, seed: int = 1):
    """T policy steps on B envs with the reference's predict_action_batch (its own torch.normal draw), values of the true
    terminal observations folded into the rewards of the envs that finished, one closing value row."""
    T, B = env["rewards"].shape
    torch.manual_seed(seed)
    rows = dict(states=[], action=[], logp=[], values=[], rewards=[], dones=[torch.zeros(B, 1, dtype=torch.bool)])
    obs = env["obs"][0]
    for t in range(T):
        _, res = pol.predict_action_batch({"states": obs}, mode="train")
        rows["states"].append(obs), rows["action"].append(res["forward_inputs"]["action"])
        rows["logp"].append(res["prev_logprobs"]), rows["values"].append(res["prev_values"])
        r = env["rewards"][t].clone().unsqueeze(-1)
        d = env["dones"][t + 1].unsqueeze(-1)
        if bool(d.any()):
            _, fin = pol.predict_action_batch({"states": env["final_obs"][t]}, mode="train")
            r[:, -1] += gamma * fin["prev_values"][:, 0] * d[:, -1]
        rows["rewards"].append(r), rows["dones"].append(d)
        obs = env["obs"][t + 1]
    _, last = pol.predict_action_batch({"states": obs}, mode="train")
    rows["values"].append(last["prev_values"])
    st = lambda k: torch.stack(rows[k], 0)  # noqa: E731
    return dict(rewards=st("rewards"), dones=st("dones"), prev_values=st("values"), prev_logprobs=st("logp"),
                states=st("states"), action=st("action"))


def advantages(ref, batch: dict, gamma: float, gae_lambda: float):
    out = ref.registry.calculate_adv_and_returns(
        task_type="embodied", adv_type="gae", rewards=batch["rewards"], dones=batch["dones"], values=batch["prev_values"],
        gamma=gamma, gae_lambda=gae_lambda, group_size=1, reward_type="action_level", loss_mask=None, loss_mask_sum=None,
        normalize_advantages=True)
    batch = dict(batch)
    batch.update(advantages=out["advantages"].contiguous(), returns=out["returns"].contiguous())
    return batch


def flatten_and_shuffle(batch: dict, seed: int):
    T, B = batch["prev_logprobs"].shape[:2]
    perm = torch.randperm(T * B, generator=torch.Generator().manual_seed(seed))
    flat = {}
    for k in ("states", "action", "prev_logprobs", "advantages", "returns"):
        flat[k] = batch[k].reshape(T * B, -1)[perm]
    flat["prev_values"] = batch["prev_values"][:-1].reshape(T * B, -1)[perm]
    return flat


def optimizer_step(ref, pol, opt, mb: dict, clip_grad: float = 0.5):
    """train_micro_batch + optimizer_step (embodied_fsdp_actor_worker.py:591-700, fsdp_model_manager.py:429-463)."""
    opt.zero_grad()
    out = pol.default_forward({"states": mb["states"], "action": mb["action"]})
    loss, metrics = ref.registry.policy_loss(
        loss_type="actor_critic", task_type="embodied", logprob_type="action_level", reward_type="action_level",
        single_action_dim=mb["action"].shape[-1], logprobs=out["logprobs"], values=out["values"], old_logprobs=mb["prev_logprobs"],
        advantages=mb["advantages"], returns=mb["returns"], prev_values=mb["prev_values"], clip_ratio_high=0.2,
        clip_ratio_low=0.2, value_clip=1.0, huber_delta=10.0, loss_mask=None, loss_mask_sum=None, max_episode_steps=50)
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(pol.parameters(), clip_grad)
    if torch.isfinite(gn):
        opt.step()
    return metrics


def timed_iteration(env: dict, *, obs_dim: int, act_dim: int, gamma: float, gae_lambda: float, global_batch: int,
                    update_epoch: int, warmup_steps: int = 3, timed_steps: int | None = 10, update_budget_s: float = 15.0):
    """-> phase timings of one iteration: the rollout and the advantage pass in full, then ``warmup_steps`` untimed optimizer
    steps and ``timed_steps`` timed ones (None: every step of the update phase -- n_mb x update_epoch -- as long as the running
    total stays under ``update_budget_s``; every minibatch step does identical work, so a truncated run extrapolates)."""
    ref, pol, opt = build(obs_dim, act_dim)
    t0 = time.perf_counter()
    batch = rollout(ref, pol, env, gamma)
    t1 = time.perf_counter()
    batch = advantages(ref, batch, gamma, gae_lambda)
    t2 = time.perf_counter()
    flat = flatten_and_shuffle(batch, 1234)
    n = flat["states"].shape[0]
    n_mb = n // global_batch
    total = n_mb * update_epoch
    want = total if timed_steps is None else timed_steps
    steps, spent = [], 0.0
    for k in range(warmup_steps + want):
        lo = (k % n_mb) * global_batch
        mb = {key: v[lo:lo + global_batch] for key, v in flat.items()}
        s0 = time.perf_counter()
        optimizer_step(ref, pol, opt, mb)
        dt = time.perf_counter() - s0
        steps.append(dt)
        if k >= warmup_steps:
            spent += dt
            if timed_steps is None and spent > update_budget_s and k + 1 - warmup_steps >= 10:
                break
    timed = steps[warmup_steps:]
    median = sorted(timed)[len(timed) // 2]
    return dict(rollout_s=t1 - t0, advantages_s=t2 - t1, update_s_per_step=median, update_steps_total=total,
                warmup_steps=warmup_steps, timed_steps=len(timed), step_times_s=[round(x, 4) for x in steps],
                timed_step_times_s=timed)

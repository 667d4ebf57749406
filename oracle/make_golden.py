"""Generate tests/golden/*.pt from the REAL reference (run in the build container only).

    python -m oracle.make_golden

TEST INFRASTRUCTURE.  Every fixture stores the seeded inputs AND the outputs the reference's own
code (executed on CPU from /root/reference, via oracle/reference_loader.py) produced for them, so the
oracle restatement and the HIP path can both be checked on machines without the reference tree.
Fixtures are deliberately small (a few hundred KB in total).
"""

from __future__ import annotations

import os

import torch

from oracle import reference_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rollout(seed, T, B, C, p_done):
    g = torch.Generator().manual_seed(seed)
    rewards = torch.rand(T, B, C, generator=g)
    values = torch.randn(T + 1, B, C, generator=g)
    dones = torch.rand(T + 1, B, C, generator=g) < p_done
    dones[0] = False
    return rewards, values, dones


def make_advantages(ref):
    cases = []
    grid = [  # (seed, T, B, C, p_done, gamma, lam, masked, normalize, adv_type, group, reward_type)
        (1234, 16, 64, 1, 0.02, 0.8, 0.9, False, True, "gae", 1, "action_level"),
        (1234, 16, 64, 1, 0.02, 0.99, 0.95, False, False, "gae", 1, "action_level"),
        (7, 50, 32, 1, 0.05, 0.8, 0.9, True, True, "gae", 1, "action_level"),
        (8, 12, 16, 4, 0.05, 0.99, 0.95, True, True, "gae", 1, "action_level"),
        (9, 12, 16, 4, 0.05, 0.99, 0.95, False, True, "gae", 1, "chunk_level"),
        (10, 33, 128, 1, 0.0, 0.99, 0.95, False, True, "gae", 1, "action_level"),
        (11, 20, 64, 1, 0.08, 1.0, 1.0, True, True, "grpo", 8, "action_level"),
        (12, 10, 24, 2, 0.10, 1.0, 1.0, True, True, "grpo", 4, "action_level"),
        (13, 128, 8, 1, 0.02, 0.8, 0.9, False, True, "gae", 1, "action_level"),  # configs[0]: 8 envs
    ]
    for seed, T, B, C, p, gamma, lam, masked, norm, adv_type, G, rtype in grid:
        rewards, values, dones = rollout(seed, T, B, C, p)
        lm = lms = None
        if masked:
            lm, lms = ref.metric_utils.compute_loss_mask(dones)
            if rtype == "chunk_level":
                lm, lms = lm.any(dim=-1, keepdim=True), lms[..., -1:]
        vals = values[..., :1] if rtype == "chunk_level" else values
        kw = dict(task_type="embodied", adv_type=adv_type, rewards=rewards, dones=dones,
                  values=vals if adv_type == "gae" else None, gamma=gamma, gae_lambda=lam, group_size=G,
                  reward_type=rtype, loss_mask=lm, loss_mask_sum=lms)
        if adv_type == "gae":
            kw["normalize_advantages"] = norm
        out = ref.registry.calculate_adv_and_returns(**kw)
        cases.append(dict(
            params=dict(seed=seed, T=T, B=B, C=C, p_done=p, gamma=gamma, gae_lambda=lam, masked=masked,
                        normalize_advantages=norm, adv_type=adv_type, group_size=G, reward_type=rtype),
            rewards=rewards, values=vals, dones=dones,
            loss_mask=None if lm is None else lm.contiguous(),
            loss_mask_sum=None if lms is None else lms[:1].contiguous(),
            advantages=out["advantages"].contiguous(),
            returns=out["returns"].contiguous() if "returns" in out else None))
    torch.save(cases, os.path.join(OUT, "advantages.pt"))


def make_loss_mask(ref):
    cases = []
    for seed, T, B, C, p in [(1, 12, 16, 1, 0.05), (2, 12, 16, 4, 0.05), (3, 8, 8, 1, 0.0), (4, 8, 8, 2, 0.6),
                             (5, 128, 64, 1, 0.02)]:
        _, _, dones = rollout(seed, T, B, C, p)
        m, s = ref.metric_utils.compute_loss_mask(dones)
        cases.append(dict(dones=dones, loss_mask=m.contiguous(), loss_mask_sum_row=s[0].contiguous()))
    torch.save(cases, os.path.join(OUT, "loss_mask.pt"))


def make_losses(ref):
    cases = []
    idx = 0
    for logprob_type in ("action_level", "token_level", "chunk_level"):
        for masked in (False, True):
            for variant in ("plain", "dual", "logclip", "warmup", "ratio_agg"):
                if variant == "ratio_agg" and not masked:
                    continue
                idx += 1
                g = torch.Generator().manual_seed(100 + idx)
                mb, C, A = 48, 2, 8
                lp = (torch.randn(mb, C * A, generator=g) * 0.3 - 1.0).requires_grad_(True)
                old = lp.detach() + torch.randn(mb, C * A, generator=g) * 0.1
                adv = torch.randn(mb, C, generator=g)
                v = torch.randn(mb, C, generator=g).requires_grad_(True)
                pv = v.detach() + torch.randn(mb, C, generator=g) * 0.7
                ret = torch.randn(mb, C, generator=g) * 3
                ret[0, 0] = 40.0  # force the linear branch of the Huber loss
                lm = (torch.rand(mb, C, generator=g) < 0.7) if masked else None
                lms = torch.randint(1, 50, (mb, 1), generator=g).expand(mb, C).contiguous() if masked else None
                if logprob_type == "chunk_level":
                    adv, pv, ret = adv[:, 0].contiguous(), pv[:, 0].contiguous(), ret[:, 0].contiguous()
                    v = v.detach()[:, 0].contiguous().requires_grad_(True)
                    lm = None if lm is None else lm[:, 0].contiguous()
                    lms = None if lms is None else lms[:, 0].contiguous()
                extra = {}
                if variant == "dual":
                    extra["clip_ratio_c"] = 3.0
                if variant == "logclip":
                    extra.update(clip_log_ratio_min=-0.05, clip_log_ratio_max=0.05)
                if variant == "warmup":
                    extra["critic_warmup"] = True
                mes = 50 if variant == "ratio_agg" else None
                loss, metrics = ref.registry.policy_loss(
                    loss_type="actor_critic", task_type="embodied", logprob_type=logprob_type,
                    reward_type="action_level", single_action_dim=A, logprobs=lp, values=v,
                    old_logprobs=old, advantages=adv, returns=ret, prev_values=pv, clip_ratio_high=0.2,
                    clip_ratio_low=0.2, value_clip=1.0, huber_delta=10.0, loss_mask=lm, loss_mask_sum=lms,
                    max_episode_steps=mes, **extra)
                g_lp, g_v = torch.autograd.grad(loss, [lp, v], allow_unused=True)
                cases.append(dict(
                    params=dict(logprob_type=logprob_type, masked=masked, variant=variant, action_dim=A,
                                clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0,
                                max_episode_steps=mes, **extra),
                    logprobs=lp.detach(), old_logprobs=old, advantages=adv, values=v.detach(),
                    prev_values=pv, returns=ret, loss_mask=lm, loss_mask_sum=lms,
                    loss=loss.detach(), grad_logprobs=g_lp, grad_values=g_v,
                    metrics={k: float(val) for k, val in metrics.items()}))
    torch.save(cases, os.path.join(OUT, "losses.pt"))


PERTURB_SEED = 99


def perturbation(shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(PERTURB_SEED + len(shape) * 1000 + shape[-1])) * 0.01


def make_policy(ref):
    torch.manual_seed(4321)
    pol = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    sd = {k: v.clone() for k, v in pol.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    states = torch.randn(96, 42, generator=g)
    torch.manual_seed(777)
    acts, res = pol.predict_action_batch({"states": states}, mode="train")
    torch.manual_seed(777)
    eps = torch.randn(96, 8)
    acts_eval, res_eval = pol.predict_action_batch({"states": states}, mode="eval")
    stored_action = res["forward_inputs"]["action"].clone()
    # perturb the policy a little so the training forward sees a non-unit ratio; the perturbation is
    # regenerated by the tests from PERTURB_SEED (keeps the fixture small)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(perturbation(p.shape))
    out = pol.default_forward({"states": states, "action": stored_action})
    adv = torch.randn(96, 1, generator=g)
    ret = torch.randn(96, 1, generator=g)
    loss, metrics = ref.registry.policy_loss(
        loss_type="actor_critic", task_type="embodied", logprob_type="action_level",
        reward_type="action_level", single_action_dim=8, logprobs=out["logprobs"], values=out["values"],
        old_logprobs=res["prev_logprobs"], advantages=adv, returns=ret, prev_values=res["prev_values"],
        clip_ratio_high=0.2, clip_ratio_low=0.2, value_clip=1.0, huber_delta=10.0, loss_mask=None,
        loss_mask_sum=None, max_episode_steps=50)
    names = [n for n, _ in pol.named_parameters()]
    opt = torch.optim.AdamW(
        [{"params": [p for n, p in pol.named_parameters() if "value_head" not in n], "lr": 3e-4,
          "betas": (0.9, 0.999)},
         {"params": [p for n, p in pol.named_parameters() if "value_head" in n], "lr": 3e-4,
          "betas": (0.9, 0.999)}], eps=1e-8, weight_decay=0.01)
    opt.zero_grad()
    loss.backward()
    grads = {n: p.grad.clone() for n, p in pol.named_parameters()}
    gnorm = torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.5)
    opt.step()
    # after-step parameters: every 16th element of each tensor (the AdamW update is elementwise)
    sd_after = {k: v.reshape(-1)[::16].clone() for k, v in pol.state_dict().items()}
    torch.save(dict(
        param_names=names, state_dict=sd, states=states, eps=eps,
        action=acts.reshape(96, 8), prev_logprobs=res["prev_logprobs"], prev_values=res["prev_values"],
        eval_action=acts_eval.reshape(96, 8), eval_logprobs=res_eval["prev_logprobs"],
        perturb_seed=PERTURB_SEED, train_logprobs=out["logprobs"].detach(),
        train_entropy=out["entropy"].detach(), train_values=out["values"].detach(),
        advantages=adv, returns=ret, loss=loss.detach(), metrics={k: float(v) for k, v in metrics.items()},
        grads=grads, grad_norm=gnorm.detach(), params_after_step_stride16=sd_after,
    ), os.path.join(OUT, "policy.pt"))


def make_shuffle(ref):
    g = torch.Generator().manual_seed(2)
    T, B = 6, 20
    batch = dict(rewards=torch.rand(T, B, 1, generator=g), dones=torch.rand(T + 1, B, 1, generator=g) < 0.1,
                 prev_values=torch.randn(T + 1, B, 1, generator=g),
                 prev_logprobs=torch.randn(T, B, 8, generator=g),
                 forward_inputs=dict(states=torch.randn(T, B, 42, generator=g),
                                     action=torch.randn(T, B, 8, generator=g)))
    perm = torch.randperm(T * B, generator=torch.Generator().manual_seed(1234))
    out = ref.nested.process_nested_dict_for_train(batch, perm)
    torch.save(dict(batch=batch, perm=perm, out=out), os.path.join(OUT, "shuffle.pt"))


def token_batch(seed, bsz, seq, vocab, p_on=0.7, zero_first=False, scale=3.0):
    """Seeded reasoning micro-batch: response-aligned logits, sampled-ish labels, rollout-time stats."""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(bsz, seq, vocab, generator=g) * scale
    labels = torch.randint(0, vocab, (bsz, seq), generator=g)
    best = logits.argmax(-1)
    labels = torch.where(torch.rand(bsz, seq, generator=g) < 0.5, best, labels)  # half the tokens are the mode
    old = torch.log_softmax(logits, -1).gather(-1, labels[..., None])[..., 0] + 0.3 * torch.randn(bsz, seq, generator=g)
    ref_lp = old + 0.2 * torch.randn(bsz, seq, generator=g)
    adv = torch.randn(bsz, seq, generator=g)
    lens = torch.randint(1, seq + 1, (bsz,), generator=g)
    mask = (torch.arange(seq)[None, :] < lens[:, None]) & (torch.rand(bsz, seq, generator=g) < p_on + 0.3)
    if zero_first:
        mask[0] = False
    rewards = torch.randn(bsz, generator=g)
    return dict(logits=logits, labels=labels, old_logprobs=old, ref_logprobs=ref_lp, advantages=adv,
                loss_mask=mask, rewards=rewards)


TOKEN_GRID = [  # (seed, bsz, seq, vocab, zero_first, agg, temperature, entropy_bonus, kl_beta, kl_type, clip_c, clips)
    (31, 8, 12, 517, False, "token-mean", 1.0, 0.0, 0.0, "low_var_kl", 3.0, (None, None)),
    (32, 8, 12, 517, False, "seq-mean-token-sum", 0.7, 0.01, 0.05, "low_var_kl", 3.0, (None, None)),
    (33, 8, 12, 517, False, "seq-mean-token-mean", 1.3, 0.02, 0.1, "k2", None, (-0.5, 0.4)),
    (34, 4, 9, 1000, True, "token-mean", 1.0, 0.01, 0.02, "k1", 3.0, (None, None)),
    (35, 4, 9, 1000, False, "token-mean", 0.6, 0.0, 0.03, "abs", 2.0, (None, 0.3)),
]


def make_token_path(ref):
    """t1-t6 of oracle/token_oracle.py: every piece produced by the reference's own functions; the micro-batch
    composition follows FSDPActor.training_step line by line (fsdp_actor_worker.py:694-781) using them."""
    U, A, L, R = ref.utils, ref.algo_utils, ref.losses, ref.registry
    cases = []
    for seed, bsz, seq, vocab, zero_first, agg_name, temp, bonus, beta, kl_type, clip_c, (cmin, cmax) in TOKEN_GRID:
        b = token_batch(seed, bsz, seq, vocab, zero_first=zero_first)
        agg = U.get_loss_agg_func(agg_name)
        out = {}
        for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            logits = b["logits"].to(dt).clone().requires_grad_(True)
            scaled = logits / temp
            logprobs = U.compute_logprobs_from_logits(scaled, b["labels"])
            entropy = U.compute_entropy_from_logits(scaled)
            loss, metrics = L.compute_ppo_actor_loss(
                logprobs=logprobs, old_logprobs=b["old_logprobs"], advantages=b["advantages"], clip_ratio_low=0.2,
                clip_ratio_high=0.28, loss_mask=b["loss_mask"], clip_ratio_c=clip_c, loss_agg_func=agg,
                clip_log_ratio_min=cmin, clip_log_ratio_max=cmax, fast_path_zero_loss_mask=True)
            entropy_loss = agg(entropy, mask=b["loss_mask"])
            if bonus > 0:
                loss = loss - bonus * entropy_loss
            kl_loss = torch.tensor(0.0)
            if beta > 0:
                kld = A.kl_penalty(b["ref_logprobs"], logprobs, kl_type)
                kl_loss = agg(kld, b["loss_mask"])
                loss = loss + kl_loss * beta
            final = loss.detach().clone()
            (loss / 2).backward()
            out[tag] = dict(logprobs=logprobs.detach(), entropy=entropy.detach(), final_loss=final,
                            entropy_loss=entropy_loss.detach(), kl_loss=kl_loss.detach(),
                            metrics={k: v.detach().clone() for k, v in metrics.items()}, d_logits=logits.grad.clone())
        adv = R.calculate_adv_and_returns(task_type="reasoning", adv_type="grpo", rewards=b["rewards"],
                                          loss_mask=b["loss_mask"], group_size=4)
        kls = {k: A.kl_penalty(b["ref_logprobs"], b["old_logprobs"], k) for k in ("k1", "abs", "k2", "k3")}
        cases.append(dict(params=dict(seed=seed, bsz=bsz, seq=seq, vocab=vocab, zero_first=zero_first, loss_agg=agg_name,
                                      temperature=temp, entropy_bonus=bonus, kl_beta=beta, kl_penalty_type=kl_type,
                                      clip_ratio_c=clip_c, clip_log_ratio_min=cmin, clip_log_ratio_max=cmax,
                                      clip_ratio_low=0.2, clip_ratio_high=0.28, gradient_accumulation=2, group_size=4),
                          out=out, grpo_advantages=adv[0], kl_terms=kls))
    torch.save(cases, os.path.join(OUT, "token_path.pt"))


def patch_states(seed, scale=1):
    """Seeded (before, after) state dicts for the weight-patch fixtures: mixed dtypes and ranks, a NaN (resent on every
    sync: NaN != NaN), a sign flip of zero (not a change under torch.ne), an untouched tensor, a dense and a sparse update."""
    g = torch.Generator().manual_seed(seed)
    R = 37 * scale
    before = {
        "backbone.weight": torch.randn(R, 300, generator=g),
        "backbone.bias": torch.randn(300, generator=g),
        "logstd": torch.tensor(-0.5),
        "conv": torch.randn(4, 5, 6, generator=g).bfloat16(),
        "steps": torch.arange(40).reshape(4, 10),
        "head.weight": torch.randn(600 * scale, 70, generator=g).half(),
        "frozen": torch.randn(11, 13, generator=g),
        "flags": torch.rand(9, 31, generator=g) < 0.5,
    }
    before["backbone.weight"][5, 5] = 0.0
    after = {k: v.clone() for k, v in before.items()}
    after["backbone.weight"][torch.rand(R, 300, generator=g) < 0.01] += 1
    after["backbone.weight"][3, 7] = float("nan")
    after["backbone.weight"][5, 5] = -0.0
    after["backbone.bias"][299] = 7.0
    after["logstd"] = torch.tensor(-0.25)
    after["conv"][1, 2, 3] = 5
    after["steps"][2, 3] = -7
    after["head.weight"][torch.rand(600 * scale, 70, generator=g) < 0.3] *= 2
    after["flags"][8, 30] = ~after["flags"][8, 30]
    return before, after


def make_weight_patches():
    """tests/golden/weight_patch.pt: what the reference's own GPUSnapshotPatchBuilder (run on CPU tensors through
    oracle/reference_loader.load_weight_syncer) emits for patch_states(seed), delta encoding on and off, two syncs each
    (the second contains only the NaN entry), plus a sender whose receiver holds bf16 copies."""
    m = reference_loader.load_weight_syncer()
    cases = []
    for seed, delta, narrow in ((1, True, False), (1, False, False), (2, True, True)):
        before, after = patch_states(seed)
        keys = list(before)
        names = [k for k in keys if k != "frozen"] + ["frozen"]
        snap = {}
        for k, v in before.items():
            view = m.as_coo_2d_view(v)[0]
            snap[k] = view.to(torch.bfloat16 if (narrow and v.dtype == torch.float32) else view.dtype, copy=True)
        builder = m.GPUSnapshotPatchBuilder(snap, keys, names, {k: v.shape for k, v in before.items()},
                                            torch.device("cpu"), delta)
        patches = []
        for version in (11, 12):
            p = builder.create_patch(after, version)
            patches.append({f: getattr(p, f).clone() for f in ("version", "ordinals", "nnz_per_tensor", "rows", "cols",
                                                                "values")})
        cases.append(dict(params=dict(seed=seed, delta=delta, narrow=narrow, keys=keys, names=names), patches=patches))
    torch.save(cases, os.path.join(OUT, "weight_patch.pt"))


BUCKET_CASES = (dict(seed=1, bucket_size=1, bucket_dtype=None, is_agent=False),          # every tensor its own bucket
                dict(seed=1, bucket_size=50_000, bucket_dtype="bf16", is_agent=False),   # a few tensors per bucket
                dict(seed=2, bucket_size=10 ** 9, bucket_dtype="fp16", is_agent=True),   # one bucket, agent key rewrite
                dict(seed=3, bucket_size=20_000, bucket_dtype="fp32", is_agent=True))


def bucket_state(seed):
    """The sender's state dict of the bucket fixtures: patch_states' "after" (mixed dtypes and ranks, a NaN, a 0-dim
    tensor, bool and int64 tensors) plus the keys the bucket syncer treats specially."""
    state = patch_states(seed)[1]
    g = torch.Generator().manual_seed(100 + seed)
    state["model.language_model.layers.0.w"] = torch.randn(16, 24, generator=g)
    state["model.visual.proj"] = torch.randn(8, 8, generator=g).bfloat16()
    state["decoder._extra_state"] = torch.zeros(3)
    state["empty"] = torch.zeros(0, 4)
    names = [k for k in state if k != "frozen"] + ["not_in_state"]
    return state, names


def make_weight_buckets():
    """tests/golden/weight_bucket.pt: the buckets the reference's own BucketWeightSyncer.sync sends (bucket_device cpu,
    run through oracle/reference_loader.load_weight_syncer) for bucket_state(seed) under BUCKET_CASES."""
    import asyncio
    import sys

    reference_loader.load_weight_syncer()
    m = sys.modules["rlinf.hybrid_engines.weight_syncer.bucket_syncer"]
    cases = []
    for case in BUCKET_CASES:
        state, names = bucket_state(case["seed"])
        syncer = m.BucketWeightSyncer(case["bucket_size"], case["bucket_dtype"], "cpu", is_agent=case["is_agent"])
        sent = []

        async def send(bucket):
            sent.append({k: v.clone() for k, v in bucket.items()})

        async def run():
            await syncer.init_sender(state, names, send)
            await syncer.sync(state, send, 7)

        asyncio.run(run())
        cases.append(dict(params=dict(case), buckets=sent))
    torch.save(cases, os.path.join(OUT, "weight_bucket.pt"))


REINPP_GRID = (dict(seed=1, bsz=8, seq=6, kl_beta=0.0, kl="", masks="prefix"),
               dict(seed=2, bsz=8, seq=37, kl_beta=0.001, kl="low_var_kl", masks="ragged"),   # first position False in places
               dict(seed=3, bsz=12, seq=1100, kl_beta=0.5, kl="kl", masks="ragged"),          # more than one 1024-token tile
               dict(seed=4, bsz=4, seq=64, kl_beta=0.05, kl="abs", masks="empty_rows"),
               dict(seed=5, bsz=6, seq=20, kl_beta=0.2, kl="mse", masks="all_false"))


def reinpp_batch(seed, bsz, seq, masks, **_):
    """rewards [bsz], loss_mask [bsz, seq], logprob / ref_logprob [bsz, seq] for the Reinforce++ fixtures."""
    g = torch.Generator().manual_seed(seed)
    rewards = torch.randn(bsz, generator=g)
    lens = torch.randint(1, seq + 1, (bsz,), generator=g)
    mask = torch.arange(seq)[None, :] < lens[:, None]
    if masks in ("ragged", "empty_rows"):
        start = torch.randint(0, 3, (bsz,), generator=g)  # some masks do not start at position 0 (prompt tokens in the window)
        mask &= torch.arange(seq)[None, :] >= start[:, None]
    if masks == "empty_rows":
        mask[1] = False
    if masks == "all_false":
        mask[:] = False
    logprob = -torch.rand(bsz, seq, generator=g) * 3
    ref_logprob = logprob + 0.3 * torch.randn(bsz, seq, generator=g)
    return rewards, mask, logprob, ref_logprob


def make_reinpp(ref):
    """tests/golden/reinpp.pt: calculate_adv_and_returns(task_type="reasoning", adv_type="reinpp") of the reference."""
    import sys
    reg = sys.modules["rlinf.algorithms.registry"]
    cases = []
    for p in REINPP_GRID:
        rewards, mask, lp, rlp = reinpp_batch(**p)
        adv, ret = reg.calculate_adv_and_returns(task_type="reasoning", adv_type="reinpp", rewards=rewards.clone(), loss_mask=mask,
                                                 group_size=2, kl_beta=p["kl_beta"], logprob=lp, ref_logprob=rlp,
                                                 kl_penalty_type=p["kl"], use_reinpp_baseline=False)
        assert ret is None
        cases.append(dict(params=dict(p), advantages=adv.clone()))
    torch.save(cases, os.path.join(OUT, "reinpp.pt"))


def main():
    ref = reference_loader.load()
    os.makedirs(OUT, exist_ok=True)
    make_advantages(ref)
    make_loss_mask(ref)
    make_losses(ref)
    make_policy(ref)
    make_shuffle(ref)
    make_token_path(ref)
    make_weight_patches()
    make_weight_buckets()
    make_reinpp(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

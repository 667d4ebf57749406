"""The learner of asynchronous (decoupled) PPO: same method names and data path as
rlinf/workers/actor/async_ppo_fsdp_worker.py -- ``compute_advantages_and_returns`` (:185-213), ``compute_proximal_logprobs``
(:215-272) and ``run_training`` (:274-497).

What differs from the synchronous learner (embodied_fsdp_actor_worker.py) is data, not machinery: trajectories carry the
policy ``versions`` they were sampled with; the loss is ``decoupled_actor_critic`` (clip against a proximal policy given,
recomputed, or interpolated from the version distance; importance-weight by exp(proximal - behaviour)); advantages are
normalised a second time over the whole shuffled buffer with cross-rank statistics (``masked_normalization``); dual clip is
on by default (clip_ratio_c 3.0).  The reference's receive thread, rollout store and staleness bookkeeping (:84-183) are
control plane and not mirrored: ``recv_rollout_trajectories`` of the base class fills ``rollout_batch`` directly.
"""

from __future__ import annotations

import torch

from ... import ops
from ..._lib import DPPO_OUT_NAMES, PPO_ACTOR_GRAD_SCALE, PPO_OUT_FLOATS
from ...algorithms.losses import _CRITIC_KEYS, _DECOUPLED_KEYS, _EV_MAP, explained_variance_from_stats
from ...algorithms.registry import calculate_adv_and_returns
from ...scheduler import all_reduce_flat_
from ...utils.pending import PendingMetrics
from .embodied_fsdp_actor_worker import CRITIC_EXPLAINED_VARIANCE_KEY, EmbodiedFSDPActor


class AsyncPPOEmbodiedFSDPActor(EmbodiedFSDPActor):
    """Embodied learner for async PPO / decoupled actor-critic training."""

    def compute_advantages_and_returns(self) -> dict:
        """:185-213 -- GAE against the proximal values when the batch carries them."""
        alg, b = self.cfg.algorithm, self.rollout_batch
        proximal_values = b.get("proximal_values")
        out = calculate_adv_and_returns(
            task_type=self.cfg.runner.task_type, adv_type=alg.adv_type, rewards=b["rewards"], dones=b["dones"],
            values=proximal_values if proximal_values is not None else b.get("prev_values"), gamma=alg.get("gamma", 1),
            gae_lambda=alg.get("gae_lambda", 1), group_size=alg.get("group_size", 8), reward_type=alg.reward_type,
            loss_mask=b.get("loss_mask"), loss_mask_sum=b.get("loss_mask_sum"))
        b.update(out)
        return self._rollout_metrics(b)

    @torch.no_grad()
    def compute_proximal_logprobs(self) -> None:
        """:215-272 -- log-probs of the stored actions under the CURRENT weights, in micro-batch sized pieces, un-shuffled."""
        b, m = self.rollout_batch, self.model
        T, B = b["prev_logprobs"].shape[:2]
        states = b["forward_inputs"]["states"].reshape(T * B, -1)
        action = b["forward_inputs"]["action"].reshape(T * B, -1)
        out = torch.empty((T * B, m.layout.act_dim), dtype=torch.float32, device=self.device)
        step = int(self.cfg.actor.micro_batch_size)
        for lo in range(0, T * B, step):
            lp = ops.mlp_train_fwd(m.flat.data, m.packed(), m.layout, states[lo:lo + step], action[lo:lo + step])[0]
            out[lo:lo + lp.shape[0]] = lp
        b["proximal_logprobs"] = out.view(T, B, *b["prev_logprobs"].shape[2:])

    # ---- update ---------------------------------------------------------------------------------------------------
    def _flatten_shuffle_normalize(self):
        """flatten_rollout_batch_for_train (:42-69) with one randperm seeded actor.seed + rank (:283-285), then
        masked_normalization of the advantages over the whole buffer, statistics summed over ranks (:292-296)."""
        b = self.rollout_batch
        T, B = b["prev_logprobs"].shape[:2]
        N = T * B
        pkey = ("perm", N)
        if pkey not in self._ws:
            g = torch.Generator()
            g.manual_seed(int(self.cfg.actor.seed) + self._rank)
            self._ws[pkey] = torch.randperm(N, generator=g).to(self.device)
        fields = {"states": b["forward_inputs"]["states"], "action": b["forward_inputs"]["action"],
                  "prev_logprobs": b["prev_logprobs"], "advantages": b["advantages"], "prev_values": b["prev_values"][:-1],
                  "returns": b.get("returns"), "loss_mask": b.get("loss_mask"), "versions": b.get("versions"),
                  "proximal_logprobs": b.get("proximal_logprobs"), "proximal_values": b.get("proximal_values")}
        if b.get("loss_mask_sum") is not None:
            fields["loss_mask_sum"] = b["loss_mask_sum"].contiguous()
        fields = {k: v for k, v in fields.items() if v is not None}
        if "versions" in fields:
            fields["versions"] = fields["versions"].float()
        src = [v.reshape(N, *v.shape[2:]).contiguous() for v in fields.values()]
        # persistent destinations, keyed by every field's name, row shape and dtype: the prepared launches and the captured
        # hipGraph of the fused path read the same addresses every iteration
        key = ("ashuf", N, tuple((n, tuple(t.shape[1:]), t.dtype) for n, t in zip(fields, src)))
        if key not in self._ws:
            self._ws[key] = [torch.empty_like(t) for t in src] + [torch.empty(N, *src[list(fields).index("advantages")].shape[1:],
                                                                              dtype=torch.float32, device=self.device)]
        flat = dict(zip(fields, ops.gather_rows(src, self._ws[pkey], self._ws[key][:-1])))
        if self.cfg.algorithm.get("normalize_advantages", True):
            adv, mask = flat["advantages"], flat.get("loss_mask")
            if mask is not None:
                assert mask.dim() == adv.dim() and mask.shape == adv.shape, (mask.shape, adv.shape)
            stats = ops.masked_stats(adv, mask)
            all_reduce_flat_(stats, self.ctx)  # factor, x_sum, x_sum_sq in ONE call (the reference issues three)
            flat["advantages"] = ops.masked_normalize(adv, mask, stats, out=self._ws[key][-1])
        return flat, N

    # ---- the fused path: rlx_ppo_step with the decoupled loss + deferred actor scale, prepared launches, hipGraph -------
    def _fused_update_ok(self, flat: dict) -> bool:
        """No critic warm-up, loss-mask sums already at the advantage shape.  (Any world size: every rank applies its own
        micro-batches' actor scales where its slabs are collapsed -- the xGMI staging launch or the slab sum in front of RCCL.)"""
        msum = flat.get("loss_mask_sum")
        return (self.fused_step and self.critic_warmup_steps == 0
                and (msum is None or (msum.dtype == torch.int64 and msum.numel() == flat["advantages"].numel())))

    def _decoupled_loss_params(self):
        alg, m = self.cfg.algorithm, self.cfg.actor.model
        return ops.make_ppo_params(
            logprob_type=alg.logprob_type, action_dim=int(m.get("action_dim", 7)), chunks=int(m.get("num_action_chunks", 1)),
            clip_ratio_low=alg.clip_ratio_low, clip_ratio_high=alg.clip_ratio_high, value_clip=alg.get("value_clip"),
            huber_delta=alg.get("huber_delta"), max_episode_steps=self.cfg.env.train.get("max_episode_steps"),
            clip_ratio_c=alg.get("clip_ratio_c", 3.0), critic_warmup=False, has_critic=True,
            reward_type=alg.get("reward_type", "action_level"))

    def _entropy_bonus_deferred(self, mb: dict, g: torch.Tensor, row: torch.Tensor, ent_row: torch.Tensor):
        """:449-462 behind a decoupled fused step: the decoupled row keeps slot 19 for the average version, so the bonus goes
        through a scratch row; the logstd slab is in sum form -> the bonus is pre-divided by the row's actor scale."""
        ent_row.zero_()
        ent_row[18] = row[DPPO_OUT_NAMES["mask_count"]]
        self._entropy_bonus(mb, g, ent_row, actor_scale=row[PPO_ACTOR_GRAD_SCALE:PPO_ACTOR_GRAD_SCALE + 1])
        row[PPO_OUT_FLOATS] = ent_row[0]
        row[PPO_OUT_FLOATS + 1] = ent_row[19]

    def _fused_plan(self, flat: dict, N: int, rows: torch.Tensor, norms: torch.Tensor, grads: torch.Tensor, ws: dict, n_global: int,
                    per_rank: int, accum: int, micro: int) -> list:
        """Every launch of the update phase marshalled once: [(micro-batch calls, AdamW call)] per optimizer step."""
        m, o, alg = self.model, self.cfg.actor.optim, self.cfg.algorithm
        epochs = int(alg.get("update_epoch", 1))
        pkey = ("aplan", N, micro, accum, epochs, tuple((k, t.data_ptr()) for k, t in flat.items()), rows.data_ptr())
        if self._ws.get("aplan_key") == pkey:
            return self._ws["aplan"]
        bf16 = m.compute_dtype == torch.bfloat16
        tiles = m.tiles() if self.optimizer_writes_tiles else None
        lp = self._decoupled_loss_params()
        bonus = float(alg.get("entropy_bonus", 0) or 0)
        ent_row = torch.zeros(PPO_OUT_FLOATS, device=self.device)
        plan, step = [], 0
        for _ in range(epochs):
            for i in range(n_global):
                calls, step_rows = [], rows[step * accum:(step + 1) * accum]
                for j in range(accum):
                    lo = i * per_rank + j * micro
                    mb = {k: v[lo:lo + micro] for k, v in flat.items()}
                    if mb.get("proximal_values") is not None:  # :419-421
                        mb["prev_values"] = mb["proximal_values"]
                    if mb.get("loss_mask") is not None:
                        mb["loss_mask"] = mb["loss_mask"].view(torch.uint8)
                    row, g = step_rows[j], grads[j * ws["slabs"]:(j + 1) * ws["slabs"]]
                    dec = ops.decoupled_step_args(lp, mb, current_version=int(self.version) + 1,
                                                  behave_weight_threshold=alg.get("behave_weight_threshold"),
                                                  current_version_dev=self._version_dev)
                    calls.append(ops.PreparedPpoStep(m.flat.data, m.layout, lp, mb, g, row[:PPO_OUT_FLOATS], ws["step_ws"],
                                                     grad_out=1.0 / accum, tiles=tiles, bf16=bf16, decoupled=dec))
                    if bonus > 0:
                        calls.append(lambda _s, mb=mb, g=g, r=row: self._entropy_bonus_deferred(mb, g, r, ent_row))
                multi, xg = self._exchange, self._xgmi
                adam = ops.PreparedAdamw(
                    m.flat.data, self.grad_flat if (multi and xg is None) else grads, self.exp_avg, self.exp_avg_sq, self.groups,
                    betas=(o.adam_beta1, o.adam_beta2), eps=o.adam_eps, weight_decay=o.weight_decay, max_grad_norm=o.clip_grad,
                    grad_scale=1.0 / self._world_size if multi else 1.0, stats=norms[step], step_state=self.step_state,
                    workspace=self.adamw_ws, tile_layout=m.layout if tiles is not None else None, tiles=tiles, xgmi=xg,
                    grad_flat=self.grad_flat if xg is not None else None,
                    deferred=ops.deferred_actor_scale(m.layout, step_rows[:, :PPO_OUT_FLOATS], accum),
                    deferred_in_caller=multi and xg is None, sync=self.adamw_sync)
                plan.append((calls, adam))
                step += 1
        self._ws["aplan_key"], self._ws["aplan"] = pkey, plan
        return plan

    def _run_training_fused(self, flat: dict, N: int, n_global: int, per_rank: int, accum: int, micro: int) -> dict:
        alg, m = self.cfg.algorithm, self.model
        n_steps = n_global * int(alg.get("update_epoch", 1))
        ws = self._minibatch_workspace(micro)
        gkey = ("grads", micro, accum)
        if gkey not in self._ws:
            self._ws[gkey] = torch.zeros((ws["slabs"] * accum, m.n_params), dtype=torch.float32, device=self.device)
        grads = self._ws[gkey]
        mkey = ("arows", n_steps, accum)
        if mkey not in self._ws:
            self._ws[mkey] = (torch.zeros(n_steps * accum, PPO_OUT_FLOATS + 2, device=self.device),
                              torch.zeros(n_steps, 2, device=self.device))
        rows, norms = self._ws[mkey]
        if getattr(self, "_version_dev", None) is None:
            self._version_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._version_dev.fill_(float(int(self.version) + 1))  # read by the launches when they EXECUTE: a replayed graph sees it
        self._grad_out_host = 1.0 / accum
        plan = self._fused_plan(flat, N, rows, norms, grads, ws, n_global, per_rank, accum, micro)
        gkey2 = ("agraph", self._ws["aplan_key"])
        # (at world_size > 1 only over the xGMI exchange -- pure kernels; RCCL capture needs the all-rank agreement of the base class)
        use_graph = self.enable_hip_graph and self.lr_scheduler.is_static and (not self._exchange or self._xgmi is not None)
        self._lr_log = []
        if use_graph and self._ws.get("agraph_key") == gkey2:
            self._ws["agraph"].replay()
            m.mark_updated(tiles_fresh=self.optimizer_writes_tiles)
            self.optimizer_steps += n_steps
        else:
            self._exec_plan(plan, grads)  # real run (also the warm-up before a capture)
            if use_graph:
                torch.cuda.synchronize(self.device)
                before, g = self.optimizer_steps, torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._exec_plan(plan, grads)
                self.optimizer_steps = before  # capture records, it does not execute
                self._ws["agraph_key"], self._ws["agraph"] = gkey2, g
        out = self._collect_decoupled_metrics(rows, norms, accum, flat.get("versions") is not None)
        if self._xgmi is not None and not self.defer_host_reads:
            self._xgmi.check_status()  # a peer that never published its gradient: raise instead of training on garbage
        self._step_lr_scheduler()  # :467
        return out

    def run_training(self) -> dict:
        a, alg, m = self.cfg.actor, self.cfg.algorithm, self.model
        assert alg.adv_type == "gae", "decoupled_actor_critic needs values: compute_values = (adv_type == 'gae') (:374)"
        lay = m.layout
        with self.timer("run_training"):
            flat, N = self._flatten_shuffle_normalize()
            world, gbs, micro = self._world_size, int(a.global_batch_size), int(a.micro_batch_size)
            assert gbs % (micro * world) == 0, (f"global_batch_size {gbs} must be divisible by micro_batch_size {micro} * "
                                                f"world_size {world}")
            per_rank = gbs // world
            accum = per_rank // micro
            self.gradient_accumulation = accum
            assert N % per_rank == 0, f"Flattened rollout size {N} must be divisible by per-rank batch size {per_rank}"
            n_global = N // per_rank
            if self._fused_update_ok(flat):
                return self._run_training_fused(flat, N, n_global, per_rank, accum, micro)
            n_steps = n_global * int(alg.get("update_epoch", 1))
            slabs = ops.mlp_bwd_slabs(micro)
            grads = torch.zeros((slabs * accum, m.n_params), dtype=torch.float32, device=self.device)
            bwd_ws = torch.empty(ops._lib.load().rlx_mlp_bwd_workspace_bytes(ops.byref(lay), micro), dtype=torch.uint8,
                                 device=self.device)
            rows = torch.zeros(n_steps * accum, PPO_OUT_FLOATS + 2, device=self.device)  # + the bonus's part of the loss, entropy loss
            norms = torch.zeros(n_steps, 2, device=self.device)
            ent_row = torch.zeros(PPO_OUT_FLOATS, device=self.device)
            self._grad_out_host = 1.0 / accum
            bonus = float(alg.get("entropy_bonus", 0) or 0)
            step = 0
            for _ in range(int(alg.get("update_epoch", 1))):
                for i in range(n_global):
                    warm = self.optimizer_steps < self.critic_warmup_steps
                    for j in range(accum):
                        lo = i * per_rank + j * micro
                        mb = {k: v[lo:lo + micro] for k, v in flat.items()}
                        row = rows[step * accum + j]
                        g = grads[j * slabs:(j + 1) * slabs]
                        logprob, _, value, mean, acts = ops.mlp_train_fwd(m.flat.data, m.packed(), lay, mb["states"], mb["action"])
                        lp, v = logprob.requires_grad_(True), value.requires_grad_(True)
                        prox_v = mb.get("proximal_values")
                        loss, out = ops.ppo_loss(
                            lp, mb["prev_logprobs"], mb["advantages"], logprob_type=alg.logprob_type, reward_type=alg.reward_type,
                            action_dim=int(a.model.get("action_dim", 7)), clip_ratio_low=alg.clip_ratio_low,
                            clip_ratio_high=alg.clip_ratio_high, values=v,
                            prev_values=prox_v if prox_v is not None else mb["prev_values"], returns=mb["returns"],
                            value_clip=alg.get("value_clip"), huber_delta=alg.get("huber_delta"), loss_mask=mb.get("loss_mask"),
                            loss_mask_sum=mb.get("loss_mask_sum"), max_episode_steps=self.cfg.env.train.get("max_episode_steps"),
                            clip_ratio_c=alg.get("clip_ratio_c", 3.0), critic_warmup=warm, has_critic=True,
                            decoupled=dict(proximal_logprobs=mb.get("proximal_logprobs"), versions=mb.get("versions"),
                                           current_version=int(self.version) + 1,
                                           behave_weight_threshold=alg.get("behave_weight_threshold")))
                        (loss / accum).backward()
                        ops.mlp_train_bwd(m.flat.data, m.packed(), lay, mb["states"], mb["action"], mean, acts, lp.grad, None,
                                          v.grad, grads=g, workspace=bwd_ws)
                        row[:PPO_OUT_FLOATS] = out
                        if bonus > 0 and not warm:  # :449-462; the decoupled row keeps slot 19 for the average version
                            ent_row.zero_()
                            ent_row[18] = out[DPPO_OUT_NAMES["mask_count"]]
                            self._entropy_bonus(mb, g, ent_row)
                            row[PPO_OUT_FLOATS] = ent_row[0]
                            row[PPO_OUT_FLOATS + 1] = ent_row[19]
                    self.optimizer_step(grads, stats=norms[step], critic_warmup=warm)
                    step += 1
            out = self._collect_decoupled_metrics(rows, norms, accum, flat.get("versions") is not None)
            self._step_lr_scheduler()  # :467
            return out

    def _collect_decoupled_metrics(self, rows, norms, accum: int, has_versions: bool):
        """Means over micro-batches (the version metrics only over those that reported them: some element unmasked,
        losses.py:156-165), AVG over ranks; EV sufficient statistics summed (:477-496).  One D2H copy -- behind the queued
        launches and read one iteration late when the runner's run-ahead loop asks for that (utils/pending.py)."""
        m = rows.mean(dim=0)
        counted = (rows[:, DPPO_OUT_NAMES["mask_count"]] > 0).float()
        ver = (rows[:, DPPO_OUT_NAMES["actor/average_version"]] * counted).sum() / counted.sum().clamp_min(1.0)
        ev = rows[:, DPPO_OUT_NAMES["ev/count"]:DPPO_OUT_NAMES["ev/errors_sq_sum"] + 1].sum(dim=0)
        avg = torch.cat([m, ver.view(1), norms[:, 0].mean().view(1)])
        if self._world_size > 1:
            all_reduce_flat_(avg, self.ctx, average=True)
            all_reduce_flat_(ev, self.ctx)
        vec = torch.cat([avg, ev, counted.sum().view(1)])
        current, lrs = float(int(self.version) + 1), tuple(self._lrs)
        xgmi = self._xgmi
        snap = xgmi is not None and self.defer_host_reads
        if snap:  # the exchange's status word travels with this step's numbers (a blocking read would wait for the whole queue)
            vec = torch.cat([vec, xgmi.status_snapshot(torch.empty(1, dtype=torch.float32, device=vec.device))])

        def finish(host: list) -> dict:
            if snap:
                timed_out, host = host[-1] != 0.0, host[:-1]
                if timed_out:
                    xgmi.check_status()  # raises: a peer never published its gradient during this step
            any_counted = host[-1] > 0
            out = {k: host[DPPO_OUT_NAMES[k]] for k in _DECOUPLED_KEYS + _CRITIC_KEYS}
            if has_versions and any_counted:
                out["actor/average_version"] = host[PPO_OUT_FLOATS + 2]
                out["actor/current_version"] = current
            n0 = PPO_OUT_FLOATS + 4
            out[CRITIC_EXPLAINED_VARIANCE_KEY] = explained_variance_from_stats(
                {name: host[n0 + i] for i, name in enumerate(_EV_MAP.values())})
            out["actor/total_loss"] = (host[DPPO_OUT_NAMES["loss"]] + host[PPO_OUT_FLOATS]) / max(accum, 1)
            out["actor/entropy_loss"] = host[PPO_OUT_FLOATS + 1]
            out["actor/grad_norm"] = host[PPO_OUT_FLOATS + 3]
            self._one_launch_expired(host[PPO_OUT_FLOATS + 3])
            out["actor/lr"], out["critic/lr"] = lrs
            return out

        if self.defer_host_reads:
            return PendingMetrics(vec, finish)
        return finish(vec.tolist())

#!/usr/bin/env python
"""Headline benchmark: ManiSkill-shaped 1024-env PPO actor-learner loop on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU.  Launched by torch.distributed.run -- RANK / LOCAL_RANK / WORLD_SIZE from the env -- or, from a bare
    shell, by this script itself: without a torchrun environment `--gpus N` spawns the N ranks and rank 0 prints the line.)

One "step" = one full iteration of the hot path over one batch of synthetic input, through the reference-shaped
runner/workers: rollout (T = 128 policy steps on B = 1024 envs incl. bootstrap-value forwards and the closing value
row) -> GAE advantages/returns -> shuffle -> update_epoch x minibatches of fused forward / PPO loss / backward /
clip+AdamW (8 x 16 = 128 optimizer steps at global_batch 8192).  Inputs (the synthetic env tensors) are resident in
HBM before the timed region.  Total work is fixed as N grows (envs and the global minibatch are sharded over ranks):
strong scaling -- the headline `value`, as north_star asks.  At N > 1 the same line also carries `weak_scaling` (1024 envs and
8192 minibatch rows PER GPU: the regime where more GPUs buy throughput on a loop this short) and `transports` (the timed region
once per gradient transport: RCCL between eager launches first -- the most conservative path --, then the hand-written xGMI exchange
and RCCL inside the captured update graph; the headline is the fastest one and says which).
Rank 0 prints ONE JSON line.

Extra objects on that line (the roofline probes themselves live in tools/bench_roofline.py):
  roofline      gae_scan (the kernel BASELINE.json grades against the HBM roofline), timed live with HIP events on the
                launch stream at the scaled shape 65536 envs x 128 steps (the 1024 x 128 buffer is 2.2 MB and lives in
                L2); algorithmic bytes = 17 B per env-step.
  roofline_widening  (N = 1) the kernels of the widening rows: token_logprob fwd / bwd at 4096 tokens x
                151936 vocab (bf16 logits), gae_seq at 4096 x 8192, patch_scan over a 622 M-element bf16 tensor, reinpp at
                4096 x 8192 and 32768 x 1024; same
                HIP-event method.
  variants      (N = 1) the exact-f32 path (the precision the reference's YAML ships) and pipeline mode, timed the same way
                inside the same run, so they are driver-timed too.
  cpu_baseline  the reference's own functions (staged under oracle/_ref; kind "reference") -- or, where those files are
                absent, the CPU oracle (kind "port") -- on this box's host cores, on a bounded sample, N = 1 only.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

T_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS, HORIZON, OBS_DIM, ACT_DIM = 1024, 128, 42, 8
GLOBAL_BATCH, UPDATE_EPOCH = 8192, 8
GAMMA, LAMBDA = 0.8, 0.9            # examples/embodiment/config/maniskill_ppo_mlp.yaml:63-64
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_cfg(world: int, use_graph: bool, precision: str = "32", pipeline: bool = False, rollout_epochs: int = 1,
              overlap: bool = False, total_envs: int = ENVS, global_batch: int = GLOBAL_BATCH, update_graph: bool | None = None):
    """``pipeline`` / ``rollout_epochs`` / ``overlap``: runner.use_training_pipeline variants (NOT the headline configuration):
    the horizon is split into ``rollout_epochs`` epochs of HORIZON / rollout_epochs steps so that the work per iteration stays
    1024 x 128 env-steps and 128 optimizer steps."""
    from rlinf_amd.config import DictConfig
    assert HORIZON % rollout_epochs == 0
    return DictConfig(dict(
        runner=dict(task_type="embodied", max_epochs=1, max_steps=-1, use_training_pipeline=pipeline, pipeline_overlap=overlap),
        algorithm=dict(update_epoch=UPDATE_EPOCH, normalize_advantages=True, group_size=1, reward_type="action_level",
                       logprob_type="action_level", entropy_type="action_level", adv_type="gae", loss_type="actor_critic",
                       bootstrap_type="always", entropy_bonus=0, clip_ratio_high=0.2, clip_ratio_low=0.2, value_clip=1.0,
                       huber_delta=10.0, gamma=GAMMA, gae_lambda=LAMBDA),
        env=dict(train=dict(rollout_epoch=rollout_epochs, total_num_envs=total_envs, auto_reset=True, ignore_terminations=False,
                            max_episode_steps=50, max_steps_per_rollout_epoch=HORIZON // rollout_epochs, seed=1234, group_size=1)),
        rollout=dict(pipeline_stage_num=1, enable_cuda_graph=use_graph),
        actor=dict(training_backend="fsdp", micro_batch_size=global_batch // world, global_batch_size=global_batch,
                   seed=1234, enable_hip_graph=use_graph if update_graph is None else update_graph, optimizer_writes_tiles=bool(int(os.environ.get("RLX_BENCH_OPT_TILES", "1"))),
                   model=dict(model_type="mlp_policy", obs_dim=OBS_DIM, action_dim=ACT_DIM, num_action_chunks=1,
                              precision=precision, add_value_head=True),
                   optim=dict(lr=3e-4, value_lr=3e-4, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                              weight_decay=0.01, clip_grad=0.5),
                   fsdp_config=dict(strategy="fsdp", sharding_strategy="no_shard"))))


def build_runner(cfg, ctx, learner: str = "sync"):
    """``learner`` "async": AsyncPPOEmbodiedFSDPActor behind the same runner (decoupled actor-critic loss: SURVEY.md 8f-4)."""
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.workers.actor import EmbodiedFSDPActor
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker
    if learner == "async":
        from rlinf_amd.workers.actor.async_ppo_fsdp_worker import AsyncPPOEmbodiedFSDPActor as EmbodiedFSDPActor  # noqa: F811
        cfg.algorithm.loss_type = "decoupled_actor_critic"
        cfg.algorithm.behave_weight_threshold = 2.0  # examples/embodiment/config/maniskill_async_ppo_*.yaml:45
    cfg = validate_cfg(cfg)
    actor = EmbodiedFSDPActor.create_group(cfg, ctx).launch(None, name="ActorGroup")
    rollout = MultiStepRolloutWorker.create_group(cfg, ctx).launch(None, name="RolloutGroup")
    env = EnvWorker.create_group(cfg, ctx).launch(None, name="EnvGroup")
    runner = EmbodiedRunner(cfg, actor, rollout, env)
    runner.init_workers()
    return runner


from tools.bench_roofline import gae_roofline, token_tier_roofline  # noqa: E402  (the roofline probes: tools/bench_roofline.py)


def _lib_handle():
    from rlinf_amd import _lib
    return _lib.load()


def token_tier_cpu_baseline(rows: int = 64, vocab: int = 151936, warmup: int = 6, timed: int = 9):
    """The token tier on the host cores, bounded sample of the same shape: log-prob + entropy from bf16 logits, forward and
    backward, 3 warm-up + 5 timed passes (median).  kind "reference": the reference's OWN compute_logprobs_from_logits /
    compute_entropy_from_logits (rlinf/utils/utils.py:454-512, loaded from the staged copy oracle/_ref); kind "port" (the oracle's
    restatement of the same two functions) only where the staged files are absent."""
    from oracle import reference_loader as RL
    kind = "port"
    try:
        if RL.available():
            ref = RL.load()
            logp_fn = lambda x, y: ref.utils.compute_logprobs_from_logits(x, y)  # noqa: E731 -- op_type "torch" (its default)
            ent_fn = ref.utils.compute_entropy_from_logits
            kind = "reference"
    except Exception:  # noqa: BLE001
        kind = "port"
    if kind == "port":
        from oracle import token_oracle as TO
        logp_fn, ent_fn = TO.logprobs_from_logits, TO.entropy_from_logits

    threads, avail = _cpu_threads()  # fixed, stated: min(32, schedulable cores)
    g = torch.Generator().manual_seed(1)
    x0 = (torch.randn(rows, vocab, generator=g) * 4).to(torch.bfloat16)
    labels = torch.randint(0, vocab, (rows,), generator=g)
    times = []
    for k in range(warmup + timed):
        x = x0.clone().requires_grad_(True)
        t0 = time.perf_counter()
        lp = logp_fn(x, labels)
        ent = ent_fn(x)
        (lp.float().sum() - 0.01 * ent.float().sum()).backward()
        times.append(time.perf_counter() - t0)
    t = sorted(times[warmup:])
    med = t[len(t) // 2]
    return {"value": round(rows / med, 1), "unit": "tokens/s (log-prob + entropy forward and backward)",
            "cores": threads, "host_cores": avail, "kind": kind,
            "sample": f"{rows} tokens x {vocab} vocab, bf16 logits; {warmup} warm-up + {timed} timed passes, median; "
                      + ("rlinf/utils/utils.py compute_logprobs_from_logits + compute_entropy_from_logits from oracle/_ref"
                         if kind == "reference" else "oracle/token_oracle.py restatement (staged reference files absent)"),
            "pass_times_s": [round(v, 4) for v in times], "value_fastest_pass": round(rows / t[0], 1),
            "value_slowest_pass": round(rows / t[-1], 1), "spread_slowest_over_fastest": round(t[-1] / t[0], 3)}


CPU_BASELINE_THREADS = 32  # a FIXED, stated torch thread count (capped by the schedulable cores): no per-run timing probe


def _cpu_threads():
    """torch's default (one thread per logical core) oversubscribes the small GEMMs of this path badly on a many-core host (256
    threads: ~2.7 s per rollout step); rounds 2-3 picked the count by a timing probe per run (it landed on 32 on every 256-core
    box, but made the baseline depend on a few noisy samples).  Now: min(32, schedulable cores), stated in the line."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(CPU_BASELINE_THREADS, avail))
    torch.set_num_threads(threads)
    return threads, avail


def cpu_baseline_reference():
    """kind "reference": the reference's OWN functions on the host cores (oracle/reference_loop.py: its MLPPolicy, its
    calculate_adv_and_returns and policy_loss with their built-in callees, torch's clip_grad_norm_ + AdamW), loaded from the
    staged copy oracle/_ref (or /root/reference).  SURVEY.md 8d's recipe: one full rollout + advantage pass, then 3 warm-up and 10
    timed optimizer steps; the update phase = MEDIAN step x 128.  The step times of a many-core host are bimodal (thread
    placement: rounds 2-3 saw mean 43 ms against median 22 ms), so `value` rests on the median and the line also carries the
    value under the lower / upper quartile step time -- that, not min / max, is the range a speed-up may be quoted with."""
    from oracle import ppo_loop as L
    from oracle import reference_loader as RL
    from oracle import reference_loop as RLoop
    if not RL.available():
        return None
    threads, avail = _cpu_threads()
    env = L.synthetic_env_tensors(1234, HORIZON, ENVS, OBS_DIM, max_episode_steps=50)
    t0 = time.perf_counter()
    t = RLoop.timed_iteration(env, obs_dim=OBS_DIM, act_dim=ACT_DIM, gamma=GAMMA, gae_lambda=LAMBDA, global_batch=GLOBAL_BATCH,
                              update_epoch=UPDATE_EPOCH, warmup_steps=3, timed_steps=10)
    spent = time.perf_counter() - t0
    total, steps = t["update_steps_total"], sorted(t["timed_step_times_s"])
    fixed = t["rollout_s"] + t["advantages_s"]
    q = lambda f: steps[min(len(steps) - 1, max(0, int(round(f * (len(steps) - 1)))))]  # noqa: E731
    med, lo, hi = q(0.5), q(0.25), q(0.75)
    val = lambda per_step: round(ENVS * HORIZON / (fixed + per_step * total), 1)  # noqa: E731
    return {"value": val(med), "unit": "env-steps/s", "cores": threads, "kind": "reference",
            "sample": f"the reference's own MLPPolicy / calculate_adv_and_returns / policy_loss + torch clip_grad_norm_ / AdamW "
                      f"(files staged under oracle/_ref): 1 full rollout of {HORIZON}x{ENVS} + GAE, then 3 warm-up + {len(steps)} timed "
                      f"optimizer steps of {GLOBAL_BATCH} rows; update phase = median step x {total}; {spent:.1f} s of CPU work; "
                      f"torch.set_num_threads({threads}) (fixed: min({CPU_BASELINE_THREADS}, {avail} schedulable cores))",
            "host_cores": avail, "value_is": "median-based",
            "value_lower_quartile_step": val(lo), "value_upper_quartile_step": val(hi),
            "value_fastest_step": val(steps[0]), "value_slowest_step": val(steps[-1]),  # true min / max of the timed steps
            "updates_per_sec": round(1.0 / med, 2), "rollout_s": round(t["rollout_s"], 3),
            "advantages_s": round(t["advantages_s"], 4),
            "update_s_per_step": {"min": round(steps[0], 4), "lower_quartile": round(lo, 4), "median": round(med, 4),
                                  "upper_quartile": round(hi, 4), "max": round(steps[-1], 4)},
            "update_steps_timed": len(steps), "iteration_s": round(fixed + med * total, 3)}


def cpu_baseline_subprocess(timeout_s: float = 240.0):
    """cpu_baseline() in a process of its own with its OpenMP team PINNED (OMP_NUM_THREADS = the stated thread count, OMP_PROC_BIND =
    close, OMP_PLACES = cores).  In-process, behind the GPU work, the step times of a many-core host were bimodal (22 ms / 70 ms:
    the OpenMP threads migrating between cores under the process's other threads): three slow samples in ten moved every quantile."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(CPU_BASELINE_THREADS, avail))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores",
               OMP_WAIT_POLICY="active", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], env=env, capture_output=True, text=True,
                       timeout=timeout_s)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"cpu baseline subprocess failed (rc {r.returncode}): {r.stderr[-300:]}")
    out = json.loads(lines[-1])
    out["sample"] += "; own process, OpenMP team pinned (OMP_PROC_BIND=close, OMP_PLACES=cores)"
    return out


def token_tier_cpu_baseline_subprocess(timeout_s: float = 240.0):
    """token_tier_cpu_baseline() under the same policy as cpu_baseline: a process of its own, a FIXED stated thread count
    (min(32, schedulable cores)), the OpenMP team pinned.  In-process with torch's default thread count (128 threads on a box with 2
    schedulable cores) round 4's pass times spread 29 x."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(CPU_BASELINE_THREADS, avail))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores",
               OMP_WAIT_POLICY="active", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--token-cpu-baseline-only"], env=env, capture_output=True,
                       text=True, timeout=timeout_s)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"token-tier cpu baseline subprocess failed (rc {r.returncode}): {r.stderr[-300:]}")
    out = json.loads(lines[-1])
    out["sample"] += f"; own process, torch.set_num_threads({out.get('cores')}), OpenMP team pinned (OMP_PROC_BIND=close, OMP_PLACES=cores)"
    return out


def cpu_baseline(budget_s: float = 20.0):
    """The reference's own functions when its files are reachable (kind "reference"), else the oracle port (kind "port")."""
    try:
        ref = cpu_baseline_reference()
    except Exception as e:  # noqa: BLE001 -- the port below still gives a baseline; say why the reference one is missing
        ref, why = None, f"{type(e).__name__}: {e}"[:200]
    else:
        why = "reference files not staged (oracle/_ref absent)"
    if ref is not None:
        return ref
    out = cpu_baseline_port(budget_s)
    out["reference_kind_unavailable"] = why
    return out


def cpu_baseline_port(budget_s: float = 20.0):
    """Oracle iteration on the host cores: one full rollout + GAE, then optimizer steps until ~budget_s of CPU work
    has been spent (the rest of the 128 is extrapolated linearly -- every minibatch step does identical work)."""
    from oracle import ppo_loop as L
    from oracle import ppo_oracle as O
    torch.manual_seed(1234)
    pol = O.OracleMLPPolicy(OBS_DIM, ACT_DIM, 1)
    opt = O.build_adamw(pol)
    threads, avail = _cpu_threads()
    env = L.synthetic_env_tensors(1234, HORIZON, ENVS, OBS_DIM, max_episode_steps=50)
    eps = torch.randn(HORIZON, ENVS, ACT_DIM, generator=torch.Generator().manual_seed(1))
    total_updates = (ENVS * HORIZON // GLOBAL_BATCH) * UPDATE_EPOCH
    t0 = time.perf_counter()
    batch = L.rollout(pol, env, eps, GAMMA, True)
    t1 = time.perf_counter()
    batch = L.advantages(batch, GAMMA, LAMBDA, True)
    t2 = time.perf_counter()
    # two timed probe steps decide how many fit in the budget
    L.update(pol, opt, batch, seed=1234, global_batch=GLOBAL_BATCH, update_epoch=UPDATE_EPOCH, max_steps=2)
    probe = (time.perf_counter() - t2) / 2
    n_steps = int(max(2, min(total_updates, (budget_s - (t2 - t0) - 2 * probe) / max(probe, 1e-6))))
    t3 = time.perf_counter()
    done = L.update(pol, opt, batch, seed=1234, global_batch=GLOBAL_BATCH, update_epoch=UPDATE_EPOCH, max_steps=n_steps)
    t4 = time.perf_counter()
    per_update = (t4 - t3) / len(done)
    iter_s = (t1 - t0) + (t2 - t1) + per_update * total_updates
    return {"value": round(ENVS * HORIZON / iter_s, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"1 full rollout of {HORIZON}x{ENVS} + GAE + {len(done)} of {total_updates} optimizer steps "
                      f"(rest of the update phase extrapolated); {t4 - t0:.1f} s of CPU work; torch.set_num_threads({threads}) "
                      f"(fixed: min({CPU_BASELINE_THREADS}, {avail} schedulable cores))",
            "host_cores": avail, "updates_per_sec": round(1.0 / per_update, 2), "rollout_s": round(t1 - t0, 3),
            "advantages_s": round(t2 - t1, 4), "update_s_per_step": round(per_update, 4),
            "iteration_s": round(iter_s, 3)}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` from a bare shell: spawn the N ranks ourselves (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, one
    GPU each), rank 0 prints the line on our stdout, every other rank's stdout goes to stderr.  A rank that dies takes the others
    with it (exit code of the first failure); a job that outlives --launch-timeout is killed."""
    import subprocess
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus and os.environ.get("RLX_BENCH_ALLOW_SHARED_GPU") != "1":  # (the switch: the launcher's own test on a one-GPU box)
        raise SystemExit(f"--gpus {args.gpus} but {n_dev} GPU(s) visible: one rank per GPU (RCCL and the xGMI exchange both need it)")
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline, rc = time.time() + args.launch_timeout, 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
        if rc != 0 or time.time() > deadline:
            for p in live:
                p.kill()
            if rc == 0:
                rc = 124
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=10)
        except Exception:  # noqa: BLE001
            pass
    return rc


def measure(ctx, *, precision: str, scaling: str = "strong", steps: int, warmup: int, use_graph: bool = True,
            pipeline: bool = False, rollout_epochs: int = 1, overlap: bool = False, transport: str | None = None,
            learner: str = "sync") -> dict:
    """One timed region: W untimed iterations, then exactly ``steps`` iterations between barrier + synchronize on both sides,
    MAX over ranks.  ``scaling``: strong = 1024 envs / 8192-row global batch in total, weak = that much PER GPU.  Also returns
    the spread of the iteration time over 10-step windows (host time stamps taken when a window's last step has landed on the host)."""
    import torch.distributed as dist
    world, dev = ctx.world_size, ctx.device
    envs = ENVS * (world if scaling == "weak" else 1)
    gb = GLOBAL_BATCH * (world if scaling == "weak" else 1)
    update_graph = None
    if transport is not None:
        # "rccl-eager": torch.distributed's all-reduce between eagerly launched (prepared) kernels -- the most conservative N > 1
        # path there is (no hand-written exchange, no collective inside a captured graph); the rollout loop stays a hipGraph
        os.environ["RLX_GRAD_ALLREDUCE"] = "rccl" if transport == "rccl-eager" else transport
        update_graph = False if transport == "rccl-eager" else None
    runner = build_runner(build_cfg(world, use_graph, precision, pipeline=pipeline, rollout_epochs=rollout_epochs, overlap=overlap,
                                    total_envs=envs, global_batch=gb, update_graph=update_graph), ctx, learner=learner)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    try:
        for _ in range(max(warmup, 2 if use_graph else 0)):  # first graph call runs eagerly and captures
            runner.run_step()
        barrier()
        t0 = time.perf_counter()
        stamps = [t0]
        # the runner's own loop (EmbodiedRunner.run with runner.defer_metrics, its default): iteration i + 1 is queued on the
        # device before iteration i's metrics are read on the host, so the device never waits for the host between iterations.
        # Every iteration's metrics ARE read, inside the timed region; a 10-step window closes when its last step has landed.
        pending, box = None, {"m": {}, "landed": 0}

        def land(step):
            box["m"] = step.result() if DEFER_METRICS else step
            box["landed"] += 1
            if box["landed"] % 10 == 0:
                stamps.append(time.perf_counter())

        for i in range(steps):
            step = runner.run_step(defer=DEFER_METRICS)
            if pending is not None:
                land(pending)
            if DEFER_METRICS:
                pending = step
            else:
                land(step)
        if pending is not None:
            land(pending)
        barrier()
        elapsed = time.perf_counter() - t0
        m = box["m"]
        assert box["landed"] == steps and len(runner.metrics_history) >= steps
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        w = runner.actor.worker
        # who actually took part: ranks from the process group, physical devices from every rank's (host, PCI location)
        ranks_seen, devices_seen = 1, 1
        if world > 1:
            from rlinf_amd.scheduler.dist import device_identity
            ident = device_identity(dev)
            idents = [None] * world
            dist.all_gather_object(idents, ident)
            ranks_seen, devices_seen = dist.get_world_size(), len(set(idents))
        xg = w._xgmi
        win = sorted((b - a) * 100.0 for a, b in zip(stamps[:-1], stamps[1:]))  # ms per iteration over each 10-step window
        updates = (envs // world * HORIZON // (gb // world)) * UPDATE_EPOCH
        return {"env_steps_per_sec": round(envs * HORIZON * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 3),
                "ms_per_step_windows": ({"window": 10, "n": len(win), "min": round(win[0], 3), "median": round(win[len(win) // 2], 3),
                                         "max": round(win[-1], 3)} if win else None),
                "ppo_updates_per_sec": round(updates * steps / elapsed, 1), "steps": steps, "total_envs": envs, "global_batch": gb,
                "grad_allreduce": w.grad_allreduce_backend + ((" (one launch per step: pushed self-validating words)" if (getattr(w._xgmi, "one_launch", False) and w.adamw_sync is not None)
                                                               else f" ({w._xgmi.algo}, {w._xgmi.wait_mode} hand-shake)") if w._xgmi is not None else ""),
                "update_graph_replayed": w._graph is not None, "elapsed_s": round(elapsed, 4),
                "ranks_seen": ranks_seen, "devices_seen": devices_seen,
                # the exchange this run REALLY used: the hand-written one only when its start-up validation against
                # torch.distributed's all-reduce passed on every rank (scheduler.xgmi.build), else RCCL -- whatever was asked for
                "xgmi": (None if xg is None else {"validated_against_torch_distributed": True, "algo": xg.algo, "wait_mode": xg.wait_mode,
                                                  "mem_kind": getattr(xg, "mem_kind", None), "ranks_share_a_device": bool(xg.shared_device),
                                                  # the exchange inside the one-launch optimizer step (pushed self-validating words):
                                                  # used when ITS start-up validation passed, else the launch chain
                                                  "one_launch_exchange": bool(getattr(xg, "one_launch", False) and w.adamw_sync is not None),
                                                  "one_launch_verdict": getattr(xg, "one_launch_verdict", None)}),
                "metrics_read": "every iteration, one iteration late (runner.defer_metrics)" if DEFER_METRICS else "every iteration, before the next is queued",
                "last_metrics": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in m.items()
                                 if k in ("train/actor/total_loss", "train/actor/grad_norm", "train/actor/approx_kl", "rollout/rewards")}}
    finally:
        runner.close()
        del runner
        torch.cuda.empty_cache()


def _step_slabs(precision: str) -> int:
    """Split-K slabs of one 8192-row optimizer step (the library's own plan)."""
    from rlinf_amd import ops
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    lay = MLPPolicy(OBS_DIM, ACT_DIM, 1, True, False).layout
    return int(ops.ppo_step_slabs(lay, GLOBAL_BATCH, bf16=precision == "bf16"))


DEFER_METRICS = os.environ.get("RLX_BENCH_DEFER_METRICS", "1") != "0"  # 0: read every step's metrics before queueing the next

WORKLOAD = ("ManiSkill PickCube-shaped PPO: 1024 envs x 128 steps, obs 42, act 8, MLP policy (3x256 tanh actor + value head), "
            "gamma 0.8 / lambda 0.9, 8 epochs x 16 minibatches of 8192 (128 optimizer steps), {prec}, synthetic env tensors "
            "resident in HBM")
PREC_TEXT = {"bf16": "bf16 MFMA operands / f32 accumulate, master weights, losses, GAE, AdamW",
             # precision "32" (the reference YAML's shipped precision): every f32 operand travels as three bf16 planes (hi + mid + lo = x
             # exactly) and every product is six of the nine plane products on the bf16 matrix pipe with f32 accumulation -- what is
             # dropped lies below the f32 accumulation's own rounding (csrc/ppo_step_f32x.hip); RLX_F32_EXACT_MFMA=1 selects the
             # exact-f32-MFMA launches of round 1-4 instead
             "32": ("f32: exact-f32 MFMA (fused launch and weight gradients; RLX_F32_EXACT_MFMA=1)" if os.environ.get("RLX_F32_EXACT_MFMA", "0") not in ("", "0")
                    else "f32: f32-accurate products as 3 x bf16 planes (six plane products, f32 accumulate) on the bf16 matrix pipe, in the "
                         "rollout, fused and weight-gradient launches")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed iterations (default 200: a ~2 s timed region at ~9 ms each)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "32"],
                    help="operand precision of the policy's dense layers: bf16 (BASELINE.json configs[1], f32 accumulate and "
                         "master weights) or 32 (f32-accurate products: three bf16 planes per operand on the bf16 matrix pipe)")
    ap.add_argument("--pipeline", action="store_true", help="runner.use_training_pipeline (statistics normalisation, per-stage "
                    "shuffles; with --rollout-epochs > 1 the learner trains on epoch e while epoch e + 1 rolls out)")
    ap.add_argument("--rollout-epochs", type=int, default=1, help="split the 128-step horizon into this many rollout epochs")
    ap.add_argument("--overlap", action="store_true", help="pipeline mode with the rollout of epoch e + 1 on its own stream (runner."
                    "pipeline_overlap; off by default since round 5: profiles/r05_pipeline_overlap_*.txt)")
    ap.add_argument("--no-overlap", action="store_true", help="(kept for old command lines: one stream is the default now)")
    ap.add_argument("--learner", default="sync", choices=["sync", "async"],
                    help="async: AsyncPPOEmbodiedFSDPActor (decoupled actor-critic loss, SURVEY.md 8f-4) as the timed loop -- a variant "
                         "line for traces and A/B runs, not the BASELINE.json configuration (the default run reports it under variants)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="what `value` measures at N > 1: strong = 1024 envs in total (north_star's regime; the default), weak = "
                         "1024 envs and an 8192-row minibatch per GPU.  The other regime rides along in the same line.")
    ap.add_argument("--transports", default="auto", help="N > 1: comma list of gradient transports to time (xgmi, rccl); auto = both")
    ap.add_argument("--variant-steps", type=int, default=50, help="timed iterations of each N = 1 variant line (f32, pipeline)")
    ap.add_argument("--no-variants", action="store_true", help="skip the exact-f32 and pipeline variant lines (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-token-tier", action="store_true", help="skip the token-tier (LLM logits) roofline rows")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--with-codec", action="store_true", help="add the opt-in patch codec (rlx_zplane) to roofline_widening")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched N > 1 job: kill it after this many seconds")
    ap.add_argument("--pair-timeout", type=float, default=240.0, help="N > 1: seconds one (regime, transport) measurement may take "
                    "before the watchdog prints the line assembled so far and ends the job")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) run the CPU baseline and print its object")
    ap.add_argument("--token-cpu-baseline-only", action="store_true", help="(internal) run the token-tier CPU baseline and print its object")
    ap.add_argument("--no-extras", action="store_true", help="skip scaling_model, in_loop_kernels and the exchange probe")
    ap.add_argument("--exchange", default="none", choices=["none", "self"],
                    help="self (N = 1): only print the measured per-step cost of the gradient exchange's own launches on this device "
                         "(tools/exchange_self.py; also part of scaling_model in the default line)")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if args.token_cpu_baseline_only:
        print(json.dumps(token_tier_cpu_baseline()), flush=True)
        return
    if args.exchange == "self":
        from tools.exchange_self import measure as exchange_self
        print(json.dumps({"metric": "gradient exchange launches per optimizer step, one device (peers aliased to self)",
                          **exchange_self("cuda:0", 287504, 10)}), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        sys.exit(self_launch(args))  # bare shell: become the launcher

    from rlinf_amd.scheduler import init_distributed
    import torch.distributed as dist

    ctx = init_distributed()
    if ctx.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    dev = ctx.device
    use_graph = not args.no_graph
    common = dict(precision=args.precision, steps=args.steps, warmup=args.warmup, use_graph=use_graph, pipeline=args.pipeline,
                  rollout_epochs=args.rollout_epochs, overlap=args.overlap and not args.no_overlap)
    if args.learner != "sync":
        common["learner"] = args.learner

    # ---- the timed regions ------------------------------------------------------------------------------------------------
    runs, errors = {}, {}

    def trusted(v):
        """A run may carry the headline `value` if its gradient exchange is torch.distributed's (RCCL) -- or the hand-written xGMI
        exchange AFTER its start-up validation passed with every rank on a device of its own (ranks sharing a device exercise the
        protocol, not the links)."""
        return v["xgmi"] is None or not v["xgmi"]["ranks_share_a_device"]

    def best(regime):
        c = [(v["env_steps_per_sec"], tr, v) for (rg, tr), v in runs.items() if rg == regime]
        ok = [x for x in c if trusted(x[2])]
        return max(ok or c, key=lambda x: x[0]) if c else None

    def assemble():
        """The JSON line from whatever has been measured so far (None before the first headline-regime run exists)."""
        top = best(args.scaling)
        if top is None:
            return None
        _, _, head = top
        line = {
            "metric": "env_steps_per_sec", "value": head["env_steps_per_sec"], "unit": "env-steps/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD.format(prec=PREC_TEXT[args.precision])
                                   + (" -- PER GPU (weak scaling)" if args.scaling == "weak" and args.gpus > 1 else ""),
                       "total_envs": head["total_envs"], "horizon": HORIZON, "global_batch": head["global_batch"],
                       "update_epoch": UPDATE_EPOCH, "parallelism": f"dp{args.gpus}", "hip_graph": use_graph,
                       "grad_allreduce": head["grad_allreduce"], "update_graph_replayed": head["update_graph_replayed"],
                       **({"pipeline": True, "rollout_epochs": args.rollout_epochs, "overlap": args.overlap and not args.no_overlap}
                          if args.pipeline else {}),
                       **({"learner": "async (decoupled actor-critic loss, behave_weight_threshold 2): a VARIANT of the BASELINE.json "
                                      "configuration, not its headline"} if args.learner != "sync" else {})},
            "ms_per_step_windows": head["ms_per_step_windows"],
            "ppo_updates_per_sec": head["ppo_updates_per_sec"],
            # end-to-end parity AT THIS configuration (1024 x 128, 8192-row minibatches, 128 optimizer steps, hipGraph replay):
            "parity_checked": ("tests/test_end_to_end_bench_config.py::test_bench_configuration_"
                               + ("bf16_matches_autocast_oracle_with_graph_replay" if args.precision == "bf16" else "f32_matches_oracle[True]")
                               + " + ::test_first_optimizer_step_gradient_at_bench_configuration + "
                                 "::test_ten_iterations_reseeded_oracle_at_bench_configuration (runner built by this file's "
                                 "build_cfg / build_runner, compared with oracle.ppo_loop.iteration); num_action_chunks = 2 whole-loop: "
                                 "tests/test_end_to_end.py::test_action_chunks_whole_loop_matches_oracle"),
            "last_metrics": head["last_metrics"],
        }
        if args.gpus > 1:
            line["ranks_seen"], line["devices_seen"] = head["ranks_seen"], head["devices_seen"]
            line["value_transport"] = {"asked": top[1], "used": head["grad_allreduce"], "trusted": trusted(head)}
            xr = [v for (rg, tr), v in runs.items() if v["xgmi"] is not None]
            asked_xgmi = [f"{rg}/{tr}" for (rg, tr), v in runs.items() if tr == "xgmi"]
            line["xgmi_validation"] = (dict(xr[0]["xgmi"], passed=True) if xr else
                                       {"passed": False, "note": "no run used the hand-written exchange" + (
                                           " (asked for in " + ", ".join(asked_xgmi) + ": validation failed or RLX_GRAD_ALLREDUCE forced RCCL; those runs used RCCL)"
                                           if asked_xgmi else "")})
            other = "weak" if args.scaling == "strong" else "strong"
            ob = best(other)
            line[f"{other}_scaling"] = None if ob is None else {k: ob[2][k] for k in (
                "env_steps_per_sec", "ms_per_step", "ms_per_step_windows", "ppo_updates_per_sec", "total_envs", "global_batch", "grad_allreduce")}
            line["transports"] = {f"{rg}/{tr}": {k: v[k] for k in ("env_steps_per_sec", "ms_per_step", "grad_allreduce", "update_graph_replayed", "devices_seen", "done_at_s") if k in v}
                                  for (rg, tr), v in runs.items()}
            if errors:
                line["transport_errors"] = dict(errors)
        return line

    if args.gpus == 1:
        runs[("strong", None)] = measure(ctx, scaling="strong", **common)
    else:
        # Every (regime, transport) pair is its own runner and its own timed region.  Order: the headline regime first, and inside
        # it the transport most likely to give a number.  A watchdog guards the rest: the N > 1 paths meet real multi-GPU hardware
        # for the first time under the driver -- if a later pair hangs (an RCCL capture that never returns, a dead link), every
        # rank's timer fires, rank 0 prints the line assembled from what WAS measured (the hang named in transport_errors) and the
        # job exits instead of taking the measured headline down with it.
        import threading
        forced = os.environ.get("RLX_GRAD_ALLREDUCE")
        names = [forced] if forced else (["xgmi", "rccl"] if args.transports == "auto" else args.transports.split(","))
        os.environ.setdefault("RLX_XGMI_TIMEOUT_MS", "30000")  # a dead exchange costs the bench one bounded wait, then raises
        other_regime = "weak" if args.scaling == "strong" else "strong"
        # the headline regime opens with the most conservative path, so that SOME number exists before anything newer runs
        first = ["rccl-eager"] if (args.transports == "auto" and not forced and use_graph) else []
        pairs = [(args.scaling, tr) for tr in first + names] + [(other_regime, tr) for tr in names]
        current = {"pair": None}

        def emergency():
            if ctx.rank == 0:
                errors[f"{current['pair']}"] = f"no progress within --pair-timeout {args.pair_timeout:.0f} s: abandoned (watchdog)"
                line = assemble()
                if line is not None:
                    print(json.dumps(line), flush=True)
            os._exit(0 if (ctx.rank != 0 or best(args.scaling) is not None) else 3)

        for rg, tr in pairs:
            current["pair"] = f"{rg}/{tr}"
            dog = threading.Timer(args.pair_timeout, emergency)
            dog.daemon = True
            dog.start()
            try:
                runs[(rg, tr)] = measure(ctx, scaling=rg, transport=tr, **common)
                runs[(rg, tr)]["done_at_s"] = round(time.perf_counter() - T_START, 1)  # since this process started
            except Exception as e:  # noqa: BLE001 -- reported in the line; collectively consistent only if it failed everywhere
                errors[f"{rg}/{tr}"] = f"{type(e).__name__}: {e}"[:300]
            # a transport failing on ONE rank only would leave the ranks in different collectives: agree on what exists
            have = torch.tensor([1 if (rg, tr) in runs else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
            dog.cancel()
            if not int(have.item()):
                runs.pop((rg, tr), None)
                errors.setdefault(f"{rg}/{tr}", "failed on another rank")
        if best(args.scaling) is None:
            raise SystemExit(f"no transport completed the {args.scaling}-scaling run: {errors}")

    line = assemble() if ctx.rank == 0 else None

    # ---- N = 1: the variant lines (own runners, same timing method), then the probes ---------------------------------------
    if args.gpus == 1:
        def extra(key, fn):
            """The headline fields above are already measured: a failure in a secondary probe is reported in the line
            instead of costing it."""
            try:
                line[key] = fn()
            except Exception as e:  # noqa: BLE001
                line[key] = None
                line.setdefault("probe_errors", {})[key] = f"{type(e).__name__}: {e}"[:300]

        if args.steps < 100 and not args.no_variants:
            # the contract's timed region is EXACTLY --steps iterations (the driver passes 20: 0.18 s); the same configuration
            # again over 200 iterations, with the spread over 10-step windows, so the headline does not rest on 20 iterations
            def sustained():
                r = measure(ctx, **{**common, "steps": 200, "warmup": 1})
                return {k: r[k] for k in ("env_steps_per_sec", "ms_per_step", "ms_per_step_windows", "ppo_updates_per_sec", "steps", "elapsed_s")}
            extra("sustained_200_steps", sustained)
        if not args.no_variants and not args.pipeline:
            def variants():
                out = []
                vs = dict(steps=args.variant_steps, warmup=args.warmup, use_graph=use_graph)
                other = "32" if args.precision == "bf16" else "bf16"
                for name, kw in ((f"precision {'f32 (' + PREC_TEXT['32'] + ': the precision the reference YAML ships)' if other == '32' else 'bf16'}",
                                  dict(precision=other)),
                                 ("pipeline mode, 4 rollout epochs, one stream (runner.pipeline_overlap false: the default)",
                                  dict(precision=args.precision, pipeline=True, rollout_epochs=4)),
                                 ("pipeline mode, 4 rollout epochs, rollout of epoch e + 1 on its own stream (runner.pipeline_overlap true; "
                                  "the two chains do not co-execute: profiles/r05_pipeline_overlap_*.txt)",
                                  dict(precision=args.precision, pipeline=True, rollout_epochs=4, overlap=True)),
                                 ("pipeline mode, 1 rollout epoch (statistics normalisation + per-stage shuffles only)",
                                  dict(precision=args.precision, pipeline=True, rollout_epochs=1)),
                                 # the learner of async / decoupled PPO at the same configuration and precision: the
                                 # decoupled loss runs inside rlx_ppo_step (actor gradients in sum form, scaled in the slab sum), the
                                 # update phase is the same prepared-launch hipGraph; behave_weight_threshold 2 as the shipped configs
                                 ("async learner (decoupled actor-critic loss inside the fused step, behave_weight_threshold 2; "
                                  "masked_normalization of the advantages over the shuffled buffer)",
                                  dict(precision=args.precision, learner="async"))):
                    try:
                        r = measure(ctx, **{**vs, **kw})
                        out.append({"variant": name, "dtype": "f32" if kw["precision"] == "32" else "bf16", "metric": "env_steps_per_sec",
                                    "value": r["env_steps_per_sec"], "ms_per_step": r["ms_per_step"], "ms_per_step_windows": r["ms_per_step_windows"],
                                    "ppo_updates_per_sec": r["ppo_updates_per_sec"], "steps": r["steps"]})
                    except Exception as e:  # noqa: BLE001
                        out.append({"variant": name, "error": f"{type(e).__name__}: {e}"[:300]})
                return out
            extra("variants", variants)
        if not args.no_extras:
            from tools import bench_extras as BX
            n_params = 287504
            extra("in_loop_kernels", lambda: BX.in_loop_kernels(args.precision, rows=GLOBAL_BATCH, envs=ENVS,
                                                                slabs=_step_slabs(args.precision), n_params=n_params))
            extra("scaling_model", lambda: BX.scaling_model(dev, args.precision, n_params, {1: _step_slabs(args.precision)}))
        if not args.no_roofline:
            extra("roofline", lambda: gae_roofline(dev, with_traffic=not args.no_traffic))
            if not args.no_extras:
                extra("roofline_normalised", lambda: BX.gae_normalised_rows(dev))
            if not args.no_token_tier:
                extra("roofline_widening", lambda: token_tier_roofline(dev, with_codec=args.with_codec))
                # SURVEY.md 8f-1 end to end: the reasoning learner's iteration (FSDPActor.run_training) around a stand-in LM
                from tools.bench_reasoning_loop import measure as reasoning_loop
                extra("reasoning_learner", reasoning_loop)
        if not args.no_cpu_baseline:
            extra("cpu_baseline", cpu_baseline_subprocess)
            if line.get("roofline_widening"):
                extra("cpu_baseline_token_tier", token_tier_cpu_baseline_subprocess)
            if line.get("cpu_baseline"):
                cb = line["cpu_baseline"]
                line["speedup_vs_cpu_baseline"] = round(line["value"] / cb["value"], 1)
                line["speedup_vs_cpu_baseline_cores"] = cb.get("cores")  # a ratio against THAT many host cores, not a many-core host
                if "value_fastest_step" in cb:  # the spread of the CPU sample carried through (true min / max and the quartiles)
                    line["speedup_vs_cpu_baseline_range"] = {
                        "cpu_fastest_to_slowest_step": [round(line["value"] / cb["value_fastest_step"], 1), round(line["value"] / cb["value_slowest_step"], 1)],
                        "cpu_lower_to_upper_quartile_step": [round(line["value"] / cb["value_lower_quartile_step"], 1),
                                                             round(line["value"] / cb["value_upper_quartile_step"], 1)]}
    elif ctx.rank == 0 and not args.no_roofline:
        try:
            line["roofline"] = gae_roofline(dev, with_traffic=False)
        except Exception as e:  # noqa: BLE001
            line["roofline"] = None
            line.setdefault("probe_errors", {})["roofline"] = f"{type(e).__name__}: {e}"[:300]
    if ctx.rank == 0:
        print(json.dumps(line), flush=True)
    if ctx.world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: ManiSkill-shaped 1024-env PPO actor-learner loop on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the env)

One "step" = one full iteration of the hot path over one batch of synthetic input, through the reference-shaped
runner/workers: rollout (T = 128 policy steps on B = 1024 envs incl. bootstrap-value forwards and the closing value
row) -> GAE advantages/returns -> shuffle -> update_epoch x minibatches of fused forward / PPO loss / backward /
clip+AdamW (8 x 16 = 128 optimizer steps at global_batch 8192).  Inputs (the synthetic env tensors) are resident in
HBM before the timed region.  Total work is fixed as N grows (envs and the global minibatch are sharded over ranks):
strong scaling.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      gae_scan (the kernel BASELINE.json grades against the HBM roofline), timed live with HIP events on the
                launch stream at the scaled shape 65536 envs x 128 steps (the 1024 x 128 buffer is 2.2 MB and lives in
                L2); algorithmic bytes = 17 B per env-step.
  roofline_widening  (N = 1) the kernels of the widening rows: token_logprob fwd / bwd at 4096 tokens x
                151936 vocab (bf16 logits), gae_seq at 4096 x 8192, patch_scan over a 622 M-element bf16 tensor, reinpp at
                4096 x 8192 and 32768 x 1024; same
                HIP-event method.
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

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS, HORIZON, OBS_DIM, ACT_DIM = 1024, 128, 42, 8
GLOBAL_BATCH, UPDATE_EPOCH = 8192, 8
GAMMA, LAMBDA = 0.8, 0.9            # examples/embodiment/config/maniskill_ppo_mlp.yaml:63-64
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_cfg(world: int, use_graph: bool, precision: str = "32", pipeline: bool = False, rollout_epochs: int = 1,
              overlap: bool = True):
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
        env=dict(train=dict(rollout_epoch=rollout_epochs, total_num_envs=ENVS, auto_reset=True, ignore_terminations=False,
                            max_episode_steps=50, max_steps_per_rollout_epoch=HORIZON // rollout_epochs, seed=1234, group_size=1)),
        rollout=dict(pipeline_stage_num=1, enable_cuda_graph=use_graph),
        actor=dict(training_backend="fsdp", micro_batch_size=GLOBAL_BATCH // world, global_batch_size=GLOBAL_BATCH,
                   seed=1234, enable_hip_graph=use_graph, optimizer_writes_tiles=bool(int(os.environ.get("RLX_BENCH_OPT_TILES", "1"))),
                   model=dict(model_type="mlp_policy", obs_dim=OBS_DIM, action_dim=ACT_DIM, num_action_chunks=1,
                              precision=precision, add_value_head=True),
                   optim=dict(lr=3e-4, value_lr=3e-4, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                              weight_decay=0.01, clip_grad=0.5),
                   fsdp_config=dict(strategy="fsdp", sharding_strategy="no_shard"))))


def build_runner(cfg, ctx):
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.workers.actor import EmbodiedFSDPActor
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker
    cfg = validate_cfg(cfg)
    actor = EmbodiedFSDPActor.create_group(cfg, ctx).launch(None, name="ActorGroup")
    rollout = MultiStepRolloutWorker.create_group(cfg, ctx).launch(None, name="RolloutGroup")
    env = EnvWorker.create_group(cfg, ctx).launch(None, name="EnvGroup")
    runner = EmbodiedRunner(cfg, actor, rollout, env)
    runner.init_workers()
    return runner


def _pmc_pass(counter: str, workdir: str):
    """One rocprofv3 counter pass (own run, --kernel-trace only: the gpurun rules) over tools/gae_traffic_probe.py ->
    {kernel-name-prefix: mean counter value per launch} or None when rocprofv3 is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    out = os.path.join(workdir, counter)
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "-f", "csv", "--", sys.executable,
                        os.path.join(ROOT, "tools", "gae_traffic_probe.py")], env=env, cwd=workdir, check=True, timeout=240,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return None
    acc = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = "copy" if "dev_stream_copy" in name else ("gae" if "gae_scan" in name else None)
            if key:
                tot, n = acc.get(key, (0.0, 0))
                acc[key] = (tot + float(row["Counter_Value"]), n + 1)
    return {k: t / n for k, (t, n) in acc.items()} if acc else None


def gae_traffic(algo_read: int, algo_write: int):
    """HBM bytes per gae_scan launch from the TCC_EA-derived counters, one --pmc pass per counter, calibrated as the
    microarch guide prescribes: FETCH_SIZE / WRITE_SIZE are in KiB-class units whose scale depends on the access width on
    gfx950, so each is scaled by (known bytes / counter) of a copy kernel with gae_scan's own access pattern (one dword
    per lane, time-major rows)."""
    import tempfile
    cal = 4 * HORIZON * 65536  # the calibration copy reads and writes exactly this many bytes per launch
    with tempfile.TemporaryDirectory(prefix="rlx_pmc_", dir=os.environ.get("TMPDIR", "/tmp")) as wd:
        fetch = _pmc_pass("FETCH_SIZE", wd)
        write = _pmc_pass("WRITE_SIZE", wd)
    if not fetch or not write or "copy" not in fetch or "gae" not in fetch or "copy" not in write or "gae" not in write:
        return None, None
    rd = fetch["gae"] * (cal / fetch["copy"])
    wr = write["gae"] * (cal / write["copy"])
    detail = {"read_bytes": round(rd), "write_bytes": round(wr), "raw_FETCH_SIZE": round(fetch["gae"], 1),
              "raw_WRITE_SIZE": round(write["gae"], 1), "calibration": "dword-per-lane copy of 33.5 MB: "
              f"FETCH_SIZE {fetch['copy']:.1f}, WRITE_SIZE {write['copy']:.1f} per launch",
              "algorithmic_read_bytes": algo_read, "algorithmic_write_bytes": algo_write}
    return round(rd + wr), detail


def gae_roofline(device, iters: int = 30, with_traffic: bool = True):
    """gae_scan (un-normalised, the variant the auto heuristic picks) on 65536 x 128, rotating over buffer sets larger than
    the 256 MiB Infinity Cache so every launch streams from HBM.  `achieved` = algorithmic bytes / average launch duration,
    the average taken with HIP events around `iters` back-to-back launches on the launch stream (per-launch event pairs
    add ~2 us of event overhead to a 28 us kernel; the per-launch numbers are reported next to it)."""
    from rlinf_amd import ops
    T, B, nbuf = HORIZON, 65536, 5
    g = torch.Generator().manual_seed(0)
    bufs = []
    for _ in range(nbuf):
        r = torch.rand(T, B, 1, generator=g).to(device)
        v = torch.randn(T + 1, B, 1, generator=g).to(device)
        d = (torch.rand(T + 1, B, 1, generator=g) < 0.02).to(device)
        bufs.append((r, v, d, torch.empty_like(r), torch.empty_like(r)))

    def launch(i):
        r, v, d, a, q = bufs[i % nbuf]
        ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=False, out=(a, q))

    for i in range(5):
        launch(i)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()  # torch's current stream == the stream the C ABI launches on
    for i in range(iters):
        launch(i)
    e1.record()
    torch.cuda.synchronize(device)
    avg_us = e0.elapsed_time(e1) * 1e3 / iters
    evs = []
    for i in range(iters):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        launch(i)
        a1.record()
        evs.append((a0, a1))
    torch.cuda.synchronize(device)
    single = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
    algo_read, algo_write = 9 * T * B + 5 * B, 8 * T * B  # r 4 + V 4 + done 1 (+ the extra V / done row); adv 4 + ret 4
    algo_bytes = 17 * T * B
    achieved = algo_bytes / avg_us / 1e3  # GB/s
    # contract shape, for reference (launch/latency bound: 2.2 MB)
    r, v, d = (t[:, :ENVS].contiguous() for t in bufs[0][:3])
    a, q = torch.empty_like(r), torch.empty_like(r)
    for _ in range(3):
        ops.gae_scan(r, v, d, None, GAMMA, LAMBDA, out=(a, q))
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(20):
        ops.gae_scan(r, v, d, None, GAMMA, LAMBDA, out=(a, q))
    c1.record()
    torch.cuda.synchronize(device)
    out = {"bound": "hbm", "kernel": "gae_scan_c1<1,1,64,nt> (streaming scan, 65536 envs x 128 steps: the HBM-sized buffer SURVEY.md 8d "
                                      "prescribes for the roofline; at the 1024 x 128 contract shape the loop runs gae_scan_c1<1,8,8> + "
                                      "standardize_kernel out of L2, see contract_shape_us_per_call_incl_normalise)",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
           "traffic": None, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": round(avg_us, 2),
           "launches_timed": iters, "single_launch_event_pair_us": {"median": round(single[len(single) // 2], 2), "min": round(single[0], 2)},
           "frac_of_measured_copy_ceiling_6.29TBps": round(achieved / 6290.0, 4),
           "contract_shape_us_per_call_incl_normalise": round(c0.elapsed_time(c1) * 1e3 / 20, 2)}
    del bufs
    torch.cuda.empty_cache()
    if with_traffic:
        traffic, detail = gae_traffic(algo_read, algo_write)
        out["traffic"] = traffic
        if detail:
            out["traffic_detail"] = detail
    return out


def token_tier_roofline(device, tokens: int = 4096, vocab: int = 151936, iters: int = 10):
    """SURVEY.md 8f item 1 (the widening after rows a-e): the reasoning learner's logits -> log-prob/entropy kernel and
    its backward at a Qwen-size vocabulary, bf16 logits, timed with HIP events on the launch stream.  Algorithmic bytes:
    forward = tokens * vocab * 2 (one read); backward = twice that (one read + one write)."""
    from rlinf_amd import token_ops

    g = torch.Generator(device=device).manual_seed(1)
    x = torch.empty(tokens, vocab, dtype=torch.bfloat16, device=device)
    for i in range(0, tokens, 1024):
        x[i:i + 1024] = (torch.randn(min(1024, tokens - i), vocab, device=device, generator=g) * 4).to(torch.bfloat16)
    labels = torch.randint(0, vocab, (tokens,), device=device, generator=g)
    dlp = torch.randn(tokens, device=device, generator=g)
    out = torch.empty_like(x)
    _, _, lse = token_ops.token_logprob_fwd(x, labels)

    def avg_us(fn):
        for _ in range(2):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize(device)
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3

    rows = []
    by = tokens * vocab * 2
    for name, fn, nbytes in (("token_logprob_fwd", lambda: token_ops.token_logprob_fwd(x, labels), by),
                             ("token_logprob_bwd", lambda: token_ops.token_logprob_bwd(x, labels, lse, None, dlp, None, out=out),
                              2 * by)):
        us = avg_us(fn)
        gbps = nbytes / us / 1e3
        rows.append({"kernel": name, "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(gbps / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1), "algorithmic_bytes": nbytes,
                     "shape": f"{tokens} tokens x {vocab} vocab, bf16 logits"})
    # reasoning GAE along the contiguous axis (12 B/token) and the weight-patch scan (tensor + snapshot read once)
    v = torch.randn(4096, 8192, device=device, generator=g)
    r = torch.randn(4096, device=device, generator=g)
    adv, ret = torch.empty_like(v), torch.empty_like(v)
    lib = _lib_handle()
    st = torch.cuda.current_stream(device).cuda_stream
    gws = torch.empty(lib.rlx_gae_seq_workspace_bytes(4096, 8192), dtype=torch.uint8, device=device)
    us = avg_us(lambda: lib.rlx_gae_seq(v.data_ptr(), r.data_ptr(), adv.data_ptr(), ret.data_ptr(), 4096, 8192, 1.0, 0.95,
                                        gws.data_ptr(), gws.numel(), st))
    nb = v.numel() * 12
    rows.append({"kernel": "gae_seq", "bound": "hbm", "achieved": round(nb / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1), "algorithmic_bytes": nb,
                 "shape": "4096 sequences x 8192 tokens, f32 values"})
    n = x.numel()
    wsb = lib.rlx_patch_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    nnz = torch.zeros(1, dtype=torch.int64, device=device)
    us = avg_us(lambda: lib.rlx_patch_scan(x.data_ptr(), 1, out.data_ptr(), 1, n, ws.data_ptr(), wsb, nnz.data_ptr(), st))
    nb = 2 * n * 2
    rows.append({"kernel": "patch_scan (+ offsets)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1), "peak": HBM_PEAK_GBPS,
                 "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1),
                 "algorithmic_bytes": nb, "shape": f"{n} bf16 elements against their snapshot"})
    # Reinforce++ advantages (returns pass 13 B/token + normalisation 8 B/token; three launches timed together)
    from rlinf_amd import token_ops as _t
    lp = -torch.rand(4096, 8192, device=device, generator=g) * 3
    rlp = lp + 0.3 * torch.randn(4096, 8192, device=device, generator=g)
    msk = torch.ones(4096, 8192, dtype=torch.bool, device=device)
    us = avg_us(lambda: _t.reinpp_seq_adv(r, msk, lp, rlp, 0.001, "low_var_kl"))
    nb = lp.numel() * 21
    rows.append({"kernel": "reinpp_seq_adv (returns + reduce + normalize)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1),
                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1),
                 "algorithmic_bytes": nb, "shape": "4096 sequences x 8192 tokens, k3 KL penalty"})
    # the same bytes as short rows (one 64-lane workgroup per sequence, four tiles each): the shape round 1 was furthest off at
    try:
        r2 = torch.randn(32768, device=device, generator=g)
        lp2, rlp2, msk2 = lp.view(32768, 1024), rlp.view(32768, 1024), msk.view(32768, 1024)
        us = avg_us(lambda: _t.reinpp_seq_adv(r2, msk2, lp2, rlp2, 0.001, "low_var_kl"))
        rows.append({"kernel": "reinpp_seq_adv (returns + reduce + normalize)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4),
                     "avg_launch_us": round(us, 1), "algorithmic_bytes": nb, "shape": "32768 sequences x 1024 tokens, k3 KL penalty"})
        del lp2, rlp2, msk2
    except Exception as e:  # noqa: BLE001 -- an extra row must never cost the others
        rows.append({"kernel": "reinpp_seq_adv", "shape": "32768 sequences x 1024 tokens", "error": f"{type(e).__name__}: {e}"[:200]})
    del lp, rlp, msk
    # bucket weight sync: 16 f32 masters of 4096 x 8192 -> one flat bf16 transport buffer (6 B per element, one launch)
    from rlinf_amd.hybrid_engines.weight_syncer.bucket_syncer import BucketPacker
    masters = [(f"w{i}", torch.randn(4096, 8192, device=device, generator=g), torch.bfloat16) for i in range(16)]
    packer = BucketPacker(masters)
    dev_t = torch.device(device)
    us = avg_us(lambda: packer.pack(masters, dev_t, None, persistent=True))
    nb = 16 * 4096 * 8192 * 6
    rows.append({"kernel": "copy_segments (bucket pack f32 -> bf16)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1),
                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1),
                 "algorithmic_bytes": nb, "shape": "16 tensors x 4096 x 8192 f32 into one bf16 bucket"})
    return rows


def _lib_handle():
    from rlinf_amd import _lib
    return _lib.load()


def token_tier_cpu_baseline(rows: int = 64, vocab: int = 151936):
    """The CPU oracle of the token tier (torch CPU ops in the reference's order: cross_entropy log-prob, log_softmax entropy,
    autograd backward) on a bounded sample of the same shape, all host cores torch picks."""
    from oracle import token_oracle as TO

    g = torch.Generator().manual_seed(1)
    x = (torch.randn(rows, vocab, generator=g) * 4).to(torch.bfloat16).requires_grad_(True)
    labels = torch.randint(0, vocab, (rows,), generator=g)
    t0 = time.perf_counter()
    lp = TO.logprobs_from_logits(x, labels)
    ent = TO.entropy_from_logits(x)
    (lp.sum() - 0.01 * ent.float().sum()).backward()
    dt = time.perf_counter() - t0
    return {"value": round(rows / dt, 1), "unit": "tokens/s (log-prob + entropy forward and backward)",
            "cores": torch.get_num_threads(), "kind": "port", "sample": f"{rows} tokens x {vocab} vocab, bf16 logits"}


def _pick_cpu_threads(pol, budget_s: float = 6.0):
    """torch's default (one thread per logical core) oversubscribes the small GEMMs of this path badly on a
    many-core host (256 threads: ~2.7 s per rollout step).  Time one minibatch-sized forward at a few thread counts and
    keep the fastest -- that is the thread count reported as `cores`."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    x = torch.randn(GLOBAL_BATCH, OBS_DIM)
    best, best_t, t_start = 1, float("inf"), time.perf_counter()
    for nt in (1, 2, 4, 8, 16, 32, 64, 128):
        if nt > avail or time.perf_counter() - t_start > budget_s:
            break
        torch.set_num_threads(nt)
        with torch.no_grad():
            pol.value_head.mlp(x)
            t0 = time.perf_counter()
            for _ in range(3):
                pol.value_head.mlp(x)
            dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    return best, avail


def cpu_baseline_reference():
    """kind "reference": the reference's OWN functions on the host cores (oracle/reference_loop.py: its MLPPolicy, its
    calculate_adv_and_returns and policy_loss with their built-in callees, torch's clip_grad_norm_ + AdamW), loaded from the
    staged copy oracle/_ref (or /root/reference).  One full rollout + advantage pass, then 3 untimed + 10 timed optimizer steps;
    the update phase is the median step time x 128 (every minibatch step does identical work)."""
    from oracle import ppo_loop as L
    from oracle import reference_loader as RL
    from oracle import reference_loop as RLoop
    if not RL.available():
        return None
    ref, pol, _ = RLoop.build(OBS_DIM, ACT_DIM)
    threads, avail = _pick_cpu_threads(pol)
    env = L.synthetic_env_tensors(1234, HORIZON, ENVS, OBS_DIM, max_episode_steps=50)
    t0 = time.perf_counter()
    t = RLoop.timed_iteration(env, obs_dim=OBS_DIM, act_dim=ACT_DIM, gamma=GAMMA, gae_lambda=LAMBDA, global_batch=GLOBAL_BATCH,
                              update_epoch=UPDATE_EPOCH, warmup_steps=3, timed_steps=10)
    spent = time.perf_counter() - t0
    iter_s = t["rollout_s"] + t["advantages_s"] + t["update_s_per_step"] * t["update_steps_total"]
    return {"value": round(ENVS * HORIZON / iter_s, 1), "unit": "env-steps/s", "cores": threads, "kind": "reference",
            "sample": f"the reference's own MLPPolicy / calculate_adv_and_returns / policy_loss + torch clip_grad_norm_ / AdamW "
                      f"(files staged under oracle/_ref): 1 full rollout of {HORIZON}x{ENVS} + GAE, then 3 warm-up + 10 timed "
                      f"optimizer steps of {GLOBAL_BATCH} rows, update phase = median step x {t['update_steps_total']}; "
                      f"{spent:.1f} s of CPU work; torch threads picked by a timing probe out of {avail} schedulable cores",
            "host_cores": avail, "updates_per_sec": round(1.0 / t["update_s_per_step"], 2), "rollout_s": round(t["rollout_s"], 3),
            "advantages_s": round(t["advantages_s"], 4), "update_s_per_step": round(t["update_s_per_step"], 4),
            "update_step_times_s": t["step_times_s"], "iteration_s": round(iter_s, 3)}


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
    threads, avail = _pick_cpu_threads(pol)
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
                      f"(rest of the update phase extrapolated); {t4 - t0:.1f} s of CPU work; torch threads picked by a "
                      f"timing probe out of {avail} schedulable cores",
            "host_cores": avail, "updates_per_sec": round(1.0 / per_update, 2), "rollout_s": round(t1 - t0, 3),
            "advantages_s": round(t2 - t1, 4), "update_s_per_step": round(per_update, 4),
            "iteration_s": round(iter_s, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "32"],
                    help="operand precision of the policy's dense layers: bf16 (BASELINE.json configs[1], f32 accumulate and "
                         "master weights) or 32 (exact-f32 MFMA)")
    ap.add_argument("--pipeline", action="store_true", help="runner.use_training_pipeline (statistics normalisation, per-stage "
                    "shuffles; with --rollout-epochs > 1 the learner trains on epoch e while epoch e + 1 rolls out)")
    ap.add_argument("--rollout-epochs", type=int, default=1, help="split the 128-step horizon into this many rollout epochs")
    ap.add_argument("--no-overlap", action="store_true", help="pipeline mode on ONE stream (the comparison line for the overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-token-tier", action="store_true", help="skip the token-tier (LLM logits) roofline rows")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes behind roofline.traffic")
    args = ap.parse_args()

    from rlinf_amd.scheduler import init_distributed
    import torch.distributed as dist

    ctx = init_distributed()
    if ctx.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world_size}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    dev = ctx.device
    use_graph = not args.no_graph
    runner = build_runner(build_cfg(ctx.world_size, use_graph, args.precision, pipeline=args.pipeline,
                                    rollout_epochs=args.rollout_epochs, overlap=not args.no_overlap), ctx)

    def barrier():
        if ctx.world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 2 if use_graph else 0)):  # first graph call runs eagerly and captures
        runner.run_step()
    barrier()
    t0 = time.perf_counter()
    phase = {"env": 0.0}
    for _ in range(args.steps):
        m = runner.run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if ctx.world_size > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = None
    if ctx.rank == 0:
        updates = (ENVS * HORIZON // GLOBAL_BATCH) * UPDATE_EPOCH
        line = {
            "metric": "env_steps_per_sec", "value": round(ENVS * HORIZON * args.steps / elapsed, 1), "unit": "env-steps/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "ManiSkill PickCube-shaped PPO: 1024 envs x 128 steps, obs 42, act 8, MLP policy "
                                   "(3x256 tanh actor + value head), gamma 0.8 / lambda 0.9, 8 epochs x 16 minibatches "
                                   "of 8192 (128 optimizer steps), " + ("bf16 MFMA operands / f32 accumulate, master weights, losses, GAE, AdamW" if args.precision == "bf16" else "exact-f32 MFMA") + ", synthetic env tensors resident in HBM",
                       "total_envs": ENVS, "horizon": HORIZON, "global_batch": GLOBAL_BATCH, "update_epoch": UPDATE_EPOCH,
                       "parallelism": f"dp{args.gpus}", "hip_graph": use_graph,
                       **({"pipeline": True, "rollout_epochs": args.rollout_epochs, "overlap": not args.no_overlap}
                          if args.pipeline else {})},
            "ppo_updates_per_sec": round(updates * args.steps / elapsed, 1),
            # end-to-end parity AT THIS configuration (1024 x 128, 8192-row minibatches, 128 optimizer steps, hipGraph replay):
            "parity_checked": ("tests/test_end_to_end_bench_config.py::test_bench_configuration_"
                               + ("bf16_matches_autocast_oracle_with_graph_replay" if args.precision == "bf16" else "f32_matches_oracle[True]")
                               + " + ::test_first_optimizer_step_gradient_at_bench_configuration (runner built by this file's "
                                 "build_cfg / build_runner, compared with oracle.ppo_loop.iteration)"),
            "last_metrics": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in m.items()
                             if k in ("train/actor/total_loss", "train/actor/grad_norm", "train/actor/approx_kl",
                                      "rollout/rewards")},
        }
        def extra(key, fn):
            """The headline fields above are already measured: a failure in a secondary probe is reported in the line
            instead of costing it."""
            try:
                line[key] = fn()
            except Exception as e:  # noqa: BLE001
                line[key] = None
                line.setdefault("probe_errors", {})[key] = f"{type(e).__name__}: {e}"[:300]

        if not args.no_roofline:
            extra("roofline", lambda: gae_roofline(dev, with_traffic=not args.no_traffic))
            if args.gpus == 1 and not args.no_token_tier:
                extra("roofline_widening", lambda: token_tier_roofline(dev))
        if args.gpus == 1 and not args.no_cpu_baseline:
            extra("cpu_baseline", cpu_baseline)
            if line.get("roofline_widening"):
                extra("cpu_baseline_token_tier", token_tier_cpu_baseline)
            if line.get("cpu_baseline"):
                line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if ctx.world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

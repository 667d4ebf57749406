"""The N > 1 path: env axis sharded over ranks, rank-local rollout / GAE / shuffle, ONE gradient
all-reduce (mean) per optimizer step, metrics reduced once per iteration.  world_size 2:
  * on CPU over gloo: the host-side arithmetic (shards, minibatch plan, routing maps, the two collectives);
  * on the GPU box (-m gpu): the whole runner as two processes against a two-shard emulation with the CPU oracle."""

import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

WORKER = os.path.join(ROOT, "tests", "helpers", "dp_worker.py")


def _launch(mode, tmp_path, *extra, port=29611, timeout=300, one_device=True, world=2, env_extra=None):
    procs, outs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0" if one_device else str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
        out = str(tmp_path / f"rank{rank}.out")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, out, *extra], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(log)
    if any(p.returncode != 0 for p in procs):  # the first rank to die is the one with the story; show every rank
        raise AssertionError("\n".join(f"---- rank {r} rc={p.returncode} ----\n{log[-2500:]}" for r, (p, log) in enumerate(zip(procs, logs))))
    return outs


def test_world_size_2_host_logic_over_gloo(tmp_path):
    r0, r1 = (json.load(open(o)) for o in _launch("cpu", tmp_path))
    assert r0["shard"] == [0, 512] and r1["shard"] == [512, 1024]
    assert r0["plan"] == r1["plan"] == [16, 4096, 1]  # 16 minibatches of 4096 per rank, no accumulation
    assert r0["split"] == 1
    assert r0["grad_mean"] == r1["grad_mean"] == 1.5
    assert r0["sums"] == r1["sums"] == [30.0, 8.0] and r0["maxs"] == r1["maxs"] == [0.0, 6.0]
    assert r0["send"] == [[0, 32, 0]] and r1["send"] == [[1, 32, 0]] and r0["recv"] == [[0, 32, 0]]
    assert r0["dst_4_to_2"] == [[0, 16]] and r1["dst_4_to_2"] == [[0, 16]]
    # the weight patch the actor rank broadcasts arrives byte for byte (header + one broadcast per field)
    assert r0["patch_ok"] and r1["patch_ok"] and r0["empty_ok"] and r1["empty_ok"]
    assert r0["bucket_ok"] and r1["bucket_ok"]  # a weight bucket: pickled layout + one broadcast of the flat byte buffer
    # FSDPActor._dp_load_balance: the 8 sequences re-dealt 4 + 4 by the Karmarkar-Karp partitions of the GATHERED lengths
    from rlinf_amd.workers.actor.fsdp_actor_worker import seqlen_balanced_partitions
    lens = r0["balance_lens_in"] + r1["balance_lens_in"]
    parts = seqlen_balanced_partitions(lens, 2, True)
    ids = [0, 1, 2, 3, 100, 101, 102, 103]
    rew = [0.0, 1.0, 2.0, 3.0, 10.0, 11.0, 12.0, 13.0]
    for r, part in zip((r0, r1), parts):
        assert r["balance_ids"] == [ids[i] for i in part] and r["balance_rewards"] == [rew[i] for i in part]
        assert r["balance_tokens"] == sum(lens[i] for i in part)
    assert abs(r0["balance_tokens"] - r1["balance_tokens"]) <= max(lens)           # balanced by tokens, not just by count
    assert abs(sum(r0["balance_lens_in"]) - sum(r1["balance_lens_in"])) >= abs(r0["balance_tokens"] - r1["balance_tokens"])


def _two_gpus():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


@pytest.mark.gpu
def test_xgmi_allreduce_two_ranks_on_one_gpu(tmp_path):
    """The transport alone (csrc/xgmi_allreduce.hip through scheduler/xgmi.py), two processes sharing the ONE GPU of the test
    box: IPC export / mapping of the fine-grained buffer, the flag hand-shake, both staging slots, slab sums, the scale, BOTH
    forms of the exchange (direct and reduce-scatter + all-gather: scheduler.xgmi._attempt validates both), eager launches and a
    replayed hipGraph of back-to-back all-reduces -- against torch.distributed's own all-reduce.  The communicator sees that the
    ranks share a device and runs the hand-shake as a one-wave launch of its own (wait_mode "kernel"), so neither rank holds the
    GPU with a device-wide spin while the other one still has to get its staging launch scheduled: no starvation escape here."""
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641",
                   RLX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", RLX_XGMI_TIMEOUT_MS="30000", RLX_XGMI_PROBE_LIGHT="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "xgmi_probe.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            logs.append(p.communicate()[0] or "")
    joined = "\n".join(l[-2000:] for l in logs)
    assert all(p.returncode == 0 for p in procs), joined
    out = logs[0]
    assert "shared device: True" in out and "wait_mode kernel" in out, out[-2000:]
    assert "mem_kind 0: OK" in out or "mem_kind 1: OK" in out, out[-2000:]  # a coherent (fine-grained / uncached) kind works
    for algo in ("direct", "rsag"):
        assert f"{algo}: result ok after graph: True" in out, out[-2000:]
    assert "result ok after graph: False" not in out, out[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("world,algo", [(2, "direct"), (2, "rsag"), (3, "direct"), (3, "rsag"), (4, "direct"), (4, "rsag"),
                                        (8, "direct"), (8, "rsag")])
def test_xgmi_exchange_in_process_group(world, algo):
    """Both forms of the exchange at W = 2 .. 8 on ONE device: W communicators of one process wired to each other directly
    (rlx_xgmi_connect_local), every "rank" on its own stream (own hardware queue: GPU_MAX_HW_QUEUES raised for that, hence the
    subprocess).  Multi-slab inputs, both staging slots, ragged n (last shard short), n not divisible by 4 (rs + ag falls back to
    the direct form on every rank alike), bit-identical results on all ranks, and against a float64 sum."""
    code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from rlinf_amd.scheduler.xgmi import LocalXgmiGroup
W, algo = {world}, {algo!r}
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
streams = [torch.cuda.Stream(dev) for _ in range(W)]   # ONE hardware queue per "rank", reused by every group below
for n in (287504, 4 * W + 4, 1030, 4 * 257 * W + 4):
    grp = LocalXgmiGroup(W, n, dev, algo=algo, timeout_ms=15000, streams=streams)
    for it in range(5):
        xs = [torch.randn(1 + (r + it) % 3, n, device=dev, generator=g) for r in range(W)]
        outs = [torch.full((n,), float("nan"), device=dev) for _ in range(W)]
        grp.all_reduce(xs, outs, scale=1.0 / W)
        torch.cuda.synchronize()
        assert grp.status_ok(), f"n={{n}} it={{it}}: a peer wait timed out"
        want = sum(x.double().sum(0) for x in xs) / W
        for r in range(W):
            assert torch.equal(outs[r], outs[0]), f"n={{n}} it={{it}}: rank {{r}} differs from rank 0"
        err = float((outs[0].double() - want).abs().max())
        assert err < 1e-5, (n, it, err)
    grp.close()
print("OK")
"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("world,algo", [(2, "direct"), (4, "rsag"), (8, "rsag")])
def test_xgmi_clip_adamw_step_in_process_group(world, algo):
    """The optimizer entry point over the exchange, W "ranks" of one process on one device, several steps in a row: the
    reduce-scatter + all-gather form runs the gather INSIDE the AdamW launch, which also advances the exchange's sequence word --
    its blocks must read the snapshot the reduce-scatter launch left, never the word being incremented (a late block would read the
    other staging slot or wait for a flag nobody publishes).  Every rank must end every step with bit-identical parameters and
    moments, equal to plain clip + AdamW on the gradient mean."""
    code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from rlinf_amd import ops
from rlinf_amd.scheduler.xgmi import LocalXgmiGroup
W, algo = {world}, {algo!r}
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
n = 287504
groups = [(0, n // 2, 3e-4), (n // 2, n, 1e-3)]
streams = [torch.cuda.Stream(dev) for _ in range(W)]
grp = LocalXgmiGroup(W, n, dev, algo=algo, timeout_ms=15000, streams=streams)
p0 = torch.randn(n, device=dev, generator=g) * 0.1
params = [p0.clone() for _ in range(W)]
m = [torch.zeros(n, device=dev) for _ in range(W)]
v = [torch.zeros(n, device=dev) for _ in range(W)]
flat = [torch.empty(n, device=dev) for _ in range(W)]
stats = [torch.zeros(2, device=dev) for _ in range(W)]
state = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(W)]
ref_p, ref_m, ref_v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
for it in range(6):
    slabs = [torch.randn(1 + (r + it) % 3, n, device=dev, generator=g) * (5.0 if it % 2 else 0.01) for r in range(W)]
    keep = grp.clip_adamw_step(params, slabs, flat, m, v, groups, stats, state)
    torch.cuda.synchronize()
    assert grp.status_ok(), f"it={{it}}: a peer wait timed out"
    for r in range(1, W):
        assert torch.equal(params[r], params[0]) and torch.equal(m[r], m[0]) and torch.equal(v[r], v[0]), f"it={{it}}: rank {{r}} diverged"
        assert torch.equal(flat[r], flat[0]) and torch.equal(stats[r], stats[0])
    mean = (sum(x.double().sum(0) for x in slabs) / W).float()
    ops.clip_adamw_step_(ref_p, mean.clone(), ref_m, ref_v, groups, it + 1, max_grad_norm=0.5)
    # (the reference adds the ranks' slabs in another order: AdamW's m / sqrt(v) amplifies the last-bit differences of tiny gradients)
    torch.testing.assert_close(params[0], ref_p, rtol=2e-3, atol=2e-5)
    assert float(stats[0][1]) == 1.0
    assert all(int(s[0]) + int(s[1]) == it + 1 for s in state)
grp.close()
print("OK")
"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def _compare_with_the_sharded_oracle(outs, world):
    """The reference arithmetic applied shard by shard for `world` ranks: per-rank rollout on its env slice, per-shard advantage
    normalisation, per-rank shuffle with seed + rank, per-rank minibatches of global_batch / world, gradient MEAN, identical
    clip + AdamW on every rank -- against what every rank dumped after its first iteration."""
    from oracle import ppo_loop as L
    from oracle import ppo_oracle as O
    T, B, GB = 12, 64, 192
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, 8, 1)
    opt = O.build_adamw(ora)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    shards = []
    for r in range(world):
        sl = slice(r * B // world, (r + 1) * B // world)
        env_r = {k: v[:, sl].contiguous() for k, v in env.items()}
        batch = L.advantages(L.rollout(ora, env_r, eps[:, sl], 0.8, True), 0.8, 0.9, True)
        perm = torch.randperm(T * B // world, generator=torch.Generator().manual_seed(1234 + r))
        shards.append((batch, O.flatten_and_shuffle(batch, perm)))
    n_mb = (T * B // world) // (GB // world)
    for _ in range(2):  # update_epoch
        chunks = [O.chunk_batch(flat, n_mb) for _, flat in shards]
        for i in range(n_mb):
            opt.zero_grad()
            for r in range(world):
                mb = chunks[r][i]
                out = ora.evaluate(mb["forward_inputs"]["states"], mb["forward_inputs"]["action"])
                shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], "action_level", 8,
                                             values=out["values"], prev_values=mb["prev_values"], returns=mb["returns"])
                loss, _ = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
                (loss / world).backward()  # DDP / FSDP average the per-rank gradients
            gn = torch.nn.utils.clip_grad_norm_(ora.parameters(), 0.5)
            if torch.isfinite(gn):
                opt.step()
    want = torch.cat([p.detach().reshape(-1) for p in ora.parameters()])
    for r, o in enumerate(sorted(outs, key=lambda d: d["rank"])):
        batch = shards[r][0]
        torch.testing.assert_close(o["actions"], batch["forward_inputs"]["action"], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(o["rewards"], batch["rewards"], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(o["returns"], batch["returns"], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(o["advantages"], batch["advantages"], rtol=1e-3, atol=1e-4)  # normalised per shard
        diff = (o["params"] - want).abs()
        steps = 2 * n_mb
        assert float(diff.max()) <= 2 * 3e-4 * steps + 1e-6
        assert float((diff > 2e-5).float().mean()) < 0.02
    for o in outs[1:]:
        assert torch.equal(outs[0]["params"], o["params"])  # every rank holds the same weights after the update


@pytest.mark.gpu
@pytest.mark.parametrize("transport,graph", [("xgmi", "1"), ("rccl", "0")])
def test_async_learner_two_ranks_match_the_sharded_oracle(tmp_path, transport, graph):
    """The async (decoupled PPO) learner on its fused path at world_size 2 (two processes sharing the GPU): the actor network's
    gradients leave rlx_ppo_step in sum form and each rank multiplies ITS slab groups by ITS micro-batches' 1 / behaviour-mask count
    where it collapses them -- the xGMI staging launch, or rlx_sum_slabs_deferred in front of the (gloo) all-reduce -- before the mean
    over ranks.  Against the reference arithmetic applied shard by shard: per-rank shuffle (seed + rank), masked_normalization of
    the advantages with statistics over BOTH shards, per-rank micro-batches, (loss / accumulation / world).backward()."""
    from oracle import ppo_loop as L
    from oracle import ppo_oracle as O
    world, T, B, GB = 2, 12, 64, 192
    outs = [torch.load(o, weights_only=False) for o in _launch("gpu_async", tmp_path, "32", transport, graph, "gloo",
                                                             port=29691 + (transport == "rccl"), timeout=600, world=world)]
    assert all(o["fused"] for o in outs) and all(o["graph"] == (graph == "1") for o in outs)
    assert all(o["backend"] == (transport if transport == "xgmi" else "rccl") for o in outs), [o["backend"] for o in outs]
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, 8, 1)
    opt = O.build_adamw(ora)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    batches = []
    for r in range(world):
        sl = slice(r * B // world, (r + 1) * B // world)
        batch = L.advantages(L.rollout(ora, {k: v[:, sl].contiguous() for k, v in env.items()}, eps[:, sl], 0.8, True), 0.8, 0.9, True)
        batch["versions"] = torch.zeros_like(batch["prev_logprobs"])
        batches.append(batch)

    def sharded_async_update(version):
        per_rank, micro = GB // world, GB // world // 2
        flats = [O.flatten_and_shuffle(b, torch.randperm(T * B // world, generator=torch.Generator().manual_seed(1234 + r)))
                 for r, b in enumerate(batches)]
        adv = O.masked_normalization(torch.cat([f["advantages"] for f in flats]), None)  # statistics over every rank's rows
        for r, f in enumerate(flats):
            f["advantages"] = adv[r * (T * B // world):(r + 1) * (T * B // world)]
        n_global, accum = (T * B // world) // per_rank, per_rank // micro
        for _ in range(2):  # update_epoch
            chunks = [O.chunk_batch(f, n_global) for f in flats]
            for i in range(n_global):
                opt.zero_grad()
                for r in range(world):
                    for mb in O.chunk_batch(chunks[r][i], accum):
                        out = ora.evaluate(mb["forward_inputs"]["states"], mb["forward_inputs"]["action"])
                        shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], "action_level", 8,
                                                     values=out["values"], prev_values=mb["prev_values"], returns=mb["returns"])
                        _, ver = O.shape_decoupled_inputs(None, mb["versions"], "action_level", 8, out["logprobs"].shape[0],
                                                          shaped["logprobs"].shape)
                        loss, _ = O.decoupled_actor_critic_loss(versions=ver, current_version=version + 1, behave_weight_threshold=1.01,
                                                                clip_ratio_low=0.2, clip_ratio_high=0.2, clip_ratio_c=3.0, value_clip=1.0,
                                                                huber_delta=10.0, **shaped)
                        (loss / accum / world).backward()
                gn = torch.nn.utils.clip_grad_norm_(ora.parameters(), 0.5)
                if torch.isfinite(gn):
                    opt.step()
        return torch.cat([p.detach().reshape(-1) for p in ora.parameters()])

    steps = 2 * ((T * B // world) // (GB // world))
    for phase, key in enumerate(("first_params", "final_params")):
        if phase == 1:
            for b in batches:
                b["versions"][T // 2:] += 1.0
        want = sharded_async_update(version=0 if phase == 0 else 2)
        for o in outs:
            diff = (o[key] - want).abs()
            assert float(diff.max()) <= 2 * 3e-4 * steps * (phase + 1) + 1e-6, (key, float(diff.max()))
            assert float((diff > 2e-5 * (phase + 1)).float().mean()) < 0.02, (key, float((diff > 2e-5).float().mean()))
        assert torch.equal(outs[0][key], outs[1][key])  # both ranks hold the same weights


@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["1"])  # (iteration 0 of the graph case runs eagerly and captures: both launch styles in one ~2-minute run)
def test_eight_ranks_on_one_gpu_match_the_sharded_oracle(tmp_path, graph):
    """The WHOLE learner at world_size 8 -- the size the scaling benchmark runs at -- as eight processes sharing the one GPU of the
    test box: env shards of 8, per-rank minibatches, the hand-written exchange in the form it takes from four ranks on
    (reduce-scatter + the all-gather fused into the AdamW launch, hand-shake as a one-wave launch because the ranks share a
    device), eager launches and the update phase as a replayed hipGraph on all eight ranks, against the reference arithmetic
    applied shard by shard.  (Until round 4 world_size 8 was exercised in-process only, on the bare transport.)"""
    outs = [torch.load(o, weights_only=False)
            for o in _launch("gpu", tmp_path, "32", "xgmi", graph, "gloo", port=29671 + 2 * (graph == "1"), timeout=900, one_device=True, world=8,
                             env_extra={"RLX_XGMI_TIMEOUT_MS": "120000"})]
    assert len(outs) == 8 and all(o["backend"] == "xgmi" for o in outs), [o["backend"] for o in outs]
    assert all(o["xgmi"] == dict(algo="rsag", wait_mode="kernel", shared_device=True) for o in outs), outs[0]["xgmi"]
    if graph == "1":
        assert all(o["iters"] == 3 and o["graph_live"] for o in outs), [(o["iters"], o["graph_live"]) for o in outs]
        assert all(torch.equal(outs[0]["final_params"], o["final_params"]) for o in outs) and torch.isfinite(outs[0]["final_params"]).all()
    assert all(o["sync_ok"] for o in outs)
    _compare_with_the_sharded_oracle(outs, world=8)


@pytest.mark.gpu
@pytest.mark.parametrize("transport,graph,backend", [
    ("rccl", "0", "gloo"),   # torch.distributed all-reduce (gloo stands in for RCCL: it needs a device per rank)
    ("rccl", "1", "gloo"),   # a gloo all-reduce cannot be stream-captured: the worker keeps the eager (prepared-launch) loop
    # one rank per GPU (the box has two): RCCL captured in the update graph (on a one-GPU box that capture is exercised by
    # test_one_rank_drives_the_real_rccl_branch below, so the case is simply absent there), and the xGMI transport for real
    *([("rccl", "1", "nccl")] if _two_gpus() else []),
    pytest.param("xgmi", "0", "nccl", marks=pytest.mark.skipif(not _two_gpus(), reason="needs one GPU per rank")),
    pytest.param("xgmi", "1", "nccl", marks=pytest.mark.skipif(not _two_gpus(), reason="needs one GPU per rank")),
    # the whole learner over the hand-written exchange with both ranks on ONE GPU (hand-shake as its own one-wave launch):
    ("xgmi", "0", "gloo"),   # eager launches
    ("xgmi", "1", "gloo"),   # the update phase as one replayed hipGraph on both ranks
])
def test_two_ranks_match_the_sharded_oracle(tmp_path, transport, graph, backend, precision="32"):
    """Two processes against the reference arithmetic applied shard by shard: per-rank rollout on its env half, per-shard
    advantage normalisation, per-rank shuffle with seed + rank, per-rank minibatches of global_batch / 2, gradient mean
    (FSDP NO_SHARD, rlinf/hybrid_engines/fsdp/strategy/fsdp.py:480-496), identical clip + AdamW on every rank.  The two
    nccl cases run when the box has two GPUs (one rank per GPU, RCCL / xGMI for real)."""
    outs = [torch.load(o, weights_only=False)
            for o in _launch("gpu", tmp_path, precision, transport, graph, backend, port=29613 + 2 * (graph == "1") + 4 * (transport == "rccl") + 8 * (backend == "nccl"),
                             timeout=300, one_device=backend != "nccl")]
    assert all(o["backend"] == transport for o in outs), [o["backend"] for o in outs]
    if graph == "1":
        assert all(o["iters"] == 3 for o in outs)
        live = [o["graph_live"] for o in outs]
        assert live[0] == live[1], "both ranks replay a graph or neither does"
        assert live[0] == (transport == "xgmi" or backend == "nccl"), live
        assert torch.equal(outs[0]["final_params"], outs[1]["final_params"]) and torch.isfinite(outs[0]["final_params"]).all()
    # the weight patch built by rank 0's kernels, broadcast, applied by rank 1's kernels onto its bf16 replica
    assert all(o["sync_ok"] for o in outs) and outs[1]["sync_version"] == 3
    assert outs[0]["patch_nnz"] == outs[1]["patch_nnz"] and sum(outs[0]["patch_nnz"]) > 0
    _compare_with_the_sharded_oracle(outs, world=2)
    assert torch.equal(outs[0]["params"], outs[1]["params"])  # both ranks hold the same weights after the update


@pytest.mark.gpu
@pytest.mark.parametrize("transport,graph", [("rccl", "0"), ("rccl", "1"), ("xgmi", "1")])
def test_one_rank_drives_the_real_rccl_branch(tmp_path, transport, graph):
    """RCCL itself, on a one-GPU box: a ONE-rank `nccl` communicator is legal, so under RLX_FORCE_EXCHANGE the learner takes its own
    multi-GPU branch at world_size 1 -- rlx_sum_slabs -> dist.all_reduce on RCCL (stream-ordered) -> rlx_clip_adamw_step with the
    reduced flat gradient -- eagerly ("0") and inside the captured update graph ("1": iteration 0 captures the update phase WITH the
    RCCL call in it, two more iterations replay it), against the single-rank oracle (rlinf/hybrid_engines/fsdp/strategy/fsdp.py:480-496
    with world 1).  "xgmi": the hand-written exchange's one-rank communicator, validated at start-up against that RCCL all-reduce."""
    outs = [torch.load(o, weights_only=False)
            for o in _launch("gpu", tmp_path, "32", transport, graph, "nccl", port=29651 + 2 * (graph == "1") + 4 * (transport == "xgmi"),
                             timeout=300, world=1, env_extra={"RLX_FORCE_EXCHANGE": transport})]
    assert len(outs) == 1 and outs[0]["backend"] == transport, outs[0]["backend"]
    assert outs[0]["dist_backend"] == "nccl"
    if graph == "1":
        assert outs[0]["iters"] == 3 and outs[0]["graph_live"], "the update phase -- RCCL call included -- was captured and replayed"
        assert torch.isfinite(outs[0]["final_params"]).all()
    _compare_with_the_sharded_oracle(outs, world=1)


@pytest.mark.gpu
def test_bench_self_launches_its_ranks_from_a_bare_shell(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a torchrun environment: bench.py spawns the two ranks itself, times every
    (regime, transport) pair and rank 0 prints ONE JSON line with the strong-scaling headline, the weak-scaling object and the
    per-transport table.  (One-GPU box: both ranks share the device -- RLX_BENCH_ALLOW_SHARED_GPU -- so gloo stands in for RCCL
    and the xGMI exchange runs with its hand-shake as a one-wave launch; the numbers mean nothing here, the launcher, the
    collective agreement on which runs exist and the line's schema are what is checked.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RLX_BENCH_ALLOW_SHARED_GPU="1", RLX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", RLX_XGMI_TIMEOUT_MS="60000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline",
                        "--launch-timeout", "600"], env=env, capture_output=True, text=True, timeout=700)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2 and line["value"] > 0
    assert line["config"]["total_envs"] == 1024 and line["config"]["global_batch"] == 8192 and line["config"]["parallelism"] == "dp2"
    assert line["weak_scaling"]["total_envs"] == 2048 and line["weak_scaling"]["global_batch"] == 16384
    assert set(line["transports"]) == {"strong/rccl-eager", "strong/xgmi", "strong/rccl", "weak/xgmi", "weak/rccl"}, line.get("transport_errors")
    assert line["transports"]["strong/rccl-eager"]["update_graph_replayed"] is False
    # who took part, and which run may carry `value`: both ranks share the box's one GPU, so the hand-written exchange -- validated,
    # but not over links -- must NOT be the headline: the value comes from a torch.distributed run
    assert line["ranks_seen"] == 2 and line["devices_seen"] == 1
    assert line["xgmi_validation"]["passed"] is True and line["xgmi_validation"]["ranks_share_a_device"] is True
    assert line["value_transport"]["trusted"] is True and line["value_transport"]["asked"] in ("rccl-eager", "rccl")
    assert line["value"] == line["transports"]["strong/" + line["value_transport"]["asked"]]["env_steps_per_sec"]
    assert line["transports"]["strong/xgmi"]["grad_allreduce"].startswith("xgmi (direct, kernel hand-shake)")
    assert line["transports"]["strong/xgmi"]["update_graph_replayed"] is True    # a pure kernel chain: captured at any world size
    assert line["transports"]["strong/rccl"]["update_graph_replayed"] is False   # gloo cannot be captured: eager fallback, both ranks


# ---- split placement (actor ranks != rollout ranks) + the weight syncers in the worker path -----------------------
def test_split_placement_host_logic_over_gloo(tmp_path):
    """component_placement ``actor: 0`` / ``env,rollout: 1`` at world 2, CPU: rank sets, per-component contexts and groups, the
    weight link's closures (metadata point to point, bucket / patch / empty patch over the sync group's broadcast) and the 1 : 1
    trajectory route (bool fields travel as bytes)."""
    r0, r1 = (json.load(open(o)) for o in _launch("split_cpu", tmp_path, port=29691))
    for r in (r0, r1):
        assert r["split"] is True and r["worlds"] == [1, 1, 1]
    assert r0["present"] == [True, False, False] and r1["present"] == [False, True, True]
    assert r0["component_ctx"] == [0, 1, [0]] and r1["component_ctx"] == [0, 1, [1]]
    assert r0["meta"]["ordered_keys"] == ["a", "b"] and r0["meta"]["receiver_dtypes"] == {"a": "torch.bfloat16"}
    assert r1["bucket_ok"] and r1["patch_ok"] and r1["empty_ok"] and r0["traj_ok"]


def test_component_placement_parsing():
    from rlinf_amd.utils.placement import parse_component_placement, parse_rank_spec
    assert parse_rank_spec("all", 4) == [0, 1, 2, 3] and parse_rank_spec(2, 4) == [2] and parse_rank_spec("0-1,3", 4) == [0, 1, 3]
    assert parse_component_placement({"actor": "0-3", "env,rollout": "4-7"}, 8) == {
        "actor": [0, 1, 2, 3], "env": [4, 5, 6, 7], "rollout": [4, 5, 6, 7]}
    assert parse_component_placement({"env,rollout,actor": 0}, 1) == {"env": [0], "rollout": [0], "actor": [0]}
    with pytest.raises(ValueError):
        parse_component_placement({"actor": "0", "actor,env": "1"}, 2)
    with pytest.raises(ValueError):
        parse_rank_spec("3-1", 4)


@pytest.mark.gpu
@pytest.mark.parametrize("syncer,init_sync", [("bucket", "0"), ("patch", "0"), ("patch", "1"), ("patch_cpu", "1")])
def test_split_placement_whole_loop_matches_the_collocated_run(tmp_path, syncer, init_sync):
    """Learner on rank 0, env + rollout on rank 1 (two processes on the one GPU, gloo standing in for RCCL) against the SAME job in
    one process: after every sync the rollout worker's weights are the learner's byte for byte (so every iteration's rollout,
    advantages and update are the collocated run's: metrics equal), versions advance with the global step, and the syncer ran
    its init hand-shake once on each side."""
    (tmp_path / "split").mkdir(), (tmp_path / "one").mkdir()
    split = [torch.load(o, weights_only=False) for o in _launch("split_gpu", tmp_path / "split", syncer, init_sync, "gloo", "split",
                                                               port=29693 + 2 * (syncer == "patch") + 4 * (init_sync == "1") + 16 * (syncer == "patch_cpu"))]
    one = [torch.load(o, weights_only=False) for o in _launch("split_gpu", tmp_path / "one", syncer, init_sync, "gloo", "collocated",
                                                             world=1, port=29701 + 2 * (syncer == "patch") + 4 * (init_sync == "1") + 16 * (syncer == "patch_cpu"))][0]
    learner, rollout = split
    assert learner["has_actor"] and not learner["has_rollout"] and rollout["has_rollout"] and not rollout["has_actor"]
    assert learner["actor_world"] == 1 and rollout["env_world"] == 1 and rollout["num_envs"] == 32
    assert learner["sender_initialized"] and rollout["receiver_initialized"] and not rollout["shares"]
    # weights: the rollout rank's copy after the closing sync == the learner's final weights, byte for byte
    assert torch.equal(rollout["rollout_params"].view(torch.int32), learner["actor_params"].view(torch.int32))
    assert rollout["versions"] == [0, 1, 2] and rollout["final_version"] == 3
    # the single-process run with its own rollout copy over the in-process link: same weights, same versions
    assert one["has_actor"] and one["has_rollout"] and not one["shares"]
    assert torch.equal(one["rollout_params"].view(torch.int32), one["actor_params"].view(torch.int32))
    assert one["versions"] == [0, 1, 2] and one["final_version"] == 3
    assert torch.equal(one["actor_params"].view(torch.int32), learner["actor_params"].view(torch.int32))
    for it in range(3):
        got, want = learner["metrics"][it], one["metrics"][it]
        keys = [k for k in want if k.startswith(("train/", "rollout/"))]
        assert keys and all(k in got for k in keys)
        for k in keys:
            assert got[k] == pytest.approx(want[k], rel=1e-6, abs=1e-9), (it, k)
        assert not any(k.startswith("train/") for k in rollout["metrics"][it])  # a rollout-only rank trains nothing


# ---- the gradient exchange INSIDE the one-launch optimizer step (pushed self-validating words; csrc/adamw_clip.hip, XchgPeers) ----
@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_xgmi_one_launch_exchange_in_process_group(world):
    """W "ranks" of one process, one stream each, the REAL protocol: non-owner blocks push their float4s into the owner's inbox row,
    the owner adds the W rows in rank order and pushes the reduced words into every other rank's gather area, the norm partials go
    to every rank's slots -- one launch per rank and step (W x 144 blocks resident together: W <= 3 on one device).  Against the
    launch chain (direct form: the same rank-order sums) on the same inputs: without clipping every buffer is bit-identical; with
    clipping the norm's partials are formed over different blocks (last bit of the coefficient).  Replicas bit-identical always."""
    code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from rlinf_amd.scheduler.xgmi import LocalXgmiGroup
W = {world}
dev = torch.device("cuda:0")
n = 287504
groups = [(0, n // 2, 3e-4), (n // 2, n, 1e-3)]
streams = [torch.cuda.Stream(dev) for _ in range(W)]
runs = []
for one in (False, True):
    g = torch.Generator(device=dev).manual_seed(11)
    grp = LocalXgmiGroup(W, n, dev, algo="direct", timeout_ms=15000, streams=streams, one_launch=one)
    p0 = torch.randn(n, device=dev, generator=g) * 0.1
    params = [p0.clone() for _ in range(W)]
    m = [torch.zeros(n, device=dev) for _ in range(W)]
    v = [torch.zeros(n, device=dev) for _ in range(W)]
    flat = [torch.empty(n, device=dev) for _ in range(W)]
    stats = [torch.zeros(2, device=dev) for _ in range(W)]
    state = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(W)]
    trace = []
    for it in range(8):
        big = it in (3, 6)
        slabs = [torch.randn(1 + (r + it) % 3, n, device=dev, generator=g) * (5.0 if big else 1e-4) for r in range(W)]
        if it == 5:
            slabs[W - 1][0, 123] = float("nan")   # one rank's non-finite gradient: every rank skips the step
        grp.clip_adamw_step(params, slabs, flat, m, v, groups, stats, state)
        torch.cuda.synchronize()
        assert grp.status_ok(), f"one={{one}} it={{it}}: a wait timed out"
        for r in range(1, W):
            assert torch.equal(params[r], params[0]) and torch.equal(m[r], m[0]) and torch.equal(v[r], v[0]), (one, it, r)
            assert torch.equal(flat[r].view(torch.int32), flat[0].view(torch.int32)) and torch.equal(stats[r].view(torch.int32), stats[0].view(torch.int32))
        trace.append((params[0].clone(), m[0].clone(), v[0].clone(), flat[0].clone(), stats[0].clone(), state[0].clone(), big))
    runs.append(trace)
    grp.close()
for it, (a, b) in enumerate(zip(*runs)):
    assert torch.equal(a[5], b[5]), it
    if it == 5:
        assert float(a[4][1]) == float(b[4][1]) == 0.0
        continue
    if it < 3:  # no clipping yet: bit for bit
        for k in range(4):
            assert torch.equal(a[k], b[k]), (it, k)
    else:
        for k in range(4):
            torch.testing.assert_close(b[k], a[k], rtol=2e-6, atol=1e-9)
    assert float(b[4][0]) == __import__("pytest").approx(float(a[4][0]), rel=1e-6)
assert int(runs[1][-1][5][0]) + int(runs[1][-1][5][1]) == 7
print("OK")
"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("image", ["none", "bf16"])
def test_xgmi_one_launch_exchange_self_aliased_equals_the_single_gpu_step(world, image):
    """rlx_xgmi_connect_self: rank 0 of a W-rank job whose peers are its own buffer -- owned blocks play their W - 1 contributors,
    the others the owner that answers them -- so the launch does a rank's pushes and polls on ONE device.  With W a power of two
    the mean of W copies of a gradient is that gradient exactly: parameters, moments, clipped gradient, norm, step counter and the
    weight image must equal the single-GPU one-launch step's, bit for bit, eagerly and over a replayed hipGraph."""
    from rlinf_amd import ops
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    from rlinf_amd.scheduler.xgmi import SelfAliasedXgmi
    if ops.adamw_sync_words(8, "cuda") is None:
        pytest.skip("RLX_ADAMW_ONE_LAUNCH=0")
    runs = []
    for xchg in (False, True):
        torch.manual_seed(4)
        pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16).to("cuda")
        n, lay = pol.n_params, pol.layout
        tiles = pol.tiles() if image != "none" else None
        m, v, flat = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        state, stats = torch.zeros(2, dtype=torch.int32, device="cuda"), torch.zeros(2, device="cuda")
        g = torch.Generator(device="cuda").manual_seed(5)
        grads = torch.empty(10, n, device="cuda")
        ws = torch.empty(ops._lib.load().rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        comm = SelfAliasedXgmi("cuda", world, n) if xchg else None
        step = ops.PreparedAdamw(pol.flat.data, grads, m, v, pol.group_ranges(3e-4, 1e-3), betas=(0.9, 0.999), eps=1e-8,
                                 weight_decay=0.01, max_grad_norm=0.5, grad_scale=1.0 / world if xchg else 1.0, stats=stats,
                                 step_state=state, workspace=ws, tile_layout=lay if tiles is not None else None, tiles=tiles,
                                 xgmi=comm, grad_flat=flat if xchg else None, sync=ops.adamw_sync_words(n, "cuda"))
        trace, stream = [], torch.cuda.current_stream().cuda_stream
        for it in range(5):
            grads.normal_(generator=g).mul_(0.02 if it % 2 else 0.0002)
            if it == 3:
                grads[2, 99] = float("inf")
            step(stream)
            trace.append((stats.clone(), state.clone()))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())  # (the eager steps' trace clones still read `grads` on the default stream)
        with torch.cuda.stream(side):
            master = torch.randn(10, n, device="cuda", generator=g) * 0.01
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(6):
                    grads.copy_(master)  # (the single-GPU step leaves its clipped gradient in slab 0: every step gets fresh slabs)
                    step(side.cuda_stream)
            for _ in range(10):
                graph.replay()
        side.synchronize()
        torch.cuda.synchronize()
        if comm is not None:
            assert comm.status_ok()
        runs.append(dict(p=pol.flat.data.clone(), m=m, v=v, trace=trace, stats=stats.clone(), state=state.clone(),
                         g=(flat if xchg else grads[0]).clone(), tiles=None if tiles is None else tiles.view(torch.int16).clone()))
        if comm is not None:
            comm.close()
    a, b = runs
    assert torch.equal(a["state"], b["state"])
    if world <= 4:  # g + g and ((g + g) + g) + g round to 2 g and 4 g exactly; from the fifth copy on a rank-order sum may be an ulp off
        for k in ("p", "m", "v"):
            assert torch.equal(a[k], b[k]), k
        assert torch.equal(a["stats"].view(torch.int32), b["stats"].view(torch.int32))
        assert torch.equal(a["g"].view(torch.int32), b["g"].view(torch.int32))
        if image != "none":
            assert torch.equal(a["tiles"], b["tiles"])
    else:
        for k, (rt, at) in dict(g=(2e-6, 1e-10), m=(2e-5, 1e-9), v=(2e-5, 1e-12), p=(2e-3, 2e-5)).items():  # (AdamW's m / sqrt(v) amplifies)
            torch.testing.assert_close(b[k], a[k], rtol=rt, atol=at, msg=lambda t, k=k: f"{k}: {t}")
    for it, ((s0, t0), (s1, t1)) in enumerate(zip(a["trace"], b["trace"])):
        assert torch.equal(t0, t1) and float(s0[1]) == float(s1[1]) == (0.0 if it == 3 else 1.0)
        assert it == 3 or float(s0[0]) == pytest.approx(float(s1[0]), rel=0 if world <= 4 else 1e-6)
    assert int(b["state"][0]) + int(b["state"][1]) == 4 + 6 * 10

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


def _launch(mode, tmp_path, *extra, port=29611, timeout=300, one_device=True):
    procs, outs = [], []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0" if one_device else str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
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


def _two_gpus():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


@pytest.mark.gpu
def test_xgmi_allreduce_two_ranks_on_one_gpu(tmp_path):
    """The transport alone (csrc/xgmi_allreduce.hip through scheduler/xgmi.py), two processes sharing the ONE GPU of the test
    box: IPC export / mapping of the fine-grained buffer, the flag hand-shake, both staging slots, slab sums, the scale, eager
    launches and a replayed hipGraph of 20 back-to-back all-reduces -- against torch.distributed's own all-reduce.
    (The whole learner over this transport is exercised when the box has a GPU per rank, below: two ranks time-slicing one
    GPU's hardware queues can starve each other's spin-waits for seconds, which says nothing about xGMI.)"""
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641",
                   RLX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", RLX_XGMI_TIMEOUT_MS="8000", RLX_XGMI_PROBE_LIGHT="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "xgmi_probe.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs, starved = [], False
    for p in procs:
        try:
            logs.append(p.communicate(timeout=100)[0])
        except subprocess.TimeoutExpired:
            starved = True
            for q in procs:
                q.kill()
            logs.append(p.communicate()[0] or "")
    joined = "\n".join(l[-2000:] for l in logs)
    if starved or "did not publish its gradient within the timeout" in joined:
        # Starvation of one process's spin wait by the other's kernels on the SAME GPU is a property of sharing a device, not
        # of the transport: inconclusive here, exercised for real by the one-GPU-per-rank cases below.  Wrong sums still fail.
        assert "result ok after graph: False" not in joined, joined
        pytest.skip("two processes time-slicing one GPU starved each other's flag waits; needs one GPU per rank")
    assert all(p.returncode == 0 for p in procs), joined
    out = logs[0]
    assert "mem_kind 0: OK" in out or "mem_kind 1: OK" in out, out[-2000:]  # a coherent (fine-grained / uncached) kind works
    assert "result ok after graph: True" in out and "result ok after graph: False" not in out, out[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("transport,graph,backend", [
    ("rccl", "0", "gloo"),   # torch.distributed all-reduce (gloo stands in for RCCL: it needs a device per rank)
    ("rccl", "1", "gloo"),   # a gloo all-reduce cannot be stream-captured: the worker keeps the eager (prepared-launch) loop
    # one rank per GPU (auto-enabled when the box has two): RCCL captured in the update graph, and the xGMI transport for real
    pytest.param("rccl", "1", "nccl", marks=pytest.mark.skipif(not _two_gpus(), reason="RCCL needs one GPU per rank")),
    pytest.param("xgmi", "0", "nccl", marks=pytest.mark.skipif(not _two_gpus(), reason="needs one GPU per rank")),
    pytest.param("xgmi", "1", "nccl", marks=pytest.mark.skipif(not _two_gpus(), reason="needs one GPU per rank")),
])
def test_two_ranks_match_the_sharded_oracle(tmp_path, transport, graph, backend, precision="32"):
    """Two processes against the reference arithmetic applied shard by shard: per-rank rollout on its env half, per-shard
    advantage normalisation, per-rank shuffle with seed + rank, per-rank minibatches of global_batch / 2, gradient mean
    (FSDP NO_SHARD, rlinf/hybrid_engines/fsdp/strategy/fsdp.py:480-496), identical clip + AdamW on every rank.  The two
    nccl cases run when the box has two GPUs (one rank per GPU, RCCL / xGMI for real)."""
    import copy

    from oracle import ppo_loop as L
    from oracle import ppo_oracle as O
    outs = [torch.load(o, weights_only=False)
            for o in _launch("gpu", tmp_path, precision, transport, graph, backend, port=29613 + 2 * (graph == "1") + 4 * (transport == "rccl"),
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
    T, B, GB = 12, 64, 192
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, 8, 1)
    opt = O.build_adamw(ora)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    shards = []
    for r in range(2):
        sl = slice(r * B // 2, (r + 1) * B // 2)
        env_r = {k: v[:, sl].contiguous() for k, v in env.items()}
        batch = L.advantages(L.rollout(ora, env_r, eps[:, sl], 0.8, True), 0.8, 0.9, True)
        perm = torch.randperm(T * B // 2, generator=torch.Generator().manual_seed(1234 + r))
        shards.append((batch, O.flatten_and_shuffle(batch, perm)))
    n_mb = (T * B // 2) // (GB // 2)
    for _ in range(2):  # update_epoch
        chunks = [O.chunk_batch(flat, n_mb) for _, flat in shards]
        for i in range(n_mb):
            opt.zero_grad()
            for r in range(2):
                mb = chunks[r][i]
                out = ora.evaluate(mb["forward_inputs"]["states"], mb["forward_inputs"]["action"])
                shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], "action_level", 8,
                                             values=out["values"], prev_values=mb["prev_values"], returns=mb["returns"])
                loss, _ = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
                (loss / 2).backward()  # DDP / FSDP average the per-rank gradients
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
    assert torch.equal(outs[0]["params"], outs[1]["params"])  # both ranks hold the same weights after the update

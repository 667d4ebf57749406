"""One rank of a world_size-W job (W = 2, or 8 for the whole-learner test; launched by tests/test_distributed.py with RANK /
WORLD_SIZE / MASTER_* set).

mode "cpu": host-side data-parallel logic on CPU tensors over gloo (no kernels).
mode "gpu": the full runner on a (shared) GPU, gradients all-reduced over gloo; dumps what the parent compares.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_main(out_path):
    from rlinf_amd.scheduler import (CommMapper, all_reduce_flat_, all_reduce_scalars, build_recv_plan, build_send_plan,
                                     compute_split_num, env_shard, init_distributed)
    from rlinf_amd.scheduler.placement import minibatch_plan
    ctx = init_distributed(device_type="cpu")
    assert ctx.world_size == 2 and dist.get_backend() == "gloo"
    res = {"rank": ctx.rank}
    # env shards are disjoint, contiguous and cover the env axis (env_worker.py:137-140)
    res["shard"] = env_shard(1024, ctx.world_size, 1, ctx.rank)
    res["plan"] = minibatch_plan(1024 * 128 // ctx.world_size, 8192, 8192 // ctx.world_size, ctx.world_size)
    res["split"] = compute_split_num(ctx.world_size, ctx.world_size)
    # C1: the gradient all-reduce of one flat buffer (mean)
    g = torch.full((1000,), float(ctx.rank + 1))
    all_reduce_flat_(g, ctx, average=True)
    res["grad_mean"] = float(g[0])
    # C3 + C4: one SUM call and one MAX call for every metric
    sums = torch.tensor([10.0 * (ctx.rank + 1), 4.0])
    maxs = torch.tensor([-float(ctx.rank), float(ctx.rank) + 5.0])
    s, m = all_reduce_scalars(sums, maxs, ctx)
    res["sums"], res["maxs"] = s.tolist(), m.tolist()
    # routing maps agree across ranks: what rank r sends to d is what d expects from r
    send = build_send_plan("env", "actor", ctx.rank, 2, 2, "traj", 64)
    recv = build_recv_plan("env", "actor", ctx.rank, 2, 2, "traj", 64)
    res["send"] = [(e.peer_rank, e.batch_size, e.offset) for e in send.entries]
    res["recv"] = [(e.peer_rank, e.batch_size, e.offset) for e in recv.entries]
    res["dst_4_to_2"] = CommMapper.get_dst_ranks(64, 4, 2, ctx.rank)
    # weight-patch transport: rank 0 broadcasts a reference-built patch (the committed fixture), rank 1 receives it intact
    from rlinf_amd.scheduler.dist import broadcast_weight_patch
    from rlinf_amd.hybrid_engines.weight_syncer import EmptyWeightPatch, WeightPatch
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "weight_patch.pt"), weights_only=False)[0]["patches"]
    fields = ("version", "ordinals", "nnz_per_tensor", "rows", "cols", "values")
    sent = WeightPatch(**{f: fx[0][f] for f in fields}) if ctx.rank == 0 else None
    got = broadcast_weight_patch(sent, ctx, src=0, device=torch.device("cpu"))
    res["patch_ok"] = all(getattr(got, f).dtype == fx[0][f].dtype and torch.equal(getattr(got, f), fx[0][f]) for f in fields)
    empty = broadcast_weight_patch(EmptyWeightPatch(torch.tensor(7)) if ctx.rank == 0 else None, ctx, device=torch.device("cpu"))
    res["empty_ok"] = isinstance(empty, EmptyWeightPatch) and int(empty.version) == 7
    # bucket transport: layout as one pickled object + ONE broadcast of the flat payload
    from rlinf_amd.hybrid_engines.weight_syncer import WeightBucket
    from rlinf_amd.scheduler.dist import broadcast_weight_bucket
    layout = (("w", torch.bfloat16, (3, 5), 0), ("steps", torch.int64, (), 256), ("flags", torch.bool, (7,), 512))
    sent_b = None
    if ctx.rank == 0:
        sent_b = WeightBucket.from_flat(torch.zeros(768, dtype=torch.uint8), layout,
                                        {"total_buckets": torch.tensor(1, dtype=torch.int32), "syncer_version": torch.tensor(9, dtype=torch.int32)})
        sent_b["w"].copy_(torch.arange(15).reshape(3, 5)), sent_b["steps"].fill_(-4), sent_b["flags"][::2].fill_(True)
    got_b = broadcast_weight_bucket(sent_b, ctx, src=0, device=torch.device("cpu"))
    res["bucket_ok"] = (list(got_b) == ["total_buckets", "syncer_version", "w", "steps", "flags"] and int(got_b["syncer_version"]) == 9
                        and got_b["w"].dtype == torch.bfloat16 and got_b["w"].float().flatten().tolist() == list(range(15))
                        and int(got_b["steps"]) == -4 and got_b["flags"].tolist() == [True, False] * 3 + [True])
    # reasoning learner: sequence-length balancing over the data-parallel group (FSDPActor._dp_load_balance): every rank ends up
    # with the same number of sequences and about the same number of tokens, from partitions all ranks compute alike
    from rlinf_amd.workers.actor.fsdp_actor_worker import FSDPActor
    rcfg = dict(runner=dict(task_type="reasoning"), algorithm=dict(group_size=2, n_minibatches=1),
                actor=dict(micro_batch_size=2, global_batch_size=8, enable_dp_load_balance=True, model=dict(encoder_seq_length=12), optim={}),
                data=dict(rollout_batch_size=4, max_prompt_length=4))
    learner = FSDPActor(rcfg, ctx)
    g = torch.Generator().manual_seed(50 + ctx.rank)
    lens = torch.randint(1, 13, (4,), generator=g) if ctx.rank == 0 else torch.randint(8, 13, (4,), generator=g)  # rank 1 drew long ones
    batch = {"input_ids": torch.arange(4).unsqueeze(1).expand(4, 12).clone() + 100 * ctx.rank,
             "attention_mask": torch.arange(12).unsqueeze(0) < lens.unsqueeze(1), "rewards": torch.arange(4.0) + 10 * ctx.rank}
    bal = learner._dp_load_balance(batch)
    res["balance_lens_in"] = lens.tolist()
    res["balance_ids"] = bal["input_ids"][:, 0].tolist()
    res["balance_tokens"] = int(bal["attention_mask"].sum())
    res["balance_rewards"] = bal["rewards"].tolist()
    json.dump(res, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def gpu_main(out_path, precision, transport="xgmi", graph="0", backend="gloo"):
    """transport: "xgmi" (hand-written peer-read all-reduce, IPC-mapped buffers; works with both ranks on ONE GPU too) or
    "rccl" (torch.distributed all-reduce: gloo when the ranks share a GPU -- RCCL needs a device per rank -- nccl otherwise);
    graph "1": actor.enable_hip_graph -- with xgmi the update phase is a pure kernel chain and is captured, with gloo the
    capture fails by construction and the worker must fall back to the eager loop on both ranks."""
    import copy

    from oracle import ppo_oracle as O
    from test_end_to_end import _build, make_cfg

    from oracle import ppo_loop as L
    os.environ["RLX_DIST_BACKEND"] = backend
    os.environ["RLX_GRAD_ALLREDUCE"] = transport
    T, B, GB = 12, 64, 192
    W = int(os.environ["WORLD_SIZE"])
    cfg = make_cfg(total_envs=B, steps=T, global_batch=GB, micro_batch=GB // W, hip_graph=graph == "1")
    cfg.actor.model.precision = precision
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    sd = copy.deepcopy(O.OracleMLPPolicy(42, 8, 1).state_dict())
    runner = _build(cfg, env, sd)
    ctx = runner.actor.worker.ctx
    assert ctx.world_size == W
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    lo, hi = ctx.rank * (B // W), (ctx.rank + 1) * (B // W)
    metrics = runner.run_step(eps[:, lo:hi].cuda())
    rb = runner.actor.worker.rollout_batch
    first = dict(params=runner.actor.worker.model.flat.detach().cpu().clone(), advantages=rb["advantages"].cpu().clone(),
                 returns=rb["returns"].cpu().clone(), actions=rb["forward_inputs"]["action"].cpu().clone(),
                 rewards=rb["rewards"].cpu().clone())
    iters = 1
    if graph == "1":  # iteration 0 ran eagerly and captured; two more are replays (or the eager fallback)
        for it in (1, 2):
            eps_i = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100 + it))
            metrics = runner.run_step(eps_i[:, lo:hi].cuda())
            iters += 1
    sync_ok, version, patch_nnz = True, None, []
    if W > 1:
        # actor -> rollout weight sync over the sparse patch format: rank 0 plays the learner, rank 1 a rollout replica that
        # holds bf16 copies; handshake and patch both travel over torch.distributed (gloo here, RCCL with a GPU per rank)
        from rlinf_amd.hybrid_engines.weight_syncer import PatchWeightSyncer
        from rlinf_amd.scheduler.dist import broadcast_weight_patch
        gsync = torch.Generator().manual_seed(5)
        master = {"w": torch.randn(96, 130, generator=gsync).cuda(), "b": torch.randn(130, generator=gsync).cuda()}
        replica = {k: v.to(torch.bfloat16) for k, v in master.items()}
        syncer = PatchWeightSyncer()
        box = [None]
        if ctx.rank == 1:
            syncer.init_receiver(replica, None, lambda meta: box.__setitem__(0, meta))
        dist.broadcast_object_list(box, src=1)
        if ctx.rank == 0:
            syncer.init_sender(master, ["w", "b"], None, lambda: box[0])
            master["w"][::3, ::7] += 0.5
            master["b"][5] = -2.0
        patch = broadcast_weight_patch(syncer.create_patch(master, 3) if ctx.rank == 0 else None, ctx, src=0)
        version = None
        if ctx.rank == 1:
            version = syncer.apply(replica, lambda: patch)
        dist.broadcast(master["w"], src=0), dist.broadcast(master["b"], src=0)
        sync_ok = ctx.rank != 1 or all(torch.equal(replica[k], master[k].to(torch.bfloat16)) for k in master)  # (rank 1 is the replica)
        patch_nnz = patch.nnz_per_tensor.tolist()
    w = runner.actor.worker
    torch.save(dict(rank=ctx.rank, metrics=metrics, params=first["params"], advantages=first["advantages"], returns=first["returns"],
                    actions=first["actions"], rewards=first["rewards"], sync_ok=bool(sync_ok), sync_version=version,
                    patch_nnz=patch_nnz, backend=w.grad_allreduce_backend, iters=iters, dist_backend=dist.get_backend(),
                    final_params=w.model.flat.detach().cpu(), graph_live=w._graph is not None,
                    graph_enabled=bool(w.enable_hip_graph),
                    xgmi=(None if w._xgmi is None else dict(algo=w._xgmi.algo, wait_mode=w._xgmi.wait_mode, shared_device=w._xgmi.shared_device))),
               out_path)
    dist.barrier()
    dist.destroy_process_group()


def gpu_async_main(out_path, precision, transport="xgmi", graph="0", backend="gloo"):
    """The ASYNC (decoupled PPO) learner at world_size W on a shared GPU: iteration 0 through the runner (every sample one version
    behind: alpha = 0), then the same buffer again with mixed behaviour versions under a behaviour-weight threshold that masks
    samples -- the fused step's actor gradients leave in sum form and every rank scales ITS slabs by ITS micro-batch's count
    before the exchange.  Dumps the parameters after each phase."""
    import copy

    from oracle import ppo_oracle as O
    from test_end_to_end import make_cfg

    from oracle import ppo_loop as L
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.scheduler import init_distributed
    from rlinf_amd.workers.actor.async_ppo_fsdp_worker import AsyncPPOEmbodiedFSDPActor
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker
    os.environ["RLX_DIST_BACKEND"] = backend
    os.environ["RLX_GRAD_ALLREDUCE"] = transport
    T, B, GB = 12, 64, 192
    W = int(os.environ["WORLD_SIZE"])
    cfg = make_cfg(total_envs=B, steps=T, global_batch=GB, micro_batch=GB // W // 2, hip_graph=graph == "1")  # two micro-batches per step
    cfg.actor.model.precision = precision
    cfg.algorithm.loss_type = "decoupled_actor_critic"
    cfg.algorithm.behave_weight_threshold = 1.01
    cfg = validate_cfg(cfg)
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    sd = copy.deepcopy(O.OracleMLPPolicy(42, 8, 1).state_dict())
    ctx = init_distributed()
    assert ctx.world_size == W
    actor = AsyncPPOEmbodiedFSDPActor.create_group(cfg, ctx).launch(None, name="ActorGroup")
    runner = EmbodiedRunner(cfg, actor, MultiStepRolloutWorker.create_group(cfg, ctx).launch(None, name="RolloutGroup"),
                            EnvWorker.create_group(cfg, ctx).launch(None, name="EnvGroup"))
    runner.init_workers(env_tensors=env)
    w = actor.worker
    w.model.load_reference_state_dict(sd)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    lo, hi = ctx.rank * (B // W), (ctx.rank + 1) * (B // W)
    runner.run_step(eps[:, lo:hi].cuda())
    first = w.model.flat.detach().cpu().clone()
    w.set_global_step(2)  # current_version 3: alpha = 2/3 for version 0, 1/2 for version 1
    w.rollout_batch["versions"][T // 2:] += 1.0
    metrics = w.run_training()
    torch.save(dict(rank=ctx.rank, first_params=first, final_params=w.model.flat.detach().cpu(), metrics=metrics,
                    backend=w.grad_allreduce_backend, fused="aplan_key" in w._ws, graph="agraph" in w._ws), out_path)
    dist.barrier()
    dist.destroy_process_group()


# ---- split placement (actor ranks != rollout ranks) and the weight syncers in the worker path -----------------------------------
def _split_cfg(syncer: str, placement: dict, *, T=12, B=32, GB=96, init_sync=False, precision="32"):
    from test_end_to_end import make_cfg
    cfg = make_cfg(total_envs=B, steps=T, global_batch=GB, update_epoch=2)
    cfg.actor.model.precision = precision
    cfg.cluster = {"num_nodes": 1, "component_placement": placement}
    if syncer == "bucket":
        cfg.weight_syncer = {"type": "bucket", "bucket": {"bucket_size": 128 * 1024, "bucket_dtype": None, "is_agent": False,
                                                           "load_instant": True}}
    elif syncer in ("patch", "patch_cpu"):  # patch_cpu: the reference's shipped default (weight_syncer/patch_syncer.yaml: host snapshot)
        cfg.weight_syncer = {"type": "patch", "patch": {"snapshot_device": "cpu" if syncer == "patch_cpu" else "cuda",
                                                         "delta_encoding": True, "compression": "none",
                                                         "init_sync": {"enabled": init_sync, "prefixes": None,
                                                                       "bucket_size": 64 * 1024}}}
    return cfg


def _launch_runner(cfg, env_tensors, state_dict):
    """The reference entry point's sequence (examples/embodiment/train_embodied_agent.py): Cluster, HybridComponentPlacement, the
    three groups launched with their placement strategies, the runner."""
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.scheduler import Cluster, init_distributed
    from rlinf_amd.utils.placement import HybridComponentPlacement
    from rlinf_amd.workers.actor import EmbodiedFSDPActor
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker
    cfg = validate_cfg(cfg)
    cluster = Cluster(cluster_cfg=cfg.cluster, ctx=init_distributed())
    placement = HybridComponentPlacement(cfg, cluster)
    actor = EmbodiedFSDPActor.create_group(cfg).launch(cluster, name="ActorGroup", placement_strategy=placement.get_strategy("actor"))
    rollout = MultiStepRolloutWorker.create_group(cfg).launch(cluster, name="RolloutGroup",
                                                              placement_strategy=placement.get_strategy("rollout"))
    env = EnvWorker.create_group(cfg).launch(cluster, name="EnvGroup", placement_strategy=placement.get_strategy("env"))
    runner = EmbodiedRunner(cfg, actor, rollout, env)
    runner.init_workers(share_weights=False, env_tensors=env_tensors)
    if actor.worker is not None:
        actor.worker.model.load_reference_state_dict(state_dict)
    return runner, placement


def split_gpu_main(out_path, syncer, init_sync="0", backend="gloo", layout="split"):
    """world 2, ``layout`` "split": rank 0 = learner, rank 1 = env + rollout (component_placement actor: 0 / env,rollout: 1), the
    weights through the configured syncer over the weight-sync group, the trajectory buffer over the 1 : 1 route; or world 1,
    ``layout`` "collocated": the same job in one process with the rollout keeping its own copy (in-process link).  Three
    iterations, then one more weight sync; dumps what the parent compares."""
    import copy

    from oracle import ppo_loop as L
    from oracle import ppo_oracle as O
    os.environ["RLX_DIST_BACKEND"] = backend
    W = int(os.environ["WORLD_SIZE"])
    T, B = 12, 32
    placement = {"actor": "0", "env,rollout": "1"} if layout == "split" else {"env,rollout,actor": "all"}
    cfg = _split_cfg(syncer, placement, T=T, B=B, init_sync=init_sync == "1")
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    sd = copy.deepcopy(O.OracleMLPPolicy(42, 8, 1).state_dict())
    runner, pl = _launch_runner(cfg, env, sd)
    assert pl.split == (layout == "split") and (W == 2) == pl.split
    has_actor, has_rollout = runner._has["actor"], runner._has["rollout"]
    metrics, versions, synced = [], [], []
    for it in range(3):
        eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100 + it)).cuda()
        metrics.append(runner.run_step(eps))
        if has_rollout:
            versions.append(int(runner.rollout.worker.version))
    runner.actor.set_global_step(runner.global_step).wait()
    runner.update_rollout_weights()  # one more sync: the rollout's copy must now equal the learner's final weights byte for byte
    torch.cuda.synchronize()
    out = dict(rank=int(os.environ["RANK"]), layout=layout, has_actor=has_actor, has_rollout=has_rollout, metrics=metrics,
               versions=versions, dist_backend=dist.get_backend() if dist.is_initialized() else None)
    if has_actor:
        a = runner.actor.worker
        out.update(actor_params=a.model.flat.detach().cpu().clone(), actor_world=a._world_size, actor_rank=a._rank,
                   grad_backend=a.grad_allreduce_backend, sender_initialized=a.weight_syncer.sender_initialized())
    if has_rollout:
        r = runner.rollout.worker
        out.update(rollout_params=r.hf_model.flat.detach().cpu().clone(), final_version=int(r.version),
                   shares=bool(r._shares_actor_weights), receiver_initialized=r.weight_syncer.receiver_initialized(),
                   env_world=runner.env.worker._world_size, num_envs=runner.env.worker.num_envs)
    torch.save(out, out_path)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def split_cpu_main(out_path):
    """The split placement's host side on CPU over gloo (world 2): rank sets and groups from component_placement, the weight link's
    four closures carrying reference-built payloads, the trajectory route."""
    from rlinf_amd.config import DictConfig
    from rlinf_amd.hybrid_engines.weight_syncer import EmptyWeightPatch, WeightBucket, WeightPatch
    from rlinf_amd.scheduler import Cluster, init_distributed
    from rlinf_amd.scheduler.dist import recv_tensors, send_tensors
    from rlinf_amd.utils.placement import HybridComponentPlacement
    from rlinf_amd.workers.weight_link import GroupLink
    ctx = init_distributed(device_type="cpu")
    cfg = DictConfig(dict(cluster=dict(num_nodes=1, component_placement={"actor": 0, "env,rollout": "1"})))
    pl = HybridComponentPlacement(cfg, Cluster(cluster_cfg=cfg.cluster, ctx=ctx))
    res = {"rank": ctx.rank, "split": pl.split, "worlds": [pl.get_world_size(c) for c in ("actor", "rollout", "env")],
           "present": [pl.get_strategy(c).present for c in ("actor", "rollout", "env")]}
    me = pl.get_strategy("actor" if ctx.rank == 0 else "rollout").ctx
    res["component_ctx"] = [me.rank, me.world_size, me.global_ranks]
    link = GroupLink(pl, torch.device("cpu"))
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "weight_patch.pt"), weights_only=False)[0]["patches"]
    fields = ("version", "ordinals", "nnz_per_tensor", "rows", "cols", "values")
    layout = (("w", torch.bfloat16, (3, 5), 0), ("steps", torch.int64, (), 256))
    if ctx.rank == 0:  # the learner: metadata in, then a bucket, a patch, an empty patch out
        meta = link.actor_recv()
        res["meta"] = meta
        b = WeightBucket.from_flat(torch.zeros(512, dtype=torch.uint8), layout,
                                   {"total_buckets": torch.tensor(1, dtype=torch.int32), "syncer_version": torch.tensor(4, dtype=torch.int32)})
        b["w"].copy_(torch.arange(15).reshape(3, 5)), b["steps"].fill_(9)
        link.actor_send(b)
        link.actor_send(WeightPatch(**{f: fx[0][f] for f in fields}))
        link.actor_send(EmptyWeightPatch(torch.tensor(12)))
        traj = [torch.empty(4, 6), torch.empty(5, 6, 1, dtype=torch.bool)]
        recv_tensors(traj, pl.peer_of("actor", "env"))
        res["traj_ok"] = bool(torch.equal(traj[0], torch.arange(24.).reshape(4, 6)) and traj[1][::2].all() and not traj[1][1::2].any())
    else:  # the rollout rank
        link.rollout_send({"ordered_keys": ["a", "b"], "receiver_dtypes": {"a": torch.bfloat16}})
        b = link.rollout_recv()
        res["bucket_ok"] = bool(isinstance(b, WeightBucket) and int(b["syncer_version"]) == 4 and int(b["steps"]) == 9
                                and b["w"].float().flatten().tolist() == list(range(15)))
        p = link.rollout_recv()
        res["patch_ok"] = bool(isinstance(p, WeightPatch) and all(torch.equal(getattr(p, f), fx[0][f]) for f in fields))
        e = link.rollout_recv()
        res["empty_ok"] = bool(isinstance(e, EmptyWeightPatch) and int(e.version) == 12)
        done = torch.zeros(5, 6, 1, dtype=torch.bool)
        done[::2] = True
        send_tensors([torch.arange(24.).reshape(4, 6), done], pl.peer_of("env", "actor"))
    res["meta"] = {k: (v if not isinstance(v, dict) else {a: str(b) for a, b in v.items()}) for k, v in res.get("meta", {}).items()}
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[1] == "cpu":
        cpu_main(sys.argv[2])
    elif sys.argv[1] == "split_cpu":
        split_cpu_main(sys.argv[2])
    elif sys.argv[1] == "split_gpu":
        split_gpu_main(*sys.argv[2:])
    elif sys.argv[1] == "gpu_async":
        gpu_async_main(*sys.argv[2:])
    else:
        gpu_main(*sys.argv[2:])

"""Host-side logic of the token tier (no GPU): the mirrors of rlinf/utils/utils.py and rlinf/algorithms/utils.py
against the CPU oracle, the registry's reasoning dispatch, the learner-step config mapping, loud failure without HIP."""

import pytest
import torch

from oracle import token_oracle as TO
from oracle.make_golden import token_batch
from rlinf_amd import _lib
from rlinf_amd.algorithms import registry
from rlinf_amd.algorithms import utils as AU
from rlinf_amd.utils import utils as UU
from rlinf_amd.workers.actor.fsdp_actor_worker import TokenLearnerStep


@pytest.mark.parametrize("agg", ["token-mean", "seq-mean-token-sum", "seq-mean-token-mean"])
def test_agg_funcs_match_oracle_and_are_tagged(agg):
    b = token_batch(201, 6, 9, 5)
    v = b["advantages"]
    fn = UU.get_loss_agg_func(agg)
    assert fn.rlx_agg == agg
    assert torch.equal(fn(v, b["loss_mask"]), TO.get_loss_agg_func(agg)(v, b["loss_mask"]))
    with pytest.raises(ValueError, match="Unsupported loss aggregation"):
        UU.get_loss_agg_func("mean")


def test_masked_mean_all_false_is_the_plain_sum():
    v = torch.randn(3, 4)
    m = torch.zeros(3, 4, dtype=torch.bool)
    assert float(UU.masked_mean(v, m)) == 0.0
    assert torch.equal(UU.masked_mean(v, None), v.mean())
    assert torch.equal(UU.masked_sum(v, ~m), v.sum())


@pytest.mark.parametrize("kind", ["kl", "k1", "abs", "mse", "k2", "low_var_kl", "k3"])
def test_kl_penalty_mirror(kind):
    b = token_batch(202, 4, 16, 5)
    a, c = b["ref_logprobs"] * 9, b["old_logprobs"]
    assert torch.equal(AU.kl_penalty(a, c, kind), TO.kl_penalty(a, c, kind))
    with pytest.raises(NotImplementedError):
        AU.kl_penalty(a, c, "full")


def test_reasoning_shaping_mirror():
    b = token_batch(203, 8, 10, 5)
    values = torch.randn(8, 10)
    for adv_type in ("gae", "grpo"):
        got = AU.preprocess_reasoning_advantages_inputs(rewards=b["rewards"], loss_mask=b["loss_mask"], values=values,
                                                        adv_type=adv_type, group_size=4)
        want = TO.preprocess_reasoning(b["rewards"], b["loss_mask"], adv_type, values=values, group_size=4)
        for k in ("rewards", "loss_mask", "dones", "values"):
            assert torch.equal(got[k], want[k]), (adv_type, k)
        assert got["adv_type"] == adv_type
    adv, ret = AU.postprocess_reasoning_advantages_outputs(torch.arange(6.).reshape(3, 2), torch.ones(3, 2))
    assert adv.shape == (2, 3) and adv.is_contiguous() and ret.is_contiguous()
    with pytest.raises(AssertionError, match="Unsupported adv_type"):
        AU.preprocess_reasoning_advantages_inputs(rewards=b["rewards"], loss_mask=b["loss_mask"], adv_type="nope")
    with pytest.raises(AssertionError, match="Unsupported reward shape"):
        AU.preprocess_reasoning_advantages_inputs(rewards=b["rewards"][None], loss_mask=b["loss_mask"], adv_type="raw")


def test_user_registered_reasoning_advantage_gets_reference_shapes():
    seen = {}
    b = token_batch(204, 8, 10, 5)
    registry.register_advantage("probe_reasoning")(lambda **kw: (None, None))
    with pytest.raises(AssertionError, match="Unsupported adv_type"):  # the reference's assert (utils.py:222)
        registry.calculate_adv_and_returns(task_type="reasoning", adv_type="probe_reasoning", rewards=b["rewards"],
                                           loss_mask=b["loss_mask"])
    registry.ADV_REGISTRY.pop("probe_reasoning")

    @registry.register_advantage("raw")
    def raw(rewards, loss_mask, dones, **kw):
        seen.update(rewards=rewards.shape, loss_mask=loss_mask.shape, dones=dones.shape)
        return torch.zeros(loss_mask.shape), None

    adv, ret = registry.calculate_adv_and_returns(task_type="reasoning", adv_type="raw", rewards=b["rewards"],
                                                  loss_mask=b["loss_mask"])
    registry.ADV_REGISTRY.pop("raw")
    assert seen == dict(rewards=(8,), loss_mask=(10, 8), dones=(11, 8))
    assert adv.shape == (8, 10) and ret is None


def test_learner_step_from_cfg():
    cfg = {"algorithm": {"ratio_clip_eps": 0.2, "clip_ratio_high": 0.28, "loss_agg_func": "seq-mean-token-mean",
                         "calculate_entropy": True, "entropy_bonus": 0.01, "kl_beta": 0.05, "kl_penalty_type": "k2",
                         "sampling_params": {"temperature": 0.6}},
           "actor": {"model": {"encoder_seq_length": 3072}}, "data": {"max_prompt_length": 1024},
           "runner": {"task_type": "reasoning"}}
    s = TokenLearnerStep.from_cfg(cfg)
    assert (s.response_len, s.clip_ratio_low, s.clip_ratio_high, s.clip_ratio_c) == (2048, 0.2, 0.28, 3.0)
    assert (s.loss_agg, s.temperature, s.kl_penalty_type, s.calculate_entropy) == ("seq-mean-token-mean", 0.6, "k2", True)


def test_token_param_validation():
    from rlinf_amd import token_ops
    with pytest.raises(ValueError, match="Unsupported loss aggregation"):
        token_ops.make_token_loss_params(loss_agg="mean", clip_ratio_low=0.2, clip_ratio_high=0.2)
    with pytest.raises(AssertionError, match="clip_ratio_c"):
        token_ops.make_token_loss_params(loss_agg="token-mean", clip_ratio_low=0.2, clip_ratio_high=0.2, clip_ratio_c=1.0)
    with pytest.raises(NotImplementedError):
        token_ops.make_token_loss_params(loss_agg="token-mean", clip_ratio_low=0.2, clip_ratio_high=0.2,
                                         kl_penalty_type="full", kl_beta=0.1)
    p = token_ops.make_token_loss_params(loss_agg="token-mean", clip_ratio_low=0.2, clip_ratio_high=0.28, clip_ratio_c=3.0,
                                         kl_penalty_type="low_var_kl", kl_beta=0.1, use_entropy=True, entropy_bonus=0.01)
    assert (p.loss_agg, p.kl_type, p.ppo.use_dual_clip, p.use_entropy) == (0, 4, 1, 1)
    assert abs(p.ppo.ratio_hi - 1.28) < 1e-6


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_token_ops_fail_loudly_without_hip():
    b = token_batch(205, 4, 6, 33)
    with pytest.raises(_lib.RlxError, match="no CPU fallback|HIP"):
        UU.compute_logprobs_from_logits(b["logits"], b["labels"])
    with pytest.raises(_lib.RlxError):
        registry.calculate_adv_and_returns(task_type="reasoning", adv_type="grpo", rewards=b["rewards"],
                                           loss_mask=b["loss_mask"], group_size=4)
    with pytest.raises(_lib.RlxError):
        registry.policy_loss(task_type="reasoning", loss_type="actor", loss_agg_func=UU.masked_mean,
                             logprobs=b["old_logprobs"], old_logprobs=b["old_logprobs"], advantages=b["advantages"],
                             clip_ratio_low=0.2, clip_ratio_high=0.2, loss_mask=b["loss_mask"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("temperature,top_k", [(1.0, -1), (0.7, 50), (1.6, 1), (1.0, 300)])
def test_categorical_oracle_is_torch_multinomial(dtype, temperature, top_k):
    """The K2 oracle against the very calls the reference makes (openvla_oft_action_model.py:379-392):
    TopKLogitsWarper, softmax, torch.multinomial -- with the generator rewound, the injected Exp(1) noise reproduces
    multinomial's choice exactly."""
    from transformers import TopKLogitsWarper

    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(16, 56, 256, generator=g) * 2).to(dtype)
    x = logits / temperature
    k = min(top_k, 256)
    if k > 0:
        x = TopKLogitsWarper(k)(None, x)
    probs = torch.softmax(x, dim=-1)
    flat = probs.reshape(-1, 256)
    ids = torch.multinomial(flat, num_samples=1, replacement=True, generator=torch.Generator().manual_seed(7)).view(16, 56)
    q = torch.empty_like(flat).exponential_(1, generator=torch.Generator().manual_seed(7)).view(16, 56, 256)
    tok, lp, processed, _ = TO.categorical_sample(logits, q, temperature, top_k)
    assert torch.equal(tok, ids) and torch.equal(processed, x)
    assert torch.equal(lp, TO.logprobs_from_logits(x, ids))
    tok0, lp0, _, act = TO.categorical_sample(logits, None, bin_centers=torch.arange(255.))
    assert torch.equal(tok0, logits.argmax(-1)) and torch.equal(act, (255 - tok0).clamp(0, 254).float())


def test_async_entry_point_selects_the_learner_by_loss_type():
    """examples/embodiment/train_async.py: config loading / validation and the reference's loss_type -> learner rule
    (examples/embodiment/train_async.py:49-80); also with the reference's own YAML where its tree exists."""
    import importlib.util
    import os

    from rlinf_amd.config import load_config, validate_cfg
    from rlinf_amd.workers.actor import AsyncPPOEmbodiedFSDPActor, EmbodiedFSDPActor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_async", os.path.join(root, "examples", "embodiment", "train_async.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg_dir = os.path.join(root, "examples", "embodiment", "config")
    cfg = validate_cfg(load_config(os.path.join(cfg_dir, "maniskill_async_ppo_mlp.yaml"), overrides=["runner.max_epochs=2"],
                                   search_paths=[cfg_dir]))
    assert cfg.algorithm.loss_type == "decoupled_actor_critic" and cfg.runner.max_epochs == 2
    cls = mod.select_actor_cls(cfg)
    assert cls is AsyncPPOEmbodiedFSDPActor and issubclass(cls, EmbodiedFSDPActor)
    n = cfg.env.train.total_num_envs * cfg.env.train.max_steps_per_rollout_epoch
    assert n % cfg.actor.global_batch_size == 0 and cfg.actor.global_batch_size % cfg.actor.micro_batch_size == 0
    cfg.algorithm.loss_type = "embodied_sac"
    with pytest.raises(NotImplementedError, match="only the decoupled-PPO learner"):
        mod.select_actor_cls(cfg)
    cfg.algorithm.loss_type = "actor_critic"
    with pytest.raises(ValueError, match="Unsupported loss type actor_critic for async embodied runner"):
        mod.select_actor_cls(cfg)
    ref_dir = "/root/reference/examples/embodiment/config"
    if os.path.exists(os.path.join(ref_dir, "maniskill_async_ppo_mlp.yaml")):
        ref_cfg = validate_cfg(load_config(os.path.join(ref_dir, "maniskill_async_ppo_mlp.yaml"), search_paths=[ref_dir]))
        assert ref_cfg.algorithm.loss_type == "decoupled_actor_critic"
        assert mod.select_actor_cls(ref_cfg) is AsyncPPOEmbodiedFSDPActor


@pytest.mark.parametrize("K", [7, 16, 37, 100, 250, 256, 1000, 1024])
def test_cpu_softmax_restatement(K):
    """oracle/softmax_replay.py -- the arithmetic csrc/token_ops.hip's categorical sampler replays on the GPU -- IS torch's CPU
    softmax: bit for bit, f32 and bf16 rows, whole chunks and chunk tails, rows masked to -inf by top-k.  (The lane count follows
    this host's torch build: 16 on AVX-512, 8 on AVX2; any other build has no vector exp and is skipped.)"""
    import numpy as np

    from oracle import softmax_replay as SR
    cap = torch.backends.cpu.get_cpu_capability().upper()
    if cap not in ("AVX2", "AVX512"):
        pytest.skip(f"torch CPU capability {cap}: no Sleef vector kernels to restate")
    W = SR.host_lanes()
    g = torch.Generator().manual_seed(K)
    X = torch.randn(64, K, generator=g) * 4
    X[::5] = X[::5].masked_fill(torch.rand(X[::5].shape, generator=g) < 0.5, float("-inf"))
    X[:, 0] = X[:, 0].clamp_min(-50.0)  # (a row needs one finite entry)
    for dtype, reduced in ((torch.float32, False), (torch.bfloat16, True)):
        Xd = X.to(dtype)
        want = torch.softmax(Xd, -1).float().numpy()
        for i in range(X.shape[0]):
            got = SR.softmax_row(Xd[i].float().numpy(), W, reduced)
            assert np.array_equal(got.view(np.uint32), want[i].view(np.uint32)), (dtype, i)

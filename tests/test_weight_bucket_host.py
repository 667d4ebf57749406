"""Bucketed weight sync without a GPU: the oracle against the committed reference buckets (bit for bit) and against the
real reference class where the reference tree exists; the mirror's host logic (bucket plan, key rewriting, transport
dtypes, the flat-bucket layout, the factory, init-sync selection); the receiver semantics of load_state_dict."""

import asyncio
import os
import sys

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import bucket_oracle as BO
from oracle.make_golden import BUCKET_CASES, bucket_state
from rlinf_amd.config import DictConfig
from rlinf_amd.hybrid_engines.weight_syncer import (BucketWeightSyncer, PatchWeightSyncer, WeightBucket, WeightSyncer,
                                                    plan_buckets)
from rlinf_amd.hybrid_engines.weight_syncer.bucket_syncer import normalize_dtype


def same_bytes(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Equal dtype, shape and bytes; a NaN may carry any payload (the f32 -> bf16 NaN pattern is the backend's choice)."""
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.is_floating_point():
        return bool(((a == b) | ((a != a) & (b != b))).all()) and torch.equal(torch.signbit(a) | (a != a), torch.signbit(b) | (b != b))
    return torch.equal(a, b)


def fixture():
    return torch.load(os.path.join(GOLDEN_DIR, "weight_bucket.pt"), weights_only=False)


def test_fixture_matches_its_generator_parameters():
    assert [c["params"] for c in fixture()] == [dict(c) for c in BUCKET_CASES]


def test_oracle_reproduces_reference_buckets():
    for case in fixture():
        p = case["params"]
        state, names = bucket_state(p["seed"])
        got = BO.make_buckets(state, names, 7, p["bucket_size"], normalize_dtype(p["bucket_dtype"]), p["is_agent"])
        assert len(got) == len(case["buckets"])
        for g, w in zip(got, case["buckets"]):
            assert list(g) == list(w), p  # same keys in the same order, metadata first
            for k in w:
                assert same_bytes(g[k], w[k]), (p, k)
        # quirks the fixture pins: a trailing zero-byte tensor is dropped, _extra_state never travels, agent keys lose
        # "language_model." only when the state has a visual tower
        sent = [k for b in case["buckets"] for k in b]
        assert "decoder._extra_state" not in sent and "not_in_state" not in sent
        assert ("model.layers.0.w" in sent) == p["is_agent"] and ("model.language_model.layers.0.w" in sent) != p["is_agent"]
    assert "empty" not in [k for b in fixture()[0]["buckets"] for k in b]


def test_oracle_receiver_is_load_state_dict():
    case = fixture()[1]  # bf16 transport
    state, _ = bucket_state(1)
    target = {k: torch.zeros_like(v) for k, v in state.items() if k != "steps"}  # an unknown received key is ignored
    target["extra_param"] = torch.ones(3)
    assert BO.apply_buckets(target, case["buckets"]) == 7
    for k, v in state.items():
        if k in ("steps", "frozen", "decoder._extra_state", "empty"):
            continue
        want = v.to(torch.bfloat16).to(v.dtype) if v.is_floating_point() else v
        assert same_bytes(target[k], want), k
    assert torch.equal(target["extra_param"], torch.ones(3)) and not target["frozen"].any()
    target["conv"] = torch.zeros(5, 5)
    with pytest.raises(RuntimeError, match="size mismatch"):
        BO.apply_buckets(target, case["buckets"])


@pytest.mark.reference
@pytest.mark.parametrize("bucket_size,bucket_dtype,is_agent", [(1, "bf16", True), (30_000, None, False), (4096, "fp16", False)])
def test_oracle_vs_reference_syncer(bucket_size, bucket_dtype, is_agent):
    from oracle import reference_loader
    if not reference_loader.available():
        pytest.skip("reference tree not present")
    reference_loader.load_weight_syncer()
    m = sys.modules["rlinf.hybrid_engines.weight_syncer.bucket_syncer"]
    state, names = bucket_state(11)
    names = list(reversed(names))
    syncer = m.BucketWeightSyncer(bucket_size, bucket_dtype, "cpu", is_agent=is_agent)
    sent = []

    async def send(b):
        sent.append(dict(b))

    async def run():
        await syncer.init_sender(state, names, send)
        await syncer.sync(state, send, torch.tensor(5))
        model = torch.nn.Module()
        model.w = torch.nn.Parameter(torch.zeros(37, 300))
        model.register_buffer("steps", torch.zeros(4, 10, dtype=torch.int64))
        queue = [{"w" if k == "backbone.weight" else k: v for k, v in b.items()} for b in sent]
        it = iter(queue)

        async def recv():
            return dict(next(it))

        return model, await syncer.apply(model, recv)

    model, version = asyncio.run(run())
    got = BO.make_buckets(state, names, 5, bucket_size, normalize_dtype(bucket_dtype), is_agent)
    assert len(got) == len(sent)
    for g, w in zip(got, sent):
        assert list(g) == list(w)
        assert all(same_bytes(g[k], w[k]) for k in w)
    target = {"w": torch.zeros(37, 300), "steps": torch.zeros(4, 10, dtype=torch.int64)}
    renamed = [{"w" if k == "backbone.weight" else k: v for k, v in b.items()} for b in got]
    assert BO.apply_buckets(target, renamed) == version == 5
    assert same_bytes(target["w"], model.w.detach()) and torch.equal(target["steps"], model.steps)


# ---- the mirror's host logic ----------------------------------------------------------------------------------------
def test_plan_matches_reference_buckets():
    for case in fixture():
        p = case["params"]
        state, names = bucket_state(p["seed"])
        syncer = BucketWeightSyncer(p["bucket_size"], p["bucket_dtype"], "cuda", is_agent=p["is_agent"])
        syncer.init_sender(state, names)
        has_visual = any("visual." in k for k in names if k in state)
        items = [(syncer._bucket_key(k, has_visual), state[k]) for k in names if k in state and syncer._bucket_key(k, has_visual)]
        plan = plan_buckets(items, p["bucket_size"], lambda _, dt: syncer._transport_dtype(dt))
        assert len(plan) == len(case["buckets"])
        for bucket_items, want in zip(plan, case["buckets"]):
            payload = {k: v for k, v in want.items() if k not in ("total_buckets", "syncer_version")}
            assert [k for k, _, _ in bucket_items] == list(payload)
            assert [dt for _, _, dt in bucket_items] == [v.dtype for v in payload.values()]


def test_plan_errors():
    with pytest.raises(ValueError, match="No parameters to sync"):
        plan_buckets([], 10)
    with pytest.raises(ValueError, match="No parameters to sync"):
        plan_buckets([("empty", torch.zeros(0))], 10)  # zero bytes never close a bucket (bucket_syncer.py:82-86)
    with pytest.raises(ValueError, match="conflicts with metadata key"):
        plan_buckets([("total_buckets", torch.zeros(3))], 10)
    with pytest.raises(ValueError, match="must not be empty"):
        BucketWeightSyncer(1, None, "cuda").init_sender({}, [])
    with pytest.raises(TypeError, match="Unsupported dtype"):
        BucketWeightSyncer(1, "int7", "cuda")


def test_flat_bucket_views():
    layout = (("a", torch.bfloat16, (2, 3), 0), ("s", torch.int64, (), 256), ("z", torch.float32, (0, 4), 512))
    flat = torch.zeros(768, dtype=torch.uint8)
    b = WeightBucket.from_flat(flat, layout, {"total_buckets": torch.tensor(1, dtype=torch.int32)})
    assert list(b) == ["total_buckets", "a", "s", "z"]  # metadata first, like the reference's first bucket
    b["a"].fill_(1.5), b["s"].fill_(-3)
    assert b["a"].shape == (2, 3) and b["s"].shape == () and b["z"].shape == (0, 4)
    again = WeightBucket.from_flat(flat.clone(), layout)
    assert torch.equal(again["a"], torch.full((2, 3), 1.5, dtype=torch.bfloat16)) and int(again["s"]) == -3
    assert flat[:12].view(torch.bfloat16).tolist() == [1.5] * 6


def test_factory_reads_the_reference_config_keys():
    s = WeightSyncer.create(DictConfig(dict(type="bucket", bucket=dict(bucket_size=1 << 20, bucket_dtype="bf16", load_instant=False),
                                            nccl_max_ctas=8)))
    assert isinstance(s, BucketWeightSyncer) and s.bucket_dtype == torch.bfloat16 and s.bucket_device.type == "cuda"
    assert s.load_instant is False and s.is_agent is False
    assert s.comm_options == dict(use_ring_broadcast=False, accel_max_ctas=8, accel_min_ctas=None)
    s = WeightSyncer.create(DictConfig(dict(type="patch", patch=dict(delta_encoding=False, compression="none",
                                                                   init_sync=dict(enabled=True, prefixes=["head"], buckets_size=99)))))
    assert isinstance(s, PatchWeightSyncer) and s.delta_encoding is False and s.comm_options is None
    assert s.init_sync_enabled and s.init_sync_prefixes == ["head"] and s.init_sync_bucket_size == 99
    with pytest.raises(ValueError, match="Unsupported weight syncer type"):
        WeightSyncer.create(DictConfig(dict(type="zip")))
    with pytest.raises(AssertionError, match="Bucket config must be provided"):
        WeightSyncer.create(DictConfig(dict(type="bucket")))


def test_init_sync_selection():
    state = {"head.weight": 1, "head.bias": 2, "header": 3, "body.0.w": 4, "logstd": 5}
    s = PatchWeightSyncer(init_sync_enabled=True, init_sync_prefixes=["head", "logstd"])
    assert s._select_init_sync_weights(state) == [("head.weight", 1), ("head.bias", 2), ("logstd", 5)]  # "header" is no child
    assert PatchWeightSyncer(init_sync_enabled=True)._select_init_sync_weights(state) == list(state.items())
    with pytest.raises(ValueError, match="did not match any state_dict keys: \\['tail'\\]"):
        PatchWeightSyncer(init_sync_enabled=True, init_sync_prefixes=["head", "tail"])._select_init_sync_weights(state)
    with pytest.raises(ValueError, match="must not be empty"):
        PatchWeightSyncer(init_sync_enabled=True, init_sync_prefixes=[])


def test_cpu_tensors_are_refused_loudly():
    from rlinf_amd._lib import RlxError
    s = BucketWeightSyncer(1 << 20, "bf16", "cuda")
    s.init_sender({}, ["w"])
    with pytest.raises(RlxError, match="no CPU path"):
        s.sync({"w": torch.zeros(4, 4)}, lambda b: None, 1)


def test_mlp_policy_state_dict_hands_out_reference_named_views():
    """syncers apply through model.state_dict(); MLPPolicy keeps ONE flat parameter, so its state_dict must hand
    out the reference's names as views that alias it (what torch's own state_dict does for ordinary modules)."""
    from oracle import ppo_oracle as O
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    torch.manual_seed(0)
    pol = MLPPolicy(42, 8, 1, True, False)
    ora = O.OracleMLPPolicy(42, 8, 1)
    sd = pol.state_dict()
    assert list(sd) == list(ora.state_dict()) and all(sd[k].shape == v.shape for k, v in ora.state_dict().items())
    sd["backbone.2.weight"].fill_(0.25)  # writes through
    o = pol.offsets["backbone.2.weight"]
    assert bool((pol.flat.data[o:o + 256 * 256] == 0.25).all())
    pol.load_state_dict(ora.state_dict())
    assert all(torch.equal(pol.state_dict()[k], v) for k, v in ora.state_dict().items())
    ora.load_state_dict(pol.state_dict())  # and the reference-shaped module accepts it back
    clone = MLPPolicy(42, 8, 1, True, False)
    clone.load_state_dict({"flat": pol.flat.detach().clone()})
    assert torch.equal(clone.flat, pol.flat)

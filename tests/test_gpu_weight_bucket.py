"""Bucketed weight sync on the GPU, through the C ABI (rlx_copy_segments): the mirror's buckets against the committed
reference buckets, both directions of interoperability (reference-made buckets applied here, buckets made here applied by
the oracle's receiver), dtype conversion on the receiver, host-staged buckets, the patch syncer's init sync, and a
full-size property check."""

import os

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import bucket_oracle as BO
from oracle.make_golden import bucket_state
from test_weight_bucket_host import same_bytes

pytestmark = pytest.mark.gpu


def _cuda(state):
    return {k: v.cuda() for k, v in state.items()}


def _sync(state, names, p, version=7, device="cuda"):
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    s = BucketWeightSyncer(p["bucket_size"], p["bucket_dtype"], device, is_agent=p["is_agent"])
    s.init_sender(state, names)
    sent = []
    s.sync(state, sent.append, version)
    return s, sent


def test_buckets_match_reference_fixture():
    for case in torch.load(os.path.join(GOLDEN_DIR, "weight_bucket.pt"), weights_only=False):
        p = case["params"]
        state, names = bucket_state(p["seed"])
        _, sent = _sync(_cuda(state), names, p)
        assert len(sent) == len(case["buckets"])
        for got, want in zip(sent, case["buckets"]):
            assert list(got) == list(want), p
            for k, w in want.items():
                assert got[k].is_cuda and same_bytes(got[k].cpu(), w), (p, k)
            # every payload tensor is a view of the one flat buffer, 256-byte aligned
            lo, hi = got.flat.data_ptr(), got.flat.data_ptr() + got.flat.numel()
            for k, _, _, off in got.layout:
                assert lo <= got[k].data_ptr() < hi or got[k].numel() == 0
                assert off % 256 == 0


@pytest.mark.parametrize("load_instant", [True, False])
@pytest.mark.parametrize("case_idx", [1, 2])
def test_round_trip_and_receiver_conversion(case_idx, load_instant):
    """sender (f32 / bf16 / f16 / int / bool masters) -> bf16 or f16 buckets -> a receiver holding OTHER dtypes: the final
    tensors equal torch's own chain value.to(transport).to(target dtype), and match the oracle receiver."""
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    case = torch.load(os.path.join(GOLDEN_DIR, "weight_bucket.pt"), weights_only=False)[case_idx]
    p = case["params"]
    state, names = bucket_state(p["seed"])
    _, sent = _sync(_cuda(state), names, p, version=12)
    flip = {torch.float32: torch.bfloat16, torch.bfloat16: torch.float32, torch.float16: torch.bfloat16}
    keys = [k for b in case["buckets"] for k in b if k not in ("total_buckets", "syncer_version")]
    target_cpu = {k: torch.zeros(v.shape, dtype=flip.get(v.dtype, v.dtype)) for b in case["buckets"] for k, v in b.items() if k in keys}
    target_cpu.pop(keys[1])  # a key the receiver does not have is ignored
    target_cpu["only_here"] = torch.full((5,), 3.0)
    target = _cuda(target_cpu)
    recv = iter(sent)
    r = BucketWeightSyncer(1, None, "cuda", load_instant=load_instant)
    r.init_receiver()
    assert r.apply(target, lambda: next(recv)) == 12
    assert BO.apply_buckets(target_cpu, case["buckets"]) == 7
    for k in target_cpu:
        assert same_bytes(target[k].cpu(), target_cpu[k]), k


def test_reference_made_buckets_on_host_and_size_mismatch():
    """What the reference's sender ships with bucket_device cpu (plain dicts of CPU tensors) loads here; a shape mismatch
    raises like load_state_dict does."""
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    case = torch.load(os.path.join(GOLDEN_DIR, "weight_bucket.pt"), weights_only=False)[3]
    state, _ = bucket_state(3)
    tgt = torch.nn.ParameterDict({k.replace(".", "_"): torch.nn.Parameter(torch.zeros_like(v), requires_grad=False)
                                  for k, v in state.items() if v.is_floating_point() and v.numel()}).cuda()
    renamed = [{k.replace(".", "_"): v.clone() for k, v in b.items()} for b in case["buckets"]]
    recv = iter(renamed)
    assert BucketWeightSyncer(1, None, "cpu").apply(tgt, lambda: next(recv)) == 7
    for b in case["buckets"]:
        for k, v in b.items():
            if k.replace(".", "_") in tgt:
                assert same_bytes(tgt[k.replace(".", "_")].detach().cpu(), v.to(tgt[k.replace(".", "_")].dtype)), k
    bad = {"total_buckets": torch.tensor(1, dtype=torch.int32), "syncer_version": torch.tensor(1, dtype=torch.int32),
           "w": torch.zeros(4, 4, device="cuda")}
    with pytest.raises(RuntimeError, match="size mismatch for w"):
        BucketWeightSyncer(1, None, "cuda").apply({"w": torch.zeros(4, 5, device="cuda")}, lambda: bad)


def test_host_staged_bucket_device():
    p = dict(bucket_size=50_000, bucket_dtype="bf16", is_agent=False)
    state, names = bucket_state(1)
    s, sent = _sync(_cuda(state), names, p, device="cpu")
    assert all(not v.is_cuda for b in sent for v in b.values()) and not sent[0].flat.is_cuda
    want = BO.make_buckets(state, names, 7, 50_000, torch.bfloat16)
    for g, w in zip(sent, want):
        assert list(g) == list(w) and all(same_bytes(g[k], w[k]) for k in w)
    target = {"backbone.weight": torch.zeros(37, 300, device="cuda")}
    recv = iter(sent)
    assert s.apply(target, lambda: next(recv)) == 7
    assert same_bytes(target["backbone.weight"].cpu(), state["backbone.weight"].bfloat16().float())


def test_non_contiguous_and_unaligned_sources():
    """Transposed parameters and views at odd storage offsets take the element-wise path; results are unchanged."""
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    g = torch.Generator().manual_seed(3)
    base = torch.randn(9000, generator=g).cuda()
    state = {"t": torch.randn(70, 130, generator=g).cuda().t(), "odd": base[1:8194], "odd16": base.bfloat16()[3:4100],
             "big": torch.randn(3 * 4096 + 5, generator=g).cuda()}
    s = BucketWeightSyncer(1 << 30, "fp16", "cuda")
    s.init_sender(state, list(state))
    sent = []
    s.sync(state, sent.append, 1)
    for k, v in state.items():
        assert same_bytes(sent[0][k].cpu(), v.cpu().half()), k
    target = {k: torch.zeros(v.shape, device="cuda")[..., :] for k, v in state.items()}
    recv = iter(sent)
    s.apply(target, lambda: next(recv))
    for k, v in state.items():
        assert same_bytes(target[k].cpu(), v.cpu().half().float()), k


def test_persistent_buckets_reuse_buffers_and_follow_the_weights():
    """persistent_buckets: the same transport buffers every sync (one launch when the parameters have not moved), fresh
    contents each time -- also after a parameter was re-allocated -- and the version in the metadata follows."""
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    g = torch.Generator().manual_seed(4)
    state = {f"p{i}": torch.randn(40 + i, 33, generator=g).cuda() for i in range(12)}
    s = BucketWeightSyncer(16_000, "bf16", "cuda", persistent_buckets=True)
    s.init_sender(state, list(state))
    replica = {k: torch.zeros(v.shape, dtype=torch.bfloat16, device="cuda") for k, v in state.items()}
    ptrs = None
    for version in (1, 2, 3):
        if version == 2:
            state["p3"] += 1.0                       # in place: same address
        if version == 3:
            state["p5"] = state["p5"] * 2.0 + 0.25   # re-allocated: the table is rebuilt
        sent = []
        s.sync(state, sent.append, version)
        assert len(sent) >= 2 and int(sent[0]["syncer_version"]) == version and list(sent[0])[:2] == ["total_buckets", "syncer_version"]
        now = [b.flat.data_ptr() for b in sent]
        assert ptrs is None or now == ptrs
        ptrs = now
        it = iter(sent)
        assert s.apply(replica, lambda: next(it)) == version
        for k, v in state.items():
            assert torch.equal(replica[k], v.bfloat16()), (version, k)


def test_unsupported_conversion_fails_loudly():
    from rlinf_amd._lib import RlxError
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    s = BucketWeightSyncer(1, "bf16", "cuda")
    s.init_sender({}, ["d"])
    with pytest.raises(RlxError):
        s.sync({"d": torch.zeros(8, dtype=torch.float64, device="cuda")}, lambda b: None, 1)


def test_patch_syncer_init_sync():
    """init_sync.enabled: the sender pushes the selected prefixes as buckets in the RECEIVER's dtypes during init, so the
    receiver starts out equal to the sender's snapshot and the first patch is empty."""
    from rlinf_amd.hybrid_engines.weight_syncer import EmptyWeightPatch, PatchWeightSyncer
    g = torch.Generator().manual_seed(9)
    master = {"head.w": torch.randn(64, 48, generator=g).cuda(), "head.b": torch.randn(48, generator=g).cuda(),
              "body.w": torch.randn(32, 32, generator=g).cuda()}
    replica = {"head.w": torch.zeros(64, 48, dtype=torch.bfloat16, device="cuda"), "head.b": torch.zeros(48, device="cuda"),
               "body.w": torch.zeros(32, 32, dtype=torch.bfloat16, device="cuda")}
    to_sender, to_receiver = [], []
    rx = PatchWeightSyncer(init_sync_enabled=True, init_sync_prefixes=["head"], init_sync_bucket_size=4096)
    tx = PatchWeightSyncer(init_sync_enabled=True, init_sync_prefixes=["head"], init_sync_bucket_size=4096)
    # the receiver announces itself first; its bucket reads happen after the sender has pushed (one process here)
    meta_box = []
    rx_init = dict(ordered_keys=list(replica), original_shapes={k: v.shape for k, v in replica.items()},
                   receiver_dtypes={k: v.dtype for k, v in replica.items()})
    tx.init_sender(master, list(master), to_receiver.append, lambda: rx_init)
    assert len(to_receiver) >= 2  # 64*48*2 bytes > 4096: more than one bucket
    it = iter(to_receiver)
    rx.init_receiver(replica, lambda: next(it), meta_box.append)
    assert meta_box[0]["ordered_keys"] == rx_init["ordered_keys"]
    assert torch.equal(replica["head.w"], master["head.w"].bfloat16()) and torch.equal(replica["head.b"], master["head.b"])
    assert not replica["body.w"].any()  # not under an init-sync prefix
    patch = tx.create_patch(master, 1)
    assert isinstance(patch, EmptyWeightPatch)


def test_full_size_property():
    """A 1.07 G-parameter state (4.3 GB of f32 masters) -> bf16 buckets of 512 MiB -> bf16 replica: equals torch's cast,
    tensor by tensor; bucket count and sizes follow the plan rule."""
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    torch.manual_seed(0)
    shapes = [(8192, 8192)] * 15 + [(32768, 2048), (8192,), (1,)]
    state = {f"layers.{i}.w": torch.randn(s, device="cuda") for i, s in enumerate(shapes)}
    s = BucketWeightSyncer(512 << 20, "bf16", "cuda")
    s.init_sender(state, list(state))
    replica = {k: torch.zeros(v.shape, dtype=torch.bfloat16, device="cuda") for k, v in state.items()}
    n_buckets = [0]

    def recv_factory():
        gen = s.iter_buckets(state, 3)

        def recv():
            n_buckets[0] += 1
            return next(gen)
        return recv

    assert s.apply(replica, recv_factory()) == 3
    assert n_buckets[0] == 5  # four 128 MiB tensors close a bucket: 16 of them -> 4 buckets, the two small ones ride in a 5th
    for k, v in state.items():
        assert torch.equal(replica[k], v.bfloat16()), k


@pytest.mark.parametrize("syncer_kind", ["bucket", "patch"])
def test_apply_into_mlp_policy_changes_weights_and_derived_images(syncer_kind):
    """A rollout replica that does NOT alias the learner's model: reference-named buckets / patches applied through
    ``apply(model)`` must land in MLPPolicy's flat buffer (its state_dict() hands out reference-named views) and the cached
    fragment-tile image must follow -- an apply that matches nothing raises instead of silently keeping the old weights."""
    from rlinf_amd import ops
    from rlinf_amd._lib import RlxError
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer, PatchWeightSyncer
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    torch.manual_seed(0)
    actor = MLPPolicy(42, 8, 1, True, False).cuda()
    torch.manual_seed(1)
    replica = MLPPolicy(42, 8, 1, True, False).cuda()
    assert not torch.equal(actor.flat, replica.flat)
    stale_tiles = replica.tiles().clone()
    names = list(actor.shapes)
    if syncer_kind == "bucket":
        tx, rx = BucketWeightSyncer(1 << 20, None, "cuda"), BucketWeightSyncer(1 << 20, None, "cuda")
        tx.init_sender(actor.state_dict(), names)
        sent = []
        tx.sync(actor.state_dict(), sent.append, 3)
        assert len(sent) > 1
        it = iter(sent)
        assert rx.apply(replica, lambda: next(it)) == 3
    else:
        tx = PatchWeightSyncer(snapshot_device="cuda", transport_device="cuda", delta_encoding=True)
        rx = PatchWeightSyncer(snapshot_device="cuda", transport_device="cuda", delta_encoding=True)
        to_rx, to_tx = [], []
        # both start from the replica's weights (init sync off), then the actor's weights travel as one patch
        start = {k: v.clone() for k, v in replica.state_dict().items()}
        rx.init_receiver(replica.state_dict(), None, to_tx.append)
        tx.init_sender(start, names, None, lambda: to_tx.pop(0))
        tx.sync(actor.state_dict(), to_rx.append, 3)
        assert rx.apply(replica, lambda: to_rx.pop(0)) == 3
    assert torch.equal(actor.flat, replica.flat)
    fresh = ops.mlp_pack_tiles(actor.flat.data, actor.layout)
    assert torch.equal(replica.tiles(), fresh) and not torch.equal(stale_tiles, fresh)
    if syncer_kind == "bucket":  # a sync of which nothing lands is a wiring error, not a no-op
        sent_again = []
        tx.sync(actor.state_dict(), sent_again.append, 4)
        it = iter(sent_again)
        with pytest.raises(RlxError, match="none matches a key"):
            rx.apply({"some.other.module.weight": torch.zeros(4, 4, device="cuda")}, lambda: next(it))

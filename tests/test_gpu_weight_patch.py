"""GPU parity of the weight-patch kernels (weight_patch.hip through the C ABI and the PatchWeightSyncer mirror):
byte-for-byte against the committed reference patches and the CPU oracle, both directions of the wire, and
encode -> apply round trips at sizes the oracle would crawl through."""

import os

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import patch_oracle as PO
from oracle.make_golden import patch_states
from rlinf_amd.hybrid_engines.weight_syncer import EmptyWeightPatch, PatchWeightSyncer, WeightPatch

pytestmark = pytest.mark.gpu
DEV = "cuda"
FIELDS = ("version", "ordinals", "nnz_per_tensor", "rows", "cols", "values")


class Pipe:
    def __init__(self):
        self.q = []

    def send(self, x):
        self.q.append(x)

    def recv(self):
        return self.q.pop(0)


def _pair(before, names, delta, narrow=False):
    """(sender syncer, receiver syncer, receiver state) initialised through the reference's handshake."""
    recv_state = {k: (v.to(torch.bfloat16) if (narrow and v.dtype == torch.float32) else v.clone()).to(DEV)
                  for k, v in before.items()}
    sender_state = {k: v.to(DEV) for k, v in before.items()}
    up, down = Pipe(), Pipe()
    rx, tx = PatchWeightSyncer(delta_encoding=delta), PatchWeightSyncer(delta_encoding=delta)
    rx.init_receiver(recv_state, down.recv, up.send)
    tx.init_sender(sender_state, names, down.send, up.recv)
    return tx, rx, recv_state, down


def _same_bits(a, b):
    return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))


def test_patches_match_reference_fixture_byte_for_byte():
    for case in torch.load(os.path.join(GOLDEN_DIR, "weight_patch.pt"), weights_only=False):
        p = case["params"]
        before, after = patch_states(p["seed"])
        tx, rx, recv_state, pipe = _pair(before, p["names"], p["delta"], p["narrow"])
        after_dev = {k: v.to(DEV) for k, v in after.items()}
        for version, want in zip((11, 12), case["patches"]):
            got = tx.create_patch(after_dev, version)
            assert isinstance(got, WeightPatch)
            for f in FIELDS:
                a = getattr(got, f).cpu()
                assert a.dtype == want[f].dtype and a.shape == want[f].shape, (p, version, f)
                if f != "values":
                    assert torch.equal(a, want[f]), (p, version, f)
            # value bytes: identical, except that the payload of a NaN produced by an f32 -> bf16 narrowing is whatever
            # the converting backend writes (torch's vectorised CPU cast, which made the fixture: 0xFFFF; c10's scalar
            # converter, torch on the GPU and this kernel: 0x7FC0) -- compared as "both NaN"
            off = 0
            for ordinal, nnz in zip(want["ordinals"].tolist(), want["nnz_per_tensor"].tolist()):
                dt = tx.snapshot[p["keys"][ordinal]].dtype
                nb = nnz * torch.empty((), dtype=dt).element_size()
                ga, wa = got.values.cpu()[off:off + nb].clone().view(dt), want["values"][off:off + nb].clone().view(dt)
                off += nb
                if dt.is_floating_point:
                    assert bool(((ga == wa) & (ga.view(torch.int16 if ga.element_size() == 2 else torch.int32)
                                               == wa.view(torch.int16 if wa.element_size() == 2 else torch.int32))
                                 | ((ga != ga) & (wa != wa))).all()), (p, version, ordinal)
                else:
                    assert torch.equal(ga, wa), (p, version, ordinal)
        # the sender's snapshot now equals the new weights in the receiver's dtypes (NaN bits included)
        for k, v in after.items():
            assert _same_bits(tx.snapshot[k].cpu(), PO.coo_2d_view(v).to(tx.snapshot[k].dtype)) or k == "backbone.weight"
        # a patch produced by the REFERENCE applied by the kernels
        pipe.send(WeightPatch(**{f: case["patches"][0][f] for f in FIELDS}))
        assert rx.apply(recv_state, pipe.recv) == 11
        for k, v in after.items():
            want = PO.coo_2d_view(v).to(recv_state[k].dtype)
            got = PO.coo_2d_view(recv_state[k].cpu())
            if want.is_floating_point():
                assert torch.equal(torch.nan_to_num(got.float(), nan=7.5), torch.nan_to_num(want.float(), nan=7.5)), k
            else:
                assert torch.equal(got, want), k


@pytest.mark.parametrize("delta", [True, False])
def test_kernel_patch_applied_by_the_oracle_and_empty_patch(delta):
    before, after = patch_states(5)
    names = list(before)
    tx, rx, recv_state, pipe = _pair(before, names, delta)
    patch = tx.create_patch({k: v.to(DEV) for k, v in after.items()}, 21)
    cpu_target = {k: v.clone() for k, v in before.items()}
    assert PO.apply_patch(cpu_target, names, {f: getattr(patch, f).cpu() for f in FIELDS}, delta) == 21
    for k, v in after.items():
        assert torch.equal(torch.nan_to_num(cpu_target[k].float(), nan=7.5), torch.nan_to_num(v.float(), nan=7.5)), k
    # nothing changed (apart from the NaN, which torch.ne always reports): remove it and the next patch is empty
    clean = {k: v.to(DEV) for k, v in after.items()}
    clean["backbone.weight"][3, 7] = 1.0
    assert isinstance(tx.create_patch(clean, 22), WeightPatch)      # the 1.0 replaces the NaN
    empty = tx.create_patch(clean, 23)
    assert isinstance(empty, EmptyWeightPatch) and int(empty.version) == 23
    pipe.send(empty)
    assert rx.apply(recv_state, pipe.recv) == 23


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16, torch.int8])
@pytest.mark.parametrize("density", [1e-4, 0.05, 1.0])
@pytest.mark.parametrize("delta", [True, False])
def test_round_trip_large(dtype, density, delta):
    """encode -> wire -> apply restores the sender's tensor exactly; nnz equals the number of changed elements; shapes
    that are not multiples of the 16384-element scan block, of the 64-element mask word, or of 8."""
    g = torch.Generator(device=DEV).manual_seed(11)
    rows, cols = 1537, 4099
    if dtype == torch.int8:
        base = torch.randint(-100, 100, (rows, cols), device=DEV, generator=g, dtype=torch.int8)
        new = torch.where(torch.rand(rows, cols, device=DEV, generator=g) < density, base + 1, base)
    else:
        base = torch.randn(rows, cols, device=DEV, generator=g).to(dtype)
        new = torch.where(torch.rand(rows, cols, device=DEV, generator=g) < density, base * 2 + 1, base)
    state = {"w": base.clone(), "b": torch.zeros(7, device=DEV)}
    up, down = Pipe(), Pipe()
    rx, tx = PatchWeightSyncer(delta_encoding=delta), PatchWeightSyncer(delta_encoding=delta)
    recv_state = {k: v.clone() for k, v in state.items()}
    rx.init_receiver(recv_state, down.recv, up.send)
    tx.init_sender(state, ["w", "b"], down.send, up.recv)
    patch = tx.create_patch({"w": new, "b": state["b"]}, 1)
    changed = int((new != base).sum())
    assert int(patch.nnz_per_tensor.sum()) == changed and patch.values.numel() == changed * base.element_size()
    r, c = (new != base).nonzero(as_tuple=True)
    if delta:
        dr, dc = PO.delta_encode(r.cpu(), c.cpu())
    else:
        dr, dc = r.cpu(), c.cpu()
    assert torch.equal(patch.rows.cpu().to(torch.int64), dr) and torch.equal(patch.cols.cpu().to(torch.int64), dc)
    assert patch.rows.dtype == PO.downscale(dr).dtype and patch.cols.dtype == PO.downscale(dc).dtype
    down.send(patch)
    assert rx.apply(recv_state, down.recv) == 1
    assert _same_bits(recv_state["w"], new) and _same_bits(tx.snapshot["w"], new)


def test_unaligned_views_and_wide_sender():
    """A sender tensor that starts 2 bytes into its storage (no 16-byte loads possible) and an f32 sender feeding a
    bf16 receiver (compare after conversion, send bf16 bytes)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    store = torch.randn(1 + 333 * 77, device=DEV, generator=g).bfloat16()
    w = store[1:].view(333, 77)
    master = torch.randn(129, 515, device=DEV, generator=g)
    state = {"w": w, "m": master}
    recv_state = {"w": w.clone(), "m": master.to(torch.bfloat16)}
    up, down = Pipe(), Pipe()
    rx, tx = PatchWeightSyncer(), PatchWeightSyncer()
    rx.init_receiver(recv_state, down.recv, up.send)
    tx.init_sender(state, ["w", "m"], down.send, up.recv)
    assert tx.snapshot["m"].dtype == torch.bfloat16
    w2 = w.clone()
    w2[::7, ::5] += 1
    m2 = master + 1e-4 * (torch.rand_like(master) < 0.5)   # most of these vanish in bf16: only real bf16 changes travel
    patch = tx.create_patch({"w": w2, "m": m2}, 9)
    want_m = int((m2.to(torch.bfloat16) != master.to(torch.bfloat16)).sum())
    assert patch.nnz_per_tensor.tolist() == [int((w2 != w).sum()), want_m]
    down.send(patch)
    rx.apply(recv_state, down.recv)
    assert _same_bits(recv_state["w"], w2) and _same_bits(recv_state["m"], m2.to(torch.bfloat16))

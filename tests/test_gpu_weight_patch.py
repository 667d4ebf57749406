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


# ---- f3: the compression stage (csrc/zplane_codec.hip behind PatchCompressor) and the host-snapshot builder ------------------
def _zplane(t: torch.Tensor):
    from rlinf_amd.hybrid_engines.weight_syncer import ZPlaneCompressor
    c = ZPlaneCompressor("rlx_zplane", "cuda")
    out, length, keep = c._launch_compress(t)
    return c, out[:int(length.item())].clone()


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int16, torch.int32, torch.int64, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n", [0, 1, 63, 4095, 4096, 4097, 70000])
def test_zplane_codec_streams_are_byte_identical_to_the_numpy_restatement(dtype, n):
    """HIP encoder == numpy encoder byte for byte (all four plane modes occur: zeros, sparse -> masks, constant runs -> XOR-filtered
    masks, noise -> raw), HIP decoder reads the numpy stream, numpy decoder reads the HIP stream, and both restore the input."""
    import numpy as np

    from oracle import zplane_oracle as Z
    g = torch.Generator().manual_seed(n + 17)
    es = torch.empty((), dtype=dtype).element_size()
    raw = torch.randint(0, 256, (n * es,), generator=g, dtype=torch.uint8)
    quarter = max(n * es // 4, 1)
    raw[quarter:2 * quarter] = 0                                                          # all-zero planes
    raw[2 * quarter:3 * quarter] *= (torch.rand(len(raw[2 * quarter:3 * quarter]), generator=g) < 0.04).to(torch.uint8)   # sparse
    raw[3 * quarter:] = 5                                                                 # a constant run
    host = raw.clone().view(dtype) if n else torch.empty(0, dtype=dtype)
    c, stream = _zplane(host.to(DEV))
    want = Z.compress(raw.numpy(), es)
    assert stream.numel() == want.size and np.array_equal(stream.cpu().numpy(), want), (dtype, n, stream.numel(), want.size)
    code = torch.tensor({torch.uint8: 0, torch.int16: 1, torch.int32: 2, torch.int64: 3, torch.bfloat16: 5, torch.float32: 6}[dtype],
                        dtype=torch.int8)
    c._pending_status = []
    back = c._decompress_tensor(torch.from_numpy(want).to(DEV), code)      # numpy stream -> HIP decoder
    assert back.dtype == dtype and _same_bits(back.cpu(), host)
    assert not c._pending_status or int(torch.cat(c._pending_status).max()) == 0
    back_np, es2 = Z.decompress(stream.cpu().numpy())                      # HIP stream -> numpy decoder
    assert es2 == es and np.array_equal(back_np, raw.numpy())


@pytest.mark.parametrize("dtype,n", [(torch.int64, 4096 * 520 + 77), (torch.uint8, 4096 * 4101 + 5), (torch.uint8, 4096 * 6 + 1000)],
                         ids=["int64-two-scan-tiles", "uint8-two-scan-tiles", "uint8-partial-workgroup"])
@pytest.mark.parametrize("single_pass", ["0", "1"], ids=["multi-launch", "single-pass"])
def test_zplane_codec_large_streams(dtype, n, single_pass, monkeypatch):
    """More directory entries than one scan tile (4096) holds -- the offsets come out of the two-level scan -- and byte streams
    whose block count is not a multiple of the blocks a workgroup takes: byte-identical to the numpy restatement, both decoders."""
    import numpy as np

    from oracle import zplane_oracle as Z
    if single_pass == "1":
        from conftest import need_dev_variants
        need_dev_variants("the single-pass zplane encoder")
    monkeypatch.setenv("RLX_ZPLANE_SINGLE_PASS", single_pass)  # "1": the one-launch encoder with the decoupled look-back (kept selectable)
    g = torch.Generator().manual_seed(n)
    es = torch.empty((), dtype=dtype).element_size()
    raw = torch.randint(0, 256, (n * es,), generator=g, dtype=torch.uint8)
    quarter = n * es // 4
    raw[quarter:2 * quarter] = 0
    raw[2 * quarter:3 * quarter] *= (torch.rand(quarter, generator=g) < 0.04).to(torch.uint8)
    raw[3 * quarter:] = 5
    c, stream = _zplane(raw.clone().view(dtype).to(DEV))
    want = Z.compress(raw.numpy(), es)
    assert stream.numel() == want.size and np.array_equal(stream.cpu().numpy(), want)
    c._pending_status = []
    code = torch.tensor({torch.uint8: 0, torch.int64: 3}[dtype], dtype=torch.int8)
    back = c._decompress_tensor(stream, code)
    assert _same_bits(back.cpu(), raw.view(dtype)) and int(torch.cat(c._pending_status).max()) == 0


def test_zplane_decoder_rejects_truncated_and_foreign_streams():
    from rlinf_amd._lib import RlxError
    c, stream = _zplane(torch.arange(20000, dtype=torch.int32, device=DEV) % 300)
    code = torch.tensor(2, dtype=torch.int8)
    c._pending_status = []
    with pytest.raises(RlxError, match="does not match"):
        c._decompress_tensor(stream[:-8].clone(), code)            # shorter than its header says
    with pytest.raises(RlxError, match="truncated"):
        c._decompress_tensor(stream[:11].clone(), code)            # shorter than the header itself: refused before the C side reads 24 bytes
    with pytest.raises(RlxError, match="does not match"):
        c._decompress_tensor(stream, torch.tensor(3, dtype=torch.int8))   # dtype code of another width
    bad = stream.clone()
    bad[0] = 0
    with pytest.raises(RlxError, match="not an RLXZ v1 stream"):
        c._decompress_tensor(bad, code)
    # a directory entry pointing past the end: the kernel reads nothing out of bounds and raises the status word
    bad = stream.clone()
    bad[24:32] = torch.tensor([0xF0, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x7F], dtype=torch.uint8, device=DEV)  # mode 1, offset 2^62 - 16
    c._pending_status = []
    c._decompress_tensor(bad, code)
    assert int(torch.cat(c._pending_status).max()) != 0


@pytest.mark.parametrize("algorithm", ["rlx_zplane"])
@pytest.mark.parametrize("density,delta", [(1e-3, True), (0.3, True), (1.0, True), (0.05, False)])
def test_compressed_patch_transport_round_trip(algorithm, density, delta):
    """PatchWeightSyncer with a compression algorithm: sync() sends a CompressedWeightPatch (the reference's fields and dtype
    codes), apply() restores the receiver bit for bit; the decompressed patch equals the uncompressed builder's patch field by
    field; sparse and DENSE updates (column deltas all 1 -> the XOR-filtered mode) really shrink the index streams."""
    from rlinf_amd.hybrid_engines.weight_syncer import CompressedWeightPatch
    g = torch.Generator(device=DEV).manual_seed(4)
    base = torch.randn(1031, 4099, device=DEV, generator=g).bfloat16()
    new = torch.where(torch.rand(base.shape, device=DEV, generator=g) < density, base * 2 + 1, base)
    state, recv_state = {"w": base.clone(), "b": torch.zeros(5, device=DEV)}, {"w": base.clone(), "b": torch.zeros(5, device=DEV)}
    up, down = Pipe(), Pipe()
    rx = PatchWeightSyncer(delta_encoding=delta, compression_algorithm=algorithm)
    tx = PatchWeightSyncer(delta_encoding=delta, compression_algorithm=algorithm)
    plain = PatchWeightSyncer(delta_encoding=delta)
    rx.init_receiver(recv_state, down.recv, up.send)
    meta = up.q[0]
    tx.init_sender(state, ["w", "b"], down.send, up.recv)
    plain.init_sender({k: v.clone() for k, v in state.items()}, ["w", "b"], None, lambda: meta)
    want = plain.create_patch({"w": new, "b": state["b"]}, 3)
    tx.sync({"w": new, "b": state["b"]}, down.send, 3)
    payload = down.q[0]
    assert isinstance(payload, CompressedWeightPatch)
    assert payload.rows_dtype_code.dtype == torch.int8 and int(payload.values_dtype_code) == 0     # value bytes travel as uint8
    assert int(payload.rows_dtype_code) == {torch.uint8: 0, torch.int32: 2, torch.int64: 3}[want.rows.dtype]
    got = rx.compressor.decompress(payload)
    for f in FIELDS:
        assert getattr(got, f).dtype == getattr(want, f).dtype and torch.equal(getattr(got, f), getattr(want, f)), f
    index_bytes = want.rows.numel() * want.rows.element_size() + want.cols.numel() * want.cols.element_size()
    packed = payload.rows_compressed.numel() + payload.cols_compressed.numel()
    if delta:
        assert packed < 0.5 * index_bytes, (density, packed, index_bytes)
    assert payload.values_compressed.numel() <= want.values.numel() + 24 + 16 * (want.values.numel() // 4096 + 1)
    assert rx.apply(recv_state, down.recv) == 3
    assert _same_bits(recv_state["w"], new)
    # an unchanged state: the EmptyWeightPatch travels uncompressed (patch_syncer.py:1036-1041)
    tx.sync({"w": new, "b": state["b"]}, down.send, 4)
    assert isinstance(down.q[0], EmptyWeightPatch) and rx.apply(recv_state, down.recv) == 4


@pytest.mark.parametrize("delta", [True, False])
def test_cpu_snapshot_builder_matches_the_same_device_builder(delta):
    """snapshot_device="cpu" (CPUSnapshotPatchBuilder, patch_syncer.py:416-640): the snapshot sits in pinned host memory, is staged
    one tensor ahead and written back only for tensors that changed -- the patches equal the same-device builder's byte for
    byte over three syncs (changes, no changes, changes again), and the host snapshot ends up equal to the sender's weights."""
    before, after = patch_states(5)
    names = list(before)
    recv_state = {k: v.clone().to(DEV) for k, v in before.items()}
    up, down = Pipe(), Pipe()
    rx = PatchWeightSyncer(delta_encoding=delta)
    rx.init_receiver(recv_state, down.recv, up.send)
    meta = up.q[0]
    dev_tx, cpu_tx = PatchWeightSyncer(delta_encoding=delta), PatchWeightSyncer(delta_encoding=delta, snapshot_device="cpu")
    dev_tx.init_sender({k: v.to(DEV) for k, v in before.items()}, names, None, lambda: meta)
    cpu_tx.init_sender({k: v.to(DEV) for k, v in before.items()}, names, None, lambda: meta)
    assert all(t.device.type == "cpu" and t.is_pinned() for t in cpu_tx.snapshot.values())
    after2 = {k: v.clone() for k, v in after.items()}
    after2["backbone.weight"][3, 7] = 1.0          # replaces the NaN (torch.ne reports a NaN as changed on every sync)
    after3 = {k: (v + 1 if v.is_floating_point() and v.numel() > 8 else v.clone()) for k, v in after2.items()}
    for version, st in ((1, after), (2, after2), (3, after2), (4, after3)):
        dev_state = {k: v.to(DEV) for k, v in st.items()}
        a, b = dev_tx.create_patch(dev_state, version), cpu_tx.create_patch(dev_state, version)
        assert type(a) is type(b), version
        if isinstance(a, WeightPatch):
            for f in FIELDS:
                assert getattr(a, f).dtype == getattr(b, f).dtype and torch.equal(getattr(a, f), getattr(b, f)), (version, f)
        else:
            assert version == 3 and int(b.version) == 3
    for k, v in after3.items():
        assert _same_bits(cpu_tx.snapshot[k], dev_tx.snapshot[k].cpu()), k
    down.send(cpu_tx.create_patch({k: v.to(DEV) for k, v in before.items()}, 5))     # and back again, applied by the kernels
    assert rx.apply(recv_state, down.recv) == 5

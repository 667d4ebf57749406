"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the
entry points include/rlx.h declares (no compute is launched here)."""

import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rlx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rlx_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from rlinf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from rlinf_amd.csrc import build
        build.build(verbose=False)
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from rlinf_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 8
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/rlx.h but not exported"
    assert sorted(_lib.PROTOTYPES) == declared, "ctypes prototype table out of sync with include/rlx.h"


def test_version_and_error_string(lib):
    assert lib.rlx_version() >= 112
    assert isinstance(lib.rlx_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    # NULL pointers / bad sizes are rejected before any HIP call, so this is safe on a CPU-only box.
    from rlinf_amd._lib import GaeParams
    p = GaeParams(0.99, 0.94, 1, 0, 1e-5, 0)
    rc = lib.rlx_gae_scan(None, None, None, None, None, None, None, 0, 4, 4, 1, ctypes.byref(p), None)
    assert rc == -22
    assert b"NULL" in lib.rlx_last_error()
    assert lib.rlx_gae_scan(None, None, None, None, None, None, None, 0, 0, 4, 1, ctypes.byref(p), None) == 0  # empty
    assert lib.rlx_gae_workspace_bytes(128, 1024, 1) == 4096 + 16 * 5 * 8  # the published normalisation words, then a partial per 64 envs
    rc = lib.rlx_grpo_group_adv(None, None, None, None, None, 4, 10, 1, 4, 1e-6, None)
    assert rc == -22 and b"group_size" in lib.rlx_last_error()


def test_ops_fail_loudly_on_cpu_tensors(lib):
    import torch
    from rlinf_amd import ops
    from rlinf_amd._lib import RlxError
    with pytest.raises(RlxError, match="no CPU fallback"):
        ops.gae_scan(torch.zeros(2, 2, 1), torch.zeros(3, 2, 1), torch.zeros(3, 2, 1, dtype=torch.bool))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rlinf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_argument_validation_of_the_widening_entries(lib):
    """Same rule for the token / patch / decoupled entries: bad arguments come back as -22 with a message, before any
    HIP call; empty problems return 0."""
    from rlinf_amd._lib import DecoupledLossParams, TokenLossParams, TokenRows
    rows = TokenRows(n_tokens=4, vocab=0, dtype=0, rows_per_seq=4, seq_stride=0, row_stride=8, temperature=1.0)
    assert lib.rlx_token_logprob_fwd(None, None, ctypes.byref(rows), None, None, None, None) == -22
    assert b"bad sizes" in lib.rlx_last_error()
    rows.vocab, rows.temperature = 8, 0.0
    assert lib.rlx_token_logprob_fwd(None, None, ctypes.byref(rows), None, None, None, None) == -22
    assert b"temperature" in lib.rlx_last_error()
    rows.temperature, rows.n_tokens = 1.0, 0
    assert lib.rlx_token_logprob_fwd(None, None, ctypes.byref(rows), None, None, None, None) == 0  # nothing to do
    rows.n_tokens, rows.vocab, rows.row_stride = 4, 2000, 2000
    assert lib.rlx_categorical_sample(None, ctypes.byref(rows), None, -1, 0, None, 0, None, None, None, None) == -22
    assert b"1024" in lib.rlx_last_error()
    tp = TokenLossParams()
    tp.loss_agg = 7
    one = ctypes.c_float(0)
    ptr = ctypes.addressof(one)
    assert lib.rlx_token_loss_fwd(ptr, ptr, ptr, None, None, None, 1, 1, ctypes.byref(tp), ptr, None, ptr, ptr, ptr, 64,
                                  None) == -22
    assert b"loss_agg" in lib.rlx_last_error()
    assert lib.rlx_grpo_seq_adv(None, None, None, 10, 4, 4, 1e-6, None) == -22 and b"group_size" in lib.rlx_last_error()
    assert lib.rlx_gae_seq(None, None, None, None, 0, 16, 1.0, 0.95, None, 0, None) == 0
    assert lib.rlx_gae_seq(None, None, None, None, 4, 16, 1.0, 0.95, None, 0, None) == -22
    assert lib.rlx_patch_scan(None, 1, None, 1, 0, None, 0, None, None) == -22
    assert lib.rlx_patch_apply(None, 99, 1, 1, ptr, 0, ptr, 0, 0, ptr, 1, None, 0, None) == -22
    # copy_segments: the plan is a host function -- it validates and numbers the chunks without a GPU
    from rlinf_amd._lib import CopySegment
    table = (CopySegment * 4)(CopySegment(4096, 8192, 10000, 0, 1, -1), CopySegment(0, 0, 0, 3, 3, -1),
                              CopySegment(256, 512, 4096, 6, 6, -1), CopySegment(64, 128, 1, 2, 0, -1))
    total = ctypes.c_int64(-1)
    assert lib.rlx_copy_segments_plan(table, 4, ctypes.byref(total)) == 0
    assert [t.first_chunk for t in table] == [0, 3, 3, 4] and total.value == 5  # ceil(10000/4096), 0, 1, 1
    table[0].dst_dtype = 6  # f32 -> 8-byte raw: no such conversion
    assert lib.rlx_copy_segments_plan(table, 4, ctypes.byref(total)) == -22
    table[0].dst_dtype, table[0].src = 1, 4098  # f32 source at a 2-byte boundary
    assert lib.rlx_copy_segments_plan(table, 4, ctypes.byref(total)) == -22
    assert lib.rlx_copy_segments(None, 0, 0, None) == 0 and lib.rlx_copy_segments(None, 3, 5, None) == -22
    dp = DecoupledLossParams()
    dp.ppo.raw_per_adv, dp.ppo.sub_per_adv, dp.proximal_mode = 4, 1, 9
    assert lib.rlx_decoupled_loss_fwd(ptr, ptr, None, None, ptr, None, None, None, None, None, 1, ctypes.byref(dp), ptr, None,
                                      ptr, ptr, 1 << 20, None) == -22
    assert b"proximal_mode" in lib.rlx_last_error()
    assert lib.rlx_reward_filter_mask(None, None, None, 4, 10, 1, 4, 0.0, 1.0, None) == -22
    assert lib.rlx_patch_workspace_bytes(1 << 20) > (1 << 20) // 8


def test_split_k_slab_plan_by_precision(lib):
    """rlx_ppo_step_slabs_for: the slab count `grads` must hold depends on the operand precision (f32: two workgroups per CU
    in one round; bf16: fewer, larger slabs -- the launch is bound by slab bytes), never leaves a slab with fewer than 256
    rows (a data-parallel rank's small minibatch), and the legacy entry is the f32 plan.  Pure host arithmetic: no GPU."""
    from ctypes import byref
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    lay = MLPPolicy(42, 8, 1, True, False).layout
    for m in (1, 5, 100, 256, 257, 1024, 2048, 4096, 8192, 65536):
        f32, bf16 = lib.rlx_ppo_step_slabs_for(byref(lay), m, 0), lib.rlx_ppo_step_slabs_for(byref(lay), m, 1)
        assert lib.rlx_ppo_step_slabs(byref(lay), m) == f32
        assert 1 <= bf16 <= f32 <= max(1, -(-m // 256)), (m, f32, bf16)
        rows = -(-m // f32)
        rows = -(-rows // 32) * 32          # rows per slab are whole 32-row k-blocks ...
        assert -(-m // rows) == f32, (m, f32)  # ... and every slab is non-empty: the launch writes each one
    assert lib.rlx_ppo_step_slabs_for(byref(lay), 0, 1) == 1 and lib.rlx_ppo_step_slabs_for(None, 8192, 1) == 1


def test_struct_mirrors_match_the_library(lib):
    """rlx_abi_struct_sizes: the ctypes mirrors of every argument struct have the size the library was compiled with (load()
    refuses a mismatch; this names the struct when it happens) and a short buffer is honoured."""
    from rlinf_amd import _lib
    mirrors = (_lib.GaeParams, _lib.PpoLossParams, _lib.GatherField, _lib.AdamwGroup, _lib.AdamwParams, _lib.MlpLayout, _lib.ValueJob,
               _lib.RolloutStep, _lib.PpoStepArgs, _lib.DecoupledLossParams, _lib.TokenRows, _lib.TokenLossParams, _lib.CopySegment)
    sizes = (ctypes.c_size_t * 16)(*([0] * 16))
    assert lib.rlx_abi_struct_sizes(ctypes.cast(sizes, ctypes.c_void_p), 16) == len(mirrors)
    assert [ctypes.sizeof(c) for c in mirrors] == list(sizes)[:len(mirrors)] and list(sizes)[len(mirrors):] == [0, 0, 0]
    short = (ctypes.c_size_t * 2)(7, 7)
    assert lib.rlx_abi_struct_sizes(ctypes.cast(short, ctypes.c_void_p), 1) == len(mirrors) and short[1] == 7
    assert lib.rlx_abi_struct_sizes(None, 0) == len(mirrors)


def test_argument_validation_of_the_decoupled_step_entries(lib):
    """The round-4 entries reject bad arguments before any HIP call: a decoupled rlx_ppo_step without the tensors its proximal mode
    needs, slab groups that do not divide the slabs, a deferred range outside the buffer."""
    from rlinf_amd import _lib
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    lay = MLPPolicy(42, 8, 1, True, False).layout
    p = _lib.AdamwParams()
    fake = ctypes.c_void_p(4096)  # never dereferenced: validation fails first
    assert lib.rlx_sum_slabs_deferred(fake, 16, 4, fake, ctypes.byref(p), None) == -22 and b"deferred_scale" in lib.rlx_last_error()
    p.deferred_scale, p.deferred_stride, p.deferred_groups = 4096, 22, 3
    assert lib.rlx_sum_slabs_deferred(fake, 16, 4, fake, ctypes.byref(p), None) == -22 and b"deferred_groups=3" in lib.rlx_last_error()
    p.deferred_groups = 2
    p.deferred_range[0][0], p.deferred_range[0][1] = 0, 17
    assert lib.rlx_sum_slabs_deferred(fake, 16, 4, fake, ctypes.byref(p), None) == -22 and b"deferred_range 0" in lib.rlx_last_error()
    assert lib.rlx_gaussian_entropy_bonus_deferred(fake, 8, fake, fake, 0.01, 1.0, 0, 1.0, None, None) == -22
    loss = _lib.PpoLossParams()
    loss.raw_per_adv, loss.sub_per_adv, loss.has_critic = 8, 1, 0
    dp = _lib.DecoupledLossParams()
    a = _lib.PpoStepArgs()
    a.layout, a.loss, a.m = ctypes.pointer(lay), ctypes.pointer(loss), 64
    for f in ("params", "states", "action", "old_logprobs", "advantages", "grads", "out", "workspace"):
        setattr(a, f, 4096)
    a.decoupled = ctypes.addressof(dp)
    dp.proximal_mode = 7
    assert lib.rlx_ppo_step(ctypes.byref(a), None) == -22 and b"proximal_mode 7" in lib.rlx_last_error()
    dp.proximal_mode = _lib.PROX_GIVEN
    assert lib.rlx_ppo_step(ctypes.byref(a), None) == -22 and b"GIVEN without proximal_logprobs" in lib.rlx_last_error()
    dp.proximal_mode = _lib.PROX_FROM_VERSIONS
    assert lib.rlx_ppo_step(ctypes.byref(a), None) == -22 and b"FROM_VERSIONS without versions" in lib.rlx_last_error()

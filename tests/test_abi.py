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
    assert lib.rlx_version() >= 100
    assert isinstance(lib.rlx_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    # NULL pointers / bad sizes are rejected before any HIP call, so this is safe on a CPU-only box.
    from rlinf_amd._lib import GaeParams
    p = GaeParams(0.99, 0.94, 1, 0, 1e-5, 0)
    rc = lib.rlx_gae_scan(None, None, None, None, None, None, None, 0, 4, 4, 1, ctypes.byref(p), None)
    assert rc == -22
    assert b"NULL" in lib.rlx_last_error()
    assert lib.rlx_gae_scan(None, None, None, None, None, None, None, 0, 0, 4, 1, ctypes.byref(p), None) == 0  # empty
    assert lib.rlx_gae_workspace_bytes(128, 1024, 1) == 16 * 5 * 8
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

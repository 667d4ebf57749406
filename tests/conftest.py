import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def free_port() -> int:
    """A free TCP port on 127.0.0.1 (one-rank gloo groups of the reference-side tests; fixed ports collide under pytest -n)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "reference: needs the reference tree under /root/reference")


def pytest_collection_modifyitems(config, items):
    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)


def dev_variants_build() -> bool:
    """True when the loaded librlx_hip was compiled with -DRLX_DEV_VARIANTS (RLX_LIB_TAG selects such a build): only then do the
    refuted / experimental kernel variants exist and are the RLX_* variant switches read.  The product build runs the measured-best
    path only; tests of the variants skip there."""
    from rlinf_amd import _lib
    return bool(_lib.load().rlx_dev_variants())


def need_dev_variants(what: str):
    if not dev_variants_build():
        pytest.skip(f"{what}: compiled into development builds only (-DRLX_DEV_VARIANTS; the product ships the measured-best path)")


@pytest.fixture(scope="session")
def ref():
    from oracle import reference_loader

    if not reference_loader.available():
        pytest.skip("reference tree not present on this machine")
    return reference_loader.load()


def synth_rollout(seed=1234, T=16, B=32, C=1, A=8, D=42, p_done=0.02, device="cpu"):
    """Seeded ManiSkill-shaped buffers: rewards~U(0,1), values~N(0,1) with T+1 rows,
    dones~Bernoulli(p) with T+1 rows and row 0 False."""
    g = torch.Generator().manual_seed(seed)
    rewards = torch.rand(T, B, C, generator=g)
    values = torch.randn(T + 1, B, C, generator=g)
    dones = torch.rand(T + 1, B, C, generator=g) < p_done
    dones[0] = False
    out = dict(rewards=rewards, values=values, dones=dones)
    return {k: v.to(device) for k, v in out.items()}

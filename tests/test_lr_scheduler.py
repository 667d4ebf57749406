"""The learner's LR schedule (host arithmetic) against the reference's own get_lr_scheduler, compiled from its source
where it lies, for every scheduler name the reference supports and the config keys build_lr_scheduler reads."""

import math

import pytest
import torch

from rlinf_amd.config import DictConfig
from rlinf_amd.utils.lr_scheduler import LearnerLRScheduler

CASES = [dict(lr_scheduler="constant"), dict(lr_scheduler="constant", lr_warmup_steps=4),
         dict(lr_scheduler="cosine", lr_warmup_steps=3, total_training_steps=20, min_lr=1e-5),
         dict(lr_scheduler="cosine", lr_warmup_steps_ratio=0.1, total_training_steps=30, min_lr_rate=0.1, min_lr=5.0),
         dict(lr_scheduler="openpi_cosine", lr_warmup_steps=5, total_training_steps=25, min_lr=3e-5),
         dict(lr_scheduler="ref_warmup_cosine", lr_warmup_steps=2, total_training_steps=8, min_lr_rate=0.2),
         dict(lr_scheduler="torch_constant"), dict(lr_scheduler="torch_cosine", total_training_steps=12), dict()]


@pytest.mark.reference
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()) or "defaults")
def test_schedule_matches_reference(case):
    from oracle import reference_loader
    if not reference_loader.available():
        pytest.skip("reference tree not present")
    pytest.importorskip("transformers")
    get_lr_scheduler = reference_loader.load_function("rlinf/hybrid_engines/fsdp/utils.py", "get_lr_scheduler", math=math, torch=torch)
    base = [3e-4, 1e-3]
    opt = torch.optim.AdamW([{"params": [torch.nn.Parameter(torch.zeros(1))], "lr": lr} for lr in base], eps=1e-8, weight_decay=0.01)
    # build_lr_scheduler's reading of the config (fsdp_model_manager.py:479-498)
    total = case.get("total_training_steps", 0)
    warm = int(case.get("lr_warmup_steps", -1))
    if warm < 0:
        warm = int(case.get("lr_warmup_steps_ratio", 0.0) * total)
    ref = get_lr_scheduler(lr_scheduler=case.get("lr_scheduler", "constant"), optimizer=opt, num_warmup_steps=warm,
                           num_training_steps=total, num_cycles=case.get("num_cycles", 0.5), min_lr=case.get("min_lr", 0.0),
                           min_lr_rate=case.get("min_lr_rate"), last_epoch=-1)
    ours = LearnerLRScheduler(DictConfig(dict(lr=3e-4, value_lr=1e-3, **case)), base)
    for _ in range(max(total, 10) + 3):
        assert ours.get_last_lr() == [float(x) for x in ref.get_last_lr()]
        opt.step(), ref.step(), ours.step()


def test_static_detection_and_errors():
    mk = lambda **kw: LearnerLRScheduler(DictConfig(kw), [1e-3])  # noqa: E731
    assert mk().is_static and mk(lr_scheduler="torch_constant").is_static
    assert not mk(lr_warmup_steps=3).is_static and not mk(lr_scheduler="torch_cosine", total_training_steps=5).is_static
    assert mk(lr_warmup_steps=2).get_last_lr() == [0.0]  # a warm-up starts from zero, as in the reference
    with pytest.raises(NotImplementedError, match="not supported"):
        mk(lr_scheduler="linear")

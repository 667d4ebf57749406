"""Learning-rate schedules of the FSDP learner: ``build_lr_scheduler`` (rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:
464-498) over ``get_lr_scheduler`` (rlinf/hybrid_engines/fsdp/utils.py:522-606), same config keys and defaults.

The optimizer here is a HIP kernel that takes its per-range learning rates as arguments, so there is no torch optimizer to
attach a scheduler to; the schedule itself is host arithmetic once per ``run_training`` (the reference steps it there too,
embodied_fsdp_actor_worker.py:568).  To produce the very same sequence the very same torch scheduler classes are driven
over a parameter-less stand-in optimizer with one group per learning rate; ``cosine`` restates transformers'
``get_cosine_with_min_lr_schedule_with_warmup`` multiplier (the reference imports it) so that transformers is not needed.
"""

from __future__ import annotations

import math

import torch
from torch.optim.lr_scheduler import ConstantLR, CosineAnnealingLR, LambdaLR


class LearnerLRScheduler:
    def __init__(self, optim_config, base_lrs: list, last_epoch: int = -1):
        get = optim_config.get
        total_steps = get("total_training_steps", 0) or 0
        warmup = int(get("lr_warmup_steps", -1) if get("lr_warmup_steps", -1) is not None else -1)
        name = get("lr_scheduler", "constant") or "constant"
        num_cycles = get("num_cycles", 0.5)
        min_lr, min_lr_rate = get("min_lr", 0.0), get("min_lr_rate", None)
        if min_lr is None:
            min_lr = 0.0
        if warmup < 0:
            warmup = int((get("lr_warmup_steps_ratio", 0.0) or 0.0) * total_steps)
        self.name, self.num_warmup_steps = name, warmup
        # AdamW without an explicit lr: defaults["lr"] = 1e-3, which is what transformers divides min_lr by
        self._opt = torch.optim.AdamW([{"params": [torch.nn.Parameter(torch.zeros(0))], "lr": float(lr)} for lr in base_lrs])
        if min_lr_rate is not None:  # "If min_lr_rate is set, min_lr will be ignored" (utils.py:532-534)
            min_lr = None
        if name == "constant":
            self._sched = LambdaLR(self._opt, lambda s: float(s) / float(max(1.0, warmup)) if s < warmup else 1.0,
                                   last_epoch=last_epoch)
        elif name == "cosine":
            rate = min_lr_rate if min_lr_rate is not None else min_lr / self._opt.defaults["lr"]

            def cosine(step):
                if step < warmup:
                    return float(step) / float(max(1, warmup))
                progress = float(step - warmup) / float(max(1, total_steps - warmup))
                factor = 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))
                return max(0, factor * (1 - rate) + rate)

            self._sched = LambdaLR(self._opt, cosine, last_epoch=last_epoch)
        elif name in ("openpi_cosine", "ref_warmup_cosine"):
            base = float(base_lrs[0])
            min_mult = min_lr_rate if min_lr_rate is not None else (min_lr / base if (min_lr and base > 0) else 0.0)

            def warmup_cosine(step):
                if step < warmup:
                    init = 1.0 / (warmup + 1)
                    return init + (1.0 - init) * step / max(1, warmup)
                progress = min(1.0, (step - warmup) / max(1, total_steps - warmup))
                return min_mult + (1.0 - min_mult) * 0.5 * (1.0 + math.cos(math.pi * progress))

            self._sched = LambdaLR(self._opt, warmup_cosine, last_epoch=last_epoch)
        elif name == "torch_constant":
            self._sched = ConstantLR(self._opt, factor=1)
        elif name == "torch_cosine":
            self._sched = CosineAnnealingLR(self._opt, T_max=total_steps, eta_min=1e-6)
        else:
            raise NotImplementedError(f"Scheduler type {name} is not supported")

    @property
    def is_static(self) -> bool:
        """The learning rates never change: captured graphs and prepared launch plans stay valid across iterations."""
        return self.name == "torch_constant" or (self.name == "constant" and self.num_warmup_steps == 0)

    def get_last_lr(self) -> list:
        return [float(x) for x in self._sched.get_last_lr()]

    def step(self) -> None:
        self._opt.step()  # (keeps torch's "scheduler before optimizer" check quiet; there is nothing to update)
        self._sched.step()

    def state_dict(self) -> dict:
        return {"scheduler": self._sched.state_dict(), "lrs": [g["lr"] for g in self._opt.param_groups]}

    def load_state_dict(self, state: dict) -> None:
        self._sched.load_state_dict(state["scheduler"])
        for g, lr in zip(self._opt.param_groups, state["lrs"]):
            g["lr"] = lr

"""Mirror of ``rlinf.algorithms`` for the rollout + PPO/GRPO hot path (registry.py, advantages.py,
losses.py, utils.py), dispatching to the HIP kernels behind include/rlx.h."""

from . import advantages, losses  # noqa: F401  (registers the built-ins)
from .registry import (  # noqa: F401
    ADV_REGISTRY,
    LOSS_REGISTRY,
    calculate_adv_and_returns,
    get_adv_and_returns,
    get_policy_loss,
    policy_loss,
    register_advantage,
    register_policy_loss,
)

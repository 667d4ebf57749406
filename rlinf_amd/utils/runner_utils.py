"""check_progress (rlinf/utils/runner_utils.py:31-63): when the runner validates and checkpoints.  Integer logic."""

from __future__ import annotations


def safe_is_divisible(a: int, b) -> bool:
    return b is not None and b > 0 and a % b == 0


def check_progress(step: int, max_steps: int, val_check_interval: int, save_interval: int, limit_val_batches=1.0,
                   run_time_exceeded: bool = False):
    """-> (run_val, save_model, is_train_end): validation / saving every ``interval`` steps (when > 0) and on the last step."""
    is_validation_enabled = limit_val_batches != 0 and val_check_interval is not None and val_check_interval > 0
    is_save_enabled = save_interval is not None and save_interval > 0
    is_train_end = step == max_steps
    if is_validation_enabled:
        assert save_interval is None or save_interval < 0 or save_interval % val_check_interval == 0, (
            f"{save_interval=} must be divisible by {val_check_interval=}")
    run_val = (safe_is_divisible(step, val_check_interval) or is_train_end or run_time_exceeded) and is_validation_enabled
    save_model = (safe_is_divisible(step, save_interval) or is_train_end or run_time_exceeded) and is_save_enabled
    return bool(run_val), bool(save_model), is_train_end

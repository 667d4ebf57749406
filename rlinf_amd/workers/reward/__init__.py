"""Reward-model workers (rlinf/workers/reward): out of scope for this path (the embodied PPO loop takes its rewards
from the env).  The names exist because the reference's entry point imports them unconditionally; constructing one raises."""


class _Unavailable:
    def __init__(self, *a, **kw):
        raise NotImplementedError(f"{type(self).__name__}: reward-model workers are not part of the MI355X hot path build "
                                  "(reward.use_reward_model must stay False)")

    @classmethod
    def create_group(cls, cfg, *a, **kw):
        raise NotImplementedError(f"{cls.__name__}: reward-model workers are not part of the MI355X hot path build")


class EmbodiedRewardWorker(_Unavailable):
    pass


class EmbodiedAPIRewardWorker(_Unavailable):
    pass

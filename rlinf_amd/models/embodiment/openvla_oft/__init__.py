"""The discrete action head of OpenVLA-OFT (the K2 categorical rollout path).  The transformer
itself is a model backend and out of scope; what is mirrored is everything after its logits."""

from .discrete_action_head import DiscreteActionHead  # noqa: F401

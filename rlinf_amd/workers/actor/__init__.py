from .embodied_fsdp_actor_worker import EmbodiedFSDPActor  # noqa: F401

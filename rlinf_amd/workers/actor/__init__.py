from .embodied_fsdp_actor_worker import EmbodiedFSDPActor  # noqa: F401
from .async_ppo_fsdp_worker import AsyncPPOEmbodiedFSDPActor  # noqa: F401,E402

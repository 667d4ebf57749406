from .embodied_runner import EmbodiedRunner  # noqa: F401

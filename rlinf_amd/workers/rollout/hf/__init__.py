from .huggingface_worker import MultiStepRolloutWorker  # noqa: F401

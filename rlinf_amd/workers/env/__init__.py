from .env_worker import EnvWorker  # noqa: F401

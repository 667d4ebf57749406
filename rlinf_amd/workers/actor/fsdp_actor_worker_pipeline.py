"""PipelineEmbodiedFSDPActor (rlinf/workers/actor/fsdp_actor_worker_pipeline.py:35-196): the learner of
``runner.use_training_pipeline``.  In the reference it pulls micro-batches from its channel while the env workers are still
producing them; with rollout and learner sharing one resident trajectory buffer the arrival order is the buffer order, and what
the mode changes is the data path -- advantages normalised from summed (count, sum, sumsq) statistics, per-stage stateful
shuffles, fixed global batches in epoch-major order -- which EmbodiedFSDPActor implements behind the same switch.  The class
exists so that the reference's entry point (`from rlinf.workers.actor.fsdp_actor_worker_pipeline import
PipelineEmbodiedFSDPActor`) selects it by name."""

from .embodied_fsdp_actor_worker import EmbodiedFSDPActor


class PipelineEmbodiedFSDPActor(EmbodiedFSDPActor):
    def __init__(self, cfg, ctx=None):
        assert bool(cfg.runner.get("use_training_pipeline", False)), "PipelineEmbodiedFSDPActor needs runner.use_training_pipeline"
        super().__init__(cfg, ctx)

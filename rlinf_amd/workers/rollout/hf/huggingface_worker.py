"""MultiStepRolloutWorker: batched obs-preprocess -> policy forward -> Gaussian sample, one fused HIP launch per
step (mirror of rlinf/workers/rollout/hf/huggingface_worker.py: predict :469-554, _build_policy_output :579-627,
get_bootstrap_values :612-627, sync_model_from_actor :629-675, generate_one_epoch :677-800)."""

from __future__ import annotations

import torch

from .... import ops
from ....models import get_model
from ...common import Worker, peer


class MultiStepRolloutWorker(Worker):
    ROLE = "rollout"

    def __init__(self, cfg, ctx=None):
        super().__init__(cfg, ctx)
        self.hf_model = None
        self.version = 0
        self._shares_actor_weights = False
        self._pending: list = []  # bootstrap-value jobs waiting for the next launch
        # cfg.weight_syncer (huggingface_worker.py:118-126): used whenever this worker holds its OWN copy of the weights -- a split
        # placement, or a collocated one with rollout.share_actor_weights off
        from ...weight_link import weight_syncer_config
        from ....hybrid_engines.weight_syncer import WeightSyncer
        ws_cfg = weight_syncer_config(cfg)
        self.weight_syncer = WeightSyncer.create(ws_cfg) if ws_cfg is not None else None
        self._weight_link = None

    def init_worker(self, model=None):
        """``model``: when rollout and learner are collocated (component_placement ``env,rollout,actor: 0``) the
        rollout worker can alias the learner's policy object -- weight sync becomes a no-op.  Called the
        reference's way (no argument, BEFORE the actor is initialised, embodied_runner.py:163-170) the decision is deferred to
        the first ``sync_model_from_actor()``: alias the in-process learner's model unless ``rollout.share_actor_weights`` is
        off (a rollout that must keep its weights frozen while the learner updates, e.g. the overlapped pipeline)."""
        if self._overlapped_pipeline():  # the rollout of epoch e + 1 runs WHILE the learner updates: it needs its own frozen copy
            self.cfg.rollout.share_actor_weights = False
            model = None
        if self._split():
            assert self.weight_syncer is not None, "rollout.weight_syncer config must be provided"  # huggingface_worker.py:122-124
            self.hf_model = self._own_model()
            return
        if model is not None:
            self.hf_model, self._shares_actor_weights = model, True
        elif not (bool(self.cfg.rollout.get("share_actor_weights", True)) and peer("actor", self.cfg) is not None):
            self.hf_model = self._own_model()

    def _split(self) -> bool:
        placement = getattr(self, "placement", None)
        return placement is not None and placement.split

    def _own_model(self):
        """This worker's own policy object, built from the learner's seed (embodied_fsdp_actor_worker.py: every rank builds
        identical initial weights) so that a patch syncer without an init sync starts from equal states on both sides."""
        torch.manual_seed(int(self.cfg.actor.get("seed", 1234)))
        return get_model(self.cfg.actor.model).to(self.device)

    def _overlapped_pipeline(self) -> bool:
        r = self.cfg.runner
        return (bool(r.get("use_training_pipeline", False)) and self.cfg.env.train.get("rollout_epoch", 1) > 1
                and bool(r.get("pipeline_overlap", False)))  # default off since round 5: see workers/env/env_worker.py

    def adopt_model(self, model):
        """Alias the collocated learner's policy object."""
        self.hf_model, self._shares_actor_weights = model, True

    def set_global_step(self, step: int):
        self.version = step

    def sync_model_from_actor(self, flat_params: torch.Tensor | None = None):
        """Apply the learner's weights (huggingface_worker.py:629-675).  Flat-buffer copy, or nothing when aliased; then
        rebuild the fragment-tile weight image HERE, eagerly: the rollout loop may be a replayed hipGraph, which must
        find fresh tiles in the same buffer (a lazy rebuild inside the captured region would be frozen out of it)."""
        if self._split():
            from ...weight_link import GroupLink
            if self._weight_link is None:
                self._weight_link = GroupLink(self.placement, self.device)
            self._receive_weights(self._weight_link)
            return
        actor = peer("actor", self.cfg)
        if self.hf_model is None:  # deferred by init_worker(): collocated -> alias the learner's policy object
            if actor is None or actor.model is None:
                raise RuntimeError("sync_model_from_actor: no initialised in-process actor to take the weights from")
            if bool(self.cfg.rollout.get("share_actor_weights", True)):
                self.adopt_model(actor.model)
            else:
                self.hf_model = self._own_model()
        own_copy = not self._shares_actor_weights
        if (own_copy and flat_params is None and self.weight_syncer is not None and actor is not None and actor.model is not None
                and getattr(actor, "weight_syncer", None) is not None):
            # an own copy next to the learner in one process: the configured syncer moves the weights, both halves driven from
            # here over an in-process link (the learner's sync_model_to_rollout, called next by the runner, has nothing left to do)
            from ...weight_link import InProcessLink
            if self._weight_link is None:
                self._weight_link = InProcessLink()
            link = self._weight_link
            link.run(lambda: actor.serve_weight_sync(link), lambda: self._receive_weights(link))
            return
        if flat_params is None and own_copy and actor is not None and actor.model is not None:
            flat_params = actor.model.flat.data  # the reference's call carries no argument: the weights come over its channel
        if not (self._shares_actor_weights or flat_params is None):
            with torch.no_grad():
                self.hf_model.flat.data.copy_(flat_params)
        self._weights_landed()

    def _receive_weights(self, link) -> None:
        """The receiver half (huggingface_worker.py:657-672): ``init_receiver`` once, ``apply``, the applied version becomes this
        worker's version."""
        syncer = self.weight_syncer
        if not syncer.receiver_initialized():
            syncer.init_receiver(state_dict=self.hf_model.state_dict(), recv=link.rollout_recv, send=link.rollout_send)
        self.version = int(syncer.apply(self.hf_model, link.rollout_recv))
        self._weights_landed()

    def _weights_landed(self):
        self.hf_model.mark_updated()
        if self.hf_model.flat.is_cuda:
            self.hf_model.tiles()

    def predict(self, env_obs: dict, out=None, eps=None, mode: str = "train", states_copy=None):
        """-> chunk_actions [B, C, A]; action / logprob / value rows land in ``out`` (the trajectory buffer).  ONE launch:
        bootstrap-value jobs queued by queue_bootstrap() ride in the same grid."""
        m = self.hf_model
        states = env_obs["states"]
        if mode == "train" and eps is None:
            eps = torch.randn((states.shape[0], m.layout.act_dim), dtype=torch.float32, device=states.device)
        jobs, self._pending = tuple(self._pending), []
        action, _, _ = ops.mlp_rollout_step(m.flat.data, m.tiles(), m.layout, states, eps if mode == "train" else None, out=out,
                                            states_copy=states_copy, value_jobs=jobs)
        return action.view(-1, m.num_action_chunks, m.action_dim)

    def queue_bootstrap(self, final_obs: dict, rewards: torch.Tensor, flags, gamma: float, env=None, rows=None,
                        flag_is_truncation: bool = False):
        """get_bootstrap_values + compute_bootstrap_rewards (huggingface_worker.py:612-627, env_worker.py:718-758)
        deferred into the next launch: rewards[:, -1] += gamma * V(final_obs)[:, 0] where flags[:, -1].  The weights do
        not change inside a rollout epoch, so running it one launch later changes nothing but the launch count.
        ``env`` = (rewards, terminations, truncations) of the env step and ``rows`` = (done, termination, truncation)
        buffer rows: the same job then also stores the step's env outputs (``rewards`` is the destination row)."""
        if len(self._pending) == 2:
            self.flush_bootstrap()
        job = dict(states=final_obs["states"], rewards=rewards, flags=flags, gamma=gamma)
        if env is not None:
            job.update(env=env, rows=rows, flag_is_truncation=flag_is_truncation)
        self._pending.append(job)

    def flush_bootstrap(self):
        if self._pending:
            jobs, self._pending = tuple(self._pending), []
            m = self.hf_model
            ops.mlp_rollout_step(m.flat.data, m.tiles(), m.layout, None, None, value_jobs=jobs)

    def get_bootstrap_values(self, final_obs: dict, out=None) -> torch.Tensor:
        """Value head only (huggingface_worker.py:612-627); carries at most one queued bootstrap job along."""
        m = self.hf_model
        st = final_obs["states"]
        if out is None:
            out = torch.empty((st.shape[0], m.layout.val_dim), dtype=torch.float32, device=st.device)
        if len(self._pending) == 2:
            self.flush_bootstrap()
        jobs, self._pending = tuple(self._pending) + (dict(states=st, values=out),), []
        ops.mlp_rollout_step(m.flat.data, m.tiles(), m.layout, None, None, value_jobs=jobs)
        return out[:, :1]

    def generate(self, input_channel=None, output_channel=None, **kwargs):
        """The reference runs this concurrently with EnvWorker.interact over channels (huggingface_worker.py:677-800);
        in-process the env worker calls predict() directly, so there is nothing left to do here."""
        return None

    def evaluate(self, input_channel=None, output_channel=None, **kwargs):
        """Eval-mode counterpart of generate (huggingface_worker.py: the policy acts with its mean, mlp_policy.py:230-236):
        EnvWorker.evaluate drives predict(mode="eval") directly."""
        return None

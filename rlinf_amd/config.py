"""Configuration surface: ``validate_cfg`` (mirror of rlinf/config.py:1455-1566 for the embodied + FSDP
branches, :905-1122 and :467-566) on a small self-contained Hydra/OmegaConf-compatible loader.

hydra-core and omegaconf are not available in this image, so ``load_config`` implements the subset the
reference's embodied example configs use: a ``defaults:`` list with ``group/name@target.path`` entries resolved
against search paths, ``${a.b}`` absolute and ``${..a}`` relative interpolation, ``${oc.env:VAR}``, and
``key=value`` dotted overrides.  ``DictConfig`` gives attribute + ``.get`` access like OmegaConf's.
"""

from __future__ import annotations

import copy
import os
import re
from typing import Any, Iterable, Optional

import yaml

SUPPORTED_EMBODIED_MODELS = {"mlp_policy"}
_INTERP = re.compile(r"\$\{([^${}]+)\}")


class DictConfig(dict):
    """dict with attribute access; nested dicts are wrapped on the way in."""

    def __init__(self, data: Optional[dict] = None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, DictConfig):
            return DictConfig(v)
        if isinstance(v, list):
            return [DictConfig._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(f"Missing key {k}") from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self and self[k] is not None else default

    def to_container(self) -> dict:
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, list):
                return [un(x) for x in v]
            return v
        return un(self)

    def __deepcopy__(self, memo):
        return DictConfig(copy.deepcopy(self.to_container(), memo))


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(root: dict, path: str, value: Any):
    keys = [k for k in path.split(".") if k]
    cur = root
    for k in keys[:-1]:
        if not isinstance(cur.get(k), dict):
            cur[k] = {}
        cur = cur[k]
    if keys:
        if isinstance(value, dict) and isinstance(cur.get(keys[-1]), dict):
            _merge(cur[keys[-1]], value)
        else:
            cur[keys[-1]] = value
    else:
        _merge(root, value)


def _get_path(root: dict, keys: list):
    cur = root
    for k in keys:
        cur = cur[k]
    return cur


def _find(name: str, search: Iterable[str]) -> str:
    for d in search:
        for ext in (".yaml", ".yml"):
            p = os.path.join(d, name + ext)
            if os.path.isfile(p):
                return p
    raise FileNotFoundError(f"config '{name}' not found in {list(search)}")


def _load_with_defaults(path: str, search: list) -> dict:
    raw = yaml.safe_load(open(path)) or {}
    defaults = raw.pop("defaults", []) or []
    raw.pop("hydra", None)
    out: dict = {}
    own_merged = False
    for entry in defaults:
        if entry == "_self_":
            _merge(out, raw)
            own_merged = True
            continue
        if isinstance(entry, dict):  # "override hydra/job_logging: stdout" and friends
            (k, v), = entry.items()
            if k.startswith("override ") or k.startswith("hydra"):
                continue
            entry = f"{k}/{v}" if "@" not in k else f"{k.split('@')[0]}/{v}@{k.split('@')[1]}"
        spec, _, target = str(entry).partition("@")
        sub = _load_with_defaults(_find(spec, search), search)
        _set_path(out, target if target else spec.rsplit("/", 1)[0].replace("/", "."), sub)
    if not own_merged:
        _merge(out, raw)
    return out


def _resolve(root: dict, node: Any, here: list):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve(root, node[k], here + [k])
        return node
    if isinstance(node, list):
        return [_resolve(root, v, here) for v in node]
    if not isinstance(node, str) or "${" not in node:
        return node

    def lookup(expr: str):
        expr = expr.strip()
        if expr.startswith("oc.env:"):
            name, _, default = expr[len("oc.env:"):].partition(",")
            return os.environ.get(name.strip(), default.strip() or None)
        if expr.startswith("."):  # relative: one dot = sibling of the current key
            up = len(expr) - len(expr.lstrip("."))
            base = here[:-up] if up <= len(here) else []
            keys = base + [k for k in expr.lstrip(".").split(".") if k]
        else:
            keys = expr.split(".")
        try:
            return _resolve(root, _get_path(root, keys), keys)
        except (KeyError, TypeError):
            return "${" + expr + "}"  # leave unresolved (e.g. hydra-only resolvers)

    m = _INTERP.fullmatch(node)
    if m:
        return lookup(m.group(1))
    return _INTERP.sub(lambda mm: str(lookup(mm.group(1))), node)


def _parse_scalar(text: str):
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def load_config(source, overrides: Optional[Iterable[str]] = None, search_paths: Optional[Iterable[str]] = None) -> DictConfig:
    """source: a YAML path, a config name resolvable in search_paths, or a dict."""
    search = list(search_paths or [])
    if isinstance(source, dict):
        data = copy.deepcopy(dict(source))
    else:
        path = source if os.path.isfile(str(source)) else _find(str(source), search)
        search = [os.path.dirname(os.path.abspath(path))] + search
        data = _load_with_defaults(path, search)
    for ov in overrides or []:
        k, _, v = ov.partition("=")
        _set_path(data, k.lstrip("+"), _parse_scalar(v))
    return DictConfig(_resolve(data, data, []))


# ------------------------------------------------------------------------------------------------------------
# validate_cfg
# ------------------------------------------------------------------------------------------------------------
def _placement_world_size(cfg, component: str) -> int:
    """Number of processes of a component.  The reference derives it from cluster.component_placement
    (rlinf/utils/placement.py:86); here every component runs in every rank of the torchrun-style world
    (collocated placement, the shipped `env,rollout,actor: 0` pattern scaled to N GPUs)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    placement = (cfg.get("cluster", None) or {}).get("component_placement", None) if hasattr(cfg, "get") else None
    if placement is not None and world > 1:
        # a SPLIT placement (utils/placement.py: actor and rollout rank sets disjoint) gives a component its own ranks only
        from .utils.placement import parse_component_placement
        ranks = parse_component_placement(placement, world)
        a, r = ranks.get("actor"), ranks.get("rollout")
        if a is not None and r is not None and not (set(a) & set(r)) and ranks.get(component) is not None:
            return len(ranks[component])
    return world


def validate_fsdp_cfg(cfg):
    """rlinf/config.py:467-566, the knobs the NO_SHARD data-parallel slice uses."""
    fsdp = cfg.get("fsdp_config", None)
    if fsdp is None:
        cfg.fsdp_config = DictConfig({})
        fsdp = cfg.fsdp_config
    fsdp.strategy = fsdp.get("strategy", "fsdp")
    fsdp.sharding_strategy = fsdp.get("sharding_strategy", "no_shard")
    assert fsdp.sharding_strategy == "no_shard", (
        "only sharding_strategy='no_shard' (pure data parallel) is served by this build; "
        f"got {fsdp.sharding_strategy!r}")
    optim = cfg.get("optim", None)
    assert optim is not None, "actor.optim is required"
    optim.lr = float(optim.get("lr", 3e-4))
    optim.value_lr = float(optim.get("value_lr", optim.lr))
    optim.adam_beta1 = float(optim.get("adam_beta1", 0.9))
    optim.adam_beta2 = float(optim.get("adam_beta2", 0.999))
    optim.adam_eps = float(optim.get("adam_eps", 1e-8))
    optim.weight_decay = float(optim.get("weight_decay", 1e-2))
    optim.clip_grad = float(optim.get("clip_grad", 1.0))
    optim.critic_warmup_steps = int(optim.get("critic_warmup_steps", 0))
    return cfg


def validate_embodied_cfg(cfg):
    """rlinf/config.py:905-1122: defaults + divisibility asserts for the embodied runner."""
    model_cfg = cfg.actor.model
    assert model_cfg.model_type in SUPPORTED_EMBODIED_MODELS, (
        f"Model type: '{model_cfg.model_type}' is not supported by the embodied runner. "
        f"Supported embodied models: {sorted(SUPPORTED_EMBODIED_MODELS)}.")
    cfg.runner.val_check_interval = cfg.runner.get("val_check_interval", -1)
    cfg.env.train.rollout_epoch = cfg.env.train.get("rollout_epoch", 1)
    if cfg.algorithm.loss_type in ("actor_critic", "decoupled_actor_critic"):
        add_value_head = model_cfg.get("add_value_head", False)
        assert add_value_head, (
            "When using PPO algorithm (algorithm.loss_type='actor_critic'), actor.model.add_value_head must be True. "
            f"Current value: {add_value_head}")
    cfg.rollout.pipeline_stage_num = cfg.rollout.get("pipeline_stage_num", 1)
    stage_num = cfg.rollout.pipeline_stage_num
    env_world_size = _placement_world_size(cfg, "env")
    tr = cfg.env.train
    tr.group_size = tr.get("group_size", 1)
    assert tr.total_num_envs > 0, "Total number of parallel environments for training must be greater than 0"
    assert tr.total_num_envs % env_world_size == 0, (
        "Total number of parallel environments for training must be divisible by the number of environment processes")
    assert tr.total_num_envs // env_world_size % stage_num == 0, (
        "Total number of parallel environments for training must be divisible by the number of environment "
        "processes and the number of pipeline stages")
    assert tr.total_num_envs // env_world_size // stage_num > 0
    assert tr.total_num_envs // env_world_size // stage_num % tr.group_size == 0, (
        "env.train.total_num_envs // env_world_size // rollout.pipeline_stage_num must be divisible by the group size")
    assert tr.max_steps_per_rollout_epoch % model_cfg.num_action_chunks == 0, (
        "env.train.max_steps_per_rollout_epoch must be divisible by actor.model.num_action_chunks")
    cfg.runner.weight_sync_interval = cfg.runner.get("weight_sync_interval", 1)
    assert cfg.runner.weight_sync_interval > 0, "weight_sync_interval must be greater than 0"
    validate_mlp_kernel_shapes(model_cfg)
    return cfg


# Shape limits of the HIP kernels behind ``mlp_policy`` (this build's own check -- the reference's torch modules take any
# width): the rollout launch and the fused optimizer step keep a row tile's activations in ONE LDS slab of 256 columns and the
# first layer's weight tiles zero-padded to 64 inputs (csrc/ppo_step_common.h: HID = 256, Tiles::K1P = 64), and the head's
# outputs in 16 accumulator columns (MAX_OUT = 16).  All of the reference's MLP configurations fit (obs_dim <= 42, act 8).
MLP_KERNEL_LIMITS = {"hidden_dim": 256, "obs_dim_max": 64, "action_outputs_max": 16}


def validate_mlp_kernel_shapes(model_cfg) -> None:
    """Reject, with the limit named, a policy shape the kernels cannot take -- at configuration time instead of as an error
    string from the first launch."""
    lim = MLP_KERNEL_LIMITS
    hidden = model_cfg.get("hidden_dim", lim["hidden_dim"])
    assert int(hidden) == lim["hidden_dim"], (
        f"actor.model.hidden_dim={hidden}: the MI355X kernels of mlp_policy are built for the reference's hidden width "
        f"{lim['hidden_dim']} (3 x 256 tanh layers); other widths are not served by this build")
    obs = int(model_cfg.obs_dim)
    assert 1 <= obs <= lim["obs_dim_max"], (
        f"actor.model.obs_dim={obs}: the fused rollout / optimizer-step kernels take 1 <= obs_dim <= {lim['obs_dim_max']} "
        f"(first-layer weight tiles are padded to {lim['obs_dim_max']} inputs)")
    chunks = int(model_cfg.get("num_action_chunks", 1))
    outs = int(model_cfg.action_dim) * chunks
    assert 1 <= outs <= lim["action_outputs_max"], (
        f"actor.model.action_dim * num_action_chunks = {model_cfg.action_dim} * {chunks} = {outs}: the fused kernels take at most "
        f"{lim['action_outputs_max']} head outputs per policy step")


def _refuse_foreign_patch_codecs(node, path=""):
    """A weight-syncer ``patch`` section that names the reference's nvCOMP container (``compression_algorithm`` / ``compression``:
    rlinf/hybrid_engines/weight_syncer/base.py:112-119) is refused when the configuration is validated, not when the first
    patch is built: this build cannot produce that wire format (hybrid_engines/weight_syncer/compressor.py says what to use)."""
    if not isinstance(node, dict):
        return
    for key, val in node.items():
        here = f"{path}.{key}" if path else str(key)
        if key == "patch" and isinstance(val, dict):
            name = val.get("compression_algorithm", val.get("compression", "none"))
            from .hybrid_engines.weight_syncer.compressor import NVCOMP_ALGORITHMS
            if name in NVCOMP_ALGORITHMS and os.environ.get("RLX_NVCOMP_LZ4_AS_ZPLANE", "0") in ("", "0"):
                raise ValueError(f"{here}.compression_algorithm={name!r}: the reference's nvCOMP LZ4 container cannot be produced or read by "
                                 "this gfx950 build. Use 'none' (what a reference peer decodes) or 'rlx_zplane' (both ends run rlinf_amd); "
                                 "RLX_NVCOMP_LZ4_AS_ZPLANE=1 maps the name to 'rlx_zplane' for configuration files that must stay unchanged.")
        _refuse_foreign_patch_codecs(val, here)


def validate_cfg(cfg) -> DictConfig:
    """Fill defaults and assert consistency; returns the config (rlinf/config.py:1455-1566)."""
    if not isinstance(cfg, DictConfig):
        cfg = DictConfig(cfg)
    _refuse_foreign_patch_codecs(cfg)
    # rlinf/config.py:1458-1464: per-worker logging defaults the entry point hands to Cluster
    cfg.runner.per_worker_log = cfg.runner.get("per_worker_log", False)
    cfg.runner.per_worker_log_path = None
    if cfg.runner.per_worker_log:
        cfg.runner.per_worker_log_path = os.path.join(cfg.runner.logger.log_path, "worker_logs")
    if cfg.get("cluster", None) is None:  # configs built in code (tests, bench.py): the collocated single-node placement
        cfg.cluster = DictConfig({"num_nodes": 1, "component_placement": {"env,rollout,actor": "all"}})
    cfg.runner.save_interval = cfg.runner.get("save_interval", -1)
    task = cfg.runner.get("task_type", None)
    assert task in ("embodied",), f"task_type {task!r} is not served by this build (embodied only)"
    alg = cfg.algorithm
    alg.adv_type = alg.get("adv_type", "gae")
    alg.loss_type = alg.get("loss_type", "actor_critic")
    if alg.adv_type == "grpo":
        assert alg.get("group_size", 1) > 1, "group_size must be greater than 1 for grpo"  # config.py:1519-1521
    alg.normalize_advantages = alg.get("normalize_advantages", True)
    alg.update_epoch = alg.get("update_epoch", 1)
    alg.entropy_bonus = alg.get("entropy_bonus", 0)
    alg.reward_type = alg.get("reward_type", "action_level")
    alg.logprob_type = alg.get("logprob_type", "action_level")
    alg.entropy_type = alg.get("entropy_type", "action_level")
    alg.bootstrap_type = alg.get("bootstrap_type", "standard")
    cfg = validate_embodied_cfg(cfg)
    assert cfg.actor.training_backend == "fsdp", "only actor.training_backend='fsdp' (data parallel) is served"
    validate_fsdp_cfg(cfg.actor)
    world = _placement_world_size(cfg, "actor")
    assert cfg.actor.global_batch_size % (cfg.actor.micro_batch_size * world) == 0, (
        "actor.global_batch_size must be divisible by micro_batch_size * actor world size")
    cfg.actor.seed = cfg.actor.get("seed", 1234)
    return cfg

"""Load the REAL reference hot-path modules from ``/root/reference`` (build container) or from the staged copy ``oracle/_ref``
(GPU box: the pure-torch files only, see oracle/stage_reference.py).

TEST INFRASTRUCTURE.  The reference package itself is not importable here (omegaconf, ray,
hydra, gymnasium ... are absent), but the arithmetic files on the hot path are pure torch and
load by file path behind a handful of empty stub packages (recipe verified in SURVEY.md 8c).

Nothing is copied: the files are executed where they lie, read-only.
"""

from __future__ import annotations

import importlib.util
import os
import sys
import types
from types import SimpleNamespace

_STAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _pick_root() -> str:
    """/root/reference in the build container; on the GPU box the staged copy of the hot-path files (oracle/_ref, written by
    oracle/stage_reference.py, shipped by gpurun like the built .so)."""
    env = os.environ.get("RLX_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isfile("/root/reference/rlinf/algorithms/advantages.py"):
        return "/root/reference"
    return _STAGED


REFERENCE_ROOT = _pick_root()

_FILES_ALGO = [
    ("rlinf.utils.metric_utils", "rlinf/utils/metric_utils.py"),
    ("rlinf.utils.utils", "rlinf/utils/utils.py"),
    ("rlinf.algorithms.utils", "rlinf/algorithms/utils.py"),
    ("rlinf.algorithms.registry", "rlinf/algorithms/registry.py"),
    ("rlinf.algorithms.advantages", "rlinf/algorithms/advantages.py"),
    ("rlinf.algorithms.losses", "rlinf/algorithms/losses.py"),
    ("rlinf.utils.nested_dict_process", "rlinf/utils/nested_dict_process.py"),
]
_FILES_POLICY = [
    ("rlinf.models.embodiment.base_policy", "rlinf/models/embodiment/base_policy.py"),
    ("rlinf.models.embodiment.modules.utils", "rlinf/models/embodiment/modules/utils.py"),
    ("rlinf.models.embodiment.modules.value_head", "rlinf/models/embodiment/modules/value_head.py"),
    ("rlinf.models.embodiment.modules.batch_renorm", "rlinf/models/embodiment/modules/batch_renorm.py"),
    ("rlinf.models.embodiment.modules.q_head", "rlinf/models/embodiment/modules/q_head.py"),
    ("rlinf.models.embodiment.mlp_policy.mlp_policy", "rlinf/models/embodiment/mlp_policy/mlp_policy.py"),
]

# files whose modules cannot be imported here (Ray, FSDP, hydra, omegaconf): single functions / classes of them are compiled
# on their own (load_function / load_class / load_models_registry); staged so that the -m gpu tests can do the same on the GPU box
_FILES_SOURCE_ONLY = [
    "rlinf/config.py",
    "rlinf/models/__init__.py",
    "rlinf/workers/actor/embodied_fsdp_actor_worker.py",
    "rlinf/hybrid_engines/fsdp/fsdp_model_manager.py",
]

_cache: SimpleNamespace | None = None


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "rlinf/algorithms/advantages.py"))


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__path__ = []  # behave like a package
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def _exec(mod_name: str, rel_path: str) -> types.ModuleType:
    spec = importlib.util.spec_from_file_location(mod_name, os.path.join(REFERENCE_ROOT, rel_path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[mod_name] = mod
    spec.loader.exec_module(mod)
    return mod


def load() -> SimpleNamespace:
    """Return a namespace with the reference modules: .advantages .losses .algo_utils .registry
    .metric_utils .utils .nested .mlp_policy (each the real module object)."""
    global _cache
    if _cache is not None:
        return _cache
    if not available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    if "rlinf" in sys.modules and not getattr(sys.modules["rlinf"], "_rlx_stub", False):
        raise RuntimeError("a real `rlinf` package is already imported; refusing to shadow it")
    for pkg in (
        "rlinf",
        "rlinf.utils",
        "rlinf.algorithms",
        "rlinf.models",
        "rlinf.models.embodiment",
        "rlinf.models.embodiment.modules",
        "rlinf.models.embodiment.mlp_policy",
    ):
        _stub(pkg, _rlx_stub=True)
    # rlinf/utils/utils.py imports `Worker` from rlinf.scheduler at module scope only.
    _stub("rlinf.scheduler", Worker=type("Worker", (), {"torch_platform": None}), _rlx_stub=True)
    mods = {}
    for name, rel in _FILES_ALGO + _FILES_POLICY:
        mods[name] = _exec(name, rel)
    _cache = SimpleNamespace(
        metric_utils=mods["rlinf.utils.metric_utils"],
        utils=mods["rlinf.utils.utils"],
        algo_utils=mods["rlinf.algorithms.utils"],
        registry=mods["rlinf.algorithms.registry"],
        advantages=mods["rlinf.algorithms.advantages"],
        losses=mods["rlinf.algorithms.losses"],
        nested=mods["rlinf.utils.nested_dict_process"],
        mlp_policy=mods["rlinf.models.embodiment.mlp_policy.mlp_policy"],
        value_head=mods["rlinf.models.embodiment.modules.value_head"],
    )
    return _cache


_ws_cache = None


def load_weight_syncer():
    """The reference's patch weight syncer (rlinf/hybrid_engines/weight_syncer/patch_syncer.py and what it imports),
    executed from /root/reference.  ``Worker.torch_device_type`` is stubbed to "cpu" so that the reference's
    GPUSnapshotPatchBuilder -- the same-device snapshot path the HIP kernels replace -- runs on CPU tensors here."""
    global _ws_cache
    if _ws_cache is not None:
        return _ws_cache
    import logging

    load()
    sched = sys.modules["rlinf.scheduler"]
    sched.CollectiveGroupOptions = type("CollectiveGroupOptions", (), {})

    class _Stream:
        def synchronize(self):
            pass

    class _Platform:
        @staticmethod
        def current_stream(*_a):
            return _Stream()

        @staticmethod
        def is_initialized():
            return True

    sched.Worker.torch_device_type = "cpu"
    sched.Worker.torch_platform = _Platform
    _stub("rlinf.hybrid_engines", _rlx_stub=True)
    _stub("rlinf.hybrid_engines.weight_syncer", _rlx_stub=True)
    if "omegaconf" not in sys.modules:
        _stub("omegaconf", DictConfig=dict, OmegaConf=type("OmegaConf", (), {}), _rlx_stub=True)
    _stub("rlinf.utils.logging", get_logger=lambda *a, **k: logging.getLogger("rlinf-reference"), _rlx_stub=True)
    mods = {}
    for name in ("base", "compressor", "bucket_syncer", "patch_syncer"):
        mods[name] = _exec(f"rlinf.hybrid_engines.weight_syncer.{name}", f"rlinf/hybrid_engines/weight_syncer/{name}.py")
    _ws_cache = mods["patch_syncer"]
    return _ws_cache


_du_cache = None


def load_distributed_utils():
    """rlinf/utils/distributed.py (masked_stats / normalize_from_stats and friends), executed in place behind stubs for
    the scheduler's Tracer and the timers module."""
    global _du_cache
    if _du_cache is not None:
        return _du_cache
    load()
    sched = sys.modules["rlinf.scheduler"]
    if not hasattr(sched, "Tracer"):
        sched.Tracer = type("Tracer", (), {})
    if "rlinf.utils.timers" not in sys.modules:
        _stub("rlinf.utils.timers", NamedTimer=type("NamedTimer", (), {}), _rlx_stub=True)
    _du_cache = _exec("rlinf.utils.distributed", "rlinf/utils/distributed.py")
    return _du_cache


def load_function(rel_path: str, name: str, **globs):
    """One top-level function (or ``Class.method``) of a reference file whose module cannot be imported here (heavy third-party imports):
    its own source, parsed where it lies and compiled on its own.  ``globs`` supplies the names it refers to."""
    import ast

    path = os.path.join(REFERENCE_ROOT, rel_path)
    tree = ast.parse(open(path).read(), filename=path)
    body = tree.body
    *owners, name = name.split(".")  # "Class.method": the method as a plain function taking ``self``
    for owner in owners:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == owner).body
    node = next(n for n in body if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)) and n.name == name)
    node.returns, node.decorator_list = None, []
    for a in node.args.args + node.args.kwonlyargs:
        a.annotation = None
    ns = dict(globs)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def load_class(rel_path: str, name: str, **globs):
    """One top-level class of a reference file whose module cannot be imported here, compiled on its own from its source where it
    lies (signature annotations dropped, decorators of its methods kept).  ``globs`` supplies the names its bodies refer to."""
    import ast

    path = os.path.join(REFERENCE_ROOT, rel_path)
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    node.decorator_list = []
    for fn in ast.walk(node):
        if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)):
            fn.returns = None
            for a in fn.args.args + fn.args.kwonlyargs:
                a.annotation = None
    ns = dict(globs)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


_mr_cache = None


def load_models_registry():
    """The reference's MODEL REGISTRY, rlinf/models/__init__.py (register_model / get_model and the built-in registrations),
    executed as the module ``rlinf.models`` behind a ``rlinf.config`` that holds the REAL ``SupportedModel`` class with all its
    module-level registrations, ``EMBODIED_MODEL`` / ``DIFFUSION_MODELS`` and ``torch_dtype_from_precision`` -- those top-level
    statements of rlinf/config.py compiled on their own (the rest of that file needs omegaconf, Ray and the env packages)."""
    global _mr_cache
    if _mr_cache is not None:
        return _mr_cache
    import ast
    import dataclasses
    import typing

    import torch

    load()
    path = os.path.join(REFERENCE_ROOT, "rlinf/config.py")
    tree = ast.parse(open(path).read(), filename=path)
    wanted = ("SupportedModel", "DIFFUSION_MODELS", "EMBODIED_MODEL")
    keep = []
    for n in tree.body:
        if isinstance(n, ast.ClassDef) and n.name == "SupportedModel":
            keep.append(n)
        elif isinstance(n, ast.FunctionDef) and n.name == "torch_dtype_from_precision":
            n.returns = None
            for a in n.args.args:
                a.annotation = None
            keep.append(n)
        elif isinstance(n, ast.Assign) and any(isinstance(x, ast.Name) and x.id in wanted for t in n.targets for x in ast.walk(t)):
            keep.append(n)
    ns = {"dataclasses": dataclasses, "ClassVar": typing.ClassVar, "torch": torch}
    # dont_inherit: this file's own `from __future__ import annotations` would turn the dataclass's ClassVar annotation into a string
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec", dont_inherit=True), ns)
    _stub("rlinf.config", _rlx_stub=True, **{k: ns[k] for k in wanted + ("torch_dtype_from_precision",)})
    if "omegaconf" not in sys.modules:
        _stub("omegaconf", DictConfig=dict, OmegaConf=type("OmegaConf", (), {}), _rlx_stub=True)
    sched = sys.modules["rlinf.scheduler"]
    if not hasattr(sched.Worker, "torch_device_type"):
        sched.Worker.torch_device_type = "cpu"
    spec = importlib.util.spec_from_file_location("rlinf.models", os.path.join(REFERENCE_ROOT, "rlinf/models/__init__.py"),
                                                  submodule_search_locations=[])
    mod = importlib.util.module_from_spec(spec)
    mod._rlx_stub = True  # (still not a real install: load() may be called again)
    sys.modules["rlinf.models"] = mod
    sys.modules["rlinf"].models = mod
    spec.loader.exec_module(mod)
    _mr_cache = SimpleNamespace(models=mod, config=sys.modules["rlinf.config"])
    return _mr_cache


_tb_cache = None


def load_trajectory_builder():
    """rlinf/data/schema/embodied_types.py and embodied_trajectory_builder.py (torch / numpy only) behind stub packages."""
    global _tb_cache
    if _tb_cache is None:
        load()
        _stub("rlinf.data", _rlx_stub=True)
        _stub("rlinf.data.schema", _rlx_stub=True)
        types_mod = _exec("rlinf.data.schema.embodied_types", "rlinf/data/schema/embodied_types.py")
        builder_mod = _exec("rlinf.data.schema.embodied_trajectory_builder", "rlinf/data/schema/embodied_trajectory_builder.py")
        _tb_cache = SimpleNamespace(types=types_mod, builder=builder_mod)
    return _tb_cache


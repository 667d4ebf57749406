"""``rlinf`` import alias: ``import rlinf.workers.env.env_worker`` resolves to ``rlinf_amd.workers.env.env_worker`` (the very
same module objects), so code written against the reference's package layout -- its own
examples/embodiment/train_embodied_agent.py first of all -- runs on this package without a changed line.

    import rlinf_amd.compat; rlinf_amd.compat.install_alias()      # or: RLINF_AMD_ALIAS=1 with sitecustomize

hydra / omegaconf are not installed in this image; ``install_config_shims()`` provides the two entry points the reference's
script touches (``hydra.main`` as a pass-through decorator whose wrapped function takes the config, ``OmegaConf.to_container``,
``open_dict``) on top of rlinf_amd.config.DictConfig.  Nothing here is on the measured path."""

from __future__ import annotations

import contextlib
import importlib
import importlib.abc
import importlib.util
import sys
import types

ALIAS, TARGET = "rlinf", "rlinf_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != ALIAS and not fullname.startswith(ALIAS + "."):
            return None
        real = TARGET + fullname[len(ALIAS):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        spec = importlib.util.spec_from_loader(fullname, self, is_package=True)
        spec._rlx_real = real
        return spec

    def create_module(self, spec):
        return importlib.import_module(spec._rlx_real)  # the SAME module object under both names

    def exec_module(self, module):
        return None


def install_alias() -> None:
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())
    if ALIAS in sys.modules and getattr(sys.modules[ALIAS], "__name__", None) != TARGET:
        raise ImportError("a real `rlinf` package is already imported; the alias would shadow it")


def uninstall_alias() -> None:
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _AliasFinder)]
    for name in [n for n in sys.modules if n == ALIAS or n.startswith(ALIAS + ".")]:
        del sys.modules[name]


def install_config_shims() -> None:
    """Minimal ``hydra`` / ``omegaconf`` stand-ins (only when the real packages are absent)."""
    from .config import DictConfig

    if importlib.util.find_spec("omegaconf") is None and "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")

        class OmegaConf:
            @staticmethod
            def to_container(cfg, resolve: bool = True, **_):
                return cfg.to_container() if isinstance(cfg, DictConfig) else cfg

            @staticmethod
            def create(data=None):
                return DictConfig(data or {})

        @contextlib.contextmanager
        def open_dict(cfg):
            yield cfg

        oc.OmegaConf, oc.open_dict, oc.DictConfig = OmegaConf, open_dict, DictConfig
        sub = types.ModuleType("omegaconf.omegaconf")
        sub.OmegaConf, sub.open_dict, sub.DictConfig = OmegaConf, open_dict, DictConfig
        oc.omegaconf = sub
        sys.modules["omegaconf"], sys.modules["omegaconf.omegaconf"] = oc, sub
    if importlib.util.find_spec("hydra") is None and "hydra" not in sys.modules:
        hy = types.ModuleType("hydra")

        def main(version_base=None, config_path=None, config_name=None):
            def deco(fn):
                def run(cfg=None, *a, **kw):
                    if cfg is None:  # hydra would compose it from argv; here: rlinf_amd.config.load_config
                        raise RuntimeError("hydra is not installed: call main(cfg) with a config loaded by "
                                           "rlinf_amd.config.load_config")
                    return fn(cfg, *a, **kw)
                run.__wrapped__, run.config_path, run.config_name = fn, config_path, config_name
                return run
            return deco

        hy.main = main
        sys.modules["hydra"] = hy

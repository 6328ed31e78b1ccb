"""Make the reference's import paths resolve to this implementation.

tests/testbed.py imports `Tree.SpecTree`, `Tree.GreedyTree`, `Engine.Engine`,
`Engine.offload_engine` and `utils` by name (tests/testbed.py:14-18).  `install()` registers
those names in sys.modules, so the harness body runs unchanged:

    import sequoia_amd.dropin as dropin; dropin.install()
    from Tree.SpecTree import SpecTree                      # -> sequoia_amd.Tree.SpecTree
    from Engine.Engine import GraphInferenceEngine          # -> sequoia_amd.Engine.Engine
    from utils import cuda_graph_for_sampling_without_replacement
"""
from __future__ import annotations

import importlib
import sys

_ALIASES = {
    "Engine": "sequoia_amd.Engine",
    "Engine.Engine": "sequoia_amd.Engine.Engine",
    "Engine.Llama_KV": "sequoia_amd.Engine.Llama_KV",
    "Engine.Llama_model": "sequoia_amd.Engine.Llama_model",
    "Engine.Llama_modules": "sequoia_amd.Engine.Llama_modules",
    "Engine.offload_engine": "sequoia_amd.Engine.offload_engine",
    "Tree": "sequoia_amd.Tree",
    "Tree.Tree": "sequoia_amd.Tree.Tree",
    "Tree.SpecTree": "sequoia_amd.Tree.SpecTree",
    "Tree.GreedyTree": "sequoia_amd.Tree.GreedyTree",
    "Tree.SpecInferTree": "sequoia_amd.Tree.SpecInferTree",
    "Tree.GreedySTree": "sequoia_amd.Tree.GreedySTree",
    "utils": "sequoia_amd.utils",
}


def install(force: bool = False) -> None:
    for alias, real in _ALIASES.items():
        if alias in sys.modules and not force:
            mod = sys.modules[alias]
            if getattr(mod, "__name__", "").startswith("sequoia_amd"):
                continue
            raise ImportError(f"module '{alias}' is already imported from {getattr(mod, '__file__', '?')}; "
                              "call sequoia_amd.dropin.install() before importing the reference's modules")
        sys.modules[alias] = importlib.import_module(real)


def uninstall() -> None:
    for alias in _ALIASES:
        mod = sys.modules.get(alias)
        if mod is not None and getattr(mod, "__name__", "").startswith("sequoia_amd"):
            del sys.modules[alias]

"""Name -> class resolution used by the reference's session layer (util/dynamic_import.py:5-51).

``import_model("agcn")`` resolves ``fusion_gcn_amd.models.agcn.agcn.Model`` — the same
``models.<name>.<name>.Model`` convention the reference's ``Session._build_model`` relies on
(torch_src/session/session.py:50) — so the MI355X models drop in under ``torch_src/main.py``.
"""
from importlib import import_module
from typing import Sequence

_PKG = __name__.rsplit(".", 2)[0]  # "fusion_gcn_amd"


def import_names(module_path: str, names: Sequence[str]) -> list:
    mod = import_module(module_path)
    return [getattr(mod, n) for n in names if hasattr(mod, n)]


def import_class(name: str) -> type:
    module_path, _, class_name = name.rpartition(".")
    return import_names(module_path, [class_name])[0]


def import_model(name: str, class_name: str = "Model") -> type:
    name = name.lower()
    return import_names(f"{_PKG}.models.{name}.{name}", [class_name])[0]


def import_dataset_constants(dataset: str, names: Sequence[str]) -> list:
    dataset = dataset.lower().replace("-", "_")
    return import_names(f"{_PKG}.datasets.{dataset}.constants", names)

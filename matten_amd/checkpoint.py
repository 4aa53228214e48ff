"""
Reading a reference checkpoint (a Lightning ``.ckpt`` written by ``ScalarTensorModel`` of wengroup/matten) without
importing matten, e3nn, Lightning or torchmetrics -- none of them exists on the MI355X box.

What such a file holds (reference predict.py:36-46 -> ``load_from_checkpoint``; model/model.py:53-99):

* ``state_dict``: the parameters under the module names of ``create_model`` PLUS e3nn-internal
  entries this build has no counterpart for: the ``output_mask`` buffer of every tensor product / linear, the empty
  ``weight`` / ``bias`` placeholders e3nn registers when a module has no internal weights or biases, the Wigner-3j
  constants of the generated fx graphs (``..._compiled_main_*._w3j_*``) and, possibly, torchmetrics state under
  ``metrics.``.  ``filter_state_dict`` drops exactly those and nothing else.
* ``hyper_parameters``: plain dicts and, under ``tasks``, a pickled ``matten.model_factory.task.TensorRegressionTask``
  whose ``normalizer`` (if the model was trained on standardised targets) is a ``matten.data.transform`` module that
  holds e3nn ``Irreps`` objects.  ``load_checkpoint`` unpickles the file with a RESTRICTED unpickler: an explicit
  allow-list of torch / numpy / container globals resolves normally, every other global (matten.*, e3nn.*, torchmetrics.*, pytorch_lightning.*, ...)
  becomes an inert placeholder that records its constructor arguments and state and executes nothing.
  ``rebuild_tasks`` then reads name, loss weight and normaliser statistics off the placeholders and builds this
  package's own Task objects (model_factory/task.py).
"""
import io
import pickle
import re
from collections import OrderedDict
from pathlib import Path
from typing import Any, Dict, List, Tuple

import torch

from .data.transform import ScalarTargetTransform, TensorTargetTransform
from .model_factory.task import ScalarRegressionTask, Task, TensorRegressionTask

# ---------------------------------------------------------------------------------------------------------------------
# restricted unpickling
# ---------------------------------------------------------------------------------------------------------------------
_SAFE_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("collections", "deque"),
    ("copyreg", "_reconstructor"), ("copyreg", "__newobj__"), ("_codecs", "encode"),
    ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"), ("pathlib", "Path"), ("pathlib", "PurePath"),
    ("pathlib", "WindowsPath"), ("pathlib", "PureWindowsPath"),
    ("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"),
}
# torch globals a state_dict / hyper-parameter pickle legitimately references: an explicit (module, name) list, the same
# set torch's own weights_only unpickler trusts.  Nothing is matched by prefix or by "is a class": e.g.
# torch.serialization._open_file is a class whose constructor truncates a file.
_SAFE_TORCH_GLOBALS = {
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor_v3"),
    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch._utils", "_rebuild_device_tensor_from_numpy"), ("torch._utils", "_rebuild_device_tensor_from_cpu_tensor"),
    ("torch._utils", "_rebuild_meta_tensor_no_storage"),
    ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.nn.parameter", "Parameter"), ("torch.nn.parameter", "Buffer"),
    ("torch.serialization", "_get_layout"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
}
_SAFE_TORCH_CONTAINERS = {"ModuleDict", "ModuleList", "Sequential", "Module", "Identity"}
# classes of this package a checkpoint written by it may carry (hyper_parameters['tasks'] and their normalisers)
_SAFE_OWN_GLOBALS = {
    ("matten_amd.model_factory.task", "Task"), ("matten_amd.model_factory.task", "TensorRegressionTask"),
    ("matten_amd.model_factory.task", "ScalarRegressionTask"), ("matten_amd.model_factory.task", "CanonicalRegressionTask"),
    ("matten_amd.data.transform", "MeanNormNormalize"), ("matten_amd.data.transform", "ScalarNormalize"),
    ("matten_amd.data.transform", "TensorTargetTransform"), ("matten_amd.data.transform", "ScalarTargetTransform"),
    ("matten_amd.o3", "Irrep"), ("matten_amd.o3", "MulIr"), ("matten_amd.o3", "Irreps"),
}


class Opaque:
    """Placeholder of an object whose class is not importable here: keeps what the pickle said about it."""
    _opaque_module = "?"
    _opaque_name = "?"

    def __new__(cls, *args, **kwargs):
        self = object.__new__(cls)
        object.__setattr__(self, "_opaque_args", args)
        object.__setattr__(self, "_opaque_kwargs", kwargs)
        return self

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):  # (dict, slots) form
            state = {**(state[0] or {}), **state[1]}
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_opaque_state"] = state

    def __call__(self, *args, **kwargs):  # a pickled *function* global used through REDUCE
        return _make_opaque(self._opaque_module, self._opaque_name + "()")(*args, **kwargs)

    def __repr__(self):
        return f"<opaque {self._opaque_module}.{self._opaque_name}>"

    # containers some pickles build through append / extend / __setitem__
    def append(self, x):
        self.__dict__.setdefault("_opaque_items", []).append(x)

    def extend(self, xs):
        self.__dict__.setdefault("_opaque_items", []).extend(xs)

    def __setitem__(self, k, v):
        self.__dict__.setdefault("_opaque_map", {})[k] = v


class OpaqueTuple(tuple):
    """tuple subclasses (e3nn Irrep / _MulIr / Irreps) are re-created through cls.__new__(cls, contents)"""
    _opaque_module = "?"
    _opaque_name = "?"

    def __new__(cls, *args):
        return tuple.__new__(cls, args[0] if len(args) == 1 and isinstance(args[0], (tuple, list)) else args)


_OPAQUE_CACHE: Dict[Tuple[str, str], type] = {}
_TUPLE_LIKE = {("e3nn.o3._irreps", "Irrep"), ("e3nn.o3._irreps", "_MulIr"), ("e3nn.o3._irreps", "Irreps")}


def _make_opaque(module: str, name: str) -> type:
    key = (module, name)
    if key not in _OPAQUE_CACHE:
        base = OpaqueTuple if key in _TUPLE_LIKE else Opaque
        _OPAQUE_CACHE[key] = type(name.split(".")[-1], (base,), {"_opaque_module": module, "_opaque_name": name})
    return _OPAQUE_CACHE[key]


def _resolve(module: str, name: str):
    if module == "builtins" and name in _SAFE_BUILTINS:
        return getattr(__import__("builtins"), name)
    if (module, name) in _SAFE_GLOBALS:
        mod = __import__(module, fromlist=["_"])
        return getattr(mod, name)
    if module == "torch":
        obj = getattr(torch, name, None)
        # dtypes and the legacy typed tensor / storage classes (torch.FloatStorage, ...) -- never functions
        if isinstance(obj, torch.dtype):
            return obj
        if isinstance(obj, type) and re.fullmatch(r"[A-Z][A-Za-z0-9]*(Storage|Tensor)", name):
            return obj
    if (module, name) in _SAFE_TORCH_GLOBALS:
        obj = getattr(__import__(module, fromlist=["_"]), name, None)
        if obj is not None:
            return obj
    if module.startswith("torch.nn.modules.") and name in _SAFE_TORCH_CONTAINERS:
        return getattr(torch.nn, name)
    if (module, name) in _SAFE_OWN_GLOBALS:  # checkpoints written by this package
        obj = getattr(__import__(module, fromlist=["_"]), name, None)
        if isinstance(obj, type):
            return obj
    return _make_opaque(module, name)


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        return _resolve(module, name)


class _PickleModule:
    """what torch.load expects of ``pickle_module``"""
    __name__ = "matten_amd.checkpoint"
    Unpickler = RestrictedUnpickler
    Pickler = pickle.Pickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL

    @staticmethod
    def load(f, **kwargs):
        return RestrictedUnpickler(f, **kwargs).load()

    @staticmethod
    def loads(b, **kwargs):
        return RestrictedUnpickler(io.BytesIO(b), **kwargs).load()


def load_checkpoint(path) -> Dict[str, Any]:
    """torch.load with the restricted unpickler -> {"state_dict": ..., "hyper_parameters": ..., ...}"""
    ckpt = torch.load(Path(path), map_location="cpu", weights_only=False, pickle_module=_PickleModule)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt or "hyper_parameters" not in ckpt:
        raise RuntimeError(f"{path}: not a model checkpoint (needs 'state_dict' and 'hyper_parameters')")
    return ckpt


# ---------------------------------------------------------------------------------------------------------------------
# state_dict: drop the e3nn / Lightning internals, and only those
# ---------------------------------------------------------------------------------------------------------------------
_INTERNAL_KEY = re.compile(r"(^|\.)_compiled_|(^|\.)_w3j_|\.output_mask$|(^|\.)metrics\.|(^|\.)loss_fns\.")


def is_foreign_internal(key: str, value) -> bool:
    """True for state_dict entries a reference checkpoint carries that have no meaning outside e3nn / Lightning"""
    if _INTERNAL_KEY.search(key):
        return True
    # e3nn registers an EMPTY tensor named weight / bias on modules without internal weights / biases
    # (o3.TensorProduct(shared_weights=False) -> '...tp.tp.weight', Gate's ElementwiseTensorProduct, o3.Linear bias)
    if key.rsplit(".", 1)[-1] in ("weight", "bias") and isinstance(value, torch.Tensor) and value.numel() == 0:
        return True
    return False


def filter_state_dict(state_dict: Dict[str, Any], expected_keys) -> Tuple["OrderedDict[str, Any]", List[str], List[str]]:
    """-> (entries to load, missing keys, unexpected keys that are NOT known internals)"""
    expected = set(expected_keys)
    keep, bad = OrderedDict(), []
    for k, v in state_dict.items():
        if k in expected:
            keep[k] = v
        elif not is_foreign_internal(k, v):
            bad.append(k)
    missing = [k for k in expected_keys if k not in keep]
    return keep, missing, bad


# ---------------------------------------------------------------------------------------------------------------------
# tasks
# ---------------------------------------------------------------------------------------------------------------------
def _attr(obj, name, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, "__dict__", {}).get(name, default) if isinstance(obj, Opaque) else getattr(obj, name, default)


def _module_child(obj, name):
    """child `name` of a pickled torch.nn.Module placeholder (lives in its _modules dict)"""
    mods = _attr(obj, "_modules") or {}
    return mods.get(name) if hasattr(mods, "get") else None


def _module_buffer(obj, name):
    bufs = _attr(obj, "_buffers") or {}
    t = bufs.get(name) if hasattr(bufs, "get") else None
    return t if isinstance(t, torch.Tensor) else None


def _irreps_str(obj) -> str:
    """'2x0e+2x2e+4e' from a pickled e3nn Irreps (tuple of (mul, (l, p))), a string, or None"""
    if isinstance(obj, str):
        return obj
    try:
        return "+".join(f"{int(mul)}x{int(l)}{'e' if int(p) == 1 else 'o'}" for mul, (l, p) in obj)
    except Exception:  # noqa: BLE001
        return None


def _fill(normalizer, src) -> bool:
    """copy mean / norm / scale of a pickled MeanNormNormalize / ScalarNormalize placeholder into ours"""
    mean, norm = _module_buffer(src, "mean"), _module_buffer(src, "norm")
    scale = _attr(src, "scale")
    if scale is not None:
        normalizer.scale = float(scale)
    if mean is None or norm is None or not _attr(src, "mean_norm_initialized", False):
        return False
    if mean.shape != normalizer.mean.shape or norm.shape != normalizer.norm.shape:
        raise RuntimeError(f"normalizer statistics of shape {tuple(mean.shape)} do not fit {tuple(normalizer.mean.shape)}")
    normalizer.load_state_dict({"mean": mean.clone().float(), "norm": norm.clone().float()})
    return True


def _rebuild_task(name_hint, t, checkpoint_dir: Path):
    if t is None or isinstance(t, str):
        return (t or name_hint), None
    if isinstance(t, Task):  # written by this package
        return t.name, t
    cls_name = getattr(t, "_opaque_name", type(t).__name__)
    name = _attr(t, "_name") or _attr(t, "name") or name_hint
    if name is None:
        raise RuntimeError(f"cannot find the task name in the checkpoint's {cls_name}")
    loss_weight = float(_attr(t, "_loss_weight", 1.0) or 1.0)
    src = _attr(t, "normalizer")
    if src is None:
        task = (ScalarRegressionTask if "Scalar" in cls_name else TensorRegressionTask)(name, loss_weight=loss_weight)
        return name, task
    stats_path = _attr(src, "dataset_statistics_path")
    if stats_path is not None:
        stats_path = Path(str(stats_path))
        if not stats_path.is_absolute() and not stats_path.exists():
            stats_path = checkpoint_dir / stats_path  # the reference writes a path relative to the training cwd
    if _attr(src, "normalizers") is not None or _module_child(src, "normalizers") is not None:
        task = ScalarRegressionTask(name, loss_weight=loss_weight, dataset_statistics_path=stats_path,
                                    normalize_target=True)
        inner = _module_child(src, "normalizers")
        inner = _module_child(inner, name) if inner is not None else None
        filled = inner is not None and _fill(task.normalizer.normalizers[name], inner)
    else:
        inner = _module_child(src, "normalizer")
        kw = {}
        irreps = _irreps_str(_attr(inner, "irreps")) if inner is not None else None
        if irreps:
            kw["irreps"] = irreps
        task = TensorRegressionTask(name, loss_weight=loss_weight, dataset_statistics_path=stats_path,
                                    normalize_target=True, normalizer_kwargs=kw)
        filled = inner is not None and _fill(task.normalizer.normalizer, inner)
    if not filled and (stats_path is None or not Path(stats_path).exists()):
        raise RuntimeError(
            f"the checkpoint's task '{name}' standardises its target, but the statistics are neither inside the "
            f"checkpoint nor at {stats_path}: predictions could only be returned in standardised units. Put the "
            "training run's dataset_statistics.pt next to the checkpoint."
        )
    return name, task


def rebuild_tasks(tasks, checkpoint_dir) -> Dict[str, Any]:
    """hyper_parameters['tasks'] (None | str | Task | list | dict, possibly placeholders) -> {name: Task | None}"""
    checkpoint_dir = Path(checkpoint_dir)
    if tasks is None:
        return {"elastic_tensor_full": None}
    if isinstance(tasks, dict):
        return dict(_rebuild_task(k, v, checkpoint_dir) for k, v in tasks.items())
    if isinstance(tasks, (list, tuple)) and not isinstance(tasks, OpaqueTuple):
        return dict(_rebuild_task(None, v, checkpoint_dir) for v in tasks)
    return dict([_rebuild_task(None, tasks, checkpoint_dir)])


def plain(obj):
    """hyper-parameter containers with placeholders / paths turned into plain Python (for model construction)"""
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, OpaqueTuple):
        return _irreps_str(obj) or tuple(plain(v) for v in obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(plain(v) for v in obj)
    if isinstance(obj, Opaque):
        d = {k: plain(v) for k, v in obj.__dict__.items() if not k.startswith("_opaque")}
        return d or None
    if isinstance(obj, torch.Tensor) and obj.numel() == 1:
        return obj.item()
    return obj

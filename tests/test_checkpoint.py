"""Loading a reference-style Lightning checkpoint without matten / e3nn / Lightning importable:
a synthetic checkpoint is written the way the reference writes one -- e3nn-internal state_dict entries, a pickled
``matten.model_factory.task.TensorRegressionTask`` holding a filled ``matten.data.transform.TensorTargetTransform``
with e3nn ``Irreps`` inside -- from stand-in classes that exist only while the file is being pickled."""
import sys
import types
from pathlib import Path

import pytest
import torch

from common import LMAX2
from matten_amd import predict as P
from matten_amd.checkpoint import Opaque, filter_state_dict, load_checkpoint, rebuild_tasks
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

DATASET_HP = {"allowed_species": (13, 29, 79), "average_num_neighbors": torch.tensor(18.5)}


def _fake_reference_modules():
    """stand-ins under the reference's module paths, alive only inside `with`"""
    mods = {}

    def module(name):
        m = types.ModuleType(name)
        mods[name] = m
        return m

    for pkg in ("matten", "matten.model_factory", "matten.data", "matten.model", "e3nn", "e3nn.o3", "torchmetrics"):
        module(pkg)
    irr = module("e3nn.o3._irreps")
    task_mod = module("matten.model_factory.task")
    tr = module("matten.data.transform")
    tm = module("torchmetrics.regression.mae")

    def cls(mod, name, base, body):
        c = type(name, (base,), dict(body, __module__=mod.__name__, __qualname__=name))
        setattr(mod, name, c)
        return c

    Irrep = cls(irr, "Irrep", tuple, {"__new__": lambda c, l, p=None: tuple.__new__(c, l if p is None else (l, p))})
    MulIr = cls(irr, "_MulIr", tuple, {"__new__": lambda c, m, i=None: tuple.__new__(c, m if i is None else (m, i))})
    Irreps = cls(irr, "Irreps", tuple, {"__new__": lambda c, x: tuple.__new__(c, x)})

    def mnn_init(self, dim):
        torch.nn.Module.__init__(self)
        self.irreps = Irreps((MulIr(2, Irrep(0, 1)), MulIr(2, Irrep(2, 1)), MulIr(1, Irrep(4, 1))))
        self.normalization, self.reduce, self.eps, self.scale = "component", "mean", 1e-5, 0.5
        self.mean_norm_initialized = True
        self.register_buffer("mean", torch.arange(dim, dtype=torch.float32) * 0.1)
        self.register_buffer("norm", torch.arange(dim, dtype=torch.float32) * 0.01 + 2.0)

    MNN = cls(tr, "MeanNormNormalize", torch.nn.Module, {"__init__": mnn_init})

    def ttt_init(self, name):
        torch.nn.Module.__init__(self)
        self.dataset_statistics_path = Path("dataset_statistics.pt")
        self.dataset_statistics_loaded = True
        self.target_name = name
        self.normalizer = MNN(21)

    TTT = cls(tr, "TensorTargetTransform", torch.nn.Module, {"__init__": ttt_init})

    def task_init(self, name, normalize):
        self._name, self._loss_weight = name, 1.0
        self.normalizer = TTT(name) if normalize else None

    Task = cls(task_mod, "TensorRegressionTask", object, {"__init__": task_init})
    cls(tm, "MeanAbsoluteError", object, {})
    return mods, Task


class _Installed:
    def __init__(self, mods):
        self.mods = mods

    def __enter__(self):
        sys.modules.update(self.mods)

    def __exit__(self, *exc):
        for k in self.mods:
            sys.modules.pop(k, None)


def _write_reference_style_checkpoint(directory: Path, normalize: bool):
    model = ScalarTensorModel(tasks=None, backbone_hparams=dict(LMAX2, num_layers=1), dataset_hparams=DATASET_HP)
    torch.manual_seed(5)
    for p in model.parameters():
        p.data.normal_()
    sd = model.state_dict()
    ref_sd = dict(sd)
    # what e3nn 0.5.1 modules add to a reference state_dict
    ref_sd["backbone.layer0_convnet.conv.tp.tp.weight"] = torch.Tensor()
    ref_sd["backbone.layer0_convnet.conv.tp.tp.output_mask"] = torch.ones(40)
    ref_sd["backbone.layer0_convnet.conv.tp.tp._compiled_main_left_right._w3j_1_1_2"] = torch.zeros(3, 3, 5)
    ref_sd["backbone.layer0_convnet.conv.lin1.output_mask"] = torch.ones(16)
    ref_sd["backbone.layer0_convnet.conv.lin1._compiled_main_left_right._w3j_0_0_0"] = torch.ones(1, 1, 1)
    ref_sd["backbone.layer0_convnet.act.gate.mul.weight"] = torch.Tensor()
    ref_sd["backbone.layer0_convnet.act.gate.mul.output_mask"] = torch.ones(8)
    ref_sd["backbone.conv_to_output_hidden.linear.bias"] = torch.Tensor()
    ref_sd["backbone.conv_to_output_hidden.linear.output_mask"] = torch.ones(35)
    ref_sd["extra_layers_dict.out_layer.bias"] = torch.Tensor()
    ref_sd["extra_layers_dict.out_layer.output_mask"] = torch.ones(21)
    ref_sd["metrics.val.elastic_tensor_full.MeanAbsoluteError.total"] = torch.tensor(3.0)
    mods, Task = _fake_reference_modules()
    with _Installed(mods):
        hp = {
            "tasks": Task("elastic_tensor_full", normalize),
            "backbone_hparams": dict(LMAX2, num_layers=1),
            "dataset_hparams": dict(DATASET_HP),
            "optimizer_hparams": {"class_path": "torch.optim.Adam", "init_args": {"lr": 0.01}},
            "lr_scheduler_hparams": None,
            "trainer_hparams": None,
            "data_hparams": None,
        }
        torch.save({"state_dict": ref_sd, "hyper_parameters": hp, "epoch": 3,
                    "pytorch-lightning_version": "2.0.6"}, directory / "model_final.ckpt")
    assert "matten" not in sys.modules and "e3nn" not in sys.modules
    return sd


@pytest.mark.parametrize("normalize", [False, True])
def test_reference_style_checkpoint_loads(tmp_path, normalize):
    sd = _write_reference_style_checkpoint(tmp_path, normalize)
    # the stock unpickler would import whatever the file names: `matten` (here the alias package of this repository,
    # i.e. real constructors run on unpickling) and, with a target normaliser inside, e3nn -- which does not exist here
    if normalize:
        with pytest.raises(ModuleNotFoundError):
            torch.load(tmp_path / "model_final.ckpt", map_location="cpu", weights_only=False)
    model = P.get_pretrained_model(str(tmp_path), device="cpu")
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert model.hparams["dataset_hparams"]["allowed_species"] == (13, 29, 79)
    assert model.hparams["dataset_hparams"]["average_num_neighbors"] == pytest.approx(18.5)
    task = model.tasks["elastic_tensor_full"]
    assert task.name == "elastic_tensor_full"
    x = torch.randn(4, 21)
    out = model.transform_prediction({"elastic_tensor_full": x})["elastic_tensor_full"]
    if normalize:
        n = task.normalizer.normalizer
        assert str(n.irreps) == "2x0e+2x2e+1x4e" and n.scale == 0.5
        mean, norm = torch.arange(21.) * 0.1, torch.arange(21.) * 0.01 + 2.0
        assert torch.allclose(out, x * (norm * 0.5) + mean)
    else:
        assert task.normalizer is None and torch.equal(out, x)


def test_unpickling_executes_nothing_foreign(tmp_path):
    """a global outside the allow-list (here os.system through REDUCE) turns into an inert placeholder"""
    import pickle

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    torch.save({"state_dict": {}, "hyper_parameters": {"tasks": None, "x": Evil()}}, tmp_path / "evil.ckpt",
               pickle_module=pickle)
    ckpt = load_checkpoint(tmp_path / "evil.ckpt")
    assert isinstance(ckpt["hyper_parameters"]["x"], Opaque)
    assert not (tmp_path / "pwned").exists()


def test_torch_classes_outside_the_allow_list_are_inert(tmp_path):
    """torch.serialization._open_file is a CLASS whose constructor opens (and with 'w' truncates) a file: a prefix or
    isinstance(type) rule would resolve it.  The allow-list is by (module, name): it must come back as a placeholder."""
    from matten_amd.checkpoint import RestrictedUnpickler, _resolve
    import io

    victim = tmp_path / "victim.txt"
    victim.write_text("keep me")
    payload = ("ctorch.serialization\n_open_file\n(V%s\nVw\ntR." % victim).encode()
    obj = RestrictedUnpickler(io.BytesIO(payload)).load()
    assert isinstance(obj, Opaque)
    assert victim.read_text() == "keep me"
    for module, name in [("torch.serialization", "_open_zipfile_writer_file"), ("torch._utils_internal", "justknobs_check"),
                         ("torch.storage", "_load_from_bytes"), ("torch.serialization", "load"),
                         ("matten_amd.predict", "predict"), ("matten_amd._lib", "MattenHipError"),
                         ("torch.nn.modules.linear", "Linear"), ("torch", "load"), ("torch", "hub")]:
        got = _resolve(module, name)
        assert isinstance(got, type) and issubclass(got, Opaque), (module, name, got)
    # what a state_dict needs still resolves to the real thing
    assert _resolve("torch._utils", "_rebuild_tensor_v2") is torch._utils._rebuild_tensor_v2
    assert _resolve("torch", "FloatStorage") is torch.FloatStorage
    assert _resolve("torch.nn.parameter", "Parameter") is torch.nn.Parameter
    assert _resolve("collections", "OrderedDict").__name__ == "OrderedDict"


def test_unknown_state_dict_entries_are_still_an_error(tmp_path):
    sd = _write_reference_style_checkpoint(tmp_path, False)
    ckpt = load_checkpoint(tmp_path / "model_final.ckpt")
    ckpt["state_dict"]["backbone.layer0_convnet.conv.lin3.weight"] = torch.ones(5)  # not an e3nn internal
    ckpt["state_dict"].pop("backbone.layer0_convnet.conv.lin2.weight")
    keep, missing, bad = filter_state_dict(ckpt["state_dict"], list(sd.keys()))
    assert missing == ["backbone.layer0_convnet.conv.lin2.weight"]
    assert bad == ["backbone.layer0_convnet.conv.lin3.weight"]
    # a non-empty tensor called weight / bias is never dropped silently
    assert filter_state_dict({"a.bias": torch.ones(2)}, [])[2] == ["a.bias"]


def test_standardised_target_without_statistics_is_refused(tmp_path):
    """hparams ask for target standardisation but neither the checkpoint nor a statistics file has the numbers:
    returning standardised predictions silently would be wrong"""
    mods, Task = _fake_reference_modules()
    with _Installed(mods):
        t = Task("elastic_tensor_full", True)
        t.normalizer.normalizer.mean_norm_initialized = False
        t.normalizer.dataset_statistics_loaded = False
        torch.save({"state_dict": {}, "hyper_parameters": {"tasks": t}}, tmp_path / "t.ckpt")
    placeholder = load_checkpoint(tmp_path / "t.ckpt")["hyper_parameters"]["tasks"]
    with pytest.raises(RuntimeError, match="standardis"):
        rebuild_tasks(placeholder, tmp_path)
    # with the training run's statistics file next to the checkpoint it loads lazily, like the reference
    torch.save({"elastic_tensor_full": {"mean": torch.zeros(21), "norm": torch.full((21,), 3.0)}},
               tmp_path / "dataset_statistics.pt")
    tasks = rebuild_tasks(placeholder, tmp_path)
    out = tasks["elastic_tensor_full"].normalizer.inverse(torch.ones(2, 21))
    assert torch.allclose(out, torch.full((2, 21), 1.5))  # scale 0.5 * norm 3


def test_own_checkpoint_round_trip(tmp_path):
    """a checkpoint written by this package (its own Task with a filled normaliser) comes back as the same objects"""
    from matten_amd.model_factory.task import TensorRegressionTask

    task = TensorRegressionTask("elastic_tensor_full", normalize_target=True, dataset_statistics_path=None)
    task.normalizer.normalizer.load_state_dict({"mean": torch.ones(21), "norm": torch.full((21,), 2.0)})
    hp = dict(LMAX2, num_layers=1)
    model = ScalarTensorModel(tasks=task, backbone_hparams=hp, dataset_hparams=DATASET_HP)
    torch.save({"state_dict": model.state_dict(), "hyper_parameters": model.hparams}, tmp_path / "model_final.ckpt")
    loaded = P.get_pretrained_model(str(tmp_path), device="cpu")
    t = loaded.tasks["elastic_tensor_full"]
    assert isinstance(t, TensorRegressionTask)
    out = loaded.transform_prediction({"elastic_tensor_full": torch.ones(1, 21)})["elastic_tensor_full"]
    assert torch.allclose(out, torch.full((1, 21), 3.0))

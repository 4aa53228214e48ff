"""
Optimisation-loop shell of the models: the part of the reference's ``BaseModel`` (model/model.py:53-480) that the
training script drives -- ``compute_loss``, ``shared_step``, ``training_step`` / ``validation_step`` / ``test_step``,
metric bookkeeping and ``configure_optimizers`` -- on top of whatever ``forward`` / ``decode`` / ``preprocess_batch`` the
concrete model defines (model_factory/tfn_*.py).

With PyTorch Lightning installed the base class IS ``pytorch_lightning.LightningModule`` and a stock ``Trainer`` runs
the model (scripts/train_materials_tensor.py:34-68 of the reference works with ``import matten``).  Without it -- the
build and GPU boxes have no Lightning -- the base is ``torch.nn.Module`` with the three hooks the shell calls
(``save_hyperparameters``, ``log``, ``log_dict``) and ``matten_amd.model.trainer.Trainer`` is a minimal loop with the same
call sequence.  Nothing here is on the hot path: every tensor operation of a step happens in ``decode``.
"""
import importlib
import time
from typing import Any, Dict, Tuple, Union

import torch
from torch import Tensor

try:  # pragma: no cover - not installed on the build / GPU boxes
    import pytorch_lightning as _pl

    _Base = _pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    _Base = torch.nn.Module
    HAVE_LIGHTNING = False

from ..model_factory.task import TaskType


def instantiate_class(args, init: Dict[str, Any]):
    """``{"class_path": "torch.optim.Adam", "init_args": {...}}`` -> object (lightning.cli.instantiate_class,
    used by the reference at model/model.py:449,478)"""
    args = args if isinstance(args, tuple) else (args,)
    module, _, name = init["class_path"].rpartition(".")
    cls = getattr(importlib.import_module(module), name)
    return cls(*args, **(init.get("init_args") or {}))


class TimeMeter:
    """reference utils.py TimeMeter: (seconds since the last update, seconds since the start)"""

    def __init__(self):
        self.t0 = self.t = time.time()

    def update(self):
        now = time.time()
        delta, self.t = now - self.t, now
        return delta, now - self.t0


class BaseModel(_Base):
    monitor_key = "val/score"

    # ---- what a subclass's __init__ calls once backbone and tasks exist (reference model/model.py:66-99) ----
    def _init_training_shell(self, hparams: Dict[str, Any]) -> None:
        if HAVE_LIGHTNING:  # pragma: no cover
            self.save_hyperparameters(hparams)
        else:
            self.hparams = dict(hparams)
        tasks = {k: t for k, t in self.tasks.items() if t is not None}
        self.loss_fns = {name: task.init_loss() for name, task in tasks.items()}
        self.metrics = torch.nn.ModuleDict()
        for mode in ("train", "val", "test"):
            self.metrics["metric_" + mode] = torch.nn.ModuleDict(
                {name: task.init_metric_as_collection() for name, task in tasks.items()})
        self.timer = TimeMeter()
        self.logged: Dict[str, Any] = {}

    if not HAVE_LIGHTNING:
        def log(self, name, value, **kwargs):
            self.logged[name] = value.detach() if isinstance(value, Tensor) else value

        def log_dict(self, d, **kwargs):
            for k, v in d.items():
                self.log(k, v)

        @property
        def device(self):
            # (the first Parameter OBJECT is stable across .to() / FlatAdam re-homing; walking parameters() costs ~8 us and
            # preprocess_batch asks once per forward)
            p = self.__dict__.get("_amd_first_param")
            if p is None:
                p = next(self.parameters())
                self.__dict__["_amd_first_param"] = p
            return p.device

    # ---- reference model/model.py:234-280 ----
    def compute_loss(self, preds: Dict[str, Tensor], labels: Dict[str, Tensor], weight: Tensor = None):
        individual_losses, total_loss = {}, 0.0
        for task_name, task in self.tasks.items():
            p = task.transform_pred_loss(preds[task_name])
            l = task.transform_target_loss(labels[task_name])
            if weight is not None:
                p, l = p * weight, l * weight
            if task.task_type == TaskType.CLASSIFICATION and task.is_binary():
                p, l = p.reshape(-1), l.reshape(-1).to(torch.get_default_dtype())
            loss = self.loss_fns[task_name](p, l)
            individual_losses[task_name] = loss
            total_loss = total_loss + task.loss_weight * loss
        return individual_losses, total_loss

    # ---- reference model/model.py:282-312 ----
    def training_step(self, batch, batch_idx):
        loss, preds, labels = self.shared_step(batch, "train")
        self.update_metrics(preds, labels, "train")
        return {"loss": loss}

    def on_training_epoch_end(self):
        self.compute_metrics("train")

    def validation_step(self, batch, batch_idx):
        loss, preds, labels = self.shared_step(batch, "val")
        self.update_metrics(preds, labels, "val")
        return {"loss": loss}

    def on_validation_epoch_end(self):
        _, score = self.compute_metrics("val")
        if score is not None:  # val/score: early stopping, checkpointing and the lr scheduler watch it
            self.log(self.monitor_key, score, on_step=False, on_epoch=True, prog_bar=True)
        delta_t, cumulative_t = self.timer.update()
        self.log("epoch time", delta_t, on_step=False, on_epoch=True, prog_bar=True)
        self.log("cumulative time", cumulative_t, on_step=False, on_epoch=True, prog_bar=True)

    def test_step(self, batch, batch_idx):
        loss, preds, labels = self.shared_step(batch, "test")
        self.update_metrics(preds, labels, "test")
        return {"loss": loss}

    def on_test_epoch_end(self):
        self.compute_metrics("test")

    # ---- reference model/model.py:316-372 ----
    def shared_step(self, batch, mode: str):
        batch_size = batch.num_graphs if hasattr(batch, "num_graphs") else int(batch["ptr"].shape[0]) - 1
        graphs, labels = self.preprocess_batch(batch)
        preds = self.decode(graphs)
        if "atom_selector" in labels:
            selector = labels["atom_selector"]
            preds = {k: v[selector] for k, v in preds.items()}
        target_weight = graphs.get("target_weight", None)
        individual_loss, total_loss = self.compute_loss(preds, labels, weight=target_weight)
        self.log_dict({f"{mode}/loss/{name}": loss for name, loss in individual_loss.items()},
                      on_step=False, on_epoch=True, prog_bar=False, batch_size=batch_size)
        self.log(f"{mode}/total_loss", total_loss, on_step=False, on_epoch=True, prog_bar=True, batch_size=batch_size)
        return total_loss, preds, labels

    # ---- reference model/model.py:374-445 ----
    def update_metrics(self, preds: Dict, labels: Dict, mode: str):
        for task_name, metric in self.metrics["metric_" + mode].items():
            task = self.tasks[task_name]
            p = task.transform_pred_metric(preds[task_name])
            l = task.transform_target_metric(labels[task_name])
            if task.task_type == TaskType.CLASSIFICATION:
                p = torch.sigmoid(p.reshape(-1)) if task.is_binary() else torch.argmax(p, dim=1)
            metric(p.detach(), l.detach())

    def compute_metrics(self, mode, log: bool = True) -> Tuple[Dict[str, Dict[str, Tensor]], Union[Tensor, None]]:
        mode = "metric_" + mode
        total_score, individual_score = None, {}
        for task_name, metric_coll in self.metrics[mode].items():
            score = metric_coll.compute()
            individual_score[task_name] = score
            if log:
                for metric_name, metric_value in score.items():
                    self.log(f"{mode}/{metric_name}/{task_name}", metric_value, on_step=False, on_epoch=True, prog_bar=False)
            agg = self.tasks[task_name].metric_aggregation()
            if agg:
                total_score = 0 if total_score is None else total_score
                for metric_name, weight in agg.items():
                    total_score = total_score + score[metric_name] * weight
            metric_coll.reset()
        return individual_score, total_score

    # ---- reference model/model.py:447-479 ----
    def configure_optimizers(self):
        params = (filter(lambda p: p.requires_grad, self.parameters()),)
        optimizer = instantiate_class(params, self.optimizer_hparams)
        scheduler = self._config_lr_scheduler(optimizer)
        if scheduler is None:
            return optimizer
        return {"optimizer": optimizer, "lr_scheduler": scheduler, "monitor": self.monitor_key}

    def _config_lr_scheduler(self, optimizer):
        hp = self.lr_scheduler_hparams or {}
        class_path = hp.get("class_path")
        if class_path is None or class_path == "none":
            return None
        init = dict(hp, init_args={k: v for k, v in (hp.get("init_args") or {}).items()
                                   if not (k == "verbose" and "ReduceLROnPlateau" in class_path)})  # removed in torch 2.7
        return instantiate_class(optimizer, init)

"""
Task descriptions of the regression models (mirror of the reference's model_factory/task.py:10-113 and the part of
model/task.py:23-248 the inference path touches).  A task names the prediction / label key and owns the optional
target ``normalizer`` that ``ScalarTensorModel.transform_prediction`` inverts (tfn_scalar_tensor.py:80-94), its loss
(``init_loss``: MSE, model/task.py:238-239) and its metric (mean absolute error, :241-248; torchmetrics is not a
dependency: ``MeanAbsoluteError`` / ``MetricCollection`` below keep the two methods the training shell calls).
"""
from enum import Enum
from pathlib import Path
from typing import Any, Dict, Union

import torch
from torch import Tensor

from ..data.transform import ScalarTargetTransform, TensorTargetTransform


class TaskType(Enum):   # reference model/task.py:17-20
    CLASSIFICATION = "classification"
    REGRESSION = "regression"


class MeanAbsoluteError(torch.nn.Module):
    """running mean of |pred - target| over everything seen since the last reset (torchmetrics.MeanAbsoluteError)"""

    def __init__(self):
        super().__init__()
        self.register_buffer("sum_abs", torch.zeros((), dtype=torch.float64), persistent=False)
        self.register_buffer("count", torch.zeros((), dtype=torch.float64), persistent=False)

    def forward(self, preds: Tensor, target: Tensor) -> None:
        self.sum_abs += (preds.double() - target.double()).abs().sum().to(self.sum_abs.device)
        self.count += preds.numel()

    update = forward

    def compute(self) -> Tensor:
        return (self.sum_abs / self.count.clamp(min=1)).float()

    def reset(self) -> None:
        self.sum_abs.zero_()
        self.count.zero_()


class MetricCollection(torch.nn.ModuleDict):
    """{metric class name: metric}; called with (preds, target) it updates every member (torchmetrics.MetricCollection)"""

    def forward(self, preds: Tensor, target: Tensor) -> None:
        for m in self.values():
            m(preds, target)

    def compute(self) -> Dict[str, Tensor]:
        return {k: m.compute() for k, m in self.items()}

    def reset(self) -> None:
        for m in self.values():
            m.reset()


class Task:
    def __init__(self, name: str, *, loss_weight: float = 1.0, **kwargs):
        self._name = name
        self._loss_weight = loss_weight
        self.__dict__.update(kwargs)

    @property
    def name(self) -> str:
        return self._name

    @property
    def loss_weight(self) -> float:
        return self._loss_weight

    def __getitem__(self, key):  # reference model/task.py: kwargs are reachable as items too
        return self.__dict__[key]


class CanonicalRegressionTask(Task):
    """reference model/task.py:226-248: MSE loss, mean absolute error as metric and as the monitored score"""
    normalizer = None

    @property
    def task_type(self) -> TaskType:
        return TaskType.REGRESSION

    def init_loss(self):
        return torch.nn.MSELoss()

    def init_metric(self):
        return MeanAbsoluteError()

    def init_metric_as_collection(self) -> MetricCollection:
        m = self.init_metric()
        return m if isinstance(m, MetricCollection) else MetricCollection({type(m).__name__: m})

    def metric_aggregation(self) -> Dict[str, float]:
        return {"MeanAbsoluteError": 1.0}   # early stopping / checkpointing therefore run in `min` mode

    def transform_pred_metric(self, t: Tensor) -> Tensor:
        return t

    def transform_target_metric(self, t: Tensor) -> Tensor:
        return t

    def transform_target_loss(self, t: Tensor) -> Tensor:
        return t

    def transform_pred_loss(self, t: Tensor) -> Tensor:
        return t


_RegressionTask = CanonicalRegressionTask


class TensorRegressionTask(_RegressionTask):
    def __init__(self, name: str, loss_weight: float = 1.0,
                 dataset_statistics_path: Union[str, Path] = "dataset_statistics.pt", normalize_target: bool = False,
                 normalizer_kwargs: Dict[str, Any] = None):
        super().__init__(name, loss_weight=loss_weight)
        self.normalizer = TensorTargetTransform(
            target_name=name, dataset_statistics_path=dataset_statistics_path, **(normalizer_kwargs or {})
        ) if normalize_target else None

    def transform_target_metric(self, t: Tensor) -> Tensor:
        return t if self.normalizer is None else self.normalizer.inverse(t)

    transform_pred_metric = transform_target_metric


class ScalarRegressionTask(_RegressionTask):
    def __init__(self, name: str, loss_weight: float = 1.0,
                 dataset_statistics_path: Union[str, Path] = "dataset_statistics.pt", normalize_target: bool = False,
                 normalizer_kwargs: Dict[str, Any] = None):
        super().__init__(name, loss_weight=loss_weight)
        self.normalizer = ScalarTargetTransform(
            target_names=[name], dataset_statistics_path=dataset_statistics_path, **(normalizer_kwargs or {})
        ) if normalize_target else None

    def transform_target_metric(self, t: Tensor) -> Tensor:
        return t if self.normalizer is None else self.normalizer.inverse(t, self.name)

    transform_pred_metric = transform_target_metric

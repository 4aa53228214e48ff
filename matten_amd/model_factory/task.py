"""
Task descriptions of the regression models (mirror of the reference's model_factory/task.py:10-113 and the part of
model/task.py:23-248 the inference path touches).  A task names the prediction / label key and owns the optional
target ``normalizer`` that ``ScalarTensorModel.transform_prediction`` inverts (tfn_scalar_tensor.py:80-94).  Loss and
metric objects belong to the Lightning training loop and are out of scope.
"""
from pathlib import Path
from typing import Any, Dict, Union

from torch import Tensor

from ..data.transform import ScalarTargetTransform, TensorTargetTransform


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


class _RegressionTask(Task):
    normalizer = None

    def transform_target_loss(self, t: Tensor) -> Tensor:
        return t

    def transform_pred_loss(self, t: Tensor) -> Tensor:
        return t


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

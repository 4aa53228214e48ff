"""Reference import path ``matten.model.task`` (model/task.py:17-248): task type, base task, the canonical regression
task (MSE loss, mean-absolute-error metric that is also the checkpoint / early-stopping score)."""
from ..model_factory.task import (CanonicalRegressionTask, MeanAbsoluteError, MetricCollection, Task,  # noqa: F401
                                  TaskType)

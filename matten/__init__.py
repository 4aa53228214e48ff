"""
``import matten`` -> the MI355X implementation.  The reference's scripts and pickled checkpoints name their classes by
the reference's import paths (``matten.predict``, ``matten.model_factory.tfn_scalar_tensor.ScalarTensorModel``,
``matten.model_factory.task.TensorRegressionTask``, ``matten.dataset.structure_scalar_tensor.TensorDataModule``,
``matten.log.set_logger`` ...: scripts/train_materials_tensor.py:11-14, predict.py:10-16).  This package makes every
``matten.x.y`` resolve to the module object ``matten_amd.x.y`` -- the same object, not a copy, so ``isinstance`` checks and
pickles agree whichever name was used -- and so lets those scripts run unmodified with this repository on ``sys.path``
in place of the reference's ``src``.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import matten_amd as _impl

__version__ = getattr(_impl, "__version__", "0.0")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    prefix = __name__ + "."

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(self.prefix):
            return None
        real = "matten_amd." + fullname[len(self.prefix):]
        try:
            found = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if found is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=found.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module("matten_amd." + spec.name[len(self.prefix):])

    def exec_module(self, module):   # already executed under its real name
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):   # matten.predict, matten.utils, ... as attributes
    try:
        return importlib.import_module(f"{__name__}.{name}")
    except ImportError as e:
        raise AttributeError(name) from e

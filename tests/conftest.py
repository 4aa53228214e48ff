import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("MATTEN_TEST_NAN_EMPTY") == "1":
        # Debugging mode: every torch.empty / empty_like float buffer starts as NaN instead of whatever the allocator hands
        # back, so that a kernel that reads memory nobody wrote (0 x garbage) fails loudly instead of by chance.  Found
        # the alignment holes of the component-major neighbour-sum row (plan.plan_agg_linear).
        import torch

        _empty, _empty_like = torch.empty, torch.empty_like

        def nan_empty(*a, **k):
            t = _empty(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() and t.device.type == "cuda" else t

        def nan_empty_like(*a, **k):
            t = _empty_like(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() and t.device.type == "cuda" else t

        torch.empty, torch.empty_like = nan_empty, nan_empty_like


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _restore_default_stream(request):
    """Eager-mode training makes its high-priority stream the thread's current stream (BaseModel.train_step); every GPU
    test starts and ends on the default stream with the device idle, so tests do not see each other's streams."""
    yield
    if 'gpu' in request.keywords and _has_gpu():
        import torch
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())

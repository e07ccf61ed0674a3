import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the round-end driver)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a GPU-less box: they are deselected by `-m "not gpu"`,
    # and when selected without a GPU they fail loudly in the `gpu` fixture below.
    pass


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "this test needs an MI355X; run it with gpurun"
    import swift_png_amd as spng
    spng.load()  # raises if the HIP extension is missing
    return spng

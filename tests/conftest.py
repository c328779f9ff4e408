import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


@pytest.fixture(autouse=True)
def _hrv_env_switches_reloaded():
    """The HIP library caches its HRV_* environment switches; a test that flips one (monkeypatch.setenv / os.environ) calls
    hr_viton_amd._lib.reload_env() after setting it, and this fixture -- torn down after monkeypatch has restored the environment --
    drops the cache again so that the next test starts from the real environment."""
    yield
    mod = sys.modules.get("hr_viton_amd._lib")
    if mod is not None:
        mod.reload_env()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-hugs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment():
    """The HIP library reads its HGS_* A/B switches once; tests flip them with monkeypatch.setenv + hgs_reload_switches().  An
    autouse fixture is set up before (so torn down after) monkeypatch: by now the environment is back, and the library re-reads it."""
    yield
    import sys
    dgr = sys.modules.get("diff_gaussian_rasterization")
    if dgr is not None and getattr(dgr, "_lib", None) is not None:
        dgr._lib.hgs_reload_switches()

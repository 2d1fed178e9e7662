import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(params=["stages", "calls"])
def frame_path(request, monkeypatch):
    """run a fused-frame test twice: stage by stage (one C-ABI entry point per stage: fused._FusedRender) and through
    gs_frame_fwd / gs_frame_bwd (one call per direction: fused._FrameRender).  "always": a frame whose overlap count
    has not been seen yet is sized by an untracked staged pass first, so that the frame calls run even on a test's
    first render of a shape."""
    from taichi_gaussian_rasterizer_amd import fused
    monkeypatch.setattr(fused, "FRAME_CALLS", False if request.param == "stages" else "always")
    return request.param


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

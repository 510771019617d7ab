import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Make sure the in-tree native libraries exist (no-op when they are up to date)."""
    from magr_ray_tracer_amd import build
    build.build_device()
    build.build_device_refb()
    build.build_host()
    build.build_oracle()
    build.build_ref()
    yield


def has_gpu():
    try:
        import ctypes
        from magr_ray_tracer_amd import _lib
        return _lib.device_lib().rt_device_count() > 0
    except Exception:
        return False

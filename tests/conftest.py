import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "a-simple-stereo-slam-system-with-deep-loop-closing_amd")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def load_package():
    """The package directory carries the repository name (not a Python identifier): load it as `myslam_amd`."""
    if "myslam_amd" in sys.modules:
        return sys.modules["myslam_amd"]
    spec = importlib.util.spec_from_file_location("myslam_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["myslam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    from pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def api(pkg):
    """The HIP product path.  GPU tests must never silently pass without it."""
    # torch ships its own HIP runtime: import it BEFORE libmyslam_hip.so so that one runtime serves the process
    # (the tests use torch only to hold device buffers for the *_batch entry points)
    import torch  # noqa: F401
    if not os.path.exists(pkg.api.LIB_PATH):
        pkg.build_library()
    a = pkg.api
    assert a.device_count() >= 1, "no HIP device visible: GPU tests need a real MI355X"
    return a

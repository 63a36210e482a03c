import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_pkg():
    """Import the package directory `hpp-fcl_amd/` (hyphenated project name) as module `hppfcl_amd`."""
    if "hppfcl_amd" in sys.modules:
        return sys.modules["hppfcl_amd"]
    pkg_dir = os.path.join(ROOT, "hpp-fcl_amd")
    spec = importlib.util.spec_from_file_location("hppfcl_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hppfcl_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    oracle_binding.build()
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def hostsim():
    """Host build of the device headers (tests/hostsim) -- validation of the device code on CPU."""
    import hostsim_binding
    hostsim_binding.lib()
    return hostsim_binding


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch

import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


if os.environ.get("SDMI_HOSTEMU") == "1":                      # the `-m gpu` tests against the host-emulated library (tests/hostemu/shim.py)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hostemu import shim as _hostemu_shim
    _hostemu_shim.install()


def pytest_collection_modifyitems(config, items):
    """SDMI_HOSTEMU_SELECT=<file of node ids>: keep only those (the CPU tier's selection of GPU tests for the emulated library)."""
    path = os.environ.get("SDMI_HOSTEMU_SELECT")
    if os.environ.get("SDMI_HOSTEMU") != "1" or not path:
        return
    keep = {ln.strip() for ln in open(path) if ln.strip() and not ln.startswith("#")}
    chosen = [it for it in items if it.nodeid in keep]
    dropped = [it for it in items if it.nodeid not in keep]
    if dropped:
        config.hook.pytest_deselected(items=dropped)
        items[:] = chosen


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so it is imported by name through importlib)."""
    return importlib.import_module("stable-diffusion-webui_amd")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def hostemu_lib():
    """Path of the host-emulated libsdmi (tests/hostemu/build.py); built once per source state, ~40 s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hostemu import build as hostemu_build
    path = os.environ.get("SDMI_LIB") if os.environ.get("SDMI_HOSTEMU") == "1" else hostemu_build.build()
    if not path:
        pytest.skip("host emulation needs clang++ on x86-64")
    return path

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def etp_opt():
    """etp_opt(name, value): set a switch of the library (csrc/options.h) for the rest of the test; restored afterwards.  (Rounds
    1-5 used monkeypatch.setenv: the library called getenv on every launch.  It reads the environment once now.)"""
    from etpnav_amd import _lib
    saved = {}

    def setter(name, value):
        name = name[4:] if name.startswith("ETP_") else name
        if name not in saved:
            saved[name] = _lib.get_option(name)
        _lib.set_option(name, value)

    yield setter
    for name, old in saved.items():
        _lib.set_option(name, old)

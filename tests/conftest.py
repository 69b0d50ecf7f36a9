import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line(
        "markers", "reference: needs /root/reference (development container only)")


@pytest.fixture(autouse=True)
def _mopoe_knobs(monkeypatch):
    """libmopoe_hip.so reads its environment knobs (MOPOE_NO_FUSE, MOPOE_QUAD, ...) ONCE;
    the tests that switch launch forms inside one process go through monkeypatch: every
    MOPOE_* change re-reads them (mopoe_reload_knobs), and every test starts from the
    environment as it is."""
    from importlib import import_module
    try:
        L = import_module("2022_cambroise_interpret_multivae_amd._lib")
    except ImportError:
        yield
        return
    L.reload_knobs()
    real_set, real_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        real_set(name, value, *a, **k)
        if name.startswith("MOPOE_"):
            L.reload_knobs()

    def delenv(name, *a, **k):
        real_del(name, *a, **k)
        if name.startswith("MOPOE_"):
            L.reload_knobs()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    L.reload_knobs()


@pytest.fixture(scope="session", autouse=True)
def _destroy_process_group_at_exit():
    """Tests that need a one-rank nccl group create it when none exists
    (test_hip_dp_onecall.py, test_hip_topology.py); whichever came first, the group is
    destroyed before the interpreter exits (c10d warns about a leaked communicator)."""
    yield
    try:
        import torch.distributed as dist
    except ImportError:
        return
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()

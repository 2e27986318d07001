import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """PyTorch ships its own HIP / ROCr runtime; libsigdigger_amd.so uses the system's.  Both live in one process during
    the tests, and torch must bring its runtime up BEFORE the library's first HIP call or it no longer sees the GPU --
    whatever order the test files are collected in."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def sdo():
    """The CPU oracle (test infrastructure)."""
    from oracle import sdo as _sdo
    _sdo.lib()
    return _sdo


@pytest.fixture(scope="session")
def ctx():
    """A libsigdigger_amd context on cuda:0 (GPU tests only)."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    from sigdigger_amd import engine
    c = engine.Context(0)
    yield c
    c.close()


class _Tune:
    """The library's tuning struct (csrc/tuning.hpp) through the C ABI, with monkeypatch's verbs: the SUAMD_* environment is
    read once per process, so an A / B test sets the FIELD (by its environment name, words as the environment takes them)
    and the fixture puts every touched field back."""
    WORDS = {"SUAMD_ST_KERNEL": {"pair": 0, "wave": 1, "wg": 2}, "SUAMD_PSD_LARGE": {"passes": 0, "twotrip": 1}}

    def __init__(self):
        from sigdigger_amd import engine
        self.engine, self.touched = engine, {}

    def setenv(self, name, value):
        if name not in self.touched:
            self.touched[name] = self.engine.tuning_get(name)
        v = self.WORDS.get(name, {}).get(str(value))
        self.engine.tuning_set(name, int(value) if v is None else v)

    def delenv(self, name, raising=True):
        if name in self.touched:
            self.engine.tuning_set(name, self.touched[name])

    def restore(self):
        for k, v in self.touched.items():
            self.engine.tuning_set(k, v)


@pytest.fixture
def tune():
    t = _Tune()
    yield t
    t.restore()

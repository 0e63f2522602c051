import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/plxo.py) -- the checker, never the product path."""
    from oracle import plxo
    plxo.build()
    return plxo


class _Tuner:
    """Plan-time tuning for tests (plx_ssfm_tuning_override, include/polmux_hip.h) behind monkeypatch's setenv / delenv
    interface: a name PLX_SSFM_<FIELD> that is a field of plx_ssfm_tuning goes into the override of every loaded copy of
    the library (the hipcc build and the emulator build alike), anything else is a real environment variable (the two
    deployment knobs PLX_SSFM_NO_FUSE / PLX_SSFM_BARRIER_TIMEOUT_MS, the emulator's PLX_EMU_*).  Everything is undone
    at the end of the test."""
    FIELDS = {"PLX_SSFM_SHORT_ROWS": "short_rows", "PLX_SSFM_NO_ROW_SPLIT": "no_row_split", "PLX_SSFM_P1": "p1",
              "PLX_SSFM_LOGW": "logW", "PLX_SSFM_COL_THREADS": "col_threads", "PLX_SSFM_ROWR": "rowr", "PLX_SSFM_ROWSM": "rowsm",
              "PLX_SSFM_ROW256_SPLIT": "row256_split", "PLX_SSFM_ROW4K_SPLIT": "row4k_split", "PLX_SSFM_ROWG_SPLIT": "rowg_split",
              "PLX_SSFM_NO_PMD_TAB": "no_pmd_tab", "PLX_SSFM_STORE_LATE": "store_late", "PLX_SSFM_ROW_REV": "row_rev",
              "PLX_SSFM_SAFE_LANDING": "safe_landing", "PLX_SSFM_NO_FUSE": "no_fuse"}

    def __init__(self, monkeypatch):
        self.mp = monkeypatch
        self.fields = {}

    def _push(self):
        import ctypes as C
        from polmux_amd import _abi
        for b in _abi.bindings():
            if self.fields:
                b.call("plx_ssfm_tuning_override", C.byref(b.tuning(**self.fields)))
            else:
                b.call("plx_ssfm_tuning_override", None)

    def setenv(self, name, value):
        if name in self.FIELDS:
            self.fields[self.FIELDS[name]] = int(value)
            self._push()
        else:
            self.mp.setenv(name, value)

    def delenv(self, name, raising=True):
        if name in self.FIELDS:
            self.fields.pop(self.FIELDS[name], None)
            self._push()
        else:
            self.mp.delenv(name, raising=raising)

    def __getattr__(self, name):          # (everything else is monkeypatch's)
        return getattr(self.mp, name)


@pytest.fixture
def tune(monkeypatch):
    t = _Tuner(monkeypatch)
    yield t
    t.fields = {}
    t._push()

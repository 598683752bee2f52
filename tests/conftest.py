import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built (reference checkout absent and no prebuilt .so)")
    return Reference()


# ---- forcing a block->hardware mapping (lz4hip_tuning_set) AND proving that it ran (lz4hip_dispatch_counts) --------
_FAMILY = {"LZ4HIP_DECODER": (0, 1), "LZ4HIP_ENCODER": (2, 3), "LZ4HIP_HC": (4, 5)}   # (wave counter, lane counter)
_KNOB = {"LZ4HIP_DECODER": "decoder", "LZ4HIP_ENCODER": "encoder", "LZ4HIP_HC": "hc"}


class ForcedMapping:
    """with ForcedMapping("LZ4HIP_DECODER", "lane"): ...  -- sets the override and, on exit, asserts that the kernel
    family named was launched at least once and its sibling mapping not at all."""

    def __init__(self, var, which, must_run=True):
        self.var, self.which, self.must_run = var, which, must_run

    def __enter__(self):
        from lz4net_amd import _lib
        self.prev = _lib.tuning_set(_KNOB[self.var], self.which)
        self.before = _lib.dispatch_counts()
        return self

    def __exit__(self, exc_type, exc, tb):
        from lz4net_amd import _lib
        _lib.tuning_set(_KNOB[self.var], self.prev)
        if exc_type is not None:
            return False
        after = _lib.dispatch_counts()
        wave, lane = _FAMILY[self.var]
        mine, other = (wave, lane) if self.which == "wave" else (lane, wave)
        assert after[other] == self.before[other], f"{self.var}={self.which}: the OTHER mapping was launched"
        if self.must_run:
            assert after[mine] > self.before[mine], f"{self.var}={self.which}: the mapping named was never launched"
        return False


def _forced_fixture(var, which, must_run=True):
    with ForcedMapping(var, which, must_run) as f:
        yield which

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Collection order: the single-process parity suites first, the environment API next,
# anything that starts other processes last - under the driver's `pytest -x` a harness-level failure of a
# multi-process test must not hide the parity evidence of the core rows.
_ORDER = ("test_hip_parity", "test_configs_gpu", "test_dm_gpu", "test_amp_shard_gpu", "test_cobyla_emulation", "test_mps2qc_gpu", "test_step_traces",
          "test_env_gpu")
_LAST = ("test_distributed_cpu", "test_zz_multirank_gpu")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if mod in _ORDER:
            return (0, _ORDER.index(mod))
        if mod in _LAST:
            return (2, _LAST.index(mod))
        return (1, 0)
    items.sort(key=key)          # stable: the order inside a module is kept

import os
import sys

import pytest

try:  # one HIP runtime per process: when torch is there it has to be loaded before libmolahip / the pybind module,
    import torch  # noqa: F401  whatever order the test files run in (capi.lib() does the same for ctypes users)
except Exception:  # noqa: BLE001
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout(seconds): per-test limit (pytest-timeout when installed; a hung device call must "
                            "not eat the whole GPU session)")


def pytest_collection_modifyitems(config, items):
    # every GPU test gets a limit of its own (a hung device call or a stuck thread fails ONE test instead of eating the
    # whole GPU session); explicit @pytest.mark.timeout marks win
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def small_workload():
    from mola_lidar_odometry_amd import synth
    return synth.workload_small()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c

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


def pytest_runtest_setup(item):
    # MH_MATCH=t|w|o name the matcher families that lost to the product kernels (tile matcher with the map records in LDS, wave
    # matcher, sorted scan): compiled into the DEVELOPMENT library only (tools/build_variants.sh; run the suite on it with
    # MOLAHIP_LIB_PATH=tools/variants/libmolahip_dev.so).  On the shipped library their parametrisations are skipped.
    cs = getattr(item, "callspec", None)
    if cs is None or item.get_closest_marker("gpu") is None:
        return
    wanted = [v for k, v in cs.params.items() if k == "match" and isinstance(v, str)]
    wanted += [v["MH_MATCH"] for v in cs.params.values() if isinstance(v, dict) and "MH_MATCH" in v]
    if any(w[:1] in ("t", "w", "o") for w in wanted):
        from mola_lidar_odometry_amd import capi
        if not capi.dev_variants():
            pytest.skip("development matcher (MH_MATCH=%s): not in the shipped library -- tools/build_variants.sh" % wanted[0])


@pytest.fixture(scope="session")
def small_workload():
    from mola_lidar_odometry_amd import synth
    return synth.workload_small()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c

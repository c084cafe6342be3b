import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the tests' mini AIR is test infrastructure (tests/cpp/mini_air_lib.cpp): registered with the host bindings, built at first use
    from tests import mini_air_host
    mini_air_host.register()
    if os.environ.get("SS_TEST_ORACLELIB"):                  # likewise a sanitizer build of the oracle's C
        from oracle import oracle_py
        oracle_py._SO = os.environ["SS_TEST_ORACLELIB"]
    if os.environ.get("SS_TEST_HOSTLIB"):
        # a sanitizer build of the C++ host (sandstorm_amd/host: plain C++) for a session of its own: see tests/hipemu/README.md
        from sandstorm_amd import hostlib
        hostlib.LIB_PATH = os.environ["SS_TEST_HOSTLIB"]
    if os.environ.get("SS_TEST_HIPEMU") == "1":
        # tests/test_device_code_on_host.py runs selected `gpu` tests in a process of their own against the HOST BUILD OF THE DEVICE
        # CODE (tests/hipemu: test infrastructure; same C ABI, kernels executed lane by lane on the CPU).  Only this test session
        # is pointed at it - the product's loader has no such switch.
        from sandstorm_amd import _lib
        _lib.LIB_PATH = os.environ.get("SS_TEST_HIPEMU_LIB", os.path.join(ROOT, "tests", "hipemu", "_build", "libsandstorm_hipemu.so"))
        # the product's Pedersen table is sized for HBM (24-bit windows: 23.6 GB); the emulated device's memory is this host's
        os.environ.setdefault("SS_PED_WINDOW", "16")
        # ... and a cross-lane read costs the emulator two workgroup barriers: the 32-lanes-per-hash kernel (a thousand of them per lane)
        # runs the tree levels of <= 128 hashes here, not <= 8192
        os.environ.setdefault("SS_PED_SMALL_MAX", "128")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/_build/liboracle.so via ctypes."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def golden():
    import json
    gdir = os.path.join(ROOT, "tests", "golden")

    def load(name):
        with open(os.path.join(gdir, name)) as f:
            return json.load(f)
    return load

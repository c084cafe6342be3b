"""TEST INFRASTRUCTURE: the tests' mini AIR (tests/mini_air.py) as an `Air` of the C++ host - tests/cpp/mini_air_lib.cpp built into
tests/_build/libsandstorm_test_air.so against the product's libsandstorm_host.so and registered with hostlib as AIR_MINI
(tests/conftest.py).  The product library itself holds the layouts' AIRs only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "mini_air_lib.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "libsandstorm_test_air.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        from sandstorm_amd import hostlib
        hostlib.load()                               # libsandstorm_hip.so (or the emulated device code) and the host library first
        build = os.path.join(ROOT, "sandstorm_amd", "_build")
        host = os.path.join(build, "libsandstorm_host.so")
        if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(host)):
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", OUT, SRC, "-L" + build, "-lsandstorm_host", "-Wl,-rpath," + build])
        _lib = C.CDLL(OUT, mode=C.RTLD_GLOBAL)
        _lib.sst_mini_air_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    return _lib


def create(ctx_handle):
    h = C.c_void_p()
    if load().sst_mini_air_create(ctx_handle, C.byref(h)) != 0:
        raise RuntimeError("mini AIR: creation failed")
    return h.value


def register():
    from sandstorm_amd import hostlib
    hostlib.register_air_kind(hostlib.AIR_MINI, create)

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "zhusuan-pytorch_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG_ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---------------------------------------------------------------------------------------------
# Kernel back-ends for product-level tests.
#   "hip"  : the real thing -- zhusuan package + libzs_hip.so on cuda:0 (marked gpu).
#   "host" : the SAME package code with the plain-C oracle (oracle/zs_oracle_c.c, identical C ABI,
#            host pointers) patched in by tests/host_backend.py (the package has no such routing), so
#            the host logic (shapes, reductions, autograd wiring, ctypes marshalling) is covered on a
#            CPU-only machine.
# ---------------------------------------------------------------------------------------------
import subprocess

import torch

import host_backend

# ZS_ORACLE_LIBRARY: an alternative build of the C oracle (tests/test_sanitizers.py points it at the ASan + UBSan build)
ORACLE_SO = os.environ.get("ZS_ORACLE_LIBRARY") or os.path.join(ROOT, "oracle", "_build", "libzs_oracle.so")


def build_oracle_lib():
    if os.environ.get("ZS_ORACLE_LIBRARY"):
        return ORACLE_SO
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("zs_oracle_c.c", "zs_oracle_impl.inc")] + \
           [os.path.join(ROOT, "include", "zs_hip.h")]
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_host_klib = None


def host_kernel_library():
    global _host_klib
    if _host_klib is None:
        from zhusuan import _hip
        _host_klib = _hip.KernelLibrary(build_oracle_lib())
    return _host_klib


@pytest.fixture(params=[pytest.param("host"), pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    from zhusuan import _hip
    if request.param == "host":
        host_backend.install(host_kernel_library())
        try:
            yield torch.device("cpu")
        finally:
            host_backend.uninstall()
    else:
        host_backend.uninstall()
        assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
        yield torch.device("cuda:0")

"""pytest plugin used only by tests/test_reference_suite.py: makes `import zhusuan` resolve to THIS package
and routes its kernel calls to the CPU oracle library, so the reference's own unittest files can run
against it on a GPU-less machine."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd")):
    if p in sys.path:
        sys.path.remove(p)
    sys.path.insert(0, p)


def pytest_configure(config):
    import zhusuan
    assert zhusuan.__file__.startswith(os.path.join(ROOT, "zhusuan-pytorch_amd")), zhusuan.__file__
    from zhusuan import _hip
    import conftest
    import host_backend
    host_backend.install(conftest.host_kernel_library())

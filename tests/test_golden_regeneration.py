"""The committed fixtures are exactly what tests/golden/gen_golden.py produces from the real reference.

Build container only (the reference checkout is not present on the GPU box): the generator is run into a temporary
directory in a subprocess (no bytecode is written next to the reference's sources) and every array of every fixture is
compared bit for bit with the committed one.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "zhusuan")), reason="reference checkout not present")
def test_goldens_regenerate_bit_identically(tmp_path):
    code = (
        "import importlib.util, sys\n"
        "sys.dont_write_bytecode = True\n"
        "spec = importlib.util.spec_from_file_location('gg', %r)\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "m.OUT = %r\n"
        "for fn in [n for n in dir(m) if n.startswith('gen_')]:\n"
        "    getattr(m, fn)()\n" % (os.path.join(GOLDEN, "gen_golden.py"), str(tmp_path)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert made == committed
    n = 0
    for f in made:
        a, b = np.load(os.path.join(tmp_path, f)), np.load(os.path.join(GOLDEN, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            x, y = a[k], b[k]
            assert x.shape == y.shape and x.dtype == y.dtype, (f, k)
            assert np.array_equal(x, y, equal_nan=(x.dtype.kind in "fc")), (f, k)
            n += 1
    assert n > 2500

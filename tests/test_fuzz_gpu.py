"""Short runs of the randomised HIP-vs-oracle tools (tools/fuzz_hotpath.py, tools/fuzz_layers.py): a few hundred random shapes of
every kernel through the C ABI per test run (fixed seed: the suite stays deterministic).  The long runs with other seeds are in
profiles/ (r03_fuzz.txt)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("tool", ["fuzz_hotpath.py", "fuzz_layers.py"])
def test_random_shapes_against_the_oracle(tool):
    seed = 3
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "12", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no mismatch" in r.stdout, (r.stdout + r.stderr)[-2000:]

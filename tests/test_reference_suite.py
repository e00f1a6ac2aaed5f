"""The reference's OWN unittest files for this path, run unmodified against this package.

Only where the reference checkout exists (the build container); skipped elsewhere.  The files
(test/variational/test_elbo.py, test_iw.py, test/distributions/test_normal.py, test_bernoulli.py, test_logistic.py,
test_uniform.py, the six pass-through families' test files, test_base.py and their helper modules) are copied to a temporary directory at run time -- nothing of them is kept in this repository --
and executed in a subprocess whose `zhusuan` is THIS package with the CPU oracle library as kernel back-end
(tests/ref_plugin.py).  They use CPU tensors, so this is the host-logic / drop-in check: same constructor
errors, shapes, dtypes (incl. float64), scipy known answers, analytic-KL gradient tests, vimco-vs-sgvb test.
"""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT, build_oracle_lib

REF = "/root/reference"
# the six torch.distributions wrapper families (off the hot path: plain pass-throughs here too) and the base class
PASS_THROUGH = ["beta", "exponential", "gamma", "laplace", "poisson", "studentT", "base"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "test")), reason="reference checkout not present")
def test_reference_unittests_pass_against_this_package(tmp_path):
    build_oracle_lib()
    work = tmp_path / "work"
    (work / "test").mkdir(parents=True)
    for rel in ["__init__.py", "variational/__init__.py", "variational/utils.py", "variational/test_elbo.py",
                "variational/test_iw.py", "distributions/__init__.py", "distributions/utils.py",
                "distributions/test_normal.py", "distributions/test_bernoulli.py", "distributions/test_logistic.py",
                "distributions/test_uniform.py"] + ["distributions/test_%s.py" % f for f in PASS_THROUGH]:
        dst = work / "test" / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copyfile(os.path.join(REF, "test", rel), dst)
    env = dict(os.environ)
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    env["MPLBACKEND"] = "Agg"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), os.path.join(ROOT, "zhusuan-pytorch_amd"), str(work)])
    cmd = [sys.executable, "-m", "pytest", "-p", "ref_plugin", "-p", "no:cacheprovider", "--rootdir", str(work), "-q",
           "-W", "ignore", str(work / "test" / "variational"), str(work / "test" / "distributions" / "test_normal.py"),
           str(work / "test" / "distributions" / "test_bernoulli.py"), str(work / "test" / "distributions" / "test_logistic.py"),
           str(work / "test" / "distributions" / "test_uniform.py")] + \
        [str(work / "test" / "distributions" / ("test_%s.py" % f)) for f in PASS_THROUGH]
    r = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    n_passed = int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    # 3 (elbo) + 3 (iw) + 11 (normal) + 7 (bernoulli) + 9 (logistic) + 9 (uniform) = 42 on the kernel-backed path,
    # + 48 for the six torch.distributions pass-through families and test_base.py
    assert n_passed >= 90, tail

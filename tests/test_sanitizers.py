"""AddressSanitizer + UndefinedBehaviorSanitizer run of the C oracle (CPU build only: GPU sanitizers are not available on
this pool).  The oracle's own tests -- golden fixtures, ragged / empty / unaligned shapes, every entry point of the C ABI,
fp32 and fp64 -- are re-run in a child process against ``oracle/_build/libzs_oracle_asan.so``; any out-of-bounds access,
use of uninitialised stack, signed overflow or misaligned access aborts that process."""
import os
import subprocess
import sys

from conftest import ROOT


def test_c_oracle_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "_build", "libzs_oracle_asan.so")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(asan_rt) and os.path.exists(asan_rt), "gcc's libasan.so not found"
    env = dict(os.environ)
    env.update({"LD_PRELOAD": asan_rt, "ZS_ORACLE_LIBRARY": lib,
                # python itself "leaks" at exit; everything else is fatal
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider",
           "-k", "oracle or host",
           os.path.join(ROOT, "tests", "test_cabi.py"), os.path.join(ROOT, "tests", "test_locscale.py"),
           os.path.join(ROOT, "tests", "test_reinforce.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail

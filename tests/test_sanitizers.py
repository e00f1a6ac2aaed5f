"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on this pool):
(1) the kernels' own per-element arithmetic -- csrc/zs_common.h and csrc/zs_iw_math.h are __host__ __device__ -- compiled for
the host and checked against double-precision restatements (tests/host_math/zs_host_math.hip); (2) the C oracle.  The oracle's own tests -- golden fixtures, ragged / empty / unaligned shapes, every entry point of the C ABI,
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


def test_kernel_arithmetic_on_the_host_under_asan_and_ubsan(tmp_path):
    """The HIP sources' per-element helpers (Philox4x32-10, the uniform and Box-Muller conversions, the Normal / Bernoulli
    / Logistic / Uniform density terms and derivatives, the VIMCO per-particle arithmetic, the index helpers) built for the HOST with
    `hipcc --cuda-host-only -fsanitize=address,undefined` and run: Random123 known answers, double-precision references,
    edge values.  (This harness found round 1's uniform reaching exactly 1.0 -- an infinite Logistic draw once in 1.7e7.)"""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "zs_host_math")
    src = os.path.join(ROOT, "tests", "host_math", "zs_host_math.hip")
    r = subprocess.run([hipcc, "-x", "hip", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=all", src, "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host math ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert int(r.stdout.split("ok:")[1].split()[0]) > 100000

"""Properties of the GENERATED CODE that the fused objective kernel's speed rests on and that a harmless-looking source edit
destroys silently (cdna_hip_programming.md: read your kernel's disassembly).  No GPU needed: hipcc cross-compiles gfx950.

k_iw1_persist (csrc/zs_iwpersist.h) streams its rows through two register buffers with loads that the COMPILER counts
(s_waitcnt vmcnt(N)): the prefetched rows stay in flight only while every path through the once-per-datapoint code (tail,
staging, the batch mean's atomics) leaves no load "possibly pending" in the compiler's bookkeeping.  One load whose use sits behind
a lane mask or a second `if` on the same flag is enough for the loop header to drain everything -- `s_waitcnt vmcnt(0)` on every
iteration, a third of the kernel's time -- with results unchanged.  It happened four times while the kernel was written."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "zhusuan-pytorch_amd", "csrc")


@pytest.fixture(scope="module")
def bernoulli_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = str(tmp_path_factory.mktemp("isa") / "zs_bernoulli.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        os.path.join(CSRC, "zs_bernoulli.hip"), "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read()


def _kernel(asm, mangled):
    i = asm.index(mangled + ":")
    j = asm.index(".end_amdhsa_kernel", i)
    return asm[i:j], asm[j:asm.index("\n\t.text", j) if "\n\t.text" in asm[j:] else j + 4000]


@pytest.mark.parametrize("mangled", ["_ZN2zs13k_iw1_persistILb0ELb0EEEvNS_7Iw1ArgsE", "_ZN2zs13k_iw1_persistILb1ELb0EEEvNS_7Iw1ArgsE"])
def test_iw1_streaming_loop_keeps_its_prefetched_rows_in_flight(bernoulli_asm, mangled):
    body, _ = _kernel(bernoulli_asm, mangled)
    lines = body.split("\n")
    headers = [n for n, l in enumerate(lines) if "Loop Header: Depth=1" in l]
    assert headers, "no loop found in %s" % mangled

    def scan(start):            # waits and row loads from a loop header to the first barrier behind it
        waits, loads = [], 0
        for l in lines[start:]:
            t = l.strip()
            if t.startswith("s_barrier"):
                break
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
            if m:
                waits.append(int(m.group(1)))
            if t.startswith("global_load_dwordx4"):
                loads += 1
        return waits, loads
    # the streaming loop: the depth-1 loop whose body requests rows (two rounds of five loads) ahead of its barrier
    found = [(h,) + scan(h) for h in headers]
    streaming = [f for f in found if f[2] == 10]
    assert len(streaming) == 1, [(f[0], f[2]) for f in found]
    _, waits, loads = streaming[0]
    # per round: the next row is requested (5 loads), then the row that has landed is waited for with those 5 left in flight
    assert waits and min(waits) >= 5, "the streaming loop drains its prefetch: s_waitcnt vmcnt(%d) before its barrier" % min(waits)
    # and no register of the loop lives in scratch
    assert not re.search(r"\bscratch_(load|store)", body), "k_iw1_persist spills to scratch"
    assert "flat_load" not in body and "flat_store" not in body and "flat_atomic" not in body, "generic-address memory instructions (lost address space)"


def test_the_laboratory_patch_applies_to_the_release_sources(tmp_path):
    """csrc/ holds the kernels that ship and nothing else: no ZS_EXPERIMENTS block outside the knob reader; the
    laboratory is tools/lab/csrc_lab.patch, which `make experiments` applies to a COPY of the sources.  It must keep applying (the
    GPU suite also builds it and compares its last-arrival finish with the release watcher bit for bit)."""
    import glob
    for f in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")):
        n = open(f).read().count("ZS_EXPERIMENTS")
        assert n == 0 or os.path.basename(f) == "zs_common.h", (f, n)
    dst = tmp_path / "zhusuan-pytorch_amd" / "csrc"
    dst.mkdir(parents=True)
    for f in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")):
        shutil.copy(f, str(dst))
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "tools", "lab", "csrc_lab.patch")], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]

#!/usr/bin/env python
"""Are the committed profiles from the kernels of this tree?  Compares the sha256 of zhusuan-pytorch_amd/lib/libzs_hip.so (built in
the tree: `make -C zhusuan-pytorch_amd/csrc`) with the `library.sha256` the bench line of profiles/<tag>_bench_n1.json recorded on
the GPU box (the built library travels there with the tree).  Exit code 1 when they differ: regenerate with
`gpurun --timeout 2400 -- 'bash tools/gpu_round_profiles.sh <tag>; bash tools/gpu_final_check.sh <tag>'` and copy gpurun_out/<tag>_* to
profiles/."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
lib = os.path.join(ROOT, "zhusuan-pytorch_amd", "lib", "libzs_hip.so")
line = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_n1.json" % tag)))
have = hashlib.sha256(open(lib, "rb").read()).hexdigest()
want = line["library"]["sha256"]
print("library in the tree      %s" % have)
print("library of the profiles  %s  (ABI %s, %s)" % (want, line["library"]["abi"], line["library"]["build"]))
print("FRESH" if have == want else "STALE: the kernels changed after the profiles were taken")
sys.exit(0 if have == want else 1)

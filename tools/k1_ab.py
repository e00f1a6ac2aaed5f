#!/usr/bin/env python
"""K1 (fused Normal sample + log-density, in-kernel Philox) at 1 M and 4.2 M rows at the sustained clock -- bench.py's own
measurement (k1_resident) -- for whichever library ZS_HIP_LIBRARY names: the harness of same-box A/B runs of generator variants."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench          # noqa: E402
import torch          # noqa: E402
from zhusuan import _hip      # noqa: E402

r = bench.k1_resident(_hip.lib(), torch.device("cuda", 0))
print(os.environ.get("ZS_HIP_LIBRARY", "release"), json.dumps(dict((k, round(v["frac_of_hbm_peak"], 4)) for k, v in r.items())),
      json.dumps(dict((k, round(v["median_us"], 1)) for k, v in r.items())))

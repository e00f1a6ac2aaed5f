#!/usr/bin/env python
"""A/B of the update step inside the graph-replayed training steps of configs C2 / C3 / C5 in ONE process (box-to-box
variance on this pool is larger than the differences): torch's multi-tensor Adam against zhusuan.optim.FlatAdam."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch

import bench

dev = torch.device("cuda:0")
for rep in range(2):
    for name in ("c2", "c3", "c5"):
        row = []
        for label, torch_adam in (("torch.optim.Adam(fused, capturable)", True), ("FlatAdam", False)):
            r = bench.run_single_gpu_config(name, dev, 100, 10, tuned=True, torch_adam=torch_adam)
            row.append("%s %.4f ms" % (label, r["ms_per_step"]))
        print(name, " | ".join(row), flush=True)

#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
for i in 1 2 3; do
  timeout 120 python tools/k1_ab.py
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_oldk1.so timeout 120 python tools/k1_ab.py
done
} 2>/dev/null | tee gpurun_out/r05_k1_ab.txt

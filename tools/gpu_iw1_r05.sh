#!/bin/bash
# round 5: the persistent IW1 forward -- correctness first, then timing against round 4's kernel (experiments build) in ONE box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_iw_fused.py tests/test_cabi.py -x -q -m gpu -k "iw" 2>&1 | tail -25 > gpurun_out/r05_iw1_tests.log
tail -5 gpurun_out/r05_iw1_tests.log
{
  echo "== release library (persistent kernel)"
  timeout 120 python tools/iw1_timing.py
  echo "== experiments build, ZS_IW1_BLOCK_KERNEL=1 (round 4's workgroup-per-datapoint kernel)"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_BLOCK_KERNEL=1 timeout 120 python tools/iw1_timing.py
  echo "== release library again"
  timeout 120 python tools/iw1_timing.py
  echo "== phases (experiments build, persistent kernel)"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so timeout 120 python tools/iw1_phases.py 256 50 512 50 1024 50 2048 10
} > gpurun_out/r05_iw1_timing.txt 2>&1
cat gpurun_out/r05_iw1_timing.txt

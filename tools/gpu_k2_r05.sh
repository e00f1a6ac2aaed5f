#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cabi.py -x -q -m gpu -k "contiguous_shares or logprob_periods" 2>&1 | tail -6
{
for i in 1 2; do
  echo "== balanced (release default)"
  timeout 200 python tools/kernel_sweep.py --only "K2" --only "L2" --only "U2" --only "eps given" --batches 41943 83886
  echo "== items (ZS_K2_BALANCED=0, experiments build)"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_K2_BALANCED=0 timeout 200 python tools/kernel_sweep.py --only "K2" --only "L2" --only "U2" --only "eps given" --batches 41943 83886
done
} 2>/dev/null | tee gpurun_out/r05_k2_balanced.txt | grep -v "^$" | tail -60

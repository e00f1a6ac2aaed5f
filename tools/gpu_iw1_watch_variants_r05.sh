#!/bin/bash
# watcher variants (experiments build): number of watching waves x every sample worked out in full, isolated timings at B = 256 / 512
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_iw_fused.py -x -q -m gpu 2>&1 | tail -2
t() {
  label="$1"; shift
  printf "%-46s" "$label"
  env "$@" timeout 300 python tools/iw1_timing.py 2>/dev/null | grep -E "^B=(256|512|2048  K=10)" | awk '{printf "  %s %s: %s", $1, $2, $10}'
  echo
}
{
for rep in 1 2 3; do
  t "release (1 watcher)" ZS_NONE=1
  t "exp, last arrival finishes" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_SHARDED=1
  t "exp, 1 watcher" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so
  t "exp, 1 watcher, every sample in full" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_WATCH_ALL=1
  t "exp, 2 watchers" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_WATCHERS=2
  t "exp, 2 watchers, every sample in full" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_WATCHERS=2 ZS_IW1_WATCH_ALL=1
  t "exp, 4 watchers" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_WATCHERS=4
  t "exp, 4 watchers, every sample in full" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_WATCHERS=4 ZS_IW1_WATCH_ALL=1
done
} | tee gpurun_out/r05_iw1_watch_variants.txt

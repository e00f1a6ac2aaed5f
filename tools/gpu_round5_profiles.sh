#!/bin/bash
# Everything under profiles/r05_* that is not the final check: IW1 timing (release vs round 4's block kernel, warm and cold), its phase
# stamps, the in-step A/B, the counter passes.   gpurun --timeout 1800 -- 'bash tools/gpu_round5_profiles.sh'
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== release library (persistent kernel)"
  timeout 120 python tools/iw1_timing.py
  echo "== experiments build, ZS_IW1_BLOCK_KERNEL=1 (round 4's workgroup-per-datapoint kernel, with this round's K-particle reduction)"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_BLOCK_KERNEL=1 timeout 120 python tools/iw1_timing.py
  echo "== experiments build, ZS_IW1_SHARDED=1 (persistent kernel, the batch mean finished by the last arrival: two-level count, as before the watcher)"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_SHARDED=1 timeout 120 python tools/iw1_timing.py
  echo "== experiments build, watcher (the release build's mode; the experiments build carries the stamps)"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so timeout 120 python tools/iw1_timing.py
  echo "== release library again"
  timeout 120 python tools/iw1_timing.py
  echo "== release library, cold (512 MB fill between launches)"
  timeout 120 python tools/iw1_timing.py --cold
  echo "== block kernel, cold"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_BLOCK_KERNEL=1 timeout 120 python tools/iw1_timing.py --cold
} 2>/dev/null > gpurun_out/r05_iw1_timing.txt
ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so timeout 200 python tools/iw1_phases.py 2>/dev/null > gpurun_out/r05_iw1_phases.txt
ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so timeout 200 python tools/iw1_phases_instep.py 2>/dev/null | grep -v "UserWarning\|run_backward" > gpurun_out/r05_iw1_phases_instep.txt
bash tools/gpu_iw1_instep_r05.sh > /dev/null
SUFFIX=_all bash tools/gpu_iw1_sizes_instep_r05.sh > /dev/null
{ timeout 400 python tools/fuzz_hotpath.py 300 5 2>&1 | grep -v amdgpu.ids | tail -3; echo ----; timeout 200 python tools/fuzz_layers.py 100 5 2>&1 | grep -v amdgpu.ids | tail -3; } > gpurun_out/r05_fuzz.txt
bash tools/gpu_pmc_r05.sh > gpurun_out/r05_pmc.log 2>&1
timeout 300 python tools/kernel_sweep.py --out gpurun_out/r05_kernel_sweep.json 2>/dev/null > gpurun_out/r05_kernel_sweep.txt
tail -3 gpurun_out/r05_iw1_timing.txt; tail -4 gpurun_out/r05_iw1_instep.txt

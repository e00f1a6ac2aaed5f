#!/bin/bash
# the batch mean finished by a watching wave (release) against the last-arrival finisher (experiments build, ZS_IW1_SHARDED=1): parity tests,
# isolated timings, inside the graph-replayed step, phase stamps
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_iw_fused.py tests/test_one_launch.py tests/test_properties_gpu.py -x -q -m gpu 2>&1 | tail -4
run() {
  label="$1"; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --allow-experiments --full-record gpurun_out/_instep_full.json > gpurun_out/_instep.json 2>/dev/null
  python - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_instep.json")); f = json.load(open("gpurun_out/_instep_full.json"))
k = f["hip_kernels"]
print("%-44s step %.4f ms | IW1 fwd %.2f us  IW1 bwd %.2f us  K1 pair %.2f us  Adam %.2f us" % (
    sys.argv[1], d["ms_per_step"], k["zs_bernoulli_iw_objective_f32"]["avg_us"], k["zs_bernoulli_iw_objective_bwd_f32"]["avg_us"],
    k["zs_normal_sample_logprob_pair_f32"]["avg_us"], k["zs_adam_step_f32"]["avg_us"]))
PY
}
{
for i in 1 2; do
  run "release (watcher)" ZS_NONE=1
  run "experiments build, watcher" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so
  run "experiments build, last arrival finishes" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_SHARDED=1
done
echo "== isolated (tools/iw1_timing.py), release"
timeout 300 python tools/iw1_timing.py 2>/dev/null | grep -v amdgpu.ids | head -12
echo "== isolated, experiments build, last arrival finishes"
ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_SHARDED=1 timeout 300 python tools/iw1_timing.py 2>/dev/null | grep -v amdgpu.ids | head -12
echo "== stamps, watcher"
ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so timeout 300 python tools/iw1_phases_instep.py 2>/dev/null | grep -v "amdgpu.ids\|UserWarning\|run_backward"
} | tee gpurun_out/r05_iw1_watcher.txt

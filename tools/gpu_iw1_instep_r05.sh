#!/bin/bash
# IW1 forward INSIDE the graph-replayed training step (the figure bench.py reports as iw1_fwd_frac): variants on one box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
run() {
  label="$1"; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --allow-experiments --full-record gpurun_out/_instep_full.json > gpurun_out/_instep.json 2>/dev/null
  python - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_instep.json")); f = json.load(open("gpurun_out/_instep_full.json"))
k = f["hip_kernels"]
print("%-34s step %.4f ms | IW1 fwd %.2f us  IW1 bwd %.2f us  K1 pair %.2f us  Adam %.2f us" % (
    sys.argv[1], d["ms_per_step"], k["zs_bernoulli_iw_objective_f32"]["avg_us"], k["zs_bernoulli_iw_objective_bwd_f32"]["avg_us"],
    k["zs_normal_sample_logprob_pair_f32"]["avg_us"], k["zs_adam_step_f32"]["avg_us"]))
PY
}
{
for i in 1 2; do
  run "release (persistent)" ZS_NONE=1
  run "round-4 block kernel" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_BLOCK_KERNEL=1
  run "persistent, last arrival finishes" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_SHARDED=1
done
} | tee gpurun_out/r05_iw1_instep.txt

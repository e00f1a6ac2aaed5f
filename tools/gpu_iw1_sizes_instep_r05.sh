#!/bin/bash
# the fused generator-side launch against K3 + K2 + K4b INSIDE the graph-replayed step at batch sizes beyond the config's (where does the
# stream stop coming out of the Infinity Cache?)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
run() {
  label="$1"; shift
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 5 --full-record gpurun_out/_sz_full.json "$@" > gpurun_out/_sz.json 2>/dev/null
  python - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_sz.json")); f = json.load(open("gpurun_out/_sz_full.json"))
k = f["hip_kernels"]
names = ["zs_bernoulli_iw_objective_f32", "zs_bernoulli_logprob_f32", "zs_normal_logprob_f32", "zs_iw_objective_f32"]
print("%-28s step %.4f ms | %s" % (sys.argv[1], d["ms_per_step"], "  ".join("%s %.1f" % (n.replace("zs_", "").replace("_f32", ""), k[n]["us_per_step"]) for n in names if n in k)))
PY
}
{
for B in ${SIZES:-512 1024 1280 1536 2048}; do
  run "B=$B fused" --batch-per-gpu $B --iw1-max-stream-bytes 1099511627776
  run "B=$B K3+K2+K4b" --batch-per-gpu $B --iw1-max-stream-bytes 0
done
} | tee gpurun_out/r05_iw1_sizes_instep${SUFFIX}.txt

#!/bin/bash
# round 5 mid-round check: full GPU suite, K1 at 1 M / 4.2 M rows, IW1 backward row-limit A/B, bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r05_gputests.log
tail -4 gpurun_out/r05_gputests.log
{
  echo "== IW1 backward: one launch up to 32768 rows (release) vs up to 200000 rows (experiments build, ZS_IW1_BWD_ROWS)"
  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so ZS_IW1_BWD_ROWS=200000 timeout 200 python tools/iw1_timing.py
} > gpurun_out/r05_iw1_bwd_rows.txt 2>&1
grep -E "B=1024|B=2048  K=50" gpurun_out/r05_iw1_bwd_rows.txt
timeout 900 python bench.py > gpurun_out/r05_bench_c.json 2> gpurun_out/r05_bench_c.err
tail -c 300 gpurun_out/r05_bench_c.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_bench_c.json"))
print(d["value"], d["ms_per_step"], {k:v for k,v in d["roofline"].items() if "frac" in k})
f=json.load(open("bench_full.json"))
for k,v in f["hip_kernels"].items(): print(k, round(v["avg_us"],2))
PY

#!/bin/bash
# developer script: kernel census of the BNN and VAE steps
R="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for c in "C5/GPU BNN" "C2 VAE SGVB B=512 K=1"; do
  tag=$(echo "$c" | tr -c 'A-Za-z0-9' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/bench_configs.py --steps 100 --no-cpu --only "$c" > $R/gpurun_out/prof_$tag.log 2>&1
  f=$(ls $R/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
  echo "== $c"; grep "ms_per_step" $R/gpurun_out/prof_$tag.log | cut -c1-200
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot_calls = sum(int(r['Calls']) for r in rows); tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernels launched:", tot_calls, " total GPU ms:", round(tot/1e6, 2))
for r in rows[:22]:
    print("  %-88s calls=%6s avg_us=%7.2f pct=%5s" % (r['Name'][:88], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done

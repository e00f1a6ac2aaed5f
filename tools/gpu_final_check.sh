# Round-end GPU verification: build + smoke, the gpu test-suite, the PMC traffic passes (first: the bench line quotes them), the
# bench line and its rocprofv3 kernel-trace summary.  Run through gpurun from the repository root:  gpurun --timeout 2700 -- 'bash tools/gpu_final_check.sh r05'
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
TAG=${1:-r05}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16
timeout 600 python tools/soak.py 5000 > gpurun_out/soak_raw.txt 2>/dev/null; echo "soak rc=$?"; grep -v amdgpu.ids gpurun_out/soak_raw.txt > gpurun_out/${TAG}_soak.txt; cat gpurun_out/${TAG}_soak.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  # (--no-gemm-tuning: the counter passes are about the hot-path kernels; TunableOp's thousands of trial GEMMs would only bloat the CSVs)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/gpurun_out/pmc_$c" -o pmc -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-extras --no-gemm-tuning > "$ROOT/gpurun_out/pmc_$c.log" 2>&1
  echo "pmc $c rc=$?"
  rm -f "$ROOT/gpurun_out/pmc_$c/pmc_kernel_trace.csv"
done
cd "$ROOT" && python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE --out gpurun_out/${TAG}_pmc_traffic.json \
  --command "python3 bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-extras --no-gemm-tuning"
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # the bench line below quotes it (roofline.traffic)
python bench.py --steps 200 --warmup 20 --full-record gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --full-record gpurun_out/${TAG}_bench_full_driver_cmd.json > gpurun_out/${TAG}_bench_n1_driver_cmd.json 2>> gpurun_out/bench_n1.err; echo "bench (driver's command) rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof" -o bench -- python3 "$ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$ROOT/gpurun_out/bench_prof.json" 2> "$ROOT/gpurun_out/bench_prof.err"; echo "rocprof rc=$?"
rm -f "$ROOT/gpurun_out/prof/bench_kernel_trace.csv"     # 8 MB of per-dispatch rows; the stats file is the summary
cp "$ROOT/gpurun_out/prof/bench_kernel_stats.csv" "$ROOT/gpurun_out/${TAG}_bench_n1_kernel_stats.csv"
cd "$ROOT"
# keep the summaries small enough to travel back (gpurun merges at most 64 MiB): hot-path rows of the counter CSVs only
python - <<PY
import csv
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = "gpurun_out/pmc_%s/pmc_counter_collection.csv" % c
    rows = list(csv.reader(open(f)))
    ki = rows[0].index("Kernel_Name")
    keep = [r for r in rows[1:] if "anonymous namespace)::k_" in r[ki] or "zs::k_" in r[ki]]
    with open("gpurun_out/${TAG}_pmc_%s_counter_collection.csv" % c.lower(), "w", newline="") as fh:
        w = csv.writer(fh); w.writerow(rows[0]); w.writerows(keep)
PY
rm -rf gpurun_out/prof gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
du -sh gpurun_out

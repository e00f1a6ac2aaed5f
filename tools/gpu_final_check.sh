# Round-end GPU verification: build + smoke, the gpu test-suite, the bench line and its rocprofv3 kernel-trace summary.
# Run through gpurun from the repository root:  gpurun --timeout 2400 -- 'bash tools/gpu_final_check.sh'
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py --steps 200 --warmup 20 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof" -o bench -- python3 "$ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$ROOT/gpurun_out/bench_prof.json" 2> "$ROOT/gpurun_out/bench_prof.err"; echo "rocprof rc=$?"
rm -f "$ROOT/gpurun_out/prof/bench_kernel_trace.csv"     # 8 MB of per-dispatch rows; the stats file is the summary
ls "$ROOT/gpurun_out/prof"

R="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/k1pmc_$c -o k1 -- python3 $R/tools/k1_pmc.py > $R/gpurun_out/k1pmc_$c.log 2>&1
  echo "pmc $c rc=$?"; ls $R/gpurun_out/k1pmc_$c | head -5
done

# K1 (in-kernel Philox) counter passes at 1 M and 4.2 M rows -> profiles/r04_pmc_k1.json.  Run through gpurun from the repo root.
R="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_k1_sq -- python3 $R/tools/k1_pmc.py > /dev/null 2>&1
for c in GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_k1_$c -- python3 $R/tools/k1_pmc.py > /dev/null 2>&1
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc_k1_sq gpurun_out/pmc_k1_GRBM_GUI_ACTIVE gpurun_out/pmc_k1_FETCH_SIZE gpurun_out/pmc_k1_WRITE_SIZE --match "k_sample_tile<0" --out gpurun_out/r04_pmc_k1.json

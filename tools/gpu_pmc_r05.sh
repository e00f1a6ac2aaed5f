# Counter passes of round 5: the persistent IW1 forward at the config size and at four datapoints per workgroup (SQ counters, FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 runs, as MI355X_MICROARCH.md prescribes).   gpurun --timeout 1200 -- 'bash tools/gpu_pmc_r05.sh'
R="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
run() {   # tag which B kernel-name-fragment
  tag=$1; which=$2; B=$3; match=$4
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -- python3 $R/tools/pmc_kernels.py $which $B > $R/gpurun_out/pmc_${tag}.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -- python3 $R/tools/pmc_kernels.py $which $B > /dev/null 2>&1
  done
  (cd $R && python tools/pmc_summary.py gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE --match "$match" --out gpurun_out/r05_pmc_${tag}.json > /dev/null)
  grep "algorithmic" $R/gpurun_out/pmc_${tag}.log
  rm -rf $R/gpurun_out/pmc_${tag}_sq $R/gpurun_out/pmc_${tag}_FETCH_SIZE $R/gpurun_out/pmc_${tag}_WRITE_SIZE
}
run iw1_c3 iw1 256 "k_iw1_persist"
run iw1_1024 iw1 1024 "k_iw1_persist"
run iw1bwd_c3 iw1_bwd 256 "k_iw1_bwd"
ls $R/gpurun_out/r05_pmc_*.json

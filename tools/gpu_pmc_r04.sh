# Counter passes of round 4: SQ counters, FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs per target.
# gpurun --timeout 2400 -- 'bash tools/gpu_pmc_r04.sh'
R="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
run() {   # tag which B kernel-name-fragment
  tag=$1; which=$2; B=$3; match=$4
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -- python3 $R/tools/pmc_kernels.py $which $B > $R/gpurun_out/pmc_${tag}.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -- python3 $R/tools/pmc_kernels.py $which $B > /dev/null 2>&1
  done
  (cd $R && python tools/pmc_summary.py gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE --match "$match" --out gpurun_out/r04_pmc_${tag}.json > /dev/null)
  grep "algorithmic" $R/gpurun_out/pmc_${tag}.log
  rm -rf $R/gpurun_out/pmc_${tag}_sq $R/gpurun_out/pmc_${tag}_FETCH_SIZE $R/gpurun_out/pmc_${tag}_WRITE_SIZE
}
run k2_4M k2 83886 "k_logprob_tile<0"
run l2_4M l2 83886 "k_logprob_tile<1, true, false>"
run u2_4M u2 83886 "k_logprob_tile<2"
run k3bwd_6GB k3_bwd 20971 "k_bern_logprob_bwd"
run l1u_1M l1_u 20971 "k_logprob_tile<1, true, true>"
run k3logits_c3 k3_logits 256 "k_bern_logprob"
run k3logits_1M k3_logits 20971 "k_bern_logprob"
run k3probs_c3 k3 256 "k_bern_logprob"
run k3probs_1M k3 20971 "k_bern_logprob"
run iw1_c3 iw1 256 "k_iw1_block"
run iw1bwd_c3 iw1_bwd 256 "k_iw1_bwd"
ls $R/gpurun_out/r04_pmc_*.json

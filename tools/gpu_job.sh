#!/bin/bash
# The round's GPU jobs, one gpurun call each (round 6: ONE parametrised script instead of a gpu_*_rNN.sh per job):
#     gpurun --timeout 1500 -- 'bash tools/gpu_job.sh <job> [tag]'
# Everything is written under gpurun_out/<tag>_* ; copy what should be judged into profiles/.
#   dp_cost     the multi-rank step on one rank over RCCL against the single graph (same process), + timelines  -> profiles/r06_dp_step.txt
#   links       what one link between two hipGraph replays costs (event record, fork / join, all_reduce forms) -> profiles/r06_stream_links.txt
#   dp_soak     3 x 10 000 steps of each multi-rank form on one rank over RCCL; the final loss against the single graph's after as many steps -> profiles/r06_dp_soak.txt
#   dp_tests    the GPU tests of the data-parallel path, the graphs, the fused objective and the bench's control flow
#   iw1_ab      IW1 forward in the step and back to back: this tree against tools/_exp/libzs_hip_prev.so
#   k2_ab       K2 / L2 / U2 at 1 M and 4.2 M rows: this tree against tools/_exp/libzs_hip_prev.so
#   bench       the default bench line + full record
#   tests       the whole -m gpu suite
#   final       round-end verification: build + smoke, the gpu suite, soak, the two PMC traffic passes of the bench step, the bench line
#               (default and the driver's 20-step command), rocprofv3 --kernel-trace --stats of the same command   -> profiles/<tag>_*
#   profiles    the round's other profiles: IW1 timings (release / lab variants, warm and cold), phase stamps, fuzz, kernel sweep
#   pmc         one-kernel counter passes (SQ, FETCH_SIZE, WRITE_SIZE in separate rocprofv3 runs): IW1 both ways
#   k1_pmc      K1 (in-kernel Philox) counter passes at 1 M and 4.2 M rows
# (Round 5's one-off A/B scripts -- gpu_iw1_*_r05.sh, gpu_k1_ab_r05.sh, gpu_k2_r05.sh, ... -- are gone; what they measured is in
#  profiles/r05_* with the commands in docs/history.md; `make -C zhusuan-pytorch_amd/csrc experiments` still builds their library.)
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
JOB=${1:?job name}
TAG=${2:-r06}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0

one_rank() {   # bench.py as ONE rank with a process group over RCCL (the driver's launcher form)
  timeout ${ONE_RANK_TIMEOUT:-900} python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 \
    --no-cpu-baseline --no-extras "$@" 2>> gpurun_out/${TAG}_${JOB}.err
}

case "$JOB" in
dp_cost)
  # what the multi-rank step costs on ONE rank over RCCL before a byte crosses xGMI: each line is one process that times its
  # form AND the same model as a single graph, alternating (same_process); then a timeline of each form
  OUT=gpurun_out/${TAG}_dp_step.txt; : > $OUT
  sp() { python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s = r['same_process']
print('%-52s %.4f ms/step | same process: single graph %.4f, this form %.4f  (+%.1f us, ratio %.4f) | %s' % (sys.argv[1], r['ms_per_step'],
      s['single_graph_ms_per_step'], s['collective_path_ms_per_step'], s['extra_us_per_step'], s['collective_path_vs_single_graph'], r['collective_path'][:60]))" "$1"; }
  for rep in 1 2 3; do
    one_rank --steps 200 --warmup 20 --force-collective-path | sp "one bucket, 2 graphs, RCCL direct (default)" >> $OUT
    one_rank --steps 200 --warmup 20 --force-collective-path --no-direct-rccl | sp "one bucket, 2 graphs, torch.distributed" >> $OUT
    one_rank --steps 200 --warmup 20 --force-collective-path --overlap | sp "staged, 3 graphs (--overlap)" >> $OUT
  done
  for f in "default:--force-collective-path" "overlap:--force-collective-path --overlap"; do
    name=${f%%:*}; flags=${f#*:}
    one_rank --steps 50 --warmup 10 $flags --timeline gpurun_out/${TAG}_tl_$name.json > /dev/null
    { echo; echo "== timeline: $name ($flags)"; python tools/timeline_gaps.py gpurun_out/${TAG}_tl_$name.json 3.0; } >> $OUT
    rm -f gpurun_out/${TAG}_tl_$name.json
  done
  cat $OUT
  ;;
links)
  timeout 600 python tools/stream_link_probe.py 2> gpurun_out/${TAG}_links.err | tee gpurun_out/${TAG}_stream_links.txt
  ;;
dp_soak)
  # the eagerly launched RCCL calls between hipGraph replays, for N steps on end: no hang, no drift -- the objective after N updates
  # is compared with the single graph's after as many updates (3 timed trials of N; same seed, same GEMM picks: world size 1 sums one
  # rank's gradients).  The objective of ONE step is a noisy sample (+- 5 % from step to step this far into training on one resident
  # minibatch: profiles/r06_dp_training.txt, where the graphed forms equal the single graph bit for bit); the bar here is 15 %.
  N=${3:-10000}
  OUT=gpurun_out/${TAG}_dp_soak.txt; : > $OUT
  PICKS=gpurun_out/${TAG}_dp_soak_picks.csv; rm -f $PICKS
  fl() { python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-46s %6d steps  %.4f ms/step  final loss %.6f  | %s' % (sys.argv[1], r['steps'], r['ms_per_step'], r['final_loss'], (r.get('collective_path') or r['config']['launch_mode'])[:70]))" "$1"; }
  timeout 600 python bench.py --steps $N --warmup 10 --no-cpu-baseline --no-extras --gemm-picks $PICKS 2>> gpurun_out/${TAG}_${JOB}.err | fl "single graph (headline form)" >> $OUT
  one_rank --steps $N --warmup 10 --force-collective-path --gemm-picks $PICKS | fl "one bucket, 2 graphs, RCCL direct (default)" >> $OUT
  one_rank --steps $N --warmup 10 --force-collective-path --no-direct-rccl --gemm-picks $PICKS | fl "one bucket, 2 graphs, torch.distributed" >> $OUT
  one_rank --steps $N --warmup 10 --force-collective-path --overlap --gemm-picks $PICKS | fl "staged, 3 graphs (--overlap)" >> $OUT
  one_rank --steps $N --warmup 10 --force-collective-path --no-graph --gemm-picks $PICKS | fl "one bucket, eager launches" >> $OUT
  python - $OUT <<'PY' | tee -a $OUT
import sys
v = [float(l.split("final loss")[1].split()[0]) for l in open(sys.argv[1]) if "final loss" in l]
rel = max(abs(x - v[0]) / abs(v[0]) for x in v)
print("forms: %d; largest relative difference of the final loss from the single graph's: %.2e  -> %s" % (len(v), rel, "ok" if len(v) == 5 and rel < 0.15 else "DIFFERENT"))
PY
  # how the SAME single graph fares three times as long on its one resident minibatch (what the graphed forms above had behind them
  # when bench.py still read the objective after its same-process twin runs: 9 N steps, not 3 N)
  timeout 900 python bench.py --steps $((3 * N)) --warmup 10 --no-cpu-baseline --no-extras --gemm-picks $PICKS 2>> gpurun_out/${TAG}_${JOB}.err | fl "single graph, three times as long" | sed 's/final loss/objective at the end/' >> $OUT
  cat $OUT
  ;;
dp_tests)
  timeout 1700 python -m pytest -x -q -m gpu tests/test_dataparallel.py tests/test_graph.py tests/test_iw_fused.py tests/test_bench_contract.py \
     -k "not single_rank and not smoke and not strong" --durations=6 2>&1 | tail -25 > gpurun_out/${TAG}_dp_tests.log
  tail -25 gpurun_out/${TAG}_dp_tests.log
  ;;
iw1_ab)
  # IW1 forward INSIDE the graph-replayed step (bench.py's iw1_fwd_frac) and back to back (tools/iw1_timing.py): the library in the
  # tree against tools/_exp/libzs_hip_prev.so (the previous commit's kernels, same ABI), alternating on one box
  OUT=gpurun_out/${TAG}_iw1_ab.txt; : > $OUT
  run() {
    label="$1"; shift
    env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --allow-experiments --full-record gpurun_out/_instep_full.json > gpurun_out/_instep.json 2>/dev/null
    python - "$label" <<'PY' >> $OUT
import json, sys
d = json.load(open("gpurun_out/_instep.json")); f = json.load(open("gpurun_out/_instep_full.json"))
k = f["hip_kernels"]
print("%-26s step %.4f ms | IW1 fwd %.2f us  IW1 bwd %.2f us  K1 pair %.2f us  Adam %.2f us" % (
    sys.argv[1], d["ms_per_step"], k["zs_bernoulli_iw_objective_f32"]["avg_us"], k["zs_bernoulli_iw_objective_bwd_f32"]["avg_us"],
    k["zs_normal_sample_logprob_pair_f32"]["avg_us"], k["zs_adam_step_f32"]["avg_us"]))
PY
  }
  for i in 1 2 3; do
    run "this tree" ZS_NONE=1
    run "previous commit" ZS_HIP_LIBRARY=tools/_exp/libzs_hip_prev.so
  done
  { echo "== back to back (tools/iw1_timing.py), this tree"; timeout 200 python tools/iw1_timing.py 2>/dev/null | grep -E "B=|IW1|fused" | head -24
    echo "== back to back, previous commit"; ZS_HIP_LIBRARY=tools/_exp/libzs_hip_prev.so timeout 200 python tools/iw1_timing.py 2>/dev/null | grep -E "B=|IW1|fused" | head -24; } >> $OUT
  cat $OUT
  ;;
k2_ab)
  # the given-value kernels (K2 / L2 / U2) at 1 M and 4.2 M rows: this tree against tools/_exp/libzs_hip_prev.so, alternating on one box
  OUT=gpurun_out/${TAG}_k2_ab.txt; : > $OUT
  for i in 1 2 3; do
    { echo "== this tree"; timeout 200 python tools/kernel_sweep.py --batches 20971 83886 --only "logprob" 2>/dev/null | grep -E "^(K2 normal logprob|L2 logistic logprob|U2 uniform logprob|K1 normal sample.lp .eps)"
      echo "== previous kernels"; ZS_HIP_LIBRARY=tools/_exp/libzs_hip_prev.so timeout 200 python tools/kernel_sweep.py --batches 20971 83886 --only "logprob" 2>/dev/null | grep -E "^(K2 normal logprob|L2 logistic logprob|U2 uniform logprob|K1 normal sample.lp .eps)"; } >> $OUT
  done
  cut -c1-175 $OUT
  ;;
bench)
  timeout 900 python bench.py --full-record gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
  tail -c 400 gpurun_out/${TAG}_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_n1.json"))
print(d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"].items() if "frac" in k})
print("dp:", d["extra_configs"].get("c3_dp_step_n1"))
PY
  ;;
tests)
  timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -30 > gpurun_out/${TAG}_gputests.log
  tail -14 gpurun_out/${TAG}_gputests.log
  ;;
final)
  python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log
  timeout 1700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -120 > gpurun_out/${TAG}_gputests.log; tail -16 gpurun_out/${TAG}_gputests.log
  timeout 600 python tools/soak.py 5000 > gpurun_out/soak_raw.txt 2>/dev/null; echo "soak rc=$?"; grep -v amdgpu.ids gpurun_out/soak_raw.txt > gpurun_out/${TAG}_soak.txt; tail -4 gpurun_out/${TAG}_soak.txt
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
  python bench.py --steps 200 --warmup 20 --full-record gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; echo "bench rc=$?"
  python bench.py --steps 20 --warmup 5 --full-record gpurun_out/${TAG}_bench_full_driver_cmd.json > gpurun_out/${TAG}_bench_n1_driver_cmd.json 2>> gpurun_out/${TAG}_bench_n1.err; echo "bench (driver's command) rc=$?"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof" -o bench -- python3 "$ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$ROOT/gpurun_out/${TAG}_bench_prof.json" 2> "$ROOT/gpurun_out/${TAG}_bench_prof.err"; echo "rocprof rc=$?"
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
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_n1.json"))
print(d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"].items() if isinstance(v, float) and "frac" in k}, "traffic", d["roofline"]["traffic"])
print("dp:", d["extra_configs"].get("c3_dp_step_n1"))
PY
  du -sh gpurun_out
  ;;
profiles)
  EXP=tools/_exp/libzs_hip_exp.so      # (make -C zhusuan-pytorch_amd/csrc experiments: the lab build, tools/lab/csrc_lab.patch)
  {
    echo "== release library (persistent kernel)"; timeout 120 python tools/iw1_timing.py
    echo "== lab build, ZS_IW1_BLOCK_KERNEL=1 (round 4's workgroup-per-datapoint kernel)"; ZS_HIP_LIBRARY=$EXP ZS_IW1_BLOCK_KERNEL=1 timeout 120 python tools/iw1_timing.py
    echo "== lab build, ZS_IW1_SHARDED=1 (the batch mean finished by the last arrival)"; ZS_HIP_LIBRARY=$EXP ZS_IW1_SHARDED=1 timeout 120 python tools/iw1_timing.py
    echo "== release library, cold (512 MB fill between launches)"; timeout 120 python tools/iw1_timing.py --cold
  } 2>/dev/null > gpurun_out/${TAG}_iw1_timing.txt
  ZS_HIP_LIBRARY=$EXP timeout 200 python tools/iw1_phases.py 2>/dev/null > gpurun_out/${TAG}_iw1_phases.txt
  ZS_HIP_LIBRARY=$EXP timeout 200 python tools/iw1_phases_instep.py 2>/dev/null | grep -v "UserWarning\|run_backward" > gpurun_out/${TAG}_iw1_phases_instep.txt
  { timeout 400 python tools/fuzz_hotpath.py 300 5 2>&1 | grep -v amdgpu.ids | tail -3; echo ----; timeout 200 python tools/fuzz_layers.py 100 5 2>&1 | grep -v amdgpu.ids | tail -3; } > gpurun_out/${TAG}_fuzz.txt
  timeout 400 python tools/kernel_sweep.py --out gpurun_out/${TAG}_kernel_sweep.json 2>/dev/null > gpurun_out/${TAG}_kernel_sweep.txt
  tail -3 gpurun_out/${TAG}_iw1_timing.txt; tail -3 gpurun_out/${TAG}_fuzz.txt; tail -5 gpurun_out/${TAG}_kernel_sweep.txt
  ;;
pmc)
  cd /tmp; export TMPDIR=/tmp
  SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
  run() {   # tag which B kernel-name-fragment
    tag=$1; which=$2; B=$3; match=$4
    rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $ROOT/gpurun_out/pmc_${tag}_sq -- python3 $ROOT/tools/pmc_kernels.py $which $B > $ROOT/gpurun_out/pmc_${tag}.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_${tag}_$c -- python3 $ROOT/tools/pmc_kernels.py $which $B > /dev/null 2>&1
    done
    (cd $ROOT && python tools/pmc_summary.py gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE --match "$match" --out gpurun_out/${TAG}_pmc_${tag}.json > /dev/null)
    grep "algorithmic" $ROOT/gpurun_out/pmc_${tag}.log
    rm -rf $ROOT/gpurun_out/pmc_${tag}_sq $ROOT/gpurun_out/pmc_${tag}_FETCH_SIZE $ROOT/gpurun_out/pmc_${tag}_WRITE_SIZE
  }
  run iw1_c3 iw1 256 "k_iw1_persist"
  run iw1_1024 iw1 1024 "k_iw1_persist"
  run iw1bwd_c3 iw1_bwd 256 "k_iw1_bwd"
  ls $ROOT/gpurun_out/${TAG}_pmc_*.json
  ;;
k1_pmc)
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $ROOT/gpurun_out/pmc_k1_sq -- python3 $ROOT/tools/k1_pmc.py > /dev/null 2>&1
  for c in GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_k1_$c -- python3 $ROOT/tools/k1_pmc.py > /dev/null 2>&1
  done
  cd $ROOT && python tools/pmc_summary.py gpurun_out/pmc_k1_sq gpurun_out/pmc_k1_GRBM_GUI_ACTIVE gpurun_out/pmc_k1_FETCH_SIZE gpurun_out/pmc_k1_WRITE_SIZE --match "k_sample_tile<0" --out gpurun_out/${TAG}_pmc_k1.json
  rm -rf gpurun_out/pmc_k1_sq gpurun_out/pmc_k1_GRBM_GUI_ACTIVE gpurun_out/pmc_k1_FETCH_SIZE gpurun_out/pmc_k1_WRITE_SIZE
  ;;
*)
  echo "unknown job $JOB"; exit 2 ;;
esac

#!/bin/bash
# The round's GPU jobs, one gpurun call each (round 6: ONE parametrised script instead of a gpu_*_rNN.sh per job):
#     gpurun --timeout 1500 -- 'bash tools/gpu_job.sh <job> [tag]'
# Everything is written under gpurun_out/<tag>_* ; copy what should be judged into profiles/.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
JOB=${1:?job name}
TAG=${2:-r06}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0

one_rank() {   # bench.py as ONE rank with a process group over RCCL (the driver's launcher form)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 \
    --no-cpu-baseline --no-extras "$@" 2>> gpurun_out/${TAG}_${JOB}.err
}
line() { python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-44s %.4f ms/step  %.3f M evals/s  %s' % (sys.argv[1], r['ms_per_step'], r['value'] / 1e6, r['config']['launch_mode'][:60]))" "$1"; }

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
dp_tests)
  timeout 1700 python -m pytest -x -q -m gpu tests/test_dataparallel.py tests/test_graph.py tests/test_iw_fused.py tests/test_bench_contract.py \
     -k "not single_rank and not smoke and not strong" --durations=6 2>&1 | tail -25 > gpurun_out/${TAG}_dp_tests.log
  tail -25 gpurun_out/${TAG}_dp_tests.log
  ;;
tests)
  timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -30 > gpurun_out/${TAG}_gputests.log
  tail -14 gpurun_out/${TAG}_gputests.log
  ;;
*)
  echo "unknown job $JOB"; exit 2 ;;
esac

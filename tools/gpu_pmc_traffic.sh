#!/bin/bash
# developer script: multi-rank code path on one GPU + PMC traffic passes for the dominant kernel
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --force-collective-path > gpurun_out/bench_coll.log 2> gpurun_out/bench_coll.err
echo "collective-path rc=$?"; tail -1 gpurun_out/bench_coll.log | cut -c1-700; grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/bench_coll.err | tail -5
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"; ls $R/gpurun_out/pmc_$c/*/ | head
done

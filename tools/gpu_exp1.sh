#!/bin/bash
# developer experiment: BLAS backend for the (out-of-scope) MLP GEMMs, fused logits
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), int(d['value']), d['roofline']['avg_launch_us'])"; }
for b in default hipblaslt rocblas; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --blas $b 2>gpurun_out/err_$b.log | show $b; done
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --blas rocblas --fused-logits 2>/dev/null | show rocblas_fused

# Everything under profiles/ that is not part of gpu_final_check.sh: the kernel size sweep, the K1 counter passes, the per-step
# kernel tables of C2 / C3 / C5, the other configs, the streaming yardsticks.  gpurun --timeout 2400 -- 'bash tools/gpu_round_profiles.sh r03'
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r03}
mkdir -p gpurun_out
python tools/kernel_sweep.py --out gpurun_out/${TAG}_kernel_sweep.json 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kernel_sweep.txt; echo "sweep rc=$?"
for c in C2 C3 C5; do
  python tools/step_kernels.py --config $c --tuned-gemm --out gpurun_out/${TAG}_step_kernels_$c.json 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_step_kernels_$c.txt
done
echo "step tables done"
# the same tables with the reference's discarded draw executed (the package default) and, for the BNN, round 2's batched-GEMM layer
python tools/step_kernels.py --config C5 --tuned-gemm --reference-draws 2>/dev/null | grep -v amdgpu.ids | head -1 > gpurun_out/${TAG}_step_variants.txt
python tools/step_kernels.py --config C5 --tuned-gemm --bnn-layer per_layer 2>/dev/null | grep -v amdgpu.ids | head -1 >> gpurun_out/${TAG}_step_variants.txt
python tools/step_kernels.py --config C5 --tuned-gemm --bnn-layer bmm 2>/dev/null | grep -v amdgpu.ids | head -1 >> gpurun_out/${TAG}_step_variants.txt
python tools/step_kernels.py --config C2 --tuned-gemm --reference-draws 2>/dev/null | grep -v amdgpu.ids | head -1 >> gpurun_out/${TAG}_step_variants.txt
python tools/small_kernels_timing.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_small_kernels.txt; echo "small kernels rc=$?"
[ -x tools/k3_variants ] && ./tools/k3_variants 256 > gpurun_out/${TAG}_k3_forward_variants.txt 2>&1
python tools/bench_configs.py --steps 200 --out gpurun_out/${TAG}_configs.json > gpurun_out/configs.log 2>&1; echo "configs rc=$?"
python tools/copy_ceiling.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_copy_ceiling.txt
python tools/adam_ab.py 2>/dev/null | grep "ms" > gpurun_out/${TAG}_adam_ab.txt
bash tools/gpu_k1_pmc.sh; echo "k1 pmc rc=$?"
rm -rf gpurun_out/pmc_k1_sq gpurun_out/pmc_k1_GRBM_GUI_ACTIVE gpurun_out/pmc_k1_FETCH_SIZE gpurun_out/pmc_k1_WRITE_SIZE
du -sh gpurun_out; tail -3 gpurun_out/${TAG}_adam_ab.txt; head -3 gpurun_out/${TAG}_step_kernels_C2.txt

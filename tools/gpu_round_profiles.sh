# Everything under profiles/ that is not part of gpu_final_check.sh: the kernel size sweep, the per-step kernel tables of C2 / C3 / C5 (both draws
# executed = the package default, and inside skip_discarded_draws), IW1 against the launches it replaces, the eager step's host profile, the counter
# passes.   gpurun --timeout 2400 -- 'bash tools/gpu_round_profiles.sh r04'
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r04}
mkdir -p gpurun_out
python tools/kernel_sweep.py --out gpurun_out/${TAG}_kernel_sweep.json 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kernel_sweep.txt; echo "sweep rc=$?"
for c in C2 C3 C5; do
  python tools/step_kernels.py --config $c --tuned-gemm --out gpurun_out/${TAG}_step_kernels_$c.json 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_step_kernels_$c.txt
  python tools/step_kernels.py --config $c --tuned-gemm --skip-discarded-draws 2>/dev/null | grep -v amdgpu.ids >> gpurun_out/${TAG}_step_kernels_$c.txt
done
echo "step tables done"
python tools/iw1_timing.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_iw1_timing_final.txt
python tools/iw1_timing.py --cold 2>/dev/null | grep -v amdgpu.ids >> gpurun_out/${TAG}_iw1_timing_final.txt
(python tools/eager_host_profile.py --config c3 --profile; python tools/eager_host_profile.py --config c5 --profile) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_eager_host_profile.txt
python tools/small_kernels_timing.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_small_kernels.txt; echo "small kernels rc=$?"
python tools/copy_ceiling.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_copy_ceiling.txt
bash tools/gpu_k1_pmc.sh; echo "k1 pmc rc=$?"
bash tools/gpu_pmc_r04.sh > gpurun_out/pmc_r04.log 2>&1; echo "pmc r04 rc=$?"
rm -rf gpurun_out/pmc_k1_sq gpurun_out/pmc_k1_GRBM_GUI_ACTIVE gpurun_out/pmc_k1_FETCH_SIZE gpurun_out/pmc_k1_WRITE_SIZE
du -sh gpurun_out; head -2 gpurun_out/${TAG}_step_kernels_C3.txt

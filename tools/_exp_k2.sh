export ZS_HIP_LIBRARY=$PWD/tools/libzs_hip_exp.so
for u in 2 4; do for n in 0 1; do
echo "== U=$u NTL=$n"; ZS_K2_U=$u ZS_K2_NTL=$n python tools/kernel_sweep.py --only "K2 normal logprob" --only "L2 logistic logprob" --only "U2 uniform" --batches 2621 20971 41943 83886 2>&1 | grep -E "K2|L2|U2" | cut -c1-160
done; done

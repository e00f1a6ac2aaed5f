export ZS_HIP_LIBRARY=$PWD/tools/libzs_hip_exp.so
for nw in 0 16; do for v in 0 2 3; do
echo "== NW=$nw VARIANT=$v"; ZS_IW1_NW=$nw ZS_IW1_VARIANT=$v python tools/iw1_timing.py 2>&1 | grep "B=" | awk '{print $1,$2,$3,$4, "K3",$5,"sum",$8,"IW1",$10}' | head -3
done; done

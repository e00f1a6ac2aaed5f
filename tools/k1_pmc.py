"""K1 (Philox) at the 4.2 M-row sweep point, five launches: run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` (separate passes) to compare the HBM traffic with the algorithmic bytes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch
from zhusuan import _hip
dev = torch.device("cuda:0"); lib = _hip.lib(); P = _hip.ptr
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
K, B, D = 50, 83886, 40
M = B * D
mu = torch.randn(M, device=dev); sg = torch.rand(M, device=dev) + 0.5
z = torch.empty(K * M, device=dev); lp = torch.empty(B * K, device=dev)
for i in range(5):
    lib.call("zs_normal_sample_logprob_f32", P(mu), P(sg), None, 1, i, None, P(z), P(lp), K, M, D, 1, K, 0, None, st)
torch.cuda.synchronize()
print("algorithmic bytes per launch", 4 * K * M + 4 * K * B + 8 * M)

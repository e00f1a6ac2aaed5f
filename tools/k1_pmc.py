"""K1 (in-kernel Philox) at the sweep points N = 1 M and 4.2 M rows (K = 50, D = 40), a few launches each: the program
rocprofv3 runs for the counter passes (program directly after `--`, counters in their own runs):

  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES \
            SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_k1_sq -- python3 tools/k1_pmc.py
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_k1_grbm -- python3 tools/k1_pmc.py
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ...  /  --pmc WRITE_SIZE ...          (HBM traffic, separate passes)

tools/pmc_summary.py turns the CSVs into profiles/r02_pmc_k1.json."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch
from zhusuan import _hip
dev = torch.device("cuda:0"); lib = _hip.lib(); P = _hip.ptr
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
K, D = 50, 40
for B in (20971, 83886):
    M = B * D
    mu = torch.randn(M, device=dev); sg = torch.rand(M, device=dev) + 0.5
    z = torch.empty(K * M, device=dev); lp = torch.empty(B * K, device=dev)
    for i in range(6):
        lib.call("zs_normal_sample_logprob_f32", P(mu), P(sg), None, 1, i, None, P(z), P(lp), K, M, D, 1, K, 0, None, st)
    torch.cuda.synchronize()
    print("B", B, "rows", K * B, "algorithmic bytes per launch", 4 * K * M + 4 * K * B + 8 * M)
    del mu, sg, z, lp

"""Per-launch durations of K1 (Philox) at the 4.2 M-row sweep point: is the mean pulled up by outliers?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch
from zhusuan import _hip
dev = torch.device("cuda:0"); lib = _hip.lib(); P = _hip.ptr
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
K, B, D = 50, int(sys.argv[1]) if len(sys.argv) > 1 else 83886, 40
M = B * D
mu = torch.randn(M, device=dev); sg = torch.rand(M, device=dev) + 0.5
z = torch.empty(K * M, device=dev); lp = torch.empty(B * K, device=dev)
other = torch.empty(K * M, device=dev)          # a second 0.7 GB tensor, touched between launches when asked to
call = lambda off: lib.call("zs_normal_sample_logprob_f32", P(mu), P(sg), None, 1, off, None, P(z), P(lp), K, M, D, 1, K, 0, None, st)
for i in range(3): call(i)
torch.cuda.synchronize()
for mode in ("back-to-back", "sync between launches", "other tensor written between launches"):
    ds = []
    for i in range(30):
        if mode != "back-to-back": torch.cuda.synchronize()
        if mode.startswith("other"): other.fill_(1.0); torch.cuda.synchronize()
        lib.prof_enable(True); call(10 + i); torch.cuda.synchronize(); lib.prof_enable(False)
        q = lib.prof_query("zs_normal_sample_logprob_f32"); ds.append(q["total_ms"] * 1e3)
    ds_sorted = sorted(ds)
    bytes_ = 4 * K * M + 4 * K * B + 8 * M
    print("%-40s mean %.1f median %.1f min %.1f max %.1f us -> mean %.1f%% median %.1f%%" % (
        mode, sum(ds) / len(ds), ds_sorted[15], ds_sorted[0], ds_sorted[-1], bytes_ / (sum(ds) / len(ds) * 1e-6) / 8e10, bytes_ / (ds_sorted[15] * 1e-6) / 8e10))
    print("   ", " ".join("%.0f" % d for d in ds))

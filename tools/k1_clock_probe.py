"""How the duration of K1 (fused Normal sample + log-prob, in-kernel Philox, VALU-issue bound) follows the clock state of the chip: the
same 4.2 M-row launch right after process start / after a second of idling / after 0.3 s and 2 s of continuous launches / after 200
GB of fills.  MI355X: 183-187 us idle-cold, 125-135 us sustained.  Why bench.py measures the kernel behind 0.3 s of its own launches."""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch
from zhusuan import _hip
dev = torch.device("cuda:0")
lib = _hip.lib(); P = _hip.ptr
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
K, D = 50, 40
def setup(B):
    N, M = K * B, B * D
    mu, sg = torch.randn(M, device=dev), torch.rand(M, device=dev) + 0.5
    z, lp = torch.empty(K * M, device=dev), torch.empty(B * K, device=dev)
    fn = lambda: lib.call("zs_normal_sample_logprob_f32", P(mu), P(sg), None, 1, 2, None, P(z), P(lp), K, M, D, 1, K, 0, None, st)
    return fn, 4 * N * D + 4 * N + 8 * M, (mu, sg, z, lp)
def measure(fn, nbytes, label, launches=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.prof_enable(True)
    for _ in range(launches): fn()
    torch.cuda.synchronize()
    lib.prof_enable(False)
    d = sorted(1e3 * v for v in lib.prof_durations("zs_normal_sample_logprob_f32"))
    print("%-40s median %7.1f us (%.1f %%)  min %7.1f  max %7.1f" % (label, d[len(d)//2], 100 * nbytes / d[len(d)//2] / 1e3 / 8000, d[0], d[-1]), flush=True)
f1, b1, k1 = setup(20971)
f4, b4, k4 = setup(83886)
measure(f4, b4, "4.2M cold (process start)")
measure(f1, b1, "1M after that")
measure(f4, b4, "4.2M again")
t0 = time.time()
while time.time() - t0 < 0.3: 
    for _ in range(50): f1()
    torch.cuda.synchronize()
measure(f4, b4, "4.2M after 0.3 s of 1M launches")
measure(f1, b1, "1M")
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50): f1()
    torch.cuda.synchronize()
measure(f4, b4, "4.2M after 2 s more")
time.sleep(1.0)
measure(f4, b4, "4.2M after 1 s idle")
x = torch.empty(1 << 28, device=dev)
for _ in range(200): x.fill_(1.0)
torch.cuda.synchronize()
measure(f4, b4, "4.2M after 200 x 1 GB fills")
measure(f4, b4, "4.2M, 100 launches", launches=100)

"""One hot-path entry point at one size, a few launches: the program rocprofv3 runs for the counter passes of round 4
(say WHY a kernel sits where it sits -- traffic ratio or issue share).

  python3 tools/pmc_kernels.py <which> <B>        which: k2 | l2 | u2 | k3 | k3_logits | k3_bwd | k3_bwd_logits | l1_u | iw1 | iw1_bwd
K = 50, D = 40 (Normal family), X = 784 (Bernoulli).  Prints the algorithmic bytes per launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch
from zhusuan import _hip
which, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0"); lib = _hip.lib(); P = _hip.ptr
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
K, D, X = 50, 40, 784
N, M = K * B, B * D
if which in ("k2", "l2", "u2", "l1_u"):
    x = torch.randn(N * D, device=dev); mu = torch.randn(M, device=dev) * 0.1; sg = torch.rand(M, device=dev) + 0.5
    lp = torch.empty(N, device=dev)
    if which == "k2":
        fn = lambda: lib.call("zs_normal_logprob_f32", P(x), N * D, P(mu), M, P(sg), M, P(lp), K, B, D, 1, K, 0, st)
    elif which == "l2":
        fn = lambda: lib.call("zs_logistic_logprob_f32", P(x), N * D, P(mu), M, P(sg), M, P(lp), K, B, D, 1, K, st)
    elif which == "u2":
        lo, hi = mu - 5.0, mu + 5.0
        fn = lambda: lib.call("zs_uniform_logprob_f32", P(x), N * D, P(lo), M, P(hi), M, P(lp), K, B, D, 1, K, st)
    else:
        u = torch.rand(N * D, device=dev) * 0.98 + 0.01; z = torch.empty(N * D, device=dev)
        fn = lambda: lib.call("zs_logistic_sample_logprob_f32", P(mu), P(sg), P(u), 0, 0, None, P(z), P(lp), K, M, D, 1, K, None, st)
    nbytes = (8 if which == "l1_u" else 4) * N * D + 4 * N + 8 * M
else:
    p = torch.rand(N * X, device=dev) * 0.96 + 0.02
    if "logits" in which:
        p = p * 8.0 - 4.0
    x = (torch.rand(B * X, device=dev) < 0.5).float()
    lp = torch.empty(N, device=dev); glp = torch.randn(N, device=dev)
    if which == "k3":
        fn = lambda: lib.call("zs_bernoulli_logprob_f32", P(p), P(x), B * X, P(lp), K, B, X, 1, K, st)
        nbytes = 4 * N * X + 4 * B * X + 4 * N
    elif which == "k3_logits":
        fn = lambda: lib.call("zs_bernoulli_logits_logprob_f32", P(p), P(x), B * X, P(lp), None, K, B, X, 1, K, st)
        nbytes = 4 * N * X + 4 * B * X + 4 * N
    elif which in ("k3_bwd", "k3_bwd_logits"):
        gp = torch.empty(N * X, device=dev)
        name = "zs_bernoulli_logits_logprob_bwd_f32" if "logits" in which else "zs_bernoulli_logprob_bwd_f32"
        fn = lambda: lib.call(name, P(p), P(x), B * X, P(glp), 1, K, P(gp), K, B, X, st)
        nbytes = 8 * N * X + 4 * B * X + 4 * N
    else:
        z = torch.randn(N * D, device=dev); mu = torch.zeros(M, device=dev); sg = torch.ones(M, device=dev)
        qmu, qsg = torch.randn(M, device=dev), torch.rand(M, device=dev) + 0.5
        logq = torch.randn(N, device=dev) - 45
        lpz, costb, bound, coef, cost = torch.empty(N, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev), torch.randn(2 * N, device=dev) / B, torch.empty(1, device=dev)
        acc = torch.zeros(64, dtype=torch.int64, device=dev)
        if which == "iw1":
            fn = lambda: lib.call("zs_bernoulli_iw_objective_f32", P(p), 0, P(x), B * X, K, B, X, P(z), P(mu), M, P(sg), M, D, 0, None, K, P(logq), K, 1, 1,
                                  P(lp), P(lpz), P(costb), P(bound), P(coef), P(cost), P(acc), st)
            nbytes = 4 * N * X + 4 * B * X + 4 * N * D + 8 * M + 4 * N + 16 * N + 8 * B + 4
        else:
            gp = torch.empty(N * X, device=dev); gm, gs = torch.empty(M, device=dev), torch.empty(M, device=dev); g = torch.ones(1, device=dev)
            fn = lambda: lib.call("zs_bernoulli_iw_objective_bwd_f32", P(p), 0, P(x), B * X, K, B, X, P(coef), P(g), 0, P(gp), P(z), P(qmu), P(qsg), D, 0,
                                  P(gm), P(gs), st)
            nbytes = (8 * N * X + 4 * B * X + 4 * N) + (4 * N * D + 4 * N + 16 * M)
for i in range(8):
    fn()
torch.cuda.synchronize()
print("which", which, "B", B, "rows", N, "algorithmic bytes per launch", nbytes)

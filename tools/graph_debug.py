"""developer tool: find which part of the training step breaks hipGraph capture (each case in a subprocess)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["fwd", "fwd_bwd", "fwd_bwd_zero", "fwd_bwd_bucket", "fwd_bwd_adam", "full", "full_small", "mlp_only", "mlp_bwd"]

def run(case):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")]
    import torch, zhusuan
    from zhusuan import dataparallel
    from examples import iwae
    dev = torch.device("cuda:0")
    small = case == "full_small"
    B, K, H = (16, 5, 32) if small else (256, 50, 500)
    model = iwae.build(n_samples=K, estimator="vimco", hidden=H, device=dev)
    x = (torch.rand(B, 784, device=dev) < 0.5).float()
    rng = zhusuan.DeviceRNG(dev, seed=1)
    bucket = dataparallel.GradientBucket(model) if case in ("fwd_bwd_bucket", "full", "full_small") else None
    opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True, capturable=True) if case in ("fwd_bwd_adam", "full", "full_small") else None
    mlp = torch.nn.Sequential(torch.nn.Linear(40, 500), torch.nn.ReLU(), torch.nn.Linear(500, 500), torch.nn.ReLU(), torch.nn.Linear(500, 784)).to(dev)
    zin = torch.randn(12800, 40, device=dev)

    def body():
        if case == "mlp_only":
            with torch.no_grad():
                return mlp(zin).sum()
        if case == "mlp_bwd":
            for p in mlp.parameters():
                p.grad = None
            l = mlp(zin).sum(); l.backward(); return l.detach()
        rng.begin_step()
        if bucket is not None:
            bucket.zero()
        elif case != "fwd":
            for p in model.parameters():
                if p.grad is not None:
                    p.grad.zero_() if case == "fwd_bwd_zero" else None
                if case != "fwd_bwd_zero":
                    p.grad = None
        if case == "fwd":
            with torch.no_grad():
                return model({"x": x})
        loss = model({"x": x})
        loss.backward()
        g = bucket.all_reduce_mean(loss) if bucket is not None else loss.detach()
        if opt is not None:
            opt.step()
        return g

    with zhusuan.device_rng(rng):
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = body()
        g.replay(); torch.cuda.synchronize()
        print("OK", case, float(out))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l and "Warning" not in l and "run_backward" not in l][-2:]
            print(c, "rc=%d" % r.returncode, " | ".join(tail), flush=True)

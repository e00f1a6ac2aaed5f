import torch
dev = torch.device("cuda:0")
z = torch.empty(50 * 83886 * 40, device=dev)
src = torch.randn(50 * 83886 * 40, device=dev)
def timed(fn, n=30):
    out = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); out.append(a.elapsed_time(b) * 1e3)
    return out
for name, fn in [("fill_ (write 671 MB)", lambda: z.fill_(1.0)), ("copy_ (read+write)", lambda: z.copy_(src)), ("sum (read 671 MB)", lambda: src.sum())]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    d = timed(fn)
    print(name, " ".join("%.0f" % x for x in d))

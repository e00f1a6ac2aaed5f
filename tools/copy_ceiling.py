"""What a plain device-to-device stream reaches on this chip, as a yardstick for the read+write kernels (K3 backward
reads p and writes gp of the same size): torch's copy_ (hipMemcpy D2D / blit kernel) and an element-wise add, sizes as in
the kernel sweep."""
import torch
dev = torch.device("cuda:0")
for mb in (81, 830, 3323, 13293):
    n = mb * 1000 * 1000 // 8          # two tensors of n floats: mb MB total traffic
    x = torch.empty(n, device=dev).normal_()
    y = torch.empty_like(x)
    for name, fn in (("copy_", lambda: y.copy_(x)), ("add(x, 1.0, out)", lambda: torch.add(x, 1.0, out=y)),
                     ("fill_ (write only)", lambda: y.fill_(1.0))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        traffic = (4 * n) if "fill" in name else (8 * n)
        print("%6d MB  %-22s %9.1f us  %6.2f TB/s  %5.1f%% of 8 TB/s" % (traffic // 1000000, name, us, traffic / us / 1e6, traffic / us / 1e6 / 8 * 100))
    del x, y

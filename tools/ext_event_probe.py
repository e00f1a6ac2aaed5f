import torch, time
dev = torch.device("cuda:0")
try:
    ev = torch.cuda.Event(external=True)
except TypeError as e:
    print("no external kwarg:", e); raise SystemExit
a = torch.zeros(1 << 20, device=dev); b = torch.zeros(64 << 20, device=dev); flag = torch.zeros(1, device=dev)
side = torch.cuda.Stream(); cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    for _ in range(3):
        a.add_(1.0); [b.mul_(1.0001) for _ in range(20)]
torch.cuda.current_stream().wait_stream(cap); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        a.add_(1.0)                 # "decoder backward + pack"
        ev.record()                 # external event node inside the graph
        for _ in range(20):
            b.mul_(1.0001)          # "encoder backward": long
except Exception as e:
    print("capture failed:", repr(e)); raise SystemExit
torch.cuda.synchronize()
out = torch.zeros(1, device=dev)
t_side = torch.cuda.Event(enable_timing=True); t_end = torch.cuda.Event(enable_timing=True); t0 = torch.cuda.Event(enable_timing=True)
for it in range(3):
    a.zero_(); torch.cuda.synchronize()
    t0.record()
    g.replay()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        out.copy_(a[:1])            # must see a == 1 (after the first node), and run BEFORE the graph's tail ends
        t_side.record()
    t_end.record()
    torch.cuda.synchronize()
    print("iter", it, "side saw a =", float(out), " side done at %.3f ms, graph done at %.3f ms" % (t0.elapsed_time(t_side), t0.elapsed_time(t_end)))

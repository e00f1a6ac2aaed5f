"""What the pieces BETWEEN the hipGraphs of a data-parallel step cost on one GPU (one rank over RCCL): a graph of small kernels is
replayed twice per iteration with one of the candidate links in between -- nothing, an event record, a fork / join through a
side stream (torch's events: system-scope release; HIP events created with hipEventReleaseToDevice), an asynchronous and a
synchronous all-reduce of a 2.7 MB bucket through torch.distributed -- and the time per iteration is compared with bare
back-to-back replays.  -> profiles/r06_stream_links.txt

    python tools/stream_link_probe.py
"""
import ctypes
import os
import socket
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402


def hip_runtime():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    lib.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    lib.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    lib.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    lib.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
    lib.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
    return lib


HIP_EVENT_DISABLE_TIMING = 0x2
HIP_EVENT_RELEASE_TO_DEVICE = 0x40000000
HIP_EVENT_DISABLE_SYSTEM_FENCE = 0x20000000


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    hip = hip_runtime()
    bucket = torch.zeros(680000, device=dev)
    dist.all_reduce(bucket)
    torch.cuda.synchronize()

    # the graph: 12 element-wise kernels over 40 MB (about 12 us each: like the step's streaming kernels, with dirty lines in L2)
    a = torch.rand(10 * 1024 * 1024, device=dev)
    b = torch.empty_like(a)

    def body():
        for _ in range(6):
            torch.mul(a, 1.0001, out=b)
            torch.add(b, 0.5, out=a)
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    torch.cuda.synchronize()

    def hip_event(flags):
        ev = ctypes.c_void_p()
        rc = hip.hipEventCreateWithFlags(ctypes.byref(ev), flags)
        assert rc == 0, rc
        return ev
    ev_sys = [hip_event(HIP_EVENT_DISABLE_TIMING) for _ in range(2)]
    ev_dev = [hip_event(HIP_EVENT_DISABLE_TIMING | HIP_EVENT_RELEASE_TO_DEVICE) for _ in range(2)]
    ev_nof = [hip_event(HIP_EVENT_DISABLE_TIMING | HIP_EVENT_DISABLE_SYSTEM_FENCE) for _ in range(2)]
    tev = [torch.cuda.Event() for _ in range(2)]

    def raw(stream):
        return ctypes.c_void_p(stream.cuda_stream)

    def link_none():
        pass

    def link_record_torch():
        tev[0].record()

    def mk_record_hip(evs):
        def f():
            hip.hipEventRecord(evs[0], raw(torch.cuda.current_stream()))
        return f

    def link_forkjoin_torch():
        cur = torch.cuda.current_stream()
        tev[0].record(cur)
        side.wait_event(tev[0])
        tev[1].record(side)
        cur.wait_event(tev[1])

    def link_fork_torch():
        cur = torch.cuda.current_stream()
        tev[0].record(cur)
        side.wait_event(tev[0])

    def mk_forkjoin_hip(evs, join=True):
        def f():
            cur = raw(torch.cuda.current_stream())
            hip.hipEventRecord(evs[0], cur)
            hip.hipStreamWaitEvent(raw(side), evs[0], 0)
            if join:
                hip.hipEventRecord(evs[1], raw(side))
                hip.hipStreamWaitEvent(cur, evs[1], 0)
        return f

    def link_allreduce_async_wait():
        dist.all_reduce(bucket, async_op=True).wait()

    pending = []

    def link_allreduce_async_nowait():
        pending.append(dist.all_reduce(bucket, async_op=True))
        if len(pending) > 1:
            pending.pop(0).wait()           # joined one link later (its collective has long finished)

    def link_allreduce_sync():
        dist.all_reduce(bucket)

    def link_allreduce_sync_on_side():
        cur = torch.cuda.current_stream()
        hip.hipEventRecord(ev_dev[0], raw(cur))
        hip.hipStreamWaitEvent(raw(side), ev_dev[0], 0)
        with torch.cuda.stream(side):
            dist.all_reduce(bucket)

    # stream memory operations instead of events: the compute stream WRITES a counter into signal memory, the side stream WAITS for
    # it (and back for the join) -- no event object, no host-visible signal
    sig = [ctypes.c_void_p(), ctypes.c_void_p()]
    have_values = all(hip.hipExtMallocWithFlags(ctypes.byref(p), 8, 0x2) == 0 for p in sig)      # hipMallocSignalMemory
    tick = [0]

    def mk_values(join=True):
        def f():
            tick[0] += 1
            cur = raw(torch.cuda.current_stream())
            rc = hip.hipStreamWriteValue32(cur, sig[0], tick[0], 0)
            rc |= hip.hipStreamWaitValue32(raw(side), sig[0], tick[0], 0, 0xFFFFFFFF)             # hipStreamWaitValueGte
            if join:
                rc |= hip.hipStreamWriteValue32(raw(side), sig[1], tick[0], 0)
                rc |= hip.hipStreamWaitValue32(cur, sig[1], tick[0], 0, 0xFFFFFFFF)
            assert rc == 0, rc
        return f

    small = torch.zeros(1, device=dev)

    def link_eager_kernel():
        small.add_(1.0)

    links = [("nothing (graph -> graph)", link_none),
             ("event record, torch.cuda.Event", link_record_torch),
             ("event record, HIP event (default flags)", mk_record_hip(ev_sys)),
             ("event record, HIP event (release to device)", mk_record_hip(ev_dev)),
             ("event record, HIP event (no system fence)", mk_record_hip(ev_nof)),
             ("fork to a side stream, torch events", link_fork_torch),
             ("fork, HIP events (release to device)", mk_forkjoin_hip(ev_dev, join=False)),
             ("fork + join, torch events", link_forkjoin_torch),
             ("fork + join, HIP events (default flags)", mk_forkjoin_hip(ev_sys)),
             ("fork + join, HIP events (release to device)", mk_forkjoin_hip(ev_dev)),
             ("fork + join, HIP events (no system fence)", mk_forkjoin_hip(ev_nof)),
             ("fork, stream write / wait value (signal memory)", mk_values(join=False)),
             ("fork + join, stream write / wait value", mk_values(join=True)),
             ("all_reduce(async_op=True).wait()", link_allreduce_async_wait),
             ("all_reduce(async_op=True), joined one link later", link_allreduce_async_nowait),
             ("all_reduce() (synchronous form: current stream)", link_allreduce_sync),
             ("fork (HIP, device) + all_reduce() on the side stream", link_allreduce_sync_on_side),
             ("one eager kernel launch", link_eager_kernel)]

    def run(link, iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
            link()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

    print("graph = 12 element-wise kernels over 40 MB; us per (replay + link), median of 5 runs of 300; RCCL %s, one rank"
          % ".".join(str(v) for v in torch.cuda.nccl.version()))
    base = None
    for name, link in links:
        if "value" in name and not have_values:
            print("%-58s (hipExtMallocWithFlags(hipMallocSignalMemory) failed: skipped)" % name)
            continue
        run(link, 50)
        vals = sorted(run(link, 300) for _ in range(5))
        med = vals[2]
        if base is None:
            base = med
        print("%-58s %8.2f us   %+6.2f us vs bare replays" % (name, med, med - base))
        sys.stdout.flush()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

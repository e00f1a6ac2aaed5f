"""Does HAVING a process group slow the single-GPU step down?  The headline step (one hipGraph) is timed in one process (a) before
anything distributed exists, (b) after init_process_group("nccl") with one rank + its first collective, (c) after this job's own
RCCL communicator (zhusuan.dataparallel.DirectAllReduce) was built, (d) after both were torn down -- same model, same graph, same
GEMM picks throughout.  -> profiles/r06_pg_overhead.txt

    python tools/pg_overhead_probe.py            (environment variables of torch's NCCL watchdog can be set from outside)
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench                              # noqa: E402  (sys.path, workload builders, timed_trials)
import numpy as np                        # noqa: E402
import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402


def main():
    import zhusuan
    from zhusuan import dataparallel
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    bench.gemm_tuning(True)
    torch.manual_seed(0)
    model, obs, evals, _ = bench.make_workload("c3", dev)
    opt = bench.make_optimizer(model, False)
    rng = zhusuan.DeviceRNG(dev, seed=1)
    one = torch.ones((), device=dev)

    def compute():
        rng.begin_step()
        for p in model.parameters():
            p.grad = None
        loss = model(obs)
        loss.backward(one)
        return loss.detach()
    with zhusuan.device_rng(rng):
        step = zhusuan.GraphedStep(compute, opt.step, rng=rng, warmup=10)
        bench.gemm_tuning(True, tune=False)

        def timed(label):
            t = []
            for _ in range(3):
                t += bench.timed_trials(step, 200, 1, dev, min_seconds=0.4)[0]
            ms = 1e3 * float(np.median(t)) / 200
            print("%-62s %.4f ms/step  (min %.4f, max %.4f over %d trials)" % (label, ms, 1e3 * min(t) / 200, 1e3 * max(t) / 200, len(t)))
            sys.stdout.flush()
            return ms
        base = timed("(a) no process group")
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        b = timed("(b) process group (nccl, one rank) + its first collective")
        rccl = dataparallel.DirectAllReduce.create(timeout_s=60)
        c = timed("(c) + this job's own RCCL communicator") if rccl is not None else None
        if rccl is not None:
            rccl.close()
        dist.destroy_process_group()
        d = timed("(d) both torn down")
        print("process group: %+.1f us per step; own communicator: %s; after tear-down: %+.1f us" % (
            1e3 * (b - base), ("%+.1f us" % (1e3 * (c - b))) if c is not None else "n/a", 1e3 * (d - base)))
    env = dict((k, v) for k, v in os.environ.items() if k.startswith(("TORCH_NCCL", "NCCL_", "RCCL_")))
    print("environment:", env or "{}")


if __name__ == "__main__":
    main()

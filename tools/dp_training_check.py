"""Does the MULTI-RANK FORM of the step TRAIN like the single graph?  One rank over RCCL; the IWAE workload of bench.py from the same
initial weights and the same Philox seed in every form; the objective every `every` steps, side by side:

    single        one hipGraph (the headline step)
    dp_eager      flat bucket filled by the backward pass, all-reduce, FlatAdam(grad_scale) -- launched from Python
    dp_graphs     the same as graph A -> all-reduce (this job's RCCL communicator) -> graph B     (bench.py's default with > 1 rank)
    dp_graphs_nc  the same two graphs with NOTHING between them                                     (isolates the collective)
    dp_graphs_pk  the two graphs with a bucket the backward does not write into (direct=False: a pack copy)
    dp_graphs_td  the two graphs around torch.distributed's all_reduce
then, on dp_graphs' model, a single-graph twin recorded AFTERWARDS and replayed alternately (what bench.py's same_process figure does).

    python tools/dp_training_check.py [steps] [every]          -> profiles/r06_dp_training.txt
"""
import copy
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench                              # noqa: E402
import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402


def main():
    import zhusuan
    from zhusuan import dataparallel, _ops
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    rccl = dataparallel.DirectAllReduce.create(timeout_s=60)
    assert rccl is not None, dataparallel.DirectAllReduce.last_error
    torch.manual_seed(0)
    model0, obs, _, _ = bench.make_workload("c3", dev)
    one = torch.ones((), device=dev)
    curves = {}

    def build(form):
        model = copy.deepcopy(model0)
        opt = bench.make_optimizer(model, False)
        rng = zhusuan.DeviceRNG(dev, seed=1)
        if form == "single":
            def compute():
                rng.begin_step()
                for p in model.parameters():
                    p.grad = None
                loss = model(obs)
                loss.backward(one)
                return loss.detach()
            with zhusuan.device_rng(rng):
                return zhusuan.GraphedStep(compute, opt.step, rng=rng, warmup=3), model, opt, rng, None
        bucket = dataparallel.GradientBucket(model, direct=form != "dp_graphs_pk")

        def compute_part():
            rng.begin_step()
            bucket.zero()
            loss = model(obs)
            loss.backward(one)
            bucket.pack(loss)
            return loss.detach()

        def exchange_part(loss):
            if form == "dp_graphs_td":
                bucket.exchange(always=True)
            elif form != "dp_graphs_nc":
                bucket.exchange(direct=rccl, always=True)
            return bucket.flat[bucket.n_grad]

        def update_part():
            opt.step(grad_scale=1.0)
        if form == "dp_eager":
            def step():
                with zhusuan.device_rng(rng):
                    g = exchange_part(compute_part())
                    update_part()
                return g
            return step, model, opt, rng, bucket
        with zhusuan.device_rng(rng):
            return (zhusuan.GraphedStep(compute_part, update_part, exchange=exchange_part, rng=rng, warmup=3, optimizer=opt),
                    model, opt, rng, bucket)

    kept = {}
    for form in ("single", "dp_eager", "dp_graphs", "dp_graphs_nc", "dp_graphs_pk", "dp_graphs_td"):
        step, model, opt, rng, bucket = build(form)
        vals = []
        for i in range(steps):
            loss = step()
            if (i + 1) % every == 0:
                vals.append(float(loss))
        curves[form] = vals
        if form == "dp_graphs":
            kept = dict(step=step, model=model, opt=opt, rng=rng, bucket=bucket)
        elif bucket is not None:
            bucket.release()
        torch.cuda.synchronize()
    print("objective after every %d steps (c3: K = 50, B = 256; same initial weights and Philox seed in every form)" % every)
    print("%-8s" % "step" + "".join("%14s" % f for f in curves))
    for j in range(len(curves["single"])):
        print("%-8d" % ((j + 1) * every) + "".join("%14.4f" % curves[f][j] for f in curves))
    ref = curves["single"][-1]
    worst = max(abs(v[-1] - ref) / abs(ref) for v in curves.values())
    print("largest relative difference from the single graph at the last row: %.2e" % worst)

    # a single-graph twin recorded on dp_graphs' model afterwards, alternating with it
    model, opt, rng, bucket, step = kept["model"], kept["opt"], kept["rng"], kept["bucket"], kept["step"]

    def compute_single():
        rng.begin_step()
        bucket.zero()
        loss = model(obs)
        with _ops.grad_destinations_paused():
            loss.backward(one)
        return loss.detach()
    with zhusuan.device_rng(rng):
        twin = zhusuan.GraphedStep(compute_single, opt.step, rng=rng, warmup=3)
    print("\ndp_graphs' model continued: a single-graph twin recorded now, %d steps of each in turn" % every)
    for rnd in range(4):
        for name, f in (("twin", twin), ("dp_graphs", step)):
            for _ in range(every):
                loss = f()
            print("  round %d  %-10s objective %.4f" % (rnd, name, float(loss)))
    ok = worst < 0.05
    print("RESULT: %s" % ("ok" if ok else "DIFFERENT"))
    rccl.close()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

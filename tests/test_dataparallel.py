"""Data-parallel minibatch shards (zhusuan.dataparallel): world_size-2 gloo run on CPU (C oracle
injected as the kernel library) must reproduce the single-process gradients / objective of the full
minibatch -- the mean of equal-size shard means is the global mean."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, host_kernel_library
import helpers as H
import host_backend

B, K, HID = 16, 5, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    rng = np.random.RandomState(77)
    x = (rng.uniform(size=(B, 784)) < 0.5).astype(np.float32)
    e1 = rng.standard_normal((K, B, 40)).astype(np.float32)
    e2 = rng.standard_normal((K, B, 40)).astype(np.float32)
    return x, e1, e2


def _run_shard(rank, world, estimator):
    """Objective + gradients of this rank's rows, averaged over the ranks through the flat bucket."""
    import zhusuan as zs
    from zhusuan import _hip, dataparallel
    from examples import iwae
    host_backend.install(host_kernel_library())
    dev = torch.device("cpu")
    model = iwae.build(n_samples=K, estimator=estimator, hidden=HID, device=dev)
    if rank == 0:
        H.load_params_into(model, 4242)          # other ranks start from different weights on purpose
    else:
        H.load_params_into(model, 9999)
    dataparallel.broadcast_parameters(model, src=0)
    bucket = dataparallel.GradientBucket(model)
    x, e1, e2 = _data()
    xs = dataparallel.shard_rows(torch.tensor(x), rank, world)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    bucket.zero()
    with zs.inject_epsilon([e1[:, sl], e2[:, sl]]):
        loss = model({"x": xs})
    loss.backward()
    g = bucket.all_reduce_mean(loss)
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    if world > 1:
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        assert torch.equal(flat, torch.cat([v.reshape(-1) for v in bucket.views]))
    return float(g), flat, [p.detach().clone() for p in model.parameters()]


def _run_shard_overlapped(rank, world, estimator, n_buckets):
    import zhusuan as zs
    from zhusuan import _hip, dataparallel
    from examples import iwae
    host_backend.install(host_kernel_library())
    dev = torch.device("cpu")
    model = iwae.build(n_samples=K, estimator=estimator, hidden=HID, device=dev)
    H.load_params_into(model, 4242 if rank == 0 else 9999)
    dataparallel.broadcast_parameters(model, src=0)
    buckets = dataparallel.OverlappedBuckets(model, n_buckets=n_buckets)
    x, e1, e2 = _data()
    xs = dataparallel.shard_rows(torch.tensor(x), rank, world)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    buckets.zero()
    with zs.inject_epsilon([e1[:, sl], e2[:, sl]]):
        loss = model({"x": xs})
    buckets.begin(loss)
    loss.backward()
    launched_during_backward = [b["launched"] for b in buckets.buckets]
    g = buckets.finish()
    assert all(launched_during_backward), "every bucket must leave from its hook, before finish()"
    for b in buckets.buckets:
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(b["params"], b["views"]))
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    return float(g), flat


def _run_shard_staged(rank, world, estimator):
    """Backward in two stages (decoder, then encoder), each stage's bucket all-reduced asynchronously as soon as it is
    packed -- the eager form of the staged hipGraph step of bench.py."""
    import zhusuan as zs
    from zhusuan import _hip, dataparallel
    from examples import iwae
    host_backend.install(host_kernel_library())
    model = iwae.build(n_samples=K, estimator=estimator, hidden=HID, device=torch.device("cpu"))
    H.load_params_into(model, 4242 if rank == 0 else 9999)
    dataparallel.broadcast_parameters(model, src=0)
    sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()])
    x, e1, e2 = _data()
    xs = dataparallel.shard_rows(torch.tensor(x), rank, world)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    sb.zero()
    with zs.inject_epsilon([e1[:, sl], e2[:, sl]]):
        loss = model({"x": xs})
    sb.backward_stage(loss, 0)
    assert all(p.grad is None for p in model.variational.parameters())      # stage 0 touched the decoder only
    sb.launch(0)
    sb.backward_stage(loss, 1)
    sb.launch(1)
    sb.wait()
    sb.scale()
    for st in sb.stages:
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(st["params"], st["views"]))
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    return float(sb.loss()), flat


def _run_shard_staged_update(rank, world, estimator, steps=3):
    """`steps` staged data-parallel training steps with the 1/world folded into FlatAdam's gradient read (no pass over
    the buckets), or -- world 1 -- the same steps on the full minibatch with torch.optim.Adam."""
    import zhusuan as zs
    from zhusuan import _hip, dataparallel
    from examples import iwae
    host_backend.install(host_kernel_library())
    model = iwae.build(n_samples=K, estimator=estimator, hidden=HID, device=torch.device("cpu"))
    H.load_params_into(model, 4242 if rank == 0 else 9999)
    dataparallel.broadcast_parameters(model, src=0)
    x, e1, e2 = _data()
    xs = dataparallel.shard_rows(torch.tensor(x), rank, world)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    losses = []
    if world == 1:
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        for _ in range(steps):
            opt.zero_grad()
            with zs.inject_epsilon([e1, e2]):
                loss = model({"x": xs})
            loss.backward()
            opt.step()
            losses.append(float(loss))
    else:
        sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()])
        opt = zs.optim.FlatAdam([list(model.generator.parameters()), list(model.variational.parameters())], lr=1e-2)
        for _ in range(steps):
            sb.zero()
            with zs.inject_epsilon([e1[:, sl], e2[:, sl]]):
                loss = model({"x": xs})
            sb.backward_stage(loss, 0)
            sb.launch(0)
            sb.backward_stage(loss, 1)
            sb.launch(1)
            sb.wait()
            sb.scale(gradients=False)
            opt.step(grad_scale=sb.grad_scale())
            losses.append(float(sb.loss()))
    return losses, [p.detach().clone() for p in model.parameters()]


def _worker_staged_update(rank, world, port, estimator, out_dir):
    for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        losses, params = _run_shard_staged_update(rank, world, estimator)
        torch.save({"losses": losses, "params": params}, os.path.join(out_dir, "u%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _worker_staged(rank, world, port, estimator, out_dir):
    for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loss, flat = _run_shard_staged(rank, world, estimator)
        torch.save({"loss": loss, "flat": flat}, os.path.join(out_dir, "s%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _worker_overlapped(rank, world, port, estimator, out_dir):
    for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loss, flat = _run_shard_overlapped(rank, world, estimator, 3)
        torch.save({"loss": loss, "flat": flat}, os.path.join(out_dir, "o%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, estimator, out_dir):
    for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loss, flat, params = _run_shard(rank, world, estimator)
        torch.save({"loss": loss, "flat": flat, "p0": params[0]}, os.path.join(out_dir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("estimator", ["vimco", "sgvb"])
def test_two_rank_gloo_matches_single_process(tmp_path, estimator):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, estimator, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(str(tmp_path / "r0.pt"))
    r1 = torch.load(str(tmp_path / "r1.pt"))
    assert r0["loss"] == r1["loss"] and torch.equal(r0["flat"], r1["flat"])      # all-reduced: identical everywhere
    assert torch.equal(r0["p0"], r1["p0"])                                      # broadcast made the replicas equal
    # single process, full minibatch
    loss, flat, _ = _run_shard(0, 1, estimator)
    from zhusuan import _hip
    host_backend.uninstall()
    assert abs(r0["loss"] - loss) <= 2e-6 * abs(loss)
    np.testing.assert_allclose(r0["flat"].numpy(), flat.numpy(), rtol=2e-4, atol=2e-6)


def test_bucket_layout_and_sharding():
    from zhusuan import dataparallel
    lin = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    bucket = dataparallel.GradientBucket(lin)
    # every slice starts on a 512-byte boundary (round 6: the slices are GEMM outputs when the backward pass fills the bucket);
    # the objective's slot is the last element; the padding is zero
    q = dataparallel.SLICE_ALIGN_BYTES // 4
    offs, off = [], 0
    for p in lin.parameters():
        offs.append(off)
        off = (off + p.numel() + q - 1) // q * q
    n = bucket.n_grad
    assert n == off and bucket.flat.numel() == n + 1 and bucket.nbytes() == 4 * (n + 1)
    bucket.zero()
    assert all(p.grad is None for p in lin.parameters())
    lin(torch.ones(5, 3)).sum().backward()
    ref = [p.grad.clone() for p in lin.parameters()]
    g = bucket.all_reduce_mean(torch.tensor(3.0))     # no process group: identity, nothing packed
    assert float(g) == 3.0
    bucket.pack(torch.tensor(3.0))                    # what a multi-rank step does before the all-reduce
    for p, r, o in zip(lin.parameters(), ref, offs):  # .grad now aliases the flat buffer: no unpack copy
        assert p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * o and p.grad.data_ptr() % dataparallel.SLICE_ALIGN_BYTES == bucket.flat.data_ptr() % dataparallel.SLICE_ALIGN_BYTES
        assert torch.equal(p.grad, r)
    assert float(bucket.flat[n]) == 3.0
    pad = torch.ones(n, dtype=torch.bool)
    for p, o in zip(lin.parameters(), offs):
        pad[o:o + p.numel()] = False
    assert int(pad.sum()) > 0 and float(bucket.flat[:n][pad].abs().sum()) == 0.0      # the padding holds zeros
    x = torch.arange(12.).view(6, 2)
    assert torch.equal(dataparallel.shard_rows(x, 1, 3), x[2:4])
    with pytest.raises(ValueError, match="does not split evenly"):
        dataparallel.shard_rows(x, 0, 4)


def test_overlapped_buckets_two_ranks_match_the_single_bucket(tmp_path):
    """OverlappedBuckets (all-reduce launched from autograd hooks while backward is still running) must give exactly the
    averaged gradients and objective of the one-bucket path."""
    world = 2
    port = _free_port()
    mp.spawn(_worker_overlapped, args=(world, port, "vimco", str(tmp_path)), nprocs=world, join=True)
    o0, o1 = torch.load(str(tmp_path / "o0.pt")), torch.load(str(tmp_path / "o1.pt"))
    assert o0["loss"] == o1["loss"] and torch.equal(o0["flat"], o1["flat"])
    loss, flat, _ = _run_shard(0, 1, "vimco")                      # single process, full minibatch
    from zhusuan import _hip
    host_backend.uninstall()
    assert abs(o0["loss"] - loss) <= 2e-6 * abs(loss)
    np.testing.assert_allclose(o0["flat"].numpy(), flat.numpy(), rtol=2e-4, atol=2e-6)


def test_overlapped_buckets_partition_and_hook_order():
    from zhusuan import dataparallel
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Tanh(), torch.nn.Linear(4, 4), torch.nn.Tanh(), torch.nn.Linear(4, 2))
    ob = dataparallel.OverlappedBuckets(net, n_buckets=2)
    params = list(net.parameters())
    assert [p for b in ob.buckets for p in b["params"]] == list(reversed(params))     # backward order
    assert len(ob.buckets) == 2 and ob.nbytes() == 4 * sum(b["flat"].numel() for b in ob.buckets)
    assert ob.buckets[0]["n"] == ob.buckets[0]["flat"].numel() - 1 and ob.buckets[1]["n"] is None      # the objective rides in the first
    order = []
    orig = ob._launch
    ob._launch = lambda bi: (order.append(bi), orig(bi))[1]
    ob.zero()
    loss = net(torch.ones(5, 3)).sum()
    ob.begin(loss)
    loss.backward()
    assert order == [0, 1]                                  # the last layers' bucket leaves first, during backward
    ref = [p.grad.clone() for p in params]
    g = ob.finish()                                         # no process group: nothing is sent, values unchanged
    assert float(g) == float(loss)
    for p, r in zip(params, ref):
        assert torch.equal(p.grad, r)
    ob.remove_hooks()


def test_bucket_dtype_and_repack_without_zero():
    """float64 models get a float64 bucket (mixed dtypes are refused), and pack() works when zero() was
    skipped and autograd has accumulated IN PLACE into the views that alias the flat buffer."""
    from zhusuan import dataparallel
    lin = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2)).double()
    bucket = dataparallel.GradientBucket(lin)
    n = bucket.n_grad
    assert bucket.flat.dtype == torch.float64 and bucket.nbytes() == 8 * (n + 1)
    x = torch.ones(5, 3, dtype=torch.float64)
    lin(x).sum().backward()
    bucket.pack(torch.tensor(1.5, dtype=torch.float64))
    first = bucket.flat.clone()
    lin(x).sum().backward()                                  # no zero(): accumulates into the aliased views
    bucket.pack(torch.tensor(2.5, dtype=torch.float64))      # used to raise (cat with overlapping out=)
    assert torch.allclose(bucket.flat[:n], 2 * first[:n]) and float(bucket.flat[n]) == 2.5
    ob = dataparallel.OverlappedBuckets(lin, n_buckets=2)
    assert all(b["flat"].dtype == torch.float64 for b in ob.buckets)
    ob.remove_hooks()
    mixed = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2).double())
    with pytest.raises(TypeError, match="share a dtype"):
        dataparallel.GradientBucket(mixed)
    with pytest.raises(TypeError, match="share a dtype"):
        dataparallel.OverlappedBuckets(mixed)


def test_overlapped_buckets_issue_collectives_in_bucket_order():
    """A bucket that completes early (its parameters got their gradients first, or none at all) still leaves after
    every earlier bucket: all ranks issue the same collective sequence."""
    from zhusuan import dataparallel
    a, b, c = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    net = torch.nn.ModuleList([a, b, c])
    ob = dataparallel.OverlappedBuckets(net, n_buckets=3)
    assert len(ob.buckets) == 3                              # bucket 0 = c's parameters, 1 = b's, 2 = a's
    order = []
    orig = ob._launch
    ob._launch = lambda bi: (order.append(bi), orig(bi))[1]
    ob.zero()
    x = torch.ones(4, 2)
    loss = a(x).sum() + c(x).sum()                           # b receives no gradient; a's arrive in whatever order
    ob.begin(loss)
    loss.backward()
    assert order in ([0], [])                                # bucket 2 (a) is complete but must wait for bucket 1 (b)
    ob.finish()
    assert order == [0, 1, 2]
    assert all(torch.equal(p.grad, torch.zeros_like(p)) for p in b.parameters())
    ob.remove_hooks()


@pytest.mark.parametrize("estimator", ["vimco", "sgvb"])
def test_staged_buckets_two_ranks_match_the_single_bucket(tmp_path, estimator):
    """StagedBuckets (backward split at the decoder / encoder boundary, one asynchronous all-reduce per stage: the
    eager form of the staged hipGraph step) gives the averaged gradients and objective of the one-bucket path.  With
    sgvb the encoder's gradients flow THROUGH the decoder: stage 1 then walks the retained graph again -- slower,
    still exact."""
    world = 2
    port = _free_port()
    mp.spawn(_worker_staged, args=(world, port, estimator, str(tmp_path)), nprocs=world, join=True)
    s0, s1 = torch.load(str(tmp_path / "s0.pt")), torch.load(str(tmp_path / "s1.pt"))
    assert s0["loss"] == s1["loss"] and torch.equal(s0["flat"], s1["flat"])
    loss, flat, _ = _run_shard(0, 1, estimator)                    # single process, full minibatch, one bucket
    from zhusuan import _hip
    host_backend.uninstall()
    assert abs(s0["loss"] - loss) <= 2e-6 * abs(loss)
    np.testing.assert_allclose(s0["flat"].numpy(), flat.numpy(), rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("world", [2, 4])
def test_staged_update_with_flat_adam_matches_full_batch_adam(tmp_path, world):
    """Three staged 2-rank (4-rank) steps whose update reads the all-reduced SUMS with grad_scale = 1/world (no scaling pass
    over the buckets) train exactly like torch.optim.Adam on the full minibatch in one process."""
    port = _free_port()
    mp.spawn(_worker_staged_update, args=(world, port, "vimco", str(tmp_path)), nprocs=world, join=True)
    u0, u1 = torch.load(str(tmp_path / "u0.pt")), torch.load(str(tmp_path / "u1.pt"))
    assert u0["losses"] == u1["losses"]
    for a, b in zip(u0["params"], u1["params"]):
        assert torch.equal(a, b)
    losses, params = _run_shard_staged_update(0, 1, "vimco")
    from zhusuan import _hip
    host_backend.uninstall()
    # (the first step's objective is the plain shard mean; later steps see parameters that went through Adam's normalisation of
    # near-zero gradients, where the order of the shard sums shows)
    np.testing.assert_allclose(u0["losses"][0], losses[0], rtol=5e-6)
    np.testing.assert_allclose(u0["losses"], losses, rtol=5e-6 if world == 2 else 5e-5)
    outliers = total = 0
    for a, b in zip(u0["params"], params):
        # Adam normalises every gradient by its own magnitude: where a gradient is ~0 the shard-sum's rounding decides the
        # update's size, so the allowance is a fraction of a step (lr = 1e-2), not of the parameter
        if world == 2:
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-4, atol=2e-4)
        else:
            # four shard sums: a handful of elements whose gradient is exactly zero up to rounding get the rounding's SIGN
            # normalised to a full step (Adam: m / sqrt(v) = +-1) -- at most steps * lr apart, and rare
            d = (a - b).abs()
            outliers += int((d > 6e-4 + 2e-4 * b.abs()).sum())
            total += d.numel()
            assert float(d.max()) <= 3 * 1e-2 + 1e-6
    assert outliers <= 1e-3 * max(total, 1), (outliers, total)


# ------------------------------------------------------------------ 8 ranks against the global-batch golden of BASELINE config 4
def _worker_c4(rank, world, port, out_dir):
    """Rank `rank` of 8: rows [256 r, 256 (r + 1)) of the B = 2048 minibatch of config 4 (IWAE-VIMCO, K = 50, hidden 500) with the
    matching slices of the golden's epsilon draws, per-rank model replica (rank 0's weights broadcast), staged buckets: decoder
    gradients all-reduced while the encoder's backward runs, then the 1/world mean."""
    for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import zhusuan as zs
        from zhusuan import dataparallel
        from examples import iwae
        host_backend.install(host_kernel_library())
        Bg, Kg = 2048, 50
        model = iwae.build(n_samples=Kg, estimator="vimco", hidden=500, device=torch.device("cpu"))
        H.load_params_into(model, (2000 + Bg + Kg) if rank == 0 else 1)       # replicas differ until the broadcast
        dataparallel.broadcast_parameters(model, src=0)
        x, e1, e2 = H.iwae_data(Bg, Kg)
        per = Bg // world
        sl = slice(rank * per, (rank + 1) * per)
        xs = dataparallel.shard_rows(torch.tensor(x), rank, world)
        assert torch.equal(xs, torch.tensor(x[sl]))
        sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()])
        sb.zero()
        with zs.inject_epsilon([e1[:, sl], e2[:, sl]]):
            loss = model({"x": xs})
        sb.backward_stage(loss, 0)
        sb.launch(0)
        sb.backward_stage(loss, 1)
        sb.launch(1)
        sb.wait()
        sb.scale()
        bound = model.last_iw_bound.detach().mean().reshape(1).clone()
        dist.all_reduce(bound)
        out = {"loss": float(sb.loss()), "bound": float(bound) / world, "n_ranks": dist.get_world_size(),
               "bound_rows": model.last_iw_bound.detach().clone()}
        if rank in (0, world - 1):
            out["grads"] = dict((n, p.grad.detach().clone()) for n, p in model.named_parameters())
        torch.save(out, os.path.join(out_dir, "c4_%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_reproduces_the_global_batch_golden_of_config_4(tmp_path):
    """BASELINE config 4 as it is meant to run -- 8 ranks x 256 rows, one all-reduce of [gradients | objective] -- against
    the golden that the real reference produced for the WHOLE B = 2048 minibatch in one process (g_iwae_vimco_c4g): the mean
    of the 8 shard means is the global mean.  CPU, gloo, C oracle as the kernel library; every rank ends up with
    the same averaged gradients."""
    from conftest import load_golden
    from test_end_to_end import _check_grads
    world = 8
    port = _free_port()
    mp.spawn(_worker_c4, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(str(tmp_path / ("c4_%d.pt" % r))) for r in range(world)]
    g = load_golden("g_iwae_vimco_c4g")
    assert all(o["n_ranks"] == 8 for o in outs)
    assert len({o["loss"] for o in outs}) == 1 and len({o["bound"] for o in outs}) == 1       # all-reduced: identical everywhere
    assert abs(outs[0]["loss"] - float(g["loss"])) <= 5e-5 * abs(float(g["loss"]))
    assert abs(outs[0]["bound"] - float(g["iw_bound"])) <= 2e-5 * abs(float(g["iw_bound"]))
    bound = torch.cat([o["bound_rows"] for o in outs]).numpy()                                 # rank order = row order
    np.testing.assert_allclose(bound[::16], g["bound_b_every16"], rtol=2e-5, atol=2e-4)
    for n in outs[0]["grads"]:
        assert torch.equal(outs[0]["grads"][n], outs[-1]["grads"][n])

    class Holder(object):            # _check_grads walks named_parameters() with .grad
        def __init__(self, grads):
            self._g = grads

        def named_parameters(self):
            for n, v in self._g.items():
                p = torch.nn.Parameter(torch.zeros_like(v))
                p.grad = v
                yield n, p
    assert _check_grads(g, Holder(outs[0]["grads"]), rtol_norm=1e-3) > 1000


def _worker_agree(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from zhusuan import dataparallel
        # the protocol of zhusuan.GraphedStages(agree=...): after every capture attempt the ranks meet; rank 1's second "capture" fails
        attempts = [(True, True), (True, False), (False, True), (True, True)]
        votes = [dataparallel.all_ranks_agree(a[rank]) for a in attempts]
        # ... and the stages launched eagerly afterwards issue the same collectives on both ranks (the fallback of bench.py)
        loss, flat = _run_shard_staged(rank, world, "vimco")
        torch.save({"votes": votes, "loss": loss, "flat": flat}, os.path.join(out_dir, "a%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_launch_mode_and_the_eager_fallback_stays_matched(tmp_path):
    """a rank whose graph capture fails must not change the number or size of its collectives alone.  The
    ranks vote (all-reduce MIN) at fixed points and fall back together; the fallback is the same two staged all-reduces."""
    from zhusuan import dataparallel
    assert dataparallel.all_ranks_agree(True) is True and dataparallel.all_ranks_agree(False) is False       # no process group
    world = 2
    port = _free_port()
    mp.spawn(_worker_agree, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a0, a1 = torch.load(str(tmp_path / "a0.pt")), torch.load(str(tmp_path / "a1.pt"))
    assert a0["votes"] == a1["votes"] == [True, False, False, True]
    assert a0["loss"] == a1["loss"] and torch.equal(a0["flat"], a1["flat"])
    loss, flat, _ = _run_shard(0, 1, "vimco")
    host_backend.uninstall()
    assert abs(a0["loss"] - loss) <= 2e-6 * abs(loss)
    np.testing.assert_allclose(a0["flat"].numpy(), flat.numpy(), rtol=2e-4, atol=2e-6)


# ---------------------------------------------------------------------------------------------
# Round 6: gradients written into the buckets by the backward pass itself (no pack copy), and backward launches narrowed to
# what the running pass is after (the engine is asked; no module state).
# ---------------------------------------------------------------------------------------------
def _staged_grads(dense, direct, estimator, spy=None):
    import zhusuan as zs
    from zhusuan import dataparallel
    from examples import iwae
    model = iwae.build(n_samples=K, estimator=estimator, hidden=HID, device=torch.device("cpu"), dense=dense)
    H.load_params_into(model, 4242)
    sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()], direct=direct)
    try:
        x, e1, e2 = _data()
        sb.zero()
        with zs.inject_epsilon([e1, e2]):
            loss = model({"x": torch.tensor(x)})
        if spy is not None:
            spy(True)
        sb.backward_stage(loss, 0)
        assert all(p.grad is None for p in model.variational.parameters())
        sb.backward_stage(loss, 1)
        if spy is not None:
            spy(False)
        for st in sb.stages:
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(st["params"], st["views"]))
        return float(sb.loss()), torch.cat([v.reshape(-1) for st in sb.stages for v in st["views"]]).clone(), float(loss.detach())
    finally:
        sb.release()


@pytest.mark.parametrize("estimator", ["vimco", "sgvb"])
def test_backward_writes_gradients_into_the_buckets_without_a_pack_copy(monkeypatch, estimator):
    """zhusuan.Linear layers: weight and bias gradients land in their bucket slices during backward (autograd adopts the
    aliases); what is left for the bucket to copy is the objective's scalar.  Same numbers as the copying bucket."""
    from zhusuan import dataparallel, _ops
    host_backend.install(host_kernel_library())
    try:
        copies = []
        real_cat, real_fc = torch.cat, torch._foreach_copy_
        on = [False]
        monkeypatch.setattr(torch, "cat", lambda ts, *a, **k: (copies.append(("cat", len(ts))) if on[0] and "out" in k else None) or real_cat(ts, *a, **k))
        monkeypatch.setattr(torch, "_foreach_copy_", lambda d, s_: (copies.append(("foreach_copy", len(d))) if on[0] else None) or real_fc(d, s_))
        loss_d, flat_d, l_d = _staged_grads("fused", True, estimator, spy=lambda v: on.__setitem__(0, v))
        assert copies == [], copies                       # no gradient was copied into a bucket
        import gc
        gc.collect()
        assert not _ops._GRAD_DEST      # release() withdrew every registration (and those of dead models went with their parameters)
        loss_c, flat_c, l_c = _staged_grads("fused", False, estimator, spy=lambda v: on.__setitem__(0, v))
        assert [c[0] for c in copies] == ["foreach_copy", "foreach_copy"]   # the copying form: one multi-tensor copy per stage
        assert loss_d == l_d == loss_c == l_c
        assert torch.equal(flat_d, flat_c)
        # torch.nn modules know nothing about destinations: their gradients still reach the bucket through the copy
        loss_t, flat_t, _ = _staged_grads("torch", True, estimator)
        assert torch.allclose(flat_t, flat_d, rtol=1e-4, atol=1e-6)
    finally:
        host_backend.uninstall()


def test_gradient_destinations_weight_used_twice_and_accumulation_without_zero():
    """One producer per parameter per pass may write the slice: a layer applied twice in one graph (the second use allocates
    and autograd adds), a parameter that still holds last step's gradient (never written over: autograd accumulates)."""
    import zhusuan as zs
    from zhusuan import dataparallel
    host_backend.install(host_kernel_library())
    try:
        torch.manual_seed(3)
        lin = zs.Linear(6, 6, activation="relu")
        ref = torch.nn.Linear(6, 6)
        ref.load_state_dict(lin.state_dict())
        x = torch.randn(5, 6)
        bucket = dataparallel.GradientBucket(lin)
        try:
            for rounds in (1, 2):          # second round: no zero() in between -> accumulation onto the aliased slices
                loss = lin(lin(x)).sum()
                loss.backward()
                bucket.pack(loss)
                rl = torch.relu(ref(torch.relu(ref(x)))).sum()
                rl.backward()
                for p, q in zip(lin.parameters(), ref.parameters()):
                    assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6), rounds
                assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
                assert float(bucket.flat[bucket.n_grad]) == float(loss)
            bucket.zero()
            ref.zero_grad()
            loss = lin(x).sum()
            loss.backward()
            torch.relu(ref(x)).sum().backward()
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))      # adopted, not copied
            for p, q in zip(lin.parameters(), ref.parameters()):
                assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6)
        finally:
            bucket.release()
    finally:
        host_backend.uninstall()


def test_a_restricted_pass_launches_only_the_sides_it_is_after():
    """torch.autograd.grad(y, [x]) through a dense layer: no bias reduction, no weight GEMM result; the engine is asked per
    pass (two passes over one retained graph get different answers), nothing is held in module state."""
    import zhusuan as zs
    from zhusuan import _ops
    klib = host_kernel_library()
    host_backend.install(klib)
    calls, real = [], klib.call
    klib.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        lin = zs.Linear(6, 4)
        x = torch.randn(5, 6, requires_grad=True)
        y = lin(x).sum()
        (gx,) = torch.autograd.grad(y, [x], retain_graph=True)
        assert calls == [] and torch.allclose(gx, lin.weight.detach().sum(0).expand(5, 6))
        torch.autograd.backward(y, inputs=[lin.bias], retain_graph=True)
        assert calls == ["zs_column_sum_f32"] and lin.weight.grad is None and x.grad is None
        y.backward()
        assert lin.weight.grad is not None and x.grad is not None and torch.allclose(lin.bias.grad, torch.full((4,), 10.0))
        assert not hasattr(_ops, "_GRAD_TARGETS")
    finally:
        klib.call = real
        host_backend.uninstall()


@pytest.mark.parametrize("estimator", ["vimco", "sgvb"])
def test_staged_backward_cut_at_the_encoder_outputs_runs_the_objective_backward_once(estimator):
    """backward_stage(..., also=cut) / backward_stage(..., roots=cut): the first stage delivers the objective's gradient at the
    variational net's outputs, the second starts there -- same gradients as two walks from the loss, IW1's backward launched
    once instead of twice (vimco), the decoder walked once instead of twice (sgvb)."""
    import zhusuan as zs
    from zhusuan import dataparallel
    from examples import iwae
    klib = host_kernel_library()
    host_backend.install(klib)
    calls, real = [], klib.call
    klib.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        out = {}
        for cut in (False, True):
            model = iwae.build(n_samples=K, estimator=estimator, hidden=HID, device=torch.device("cpu"), dense="fused")
            H.load_params_into(model, 4242)
            sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()])
            try:
                x, e1, e2 = _data()
                sb.zero()
                with zs.inject_epsilon([e1, e2]):
                    loss = model({"x": torch.tensor(x)})
                del calls[:]
                if cut:
                    q = model.variational.nodes["z"].dist
                    boundary = [q.mean, q.std]
                    sb.backward_stage(loss, 0, also=boundary)
                    assert all(t.grad is not None for t in boundary)
                    assert all(p.grad is None for p in model.variational.parameters())
                    n0 = len(calls)
                    sb.backward_stage(None, 1, roots=boundary)
                    assert all(t.grad is None for t in boundary)
                    # the second stage is the encoder's MLP and nothing else: no objective, no sampling, no decoder kernels
                    assert set(calls[n0:]) <= {"zs_dense_act_bwd_f32", "zs_column_sum_f32"}, calls[n0:]
                else:
                    sb.backward_stage(loss, 0)
                    sb.backward_stage(loss, 1)
                for st in sb.stages:
                    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(st["params"], st["views"]))
                out[cut] = (torch.cat([v.reshape(-1) for st in sb.stages for v in st["views"]]).clone(), list(calls))
            finally:
                sb.release()
        assert torch.allclose(out[True][0], out[False][0], rtol=1e-6, atol=1e-7)
        n_obj = lambda names: sum(n.startswith(("zs_bernoulli_iw_objective_bwd", "zs_iw_objective")) for n in names)
        assert len(out[True][1]) < len(out[False][1])
        if estimator == "vimco":
            assert sum(n == "zs_bernoulli_iw_objective_bwd_f32" for n in out[True][1]) == 1
            assert sum(n == "zs_bernoulli_iw_objective_bwd_f32" for n in out[False][1]) == 2
    finally:
        klib.call = real
        host_backend.uninstall()


@pytest.mark.gpu
def test_direct_rccl_communicator_on_one_rank():
    """zhusuan.dataparallel.DirectAllReduce with a process group of one over RCCL (a child process: this one keeps no process
    group): the communicator comes up, an all-reduce on a side stream is stream-ordered and leaves a one-rank sum unchanged,
    the bucket's exchange() goes through it, close() tears it down."""
    import subprocess
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "zhusuan-pytorch_amd"))
import torch, torch.distributed as dist
from zhusuan import dataparallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1, device_id=torch.device("cuda", 0))
d = dataparallel.DirectAllReduce.create(timeout_s=60)
assert d is not None, dataparallel.DirectAllReduce.last_error
assert d.world == 1
x = torch.arange(1000, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    y = x * 2
    d.all_reduce_sum_(y)
    z = y + 1
side.synchronize()
assert torch.equal(z, x * 2 + 1)
lin = torch.nn.Linear(8, 4).cuda()
b = dataparallel.GradientBucket(lin)
lin(torch.ones(3, 8, device="cuda")).sum().backward()
b.pack(torch.tensor(2.5, device="cuda"))
before = b.flat.clone()
b.exchange(direct=d, always=True)
torch.cuda.synchronize()
assert torch.equal(b.flat, before) and float(b.loss()) == 2.5 and b.grad_scale() == 1.0
d.close()
dist.destroy_process_group()
print("direct rccl ok")
''' % (ROOT, _free_port())
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "direct rccl ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def _worker_direct_setup(rank, world, port, fail_rank, out_dir):
    """Two real processes on GPU 0 over gloo; RCCL's four entry points replaced by stand-ins (RCCL refuses two ranks on one device):
    what is exercised is DirectAllReduce.create's COLLECTIVE logic -- the unique id travelling from rank 0, the bounded set-up
    thread, the probe's element-wise check, the ranks' agreement."""
    sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
    from zhusuan import dataparallel, _rccl
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    seen = {}

    def fake_init(nranks, uid, r):
        seen["uid"] = uid
        if r == fail_rank:
            raise RuntimeError("injected set-up failure on rank %d" % r)
        return "comm-%d" % r

    def fake_all_reduce(t, comm, stream):
        if fail_rank >= 0:           # (a failing peer never joins: a real collective would wait; the stand-in returns garbage)
            t.fill_(-1.0)
            return
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)
    _rccl.comm_init_rank, _rccl.all_reduce_sum_, _rccl.comm_destroy = fake_init, fake_all_reduce, lambda comm: None
    dataparallel.DirectAllReduce.REQUIRED_BACKEND = "gloo"
    d = dataparallel.DirectAllReduce.create(timeout_s=30)
    uids = [None] * world
    dist.all_gather_object(uids, seen.get("uid"))
    ok = {"created": d is not None, "same_uid": all(u == uids[0] and u is not None and len(u) == 128 for u in uids),
          "error": dataparallel.DirectAllReduce.last_error}
    if d is not None:
        x = torch.full((1000,), float(rank + 1), device="cuda")
        d.all_reduce_sum_(x)
        ok["sum"] = float(x[0])
        d.close()
    torch.save(ok, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_direct_communicator_set_up_is_a_joint_decision(tmp_path, fail_rank):
    """Every rank gets a communicator, or none does (then all fall back to torch.distributed's all_reduce together): rank 1's
    set-up made to fail takes rank 0's perfectly good communicator away as well."""
    world = 2
    mp.spawn(_worker_direct_setup, args=(world, _free_port(), fail_rank, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    assert all(r["same_uid"] for r in res)                       # rank 0's id reached everybody
    if fail_rank < 0:
        assert all(r["created"] and r["sum"] == 3.0 for r in res)
    else:
        assert not any(r["created"] for r in res), res
        assert "injected" in res[1]["error"] and res[0]["error"]


@pytest.mark.gpu
@pytest.mark.parametrize("direct", [True, False])
def test_the_two_graph_bucket_step_trains_bit_for_bit_like_the_single_graph(direct):
    """The multi-rank form of the step as bench.py records it -- graph A = forward + backward INTO the flat bucket + the
    objective's slot, (the collective's place), graph B = FlatAdam reading the bucket with grad_scale -- against the single
    graph of the same model, seed and optimizer: the same objective on each of 40 replays and the same parameters at the end,
    exactly (a one-rank sum changes nothing; the gradients' arithmetic does not depend on where they are written)."""
    import zhusuan as zs
    from examples import iwae
    from zhusuan import dataparallel
    dev = torch.device("cuda:0")

    def make():
        torch.manual_seed(11)
        model = iwae.build(5, "vimco", hidden=64, device=dev, dense="fused")
        return model, zs.optim.FlatAdam(model.parameters(), lr=1e-3), zs.DeviceRNG(dev, seed=7)
    x = {"x": (torch.rand(16, 784, device=dev) < 0.5).float()}
    one = torch.ones((), device=dev)
    m1, o1, r1 = make()

    def compute_single():
        r1.begin_step()
        for p in m1.parameters():
            p.grad = None
        loss = m1(x)
        loss.backward(one)
        return loss.detach()
    single = zs.GraphedStep(compute_single, o1.step, rng=r1, warmup=3, restore=True)
    m2, o2, r2 = make()
    bucket = dataparallel.GradientBucket(m2, direct=direct)
    seen = []

    def compute_part():
        r2.begin_step()
        bucket.zero()
        loss = m2(x)
        loss.backward(one)
        bucket.pack(loss)
        return loss.detach()

    def exchange_part(loss):
        seen.append(1)
        return bucket.flat[bucket.n_grad]
    try:
        two = zs.GraphedStep(compute_part, lambda: o2.step(grad_scale=1.0), exchange=exchange_part, rng=r2, warmup=3, restore=True,
                             optimizer=o2)
        assert len(two.graphs) == 2
        if direct:       # the backward wrote every dense layer's gradients where the bucket keeps them
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        seen.clear()
        for i in range(40):
            a, b = single(), two()
            assert float(a) == float(b), (i, float(a), float(b))
        assert len(seen) == 40
        for p1, p2 in zip(m1.parameters(), m2.parameters()):
            assert torch.equal(p1, p2)
    finally:
        bucket.release()

"""zhusuan.optim.FlatAdam: torch.optim.Adam's update (the reference callers' optimizer: examples vae_mnist.py:104,
iwae.py:141, bnn_vi.py:135) in one launch per 32 parameter tensors.  Product-level tests on both back-ends
(conftest.py: `host` = the package's Python logic over the C oracle, `hip` = the real library)."""
import copy

import numpy as np
import pytest
import torch

import host_backend
import zhusuan
from zhusuan import dataparallel
from zhusuan.optim import FlatAdam


def _model(dev, dtype=torch.float32, seed=0):
    torch.manual_seed(seed)
    m = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.ReLU(), torch.nn.Linear(13, 5), torch.nn.Tanh(),
                            torch.nn.Linear(5, 3))
    return m.to(device=dev, dtype=dtype)


def _loss(m, x):
    return (m(x) ** 2).sum() + sum((p ** 3).sum() for p in m.parameters()) * 1e-2


def _train(m, opt, x, steps, scale=1.0):
    for _ in range(steps):
        opt.zero_grad()
        loss = _loss(m, x) * scale
        loss.backward()
        opt.step()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-6), (torch.float64, 1e-12)])
def test_flat_adam_equals_torch_adam(dev, dtype, tol):
    a, b = _model(dev, dtype), _model(dev, dtype)
    x = torch.randn(11, 7, device=dev, dtype=dtype)
    ref = torch.optim.Adam(a.parameters(), lr=3e-3)
    opt = FlatAdam(b.parameters(), lr=3e-3)
    assert len(opt.buckets) == 1 and opt.buckets[0].n == sum(p.numel() for p in b.parameters())
    where = [p.data_ptr() for p in b.parameters()]
    _train(a, ref, x, 8)
    _train(b, opt, x, 8)
    for p, q in zip(a.parameters(), b.parameters()):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=tol, atol=tol * 1e-1)
    assert opt.buckets[0].step.tolist() == [8] * len(list(b.parameters()))
    assert where == [p.data_ptr() for p in b.parameters()], "parameters are updated where they live"


def test_flat_adam_two_buckets_and_grad_scale(dev):
    a, b = _model(dev, seed=1), _model(dev, seed=1)
    x = torch.randn(9, 7, device=dev)
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, betas=(0.8, 0.95), eps=1e-6)
    pb = list(b.parameters())
    opt = FlatAdam([pb[4:], pb[:4]], lr=1e-2, betas=(0.8, 0.95), eps=1e-6)
    assert [bk.n for bk in opt.buckets] == [sum(p.numel() for p in pb[4:]), sum(p.numel() for p in pb[:4])]       # a launch each
    for _ in range(5):
        ref.zero_grad()
        _loss(a, x).backward()
        ref.step()
        opt.zero_grad()
        (_loss(b, x) * 4.0).backward()          # gradients four times too large ...
        opt.step(grad_scale=0.25)               # ... read as a quarter (the 1/world of a gradient mean)
    for p, q in zip(a.parameters(), b.parameters()):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_flat_adam_reads_a_gradient_bucket_in_place(dev):
    """Gradients that are slices of a data-parallel flat bucket (zhusuan.dataparallel re-points p.grad so) are read where
    they are."""
    a, b = _model(dev, seed=2), _model(dev, seed=2)
    x = torch.randn(6, 7, device=dev)
    ref = torch.optim.Adam(a.parameters(), lr=1e-3)
    bucket = dataparallel.GradientBucket(b)
    opt = FlatAdam(b.parameters(), lr=1e-3)
    for _ in range(3):
        ref.zero_grad()
        _loss(a, x).backward()
        ref.step()
        bucket.zero()
        loss = _loss(b, x)
        loss.backward()
        bucket.pack(loss)                        # what all_reduce_mean does with more than one rank
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        opt.step()
    for p, q in zip(a.parameters(), b.parameters()):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=5e-6, atol=1e-7)


def test_flat_adam_many_parameters_take_several_launches(dev):
    """More tensors than the kernel's pointer table holds (32): one launch per 32, same result; float64 parameters of
    the same model form their own launches."""
    torch.manual_seed(4)
    ws_a = [torch.nn.Parameter(torch.randn(5, 3, device=dev)) for _ in range(40)] + \
           [torch.nn.Parameter(torch.randn(4, device=dev, dtype=torch.float64)) for _ in range(2)]
    ws_b = [torch.nn.Parameter(w.detach().clone()) for w in ws_a]
    ref, opt = torch.optim.Adam(ws_a, lr=1e-2), FlatAdam(ws_b, lr=1e-2)
    assert [len(bk.params) for bk in opt.buckets] == [32, 8, 2]
    for it in range(4):
        for ws, o in ((ws_a, ref), (ws_b, opt)):
            o.zero_grad()
            sum(((w * (i + 1 + it)).sin() ** 2).sum() for i, w in enumerate(ws)).backward()
            o.step()
    for p, q in zip(ws_a, ws_b):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_flat_adam_leaves_parameters_without_a_gradient_alone(dev):
    """torch.optim.Adam skips a parameter whose ``grad is None``: no moment decay, no movement, its own step count does
    not advance.  Alternating training of two parameter subsets (non-zero moments when a tensor is skipped) must therefore
    equal torch.optim.Adam exactly, step for step."""
    a, b = _model(dev, seed=3), _model(dev, seed=3)
    ref, opt = torch.optim.Adam(a.parameters(), lr=1e-2), FlatAdam(b.parameters(), lr=1e-2)
    x = torch.randn(4, 7, device=dev)
    for it in range(6):
        for m, o in ((a, ref), (b, opt)):
            o.zero_grad(set_to_none=True)
            if it % 2 == 0:
                _loss(m, x).backward()                                # every parameter
            else:
                (m[2](torch.relu(m[0](x))) ** 2).sum().backward()     # the last layer receives no gradient
            o.step()
        for p, q in zip(a.parameters(), b.parameters()):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=5e-6, atol=5e-7)
    assert opt.buckets[0].step.tolist() == [6, 6, 6, 6, 3, 3]
    assert [int(ref.state[p]["step"]) for p in a.parameters()] == [6, 6, 6, 6, 3, 3]


def test_flat_adam_param_groups_schedulers_and_checkpoints(dev):
    """param_groups is persistent (the `for g in opt.param_groups: g['lr'] = ...` idiom acts on the
    update), state_dict / load_state_dict resume a run exactly."""
    a, b = _model(dev, seed=6), _model(dev, seed=6)
    x = torch.randn(5, 7, device=dev)
    ref, opt = torch.optim.Adam(a.parameters(), lr=1e-2), FlatAdam(b.parameters(), lr=1e-2)
    assert opt.param_groups is opt.param_groups and opt.param_groups[0]["lr"] == 1e-2
    sa = torch.optim.lr_scheduler.StepLR(ref, step_size=2, gamma=0.5)
    for it in range(5):
        _train(a, ref, x, 1)
        sa.step()
        _train(b, opt, x, 1)
        if it % 2 == 1:
            for g in opt.param_groups:
                g["lr"] = g["lr"] * 0.5
    assert opt.param_groups[0]["lr"] == ref.param_groups[0]["lr"] == opt.lr
    with pytest.raises(TypeError):          # torch's scheduler classes want a torch.optim.Optimizer: documented, not claimed
        torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
    for p, q in zip(a.parameters(), b.parameters()):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=5e-6, atol=5e-7)
    # checkpoint, continue, and resume a copy from the checkpoint: same parameters
    ck = opt.state_dict()
    params_at_ck = [q.detach().clone() for q in b.parameters()]
    _train(b, opt, x, 3)
    c = _model(dev, seed=6)
    with torch.no_grad():
        for q, v in zip(c.parameters(), params_at_ck):
            q.copy_(v)
    opt_c = FlatAdam(c.parameters(), lr=123.0)
    opt_c.load_state_dict(ck)
    assert opt_c.lr == opt.lr and opt_c.buckets[0].step.tolist() == [5] * 6
    _train(c, opt_c, x, 3)
    for p, q in zip(b.parameters(), c.parameters()):
        assert torch.equal(p.detach(), q.detach())
    with pytest.raises(ValueError, match="another parameter layout"):
        FlatAdam([torch.nn.Parameter(torch.zeros(3, device=dev))]).load_state_dict(ck)
    with pytest.raises(ValueError, match="invalid hyper-parameters"):
        opt.lr = -1.0
        opt.step()


def test_flat_adam_argument_errors(dev):
    m = _model(dev)
    with pytest.raises(ValueError, match="empty parameter list"):
        FlatAdam([])
    with pytest.raises(ValueError, match="Invalid learning rate"):
        FlatAdam(m.parameters(), lr=-1.0)
    with pytest.raises(ValueError, match="Invalid beta"):
        FlatAdam(m.parameters(), betas=(1.0, 0.9))
    with pytest.raises(ValueError, match="Invalid epsilon"):
        FlatAdam(m.parameters(), eps=-1e-8)
    ps = list(m.parameters())
    with pytest.raises(ValueError, match="more than one group"):
        FlatAdam([ps[:2], ps[1:]])
    with pytest.raises(TypeError, match="float32 or float64"):
        FlatAdam([torch.nn.Parameter(torch.zeros(3, device=dev, dtype=torch.float16))])


def test_flat_adam_refuses_cpu_parameters_without_the_library():
    """No CPU path: parameters that are not on the GPU are refused (outside the tests' host-library hook)."""
    from zhusuan import _hip
    host_backend.uninstall()
    with pytest.raises(RuntimeError, match="MI355X build"):
        FlatAdam(_model(torch.device("cpu")).parameters())


@pytest.mark.gpu
def test_flat_adam_in_a_graphed_step():
    """The step count is device state: opt.step can be captured; graph replays equal eager steps, and `restore` puts
    moments, step count and parameters back after the warm-up."""
    dev = torch.device("cuda:0")
    a, b = _model(dev, seed=5), _model(dev, seed=5)
    x = torch.randn(16, 7, device=dev)
    ea, eb = FlatAdam(a.parameters(), lr=2e-3), FlatAdam(b.parameters(), lr=2e-3)
    _train(a, ea, x, 6)

    def compute():
        for p in b.parameters():
            p.grad = None
        loss = _loss(b, x)
        loss.backward()
        return loss.detach()

    step = zhusuan.GraphedStep(compute, eb.step, warmup=3, restore=True)
    assert eb.buckets[0].step.tolist() == [0] * 6
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    assert eb.buckets[0].step.tolist() == [6] * 6
    for p, q in zip(a.parameters(), b.parameters()):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_graphed_flat_adam_follows_learning_rate_changes():
    """lr / betas / eps are read from device memory by the captured launch: changing param_groups between replays acts."""
    dev = torch.device("cuda:0")
    a, b = _model(dev, seed=8), _model(dev, seed=8)
    x = torch.randn(16, 7, device=dev)
    ea, eb = FlatAdam(a.parameters(), lr=2e-3), FlatAdam(b.parameters(), lr=2e-3)

    def compute():
        for p in b.parameters():
            p.grad = None
        loss = _loss(b, x)
        loss.backward()
        return loss.detach()

    step = zhusuan.GraphedStep(compute, eb.step, warmup=3, restore=True)
    for it in range(6):
        if it == 3:
            ea.lr = 5e-4
            for g in eb.param_groups:
                g["lr"] = 5e-4
        _train(a, ea, x, 1)
        step()
    torch.cuda.synchronize()
    for p, q in zip(a.parameters(), b.parameters()):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)

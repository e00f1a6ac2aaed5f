"""Adam in one kernel launch per (up to) 32 parameter tensors (C ABI: zs_adam_step, include/zs_hip.h).

The reference's callers train with ``torch.optim.Adam(model.parameters(), lr)`` (examples
variational_autoencoder/vae_mnist.py:104, iwae.py:141, bayesian_neural_nets/bnn_vi.py:135).  ``FlatAdam`` applies the
same update (defaults betas (0.9, 0.999), eps 1e-8; no weight decay, no amsgrad).  The parameters of a bucket form one
flat index space; parameters and gradients are read where they live through a pointer table in the kernel arguments
(nothing is re-pointed or packed: a data-parallel gradient bucket's slices and a single process's per-parameter
gradient tensors go in alike), both moments are flat buffers, a thread owns four consecutive elements.  PyTorch's
multi-tensor kernel gives every 64 K-element chunk one workgroup -- the 1.35 M parameters of the VAE / IWAE models: 21
workgroups on 256 CUs, 43 us per step against 11-14 us here -- and the 1/world factor of a gradient mean rides along in
the read.  Step counts (one per tensor) and the hyper-parameters live on the device: the update can be captured in a
hipGraph (``zhusuan.GraphedStep(compute, opt.step)``) and still follows learning-rate changes.  A parameter without a
gradient (``p.grad is None``) is left alone, as by ``torch.optim.Adam``.

    opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    opt.zero_grad(); loss = model(obs); loss.backward(); opt.step()

    # data-parallel: after the all-reduce (SUM) of the gradient buckets
    opt.step(grad_scale=1.0 / world)
"""
import ctypes

import torch

from . import _hip

__all__ = ['FlatAdam']


class _Bucket(object):
    """At most 32 parameter tensors of one dtype on one device: one launch."""

    def __init__(self, params, group):
        self.params = params
        self.group = group                 # the param_groups dict whose lr / betas / eps this bucket follows
        self.dtype, self.device = params[0].dtype, params[0].device
        starts, off = [], 0
        for p in params:
            starts.append(off)
            off += p.numel()
        self.n = off
        self.starts = (ctypes.c_int64 * (len(params) + 1))(*(starts + [off]))
        self.exp_avg = torch.zeros(self.n, dtype=self.dtype, device=self.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step = torch.zeros(len(params), dtype=torch.int64, device=self.device)     # one count per tensor, as torch.optim.Adam
        self.ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
        # {lr, beta1, beta2, eps} in device memory: a captured launch follows `for g in opt.param_groups: g['lr'] = ...`
        self.hyper = torch.zeros(4, dtype=torch.float64, device=self.device)
        self._hyper_host = None

    def hyper_values(self):
        g = self.group
        return (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']))

    def sync_hyper(self):
        """Upload the group's current hyper-parameters when they changed since the last upload (a host-to-device copy:
        never during stream capture -- a captured step replays the launch, and this runs before each replay)."""
        now = self.hyper_values()
        if now != self._hyper_host:
            if not (0.0 <= now[0] and 0.0 <= now[1] < 1.0 and 0.0 <= now[2] < 1.0 and 0.0 <= now[3]):
                raise ValueError("zhusuan.optim.FlatAdam: invalid hyper-parameters lr=%r betas=(%r, %r) eps=%r" % now)
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("zhusuan.optim.FlatAdam: hyper-parameters changed inside a stream capture; change them "
                                   "between replays (GraphedStep uploads them before each replay)")
            self.hyper.copy_(torch.tensor(now, dtype=torch.float64))
            self._hyper_host = now

    def pointer_tables(self):
        """(param pointers, gradient pointers, tensors kept alive for the call).  A gradient the kernel cannot read in
        place (other dtype, not contiguous) is copied first; a missing one goes in as NULL: the kernel then leaves that
        tensor alone -- values, moments and step count -- like torch.optim.Adam does for ``p.grad is None``."""
        keep, gptr = [], []
        for p in self.params:
            if not p.is_contiguous():
                raise RuntimeError("zhusuan.optim.FlatAdam: parameters must be contiguous")
            g = p.grad
            if g is not None and (g.dtype != self.dtype or not g.is_contiguous() or g.device != self.device):
                g = g.to(device=self.device, dtype=self.dtype).contiguous()
            keep.append(g)
            gptr.append(None if g is None else g.data_ptr())
        pptr = (ctypes.c_void_p * len(self.params))(*[p.data_ptr() for p in self.params])
        return pptr, (ctypes.c_void_p * len(self.params))(*gptr), keep


class FlatAdam(object):
    """
    :param params: an iterable of parameters, or a list of such iterables (launch boundaries follow the groups, e.g.
        the stages of ``dataparallel.StagedBuckets``; within a group: one launch per 32 tensors of one dtype).
    :param lr, betas, eps: as ``torch.optim.Adam``.

    ``param_groups`` is a persistent list of dicts (one per group given to the constructor) with the keys ``params``,
    ``lr``, ``betas``, ``eps``: the idiom ``for g in opt.param_groups: g['lr'] = ...`` works (a ``torch.optim.lr_scheduler`` does
    NOT attach: FlatAdam is not a ``torch.optim.Optimizer`` subclass -- set ``g['lr']`` from the schedule yourself), also
    for a step captured in a hipGraph (the kernel reads the hyper-parameters from device memory; ``sync_hyperparameters()``
    -- called by ``step()`` and by ``GraphedStep`` before each replay -- uploads changes).  ``state_dict()`` /
    ``load_state_dict()`` hold the moments and the per-tensor step counts.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        groups = [params] if isinstance(params[0], torch.Tensor) else [list(g) for g in params]
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("Invalid beta parameters: {}".format(betas))
        self.defaults = {'lr': float(lr), 'betas': (float(betas[0]), float(betas[1])), 'eps': float(eps)}
        seen = set()
        for g in groups:
            for p in g:
                if id(p) in seen:
                    raise ValueError("some parameters appear in more than one group")
                seen.add(id(p))
        _hip.require_device(*[p for g in groups for p in g])
        self.buckets = []
        self.param_groups = []
        for g in groups:
            g = [p for p in g if p.requires_grad and p.numel() > 0]
            for p in g:
                if p.dtype not in (torch.float32, torch.float64):
                    raise TypeError("zhusuan.optim.FlatAdam: float32 or float64 parameters, got %s" % p.dtype)
            group = dict(self.defaults, params=g)
            self.param_groups.append(group)
            by_kind = {}
            for p in g:                                   # registration order is kept inside a (dtype, device) class
                by_kind.setdefault((p.dtype, p.device), []).append(p)
            for ps in by_kind.values():
                for i in range(0, len(ps), _hip.ADAM_MAX_TENSORS):
                    self.buckets.append(_Bucket(ps[i:i + _hip.ADAM_MAX_TENSORS], group))

    # ---- the torch.optim surface that training loops (and zhusuan.GraphedStep's restore) use
    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, value):
        for g in self.param_groups:
            g['lr'] = float(value)

    def state_tensors(self):
        """[(key, tensor)] of everything a step mutates besides the parameters (moments, step counts)."""
        out = []
        for i, b in enumerate(self.buckets):
            out += [((i, 'exp_avg'), b.exp_avg), ((i, 'exp_avg_sq'), b.exp_avg_sq), ((i, 'step'), b.step)]
        return out

    def state_dict(self):
        """Moments and per-tensor step counts per bucket plus the groups' hyper-parameters (clones: a checkpoint)."""
        return {'buckets': [{'exp_avg': b.exp_avg.detach().clone(), 'exp_avg_sq': b.exp_avg_sq.detach().clone(),
                             'step': b.step.detach().clone(), 'n_tensors': len(b.params), 'n': b.n} for b in self.buckets],
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def load_state_dict(self, state):
        if len(state['buckets']) != len(self.buckets) or len(state['param_groups']) != len(self.param_groups):
            raise ValueError("zhusuan.optim.FlatAdam.load_state_dict: the state belongs to another parameter layout")
        for b, sb in zip(self.buckets, state['buckets']):
            if sb['n'] != b.n or sb['n_tensors'] != len(b.params):
                raise ValueError("zhusuan.optim.FlatAdam.load_state_dict: the state belongs to another parameter layout")
        with torch.no_grad():
            for b, sb in zip(self.buckets, state['buckets']):
                b.exp_avg.copy_(sb['exp_avg'])
                b.exp_avg_sq.copy_(sb['exp_avg_sq'])
                b.step.copy_(sb['step'])
        for g, sg in zip(self.param_groups, state['param_groups']):
            g.update({k: (tuple(v) if k == 'betas' else v) for k, v in sg.items()})

    def zero_grad(self, set_to_none=True):
        for b in self.buckets:
            for p in b.params:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.detach_()
                    p.grad.zero_()

    def sync_hyperparameters(self):
        for b in self.buckets:
            b.sync_hyper()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        """One Adam update of every parameter that has a gradient, read as ``grad_scale * grad``."""
        lib = _hip.lib()
        for b in self.buckets:
            b.sync_hyper()
            lr, b1, b2, eps = b.hyper_values()
            pptr, gptr, keep = b.pointer_tables()
            sfx = "_f32" if b.dtype == torch.float32 else "_f64"
            lib.call("zs_adam_step" + sfx, pptr, gptr, b.starts, len(b.params), _hip.ptr(b.exp_avg), _hip.ptr(b.exp_avg_sq),
                     _hip.ptr(b.step), _hip.ptr(b.ticket), b.n, lr, b1, b2, eps, float(grad_scale), _hip.ptr(b.hyper),
                     _hip.stream_for(b.exp_avg))
            del keep

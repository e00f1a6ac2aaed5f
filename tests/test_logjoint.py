"""The one-launch pieces for the launch-bound configurations (include/zs_hip.h: LJ1 scalar log-joint objective, MS1 fused
sampling of several Normal nodes, PL1 particle-batched dense layer).

not gpu : the C oracle against float64 truths computed with torch on the same inputs (and, for PL1, against the
          reference caller's own op sequence: repeat + cat + matmul + div + relu);
gpu     : libzs_hip.so against the C oracle through raw ABI calls -- every family, full / scalar / periodic operands,
          aligned and unaligned, empty and ragged, up to 2^20 elements, float32 and float64.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import host_kernel_library
from zhusuan import _hip
from zhusuan.layers import _fits_lds

LJ = _hip


class Raw(object):
    def __init__(self, klib, device, dtype=torch.float32):
        self.k, self.dev, self.dtype = klib, torch.device(device), dtype
        self.sfx = "_f32" if dtype == torch.float32 else "_f64"

    def t(self, a):
        if a is None:
            return None
        return torch.as_tensor(np.ascontiguousarray(a), dtype=self.dtype).to(self.dev)

    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream) if self.dev.type == "cuda" else None

    def sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    # ---- LJ1.  terms: list of dicts(family, x, a, b, coef, n, want=(gx, ga, gb))
    def _table(self, terms, grads=None):
        tab = (_hip.LJTerm * len(terms))()
        keep = []
        for i, tm in enumerate(terms):
            e = tab[i]
            e.family, e.n, e.coef = tm["family"], tm["n"], tm["coef"]
            for name, pname in (("x", "px"), ("a", "pa"), ("b", "pb")):
                v = tm.get(name)
                if v is not None:
                    tv = self.t(v)
                    if tm.get("misalign"):
                        tv = torch.cat([tv.new_zeros(1), tv.reshape(-1)])[1:]
                    keep.append(tv)
                    setattr(e, name, tv.data_ptr())
                    setattr(e, pname, tv.numel())
                else:
                    setattr(e, pname, 1)
            if grads is not None:
                for j, (name, pname) in enumerate((("gx", "px"), ("ga", "pa"), ("gb", "pb"))):
                    if tm.get("want", (False,) * 3)[j]:
                        gt = torch.full((getattr(e, pname) + 1,), float("nan"), dtype=self.dtype, device=self.dev)[1:] \
                            if tm.get("misalign") else torch.full((getattr(e, pname),), float("nan"), dtype=self.dtype, device=self.dev)
                        grads[(i, j)] = gt
                        setattr(e, name, gt.data_ptr())
        return tab, keep

    def lj_fwd(self, terms):
        tab, keep = self._table(terms)
        out = torch.full((1,), float("nan"), dtype=self.dtype, device=self.dev)
        ws = torch.zeros(_hip.LJ_WORKSPACE, dtype=torch.float64, device=self.dev)
        ticket = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # twice on ONE workspace, the first time with other coefficients: the ticket must come back at zero, and the second
        # launch must not see the first one's partial sums (stale lines in an L1 / another XCD's L2)
        for scale in (3.0, 1.0):
            for i, tm in enumerate(terms):
                tab[i].coef = tm["coef"] * scale
            self.k.call("zs_logjoint_scalar" + self.sfx, ctypes.byref(tab), len(terms), _hip.ptr(out), _hip.ptr(ws), ws.numel(),
                        _hip.ptr(ticket), self.stream())
        self.sync()
        assert int(ticket.item()) == 0
        return float(out.item())

    def lj_bwd(self, terms, g):
        grads = {}
        tab, keep = self._table(terms, grads)
        gout = self.t(np.array([g]))
        gcoef = torch.full((len(terms),), float("nan"), dtype=self.dtype, device=self.dev)
        ws = torch.zeros(_hip.LJ_WORKSPACE, dtype=torch.float64, device=self.dev)
        ticket = torch.zeros(1, dtype=torch.int32, device=self.dev)
        for scale in (3.0, 1.0):      # (as in lj_fwd: same workspace, other numbers first)
            for i, tm in enumerate(terms):
                tab[i].coef = tm["coef"] * scale
            self.k.call("zs_logjoint_scalar_bwd" + self.sfx, ctypes.byref(tab), len(terms), _hip.ptr(gout), _hip.ptr(gcoef),
                        _hip.ptr(ws), ws.numel(), _hip.ptr(ticket), self.stream())
        self.sync()
        assert int(ticket.item()) == 0
        return dict((k, v.cpu().numpy()) for k, v in grads.items()), gcoef.cpu().numpy()

    # ---- MS1
    def ms(self, nodes, seed=0, rs=None, want_used=False):
        """nodes: list of dicts(mu, sigma, eps|None, K, D, offset, ls, kfast)."""
        tab = (_hip.MSTerm * len(nodes))()
        keep, outs = [], []
        for i, nd in enumerate(nodes):
            mu, sg, eps = self.t(nd["mu"]), self.t(nd["sigma"]), self.t(nd.get("eps"))
            K, D, M = nd["K"], nd["D"], nd["mu"].size
            R = M // D
            z = torch.full((K, M), float("nan"), dtype=self.dtype, device=self.dev)
            kfast = nd.get("kfast", False)
            lp = torch.full((R, K) if kfast else (K, R), float("nan"), dtype=self.dtype, device=self.dev)
            e = tab[i]
            e.mu, e.sigma, e.eps = mu.data_ptr(), sg.data_ptr(), (eps.data_ptr() if eps is not None else None)
            e.z, e.lp = z.data_ptr(), lp.data_ptr()
            e.K, e.M, e.D = K, M, D
            e.lp_stride_k, e.lp_stride_r = (1, K) if kfast else (R, 1)
            e.offset, e.sigma_is_logstd = nd.get("offset", 0), nd.get("ls", 0)
            keep += [mu, sg, eps]
            outs.append((z, lp, kfast))
        used = torch.zeros(2, dtype=torch.int64, device=self.dev) if want_used else None
        self.k.call("zs_normal_sample_logprob_multi" + self.sfx, ctypes.byref(tab), len(nodes), seed, _hip.ptr(rs), _hip.ptr(used),
                    self.stream())
        self.sync()
        res = [(z.cpu().numpy(), (lp.t() if kf else lp).cpu().numpy()) for z, lp, kf in outs]
        return (res, used.cpu().numpy()) if want_used else res

    def ms_bwd(self, nodes, seed=0, rs=None):
        tab = (_hip.MSTerm * len(nodes))()
        keep, outs = [], []
        for i, nd in enumerate(nodes):
            sg, eps = self.t(nd["sigma"]), self.t(nd.get("eps"))
            gz, glp = self.t(nd.get("gz")), self.t(nd.get("glp"))
            K, D, M = nd["K"], nd["D"], nd["sigma"].size
            R = M // D
            gmu = torch.full((M,), float("nan"), dtype=self.dtype, device=self.dev)
            gs = torch.full((M,), float("nan"), dtype=self.dtype, device=self.dev)
            e = tab[i]
            e.sigma, e.eps = sg.data_ptr(), (eps.data_ptr() if eps is not None else None)
            e.K, e.M, e.D = K, M, D
            e.offset, e.sigma_is_logstd = nd.get("offset", 0), nd.get("ls", 0)
            e.gz = gz.data_ptr() if gz is not None else None
            gz2 = self.t(nd.get("gz2"))
            e.gz2 = gz2.data_ptr() if gz2 is not None else None
            keep.append(gz2)
            e.glp = glp.data_ptr() if glp is not None else None
            e.glp_stride_k, e.glp_stride_r = R, 1
            e.gmu, e.gsigma = gmu.data_ptr(), gs.data_ptr()
            keep += [sg, eps, gz, glp]
            outs.append((gmu, gs))
        self.k.call("zs_normal_sample_logprob_multi_bwd" + self.sfx, ctypes.byref(tab), len(nodes), seed, _hip.ptr(rs), self.stream())
        self.sync()
        return [(a.cpu().numpy(), b.cpu().numpy()) for a, b in outs]

    # ---- single-node K1 (to compare MS1 with)
    def k1(self, mu, sigma, eps, K, D, seed, off, ls=0):
        M = mu.size
        R = M // D
        z = torch.full((K, M), float("nan"), dtype=self.dtype, device=self.dev)
        lp = torch.full((K, R), float("nan"), dtype=self.dtype, device=self.dev)
        mu, sigma, eps = self.t(mu), self.t(sigma), self.t(eps)          # (named: they must outlive the call)
        self.k.call("zs_normal_sample_logprob" + self.sfx, _hip.ptr(mu), _hip.ptr(sigma), _hip.ptr(eps), seed, off,
                    None, _hip.ptr(z), _hip.ptr(lp), K, M, D, R, 1, ls, None, self.stream())
        self.sync()
        return z.cpu().numpy(), lp.cpu().numpy()

    # ---- CS1
    def colsum(self, x):
        rows, cols = x.shape
        xt = self.t(x)
        ctiles = max((cols + 63) // 64, 1)
        ws = torch.full((128 * (cols + 256),), float("nan"), dtype=self.dtype, device=self.dev)
        tk = torch.zeros(ctiles, dtype=torch.int32, device=self.dev)
        out = torch.full((cols,), float("nan"), dtype=self.dtype, device=self.dev)
        for scale in (3.0, 1.0):       # twice on one workspace, other numbers first (tickets back at zero, no stale partials)
            xs = xt * scale
            self.k.call("zs_column_sum" + self.sfx, _hip.ptr(xs), _hip.ptr(out), rows, cols, _hip.ptr(ws), ws.numel(), _hip.ptr(tk),
                        tk.numel(), self.stream())
            self.sync()
        assert int(tk.abs().sum().item()) == 0
        return out.cpu().numpy()

    # ---- AB1
    def actbwd(self, g, y, act, in_place=False):
        rows, cols = g.shape
        gt, yt = self.t(g), self.t(y)
        ctiles = max((cols + 63) // 64, 1)
        ws = torch.full((128 * (cols + 256),), float("nan"), dtype=self.dtype, device=self.dev)
        tk = torch.zeros(ctiles, dtype=torch.int32, device=self.dev)
        gb = torch.full((cols,), float("nan"), dtype=self.dtype, device=self.dev)
        gpre = None
        for scale in (3.0, 1.0):       # twice on one workspace (tickets back at zero, no stale partials)
            gs = gt * scale
            gpre = gs if in_place else torch.full((rows, cols), float("nan"), dtype=self.dtype, device=self.dev)
            self.k.call("zs_dense_act_bwd" + self.sfx, _hip.ptr(gs), _hip.ptr(yt), act, _hip.ptr(gpre), _hip.ptr(gb), rows, cols,
                        _hip.ptr(ws), ws.numel(), _hip.ptr(tk), tk.numel(), self.stream())
            self.sync()
        assert int(tk.abs().sum().item()) == 0
        return gpre.cpu().numpy(), gb.cpu().numpy()

    # ---- PM1
    def _pm_table(self, ws, outs, gws):
        L = len(ws)
        table = (_hip.PMLayer * L)()
        for l in range(L):
            table[l].w, table[l].out = _hip.ptr(ws[l]), _hip.ptr(outs[l])
            table[l].gw = _hip.ptr(gws[l]) if gws is not None else None
            table[l].n_in, table[l].n_out = ws[l].shape[2] - 1, ws[l].shape[1]
        return table

    def pm(self, x, ws):
        K, B = ws[0].shape[0], x.shape[-2]
        shared = x.ndim == 2
        xt, wt = self.t(x), [self.t(w) for w in ws]
        outs = [torch.full((K, B, w.shape[1]), float("nan"), dtype=self.dtype, device=self.dev) for w in ws]
        table = self._pm_table(wt, outs, None)
        self.k.call("zs_particle_mlp" + self.sfx, _hip.ptr(xt), 0 if shared else B * x.shape[-1], ctypes.byref(table), len(ws), K, B,
                    self.stream())
        self.sync()
        return [o.cpu().numpy() for o in outs]

    def pm_bwd(self, x, ws, outs, gout, want_gx):
        K, B = ws[0].shape[0], x.shape[-2]
        shared = x.ndim == 2
        xt, wt, ot, gt = self.t(x), [self.t(w) for w in ws], [self.t(o) for o in outs], self.t(gout)
        slab = sum(w.shape[1] * w.shape[2] + 3 for w in ws)
        part = torch.full((K * ((B + 15) // 16) * slab + 4,), float("nan"), dtype=self.dtype, device=self.dev)
        tk = torch.zeros(max(K, 1), dtype=torch.int32, device=self.dev)
        gx = gws = None
        for scale in (3.0, 1.0):       # twice on one workspace (tickets back at zero, no stale partials)
            gs = gt * scale
            gws = [torch.full(tuple(w.shape), float("nan"), dtype=self.dtype, device=self.dev) for w in ws]
            gx = torch.full((K, B, x.shape[-1]), float("nan"), dtype=self.dtype, device=self.dev) if want_gx else None
            table = self._pm_table(wt, ot, gws)
            self.k.call("zs_particle_mlp_bwd" + self.sfx, _hip.ptr(xt), 0 if shared else B * x.shape[-1], ctypes.byref(table), len(ws),
                        _hip.ptr(gs), _hip.ptr(gx), K, B, _hip.ptr(part), part.numel(), _hip.ptr(tk), self.stream())
            self.sync()
        assert int(tk.abs().sum().item()) == 0
        return (gx.cpu().numpy() if want_gx else None), [g.cpu().numpy() for g in gws]

    # ---- PR1
    def rmse(self, pred, y):
        K, B = pred.shape
        pt, yt = self.t(pred), self.t(y)
        ws = torch.full((1024,), float("nan"), dtype=torch.float64, device=self.dev)
        tk = torch.zeros(1, dtype=torch.int32, device=self.dev)
        out = torch.full((1,), 123.0, dtype=self.dtype, device=self.dev)
        for scale in (3.0, 1.0):       # twice on one workspace (ticket back at zero, no stale partials)
            ps = pt * scale
            self.k.call("zs_particle_rmse" + self.sfx, _hip.ptr(ps), _hip.ptr(yt), _hip.ptr(out), K, B, _hip.ptr(ws), ws.numel(),
                        _hip.ptr(tk), self.stream())
            self.sync()
        assert int(tk.abs().sum().item()) == 0
        return float(out.cpu().numpy()[0])

    # ---- PL1
    def pl(self, h, w, relu):
        K, n_out, n_in1 = w.shape
        shared = h.ndim == 2
        B = h.shape[-2]
        out = torch.full((K, B, n_out), float("nan"), dtype=self.dtype, device=self.dev)
        h, w = self.t(h), self.t(w)
        self.k.call("zs_particle_linear" + self.sfx, _hip.ptr(h), 0 if shared else B * (n_in1 - 1), _hip.ptr(w),
                    _hip.ptr(out), K, B, n_in1 - 1, n_out, int(relu), self.stream())
        self.sync()
        return out.cpu().numpy()

    def pl_bwd(self, h, w, out, gout, relu, want_gh=True):
        K, n_out, n_in1 = w.shape
        shared = h.ndim == 2
        B = h.shape[-2]
        gh = torch.full((K, B, n_in1 - 1), float("nan"), dtype=self.dtype, device=self.dev)
        gw = torch.full((K, n_out, n_in1), float("nan"), dtype=self.dtype, device=self.dev)
        h, w, out, gout = self.t(h), self.t(w), self.t(out), self.t(gout)
        part = torch.full((K * ((B + 15) // 16) * n_out * n_in1 + 1,), float("nan"), dtype=self.dtype, device=self.dev)
        tickets = torch.zeros(max(K, 1), dtype=torch.int32, device=self.dev)
        # twice on ONE workspace, the first time with another upstream gradient: the tickets must come back at zero, and the
        # second launch must not see the first one's tile partials (stale lines in an L1 / another XCD's L2)
        other = gout * 3.0 + 1.0
        for go in (other, gout):
            self.k.call("zs_particle_linear_bwd" + self.sfx, _hip.ptr(h), 0 if shared else B * (n_in1 - 1), _hip.ptr(w),
                        _hip.ptr(out), _hip.ptr(go), _hip.ptr(gh) if want_gh else None, _hip.ptr(gw), K, B, n_in1 - 1,
                        n_out, int(relu), _hip.ptr(part), part.numel(), _hip.ptr(tickets), self.stream())
        self.sync()
        assert int(tickets.abs().sum().item()) == 0
        return (gh.cpu().numpy() if want_gh else None), gw.cpu().numpy()


@pytest.fixture(scope="module")
def orc():
    return Raw(host_kernel_library(), "cpu")


@pytest.fixture(scope="module")
def orc64():
    return Raw(host_kernel_library(), "cpu", torch.float64)


@pytest.fixture(scope="module")
def hip():
    return Raw(_hip.KernelLibrary(_hip.LIB_PATH), "cuda:0")


@pytest.fixture(scope="module")
def hip64():
    return Raw(_hip.KernelLibrary(_hip.LIB_PATH), "cuda:0", torch.float64)


# ------------------------------------------------------------------------------------------------ LJ1 inputs
def _operand(rng, kind, n, period, lo=None):
    """Values of one operand with `period` elements (a divisor of n)."""
    if kind == "value":
        return rng.standard_normal(period)
    if kind == "mean":
        return 0.3 * rng.standard_normal(period)
    if kind == "std":
        return rng.uniform(0.5, 1.5, size=period)
    if kind == "logstd":
        return rng.uniform(-0.7, 0.4, size=period)
    if kind == "probs":
        v = rng.uniform(0.02, 0.98, size=period)
        v[:3] = [0.0, 1.0, 1e-9][:min(3, period)]                       # the +1e-8 edges (bernoulli.py:94)
        return v
    if kind == "logits":
        return 3.0 * rng.standard_normal(period)
    if kind == "bits":
        return (rng.uniform(size=period) < 0.5).astype(np.float64)
    raise ValueError(kind)


def _term(rng, family, n, periods=(None, None, None), coef=1.0, want=(False, False, False), misalign=False, frac=False):
    px, pa, pb = [n if p is None else p for p in periods]
    tm = {"family": family, "n": n, "coef": coef, "want": want, "misalign": misalign}
    if family == LJ.LJ_ROWS:
        tm["x"] = 5.0 * rng.standard_normal(n)
    elif family in (LJ.LJ_NORMAL, LJ.LJ_NORMAL_LOGSTD):
        tm["x"], tm["a"] = _operand(rng, "value", n, px), _operand(rng, "mean", n, pa)
        tm["b"] = _operand(rng, "logstd" if family == LJ.LJ_NORMAL_LOGSTD else "std", n, pb)
    else:
        tm["x"] = rng.uniform(0.05, 0.95, size=px) if frac else _operand(rng, "bits", n, px)     # (frac: a differentiable observation)
        tm["a"] = _operand(rng, "logits" if family == LJ.LJ_BERNOULLI_LOGITS else "probs", n, pa)
        if frac and family == LJ.LJ_BERNOULLI_LOGITS:
            # (milder logits: with a fractional observation BOTH logarithms carry weight, and 1 - sigmoid(9) in fp32 -- the
            #  reference's own arithmetic, bernoulli.py:50,94 -- is good to 1e-3 only: the float64 truth is not the fp32 target there)
            tm["a"] = 1.2 * rng.standard_normal(pa)
    return tm


def _truth(terms, g=None):
    """float64 torch evaluation of the objective and (with g) of every requested gradient."""
    total = torch.zeros((), dtype=torch.float64)
    leaves = {}
    for i, tm in enumerate(terms):
        n = tm["n"]

        def full(name, j):
            v = tm.get(name)
            if v is None:
                return None
            t = torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=bool(tm["want"][j]))
            if tm["want"][j]:
                leaves[(i, j)] = t
            return t.repeat(n // t.numel()) if t.numel() != n else t
        x, a, b = full("x", 0), full("a", 1), full("b", 2)
        fam = tm["family"]
        if n == 0:
            continue
        if fam == LJ.LJ_ROWS:
            s = x.sum()
        elif fam in (LJ.LJ_NORMAL, LJ.LJ_NORMAL_LOGSTD):
            sd = torch.exp(b) if fam == LJ.LJ_NORMAL_LOGSTD else b
            s = torch.distributions.Normal(a, sd).log_prob(x).sum()
        else:
            p = torch.sigmoid(a) if fam == LJ.LJ_BERNOULLI_LOGITS else a
            s = (x * torch.log(p + 1e-8) + (1 - x) * torch.log(1 - p + 1e-8)).sum()
        total = total + tm["coef"] * s
    grads = {}
    if g is not None and leaves:
        keys = list(leaves.keys())
        gs = torch.autograd.grad(total * g, [leaves[k] for k in keys], allow_unused=True)
        grads = dict((k, (v.numpy() if v is not None else np.zeros(leaves[k].shape))) for k, v in zip(keys, gs))
    return float(total), grads


def _cases(rng, small=False):
    """Lists of terms covering families x operand classes (full / scalar / periodic) x alignment x sizes."""
    big = 4099 if small else (1 << 20) + 3
    N, NL, B, BL, R = LJ.LJ_NORMAL, LJ.LJ_NORMAL_LOGSTD, LJ.LJ_BERNOULLI, LJ.LJ_BERNOULLI_LOGITS, LJ.LJ_ROWS
    all3, a_only = (True, True, True), (False, True, False)
    cases = [
        # the VAE objective's shape: prior on z, Bernoulli likelihood, the sampler's rows
        [_term(rng, N, 16 * 40, coef=-1 / 16., want=(True, False, False)), _term(rng, B, 16 * 784, coef=-1 / 16., want=a_only),
         _term(rng, R, 16, coef=1 / 16.)],
        # the BNN objective's shape: two weight priors (parameters of period 700 / 51), the likelihood with a scalar log std
        [_term(rng, N, 10 * 700, (None, 700, 700), coef=-0.1, want=(True, False, False)),
         _term(rng, N, 10 * 51, (None, 51, 51), coef=-0.1, want=(True, False, False)),
         _term(rng, NL, 10 * 64, (64, None, 1), coef=-456. / 640, want=(False, True, True)),
         _term(rng, R, 10, coef=0.1), _term(rng, R, 10, coef=0.1)],
        # every operand periodic with a gradient (fold jobs), unaligned bases, a 1-element term, an empty term
        [_term(rng, N, 6 * 35, (35, 7 * 5, 5), coef=0.7, want=all3, misalign=True), _term(rng, NL, 12 * 8, (8, 96, 4), coef=-1.3, want=all3),
         _term(rng, BL, 9 * 20, (20, 60, None), coef=2.0, want=a_only), _term(rng, N, 1, coef=3.0, want=all3),
         _term(rng, B, 0, coef=1.0), _term(rng, R, 0, coef=1.0)],
        # scalars everywhere
        [_term(rng, N, 1000, (None, 1, 1), coef=1e-3, want=all3), _term(rng, NL, 777, (1, None, 1), coef=-2e-3, want=all3),
         _term(rng, B, 333, (None, 1, None), coef=0.5, want=a_only)],
        # ragged / large: more than one workgroup per term, a tail that is not a multiple of 4
        [_term(rng, N, big, coef=1.0 / big, want=all3), _term(rng, BL, big - 1, coef=-1.0 / big, want=a_only, misalign=True),
         _term(rng, R, big // 3, coef=1e-3)],
        # the OBSERVATION of a Bernoulli term receives a gradient too (bernoulli.py:94 is differentiable in `sample`): full size,
        # periodic (a fold job: the sum over the rows that read it), scalar; both parameterisations; unaligned
        [_term(rng, B, 16 * 48, coef=-1 / 16., want=(True, True, False), frac=True),
         _term(rng, BL, 9 * 20, (20, None, None), coef=2.0, want=(True, True, False), frac=True),
         _term(rng, B, 333, (1, None, None), coef=0.5, want=(True, False, False), frac=True),
         _term(rng, BL, 8 * 36, (36, 72, None), coef=-0.3, want=(True, True, False), frac=True, misalign=True)],
        # eight terms (the table's capacity)
        [_term(rng, f, 50 + 7 * i, coef=(-1.0) ** i * 0.25, want=(f in (N, NL), f != R, f in (N, NL)))
         for i, f in enumerate([N, NL, B, BL, R, N, B, R])],
    ]
    return cases


def _compare_bwd(got, truth, terms, rtol, atol_scale):
    for key, ref in truth.items():
        a = got[key]
        scale = max(np.abs(ref).max(), 1e-30)
        np.testing.assert_allclose(a, ref, rtol=rtol, atol=atol_scale * scale, err_msg="term %d operand %d" % key)


# ------------------------------------------------------------------------------------------------ CPU: oracle vs float64 truth
def test_c_oracle_logjoint_against_float64_truth(orc, orc64):
    rng = np.random.RandomState(11)
    for terms in _cases(rng, small=True):
        ref, gref = _truth(terms, g=0.75)
        mag = max(sum(abs(tm["coef"]) * max(tm["n"], 1) for tm in terms), abs(ref))
        assert abs(orc64.lj_fwd(terms) - ref) <= 1e-12 * mag
        assert abs(orc.lj_fwd(terms) - ref) <= 3e-7 * mag
        g64, c64 = orc64.lj_bwd(terms, 0.75)
        _compare_bwd(g64, gref, terms, 1e-10, 1e-12)
        np.testing.assert_allclose(c64, [0.75 * tm["coef"] for tm in terms], rtol=1e-14)
        g32, c32 = orc.lj_bwd(terms, 0.75)
        _compare_bwd(g32, gref, terms, 2e-4, 2e-6)
        np.testing.assert_allclose(c32, [0.75 * tm["coef"] for tm in terms], rtol=1e-6)


def test_c_oracle_logjoint_rejects_bad_tables(orc):
    rng = np.random.RandomState(1)
    with pytest.raises(RuntimeError, match="code -2"):
        orc.lj_fwd([_term(rng, LJ.LJ_ROWS, 4) for _ in range(9)])                # more than ZS_LJ_MAX_TERMS
    bad = _term(rng, LJ.LJ_NORMAL, 12, (None, 5, None))                           # period does not divide n
    with pytest.raises(RuntimeError, match="code -1"):
        orc.lj_fwd([bad])
    with pytest.raises(RuntimeError, match="code -1"):
        orc.lj_fwd([dict(_term(rng, LJ.LJ_NORMAL, 8), family=7)])
    # (d/d observation of a Bernoulli term was refused with code -2 until ABI 15; now it is an ordinary gradient)
    tm = _term(rng, LJ.LJ_BERNOULLI, 8, want=(True, False, False), frac=True)
    got, _ = orc.lj_bwd([tm], 1.0)
    _compare_bwd(got, _truth([tm], g=1.0)[1], [tm], 2e-4, 2e-6)


def test_c_oracle_multi_sampler_is_k1_per_node(orc, orc64):
    """MS1 == K1 node by node (the oracle delegates: checked against explicit calls), with given eps and with Philox draws
    whose call ids are base + offset."""
    rng = np.random.RandomState(5)
    for raw in (orc, orc64):
        nodes = []
        for (K, R, D, ls, given) in [(10, 1, 700, 1, False), (10, 1, 51, 1, False), (3, 8, 5, 0, True), (1, 4, 4, 0, False)]:
            M = R * D
            nd = {"mu": rng.standard_normal(M), "sigma": rng.uniform(-0.5, 0.3, M) if ls else rng.uniform(0.5, 1.5, M),
                  "K": K, "D": D, "ls": ls, "offset": len(nodes) + 3, "kfast": bool(len(nodes) % 2)}
            if given:
                nd["eps"] = rng.standard_normal(K * M)
            nodes.append(nd)
        st = torch.tensor([77, 1000], dtype=torch.int64)
        res, used = raw.ms(nodes, seed=123456, rs=st, want_used=True)
        assert used.tolist() == [77, 1000]
        for nd, (z, lp) in zip(nodes, res):
            z1, lp1 = raw.k1(nd["mu"], nd["sigma"], nd.get("eps"), nd["K"], nd["D"], 77, 1000 + nd["offset"], nd["ls"])
            assert np.array_equal(z, z1) and np.array_equal(lp, lp1)
        for nd in nodes:
            nd["gz"] = rng.standard_normal(nd["K"] * nd["mu"].size)
            nd["glp"] = rng.standard_normal(nd["K"] * (nd["mu"].size // nd["D"]))
        out = raw.ms_bwd(nodes, seed=1, rs=st)
        for nd, (gmu, gs) in zip(nodes, out):
            K, M, D = nd["K"], nd["mu"].size, nd["D"]
            eps = nd.get("eps")
            if eps is None:
                # regenerate the draw: z = mu + sigma * eps
                (z, _), = raw.ms([dict(nd, gz=None, glp=None)], rs=st)
                sg = np.exp(nd["sigma"]) if nd["ls"] else nd["sigma"]
                eps = ((z.reshape(K, M) - nd["mu"]) / sg).reshape(-1)
            gz, glp = nd["gz"].reshape(K, M), nd["glp"].reshape(K, M // D)
            sg = np.exp(nd["sigma"]) if nd["ls"] else nd["sigma"]
            b = (gz * np.asarray(eps).reshape(K, M)).sum(0)
            gl = np.repeat(glp.sum(0), D)
            tol = 1e-9 if raw.dtype == torch.float64 else 2e-4
            np.testing.assert_allclose(gmu, gz.sum(0), rtol=tol, atol=tol)
            np.testing.assert_allclose(gs, b * sg - gl if nd["ls"] else b - gl / sg, rtol=tol, atol=tol * 10)
        # two gradients w.r.t. one sample (gz + gz2, either may be missing) == their sum handed over as gz
        split = [dict(nd, gz=(None if i == 1 else 0.25 * nd["gz"]), gz2=(nd["gz"] if i == 1 else 0.75 * nd["gz"])) for i, nd in enumerate(nodes)]
        for (gmu, gs), (gmu2, gs2) in zip(out, raw.ms_bwd(split, seed=1, rs=st)):
            tol = 1e-12 if raw.dtype == torch.float64 else 3e-6
            np.testing.assert_allclose(gmu2, gmu, rtol=tol, atol=tol * 10)
            np.testing.assert_allclose(gs2, gs, rtol=tol, atol=tol * 10)


def _pl_reference(h, w, relu):
    """The reference caller's own op sequence (examples/bayesian_neural_nets/bnn_vi.py:36-48) in float64 torch."""
    K, n_out, n_in1 = w.shape
    ht = torch.tensor(h, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    hh = ht.repeat([K, 1, 1]) if ht.dim() == 2 else ht                       # x.repeat([n_particles, 1, 1])
    B = hh.shape[1]
    wr = torch.unsqueeze(wt, 1).repeat([1, B, 1, 1])
    hh = torch.cat((hh, torch.ones([*hh.shape[:-1], 1], dtype=torch.float64)), -1)
    hh = torch.unsqueeze(hh, -1)
    p = torch.sqrt(torch.as_tensor(hh.shape[2], dtype=torch.float64))
    out = torch.squeeze(torch.matmul(wr, hh) / p, -1)
    if relu:
        out = torch.nn.ReLU()(out)
    return ht, wt, out


PL_SHAPES = [  # (K, B, n_in, n_out, shared, relu)
    (10, 512, 13, 50, True, True), (10, 512, 50, 1, False, False),       # the layers of BASELINE config 5, per GPU
    (4, 16, 13, 50, True, True), (4, 16, 50, 1, False, False),           # the small golden's
    (1, 1, 1, 1, False, True), (3, 65, 7, 5, False, True), (2, 130, 100, 50, True, False), (5, 63, 31, 33, False, True),
    (2, 3, 255, 4, False, False), (2, 70, 3, 200, True, True), (7, 0, 4, 4, False, True),
    (10, 4096, 13, 50, True, True), (3, 2000, 50, 1, False, False)]      # many tiles per particle: the cross-workgroup hand-off


def test_c_oracle_particle_linear_is_the_reference_layer(orc, orc64):
    rng = np.random.RandomState(9)
    for (K, B, n_in, n_out, shared, relu) in PL_SHAPES:
        h = rng.standard_normal((B, n_in) if shared else (K, B, n_in))
        w = rng.standard_normal((K, n_out, n_in + 1))
        ht, wt, ref = _pl_reference(h, w, relu)
        gout = rng.standard_normal((K, B, n_out))
        for raw, tol in ((orc64, 1e-12), (orc, 2e-5)):
            out = raw.pl(h, w, relu)
            np.testing.assert_allclose(out, ref.detach().numpy(), rtol=tol, atol=tol)
            if B == 0:
                continue
            gh_ref, gw_ref = torch.autograd.grad(ref, [ht, wt], torch.tensor(gout), retain_graph=True)
            gh, gw = raw.pl_bwd(h, w, ref.detach().numpy(), gout, relu)
            np.testing.assert_allclose(gw, gw_ref.numpy(), rtol=tol * 10, atol=tol * 10 * max(np.abs(gw_ref.numpy()).max(), 1))
            gh_full = gh.sum(0) if shared else gh
            np.testing.assert_allclose(gh_full, gh_ref.numpy(), rtol=tol * 10, atol=tol * 10 * max(np.abs(gh_ref.numpy()).max(), 1))
    with pytest.raises(RuntimeError, match="code -2"):
        orc.pl(np.zeros((1, 2, 300)), np.zeros((1, 2, 301)), False)               # n_in > 255
    with pytest.raises(RuntimeError, match="code -1"):
        orc.k.call("zs_particle_linear_f32", None, 5, None, None, 1, 2, 3, 4, 0, None)      # h_stride_k neither 0 nor B * n_in


# ------------------------------------------------------------------------------------------------ GPU: HIP vs the C oracle
@pytest.mark.gpu
def test_hip_logjoint(hip, orc, hip64, orc64):
    rng = np.random.RandomState(21)
    for terms in _cases(rng):
        ref, gref = _truth(terms, g=-1.25)
        mag = max(sum(abs(tm["coef"]) * max(tm["n"], 1) for tm in terms), abs(ref))
        a, b = hip.lj_fwd(terms), orc.lj_fwd(terms)
        # (against float64: 10^6 fp32 terms carry their own rounding, a few 1e-7 of the summed magnitude)
        assert abs(a - b) <= 2e-7 * mag and abs(a - ref) <= 3e-6 * mag, (a, b, ref)
        assert hip.lj_fwd(terms) == a                                           # deterministic: fixed combination order
        ga, ca = hip.lj_bwd(terms, -1.25)
        gb, cb = orc.lj_bwd(terms, -1.25)
        assert ga.keys() == gb.keys() == gref.keys()
        _compare_bwd(ga, gref, terms, 3e-4, 3e-6)
        for key in ga:
            scale = max(np.abs(gb[key]).max(), 1e-30)
            np.testing.assert_allclose(ga[key], gb[key], rtol=3e-4, atol=3e-6 * scale, err_msg=str(key))
        np.testing.assert_allclose(ca, cb, rtol=1e-6)
        ga2, _ = hip.lj_bwd(terms, -1.25)
        assert all(np.array_equal(ga[k], ga2[k]) for k in ga)                   # periodic / scalar sums are deterministic too
    for terms in _cases(np.random.RandomState(22), small=True):
        mag = max(sum(abs(tm["coef"]) * max(tm["n"], 1) for tm in terms), abs(orc64.lj_fwd(terms)))
        assert abs(hip64.lj_fwd(terms) - orc64.lj_fwd(terms)) <= 1e-13 * mag
        ga, _ = hip64.lj_bwd(terms, 0.5)
        gb, _ = orc64.lj_bwd(terms, 0.5)
        for key in ga:
            np.testing.assert_allclose(ga[key], gb[key], rtol=1e-11, atol=1e-12 * max(np.abs(gb[key]).max(), 1e-30))
    with pytest.raises(RuntimeError, match="code -2"):
        hip.lj_fwd([_term(rng, LJ.LJ_ROWS, 4) for _ in range(9)])
    with pytest.raises(RuntimeError, match="code -1"):
        hip.lj_fwd([_term(rng, LJ.LJ_NORMAL, 12, (None, 5, None))])
    tm = _term(rng, LJ.LJ_BERNOULLI, 8, want=(True, False, False), frac=True)          # (refused with code -2 until ABI 15)
    _compare_bwd(hip.lj_bwd([tm], 1.0)[0], _truth([tm], g=1.0)[1], [tm], 3e-4, 3e-6)
    assert hip.lj_fwd([_term(rng, LJ.LJ_ROWS, 0)]) == 0.0                       # all terms empty


@pytest.mark.gpu
def test_hip_multi_sampler(hip, orc, hip64, orc64):
    rng = np.random.RandomState(31)
    for h, o, tol in ((hip, orc, 2e-5), (hip64, orc64, 1e-12)):
        for trial in range(3):
            nodes = []
            for (K, R, D, ls, given) in [(10, 1, 700, 1, False), (10, 1, 51, 1, False), (3, 8, 5, 0, True), (1, 4, 4, 0, False),
                                         (50, 16, 40, 0, False), (2, 3, 1, 0, False), (4, 0, 3, 0, False)][trial:trial + 5]:
                M = R * D
                nd = {"mu": rng.standard_normal(M), "sigma": rng.uniform(-0.5, 0.3, M) if ls else rng.uniform(0.5, 1.5, M),
                      "K": K, "D": D, "ls": ls, "offset": 5 * len(nodes) + trial, "kfast": bool(len(nodes) % 2)}
                if given:
                    nd["eps"] = rng.standard_normal(K * M)
                nodes.append(nd)
            st_h = torch.tensor([99, 12345], dtype=torch.int64, device=h.dev)
            st_o = torch.tensor([99, 12345], dtype=torch.int64)
            (ra, ua), (rb, ub) = h.ms(nodes, rs=st_h, want_used=True), o.ms(nodes, rs=st_o, want_used=True)
            assert ua.tolist() == ub.tolist() == [99, 12345]
            for nd, (za, la), (zb, lb) in zip(nodes, ra, rb):
                if nd.get("eps") is not None:
                    assert np.array_equal(za, zb)                               # z = mu + sigma * eps rounds like the reference
                else:
                    np.testing.assert_allclose(za, zb, rtol=0, atol=2e-5)       # device sin / cos / log vs libm
                np.testing.assert_allclose(la, lb, rtol=2e-5 if tol > 1e-9 else 1e-6, atol=2e-4 if tol > 1e-9 else 1e-4)
                # ... and MS1 == K1 on the device itself, bit for bit (same stream definition, same arithmetic)
                if nd["mu"].size:
                    z1, _ = h.k1(nd["mu"], nd["sigma"], nd.get("eps"), nd["K"], nd["D"], 99, 12345 + nd["offset"], nd["ls"])
                    assert np.array_equal(za, z1)
            for nd in nodes:
                nd["gz"] = rng.standard_normal(nd["K"] * nd["mu"].size)
                nd["glp"] = rng.standard_normal(nd["K"] * (nd["mu"].size // nd["D"]))
            for (ga, sa), (gb, sb), nd in zip(h.ms_bwd(nodes, rs=st_h), o.ms_bwd(nodes, rs=st_o), nodes):
                np.testing.assert_allclose(ga, gb, rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(sa, sb, rtol=2e-4, atol=2e-4 * max(np.abs(sb).max(), 1) if sb.size else 0)
            split = [dict(nd, gz=(None if i == 1 else 0.25 * nd["gz"]), gz2=(nd["gz"] if i == 1 else 0.75 * nd["gz"])) for i, nd in enumerate(nodes)]
            for (ga, sa), (gb, sb) in zip(h.ms_bwd(split, rs=st_h), o.ms_bwd(split, rs=st_o)):
                np.testing.assert_allclose(ga, gb, rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(sa, sb, rtol=2e-4, atol=2e-4 * max(np.abs(sb).max(), 1) if sb.size else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("K,B,n_in,n_out,shared,relu", PL_SHAPES)
def test_hip_particle_linear(hip, orc, hip64, orc64, K, B, n_in, n_out, shared, relu):
    rng = np.random.RandomState(K * 1000 + B + n_in)
    h = rng.standard_normal((B, n_in) if shared else (K, B, n_in))
    w = rng.standard_normal((K, n_out, n_in + 1))
    gout = rng.standard_normal((K, B, n_out))
    for a, b, tol, itemsize in ((hip, orc, 3e-5, 4), (hip64, orc64, 1e-12, 8)):
        if not _fits_lds(n_in, n_out, itemsize):
            # a particle's weights + a 64-row tile must fit 60 KB of LDS (the double twin: half as many elements);
            # zhusuan.particle_linear sends such layers through torch's batched GEMM
            with pytest.raises(RuntimeError, match="code -2"):
                a.pl(h, w, relu)
            continue
        oa, ob = a.pl(h, w, relu), b.pl(h, w, relu)
        np.testing.assert_allclose(oa, ob, rtol=tol, atol=tol)
        for want_gh in (True, False):
            gha, gwa = a.pl_bwd(h, w, ob, gout, relu, want_gh)
            ghb, gwb = b.pl_bwd(h, w, ob, gout, relu, want_gh)
            np.testing.assert_allclose(gwa, gwb, rtol=tol * 10, atol=tol * 10 * max(np.abs(gwb).max(), 1) if gwb.size else 0)
            if want_gh and B:
                np.testing.assert_allclose(gha, ghb, rtol=tol * 10, atol=tol * 10 * max(np.abs(ghb).max(), 1))
    with pytest.raises(RuntimeError, match="code -2"):
        hip.pl(np.zeros((1, 2, 300)), np.zeros((1, 2, 301)), False)


CS_SHAPES = [(12800, 500), (12800, 784), (12800, 40), (512, 500), (512, 784), (1, 7), (3, 1), (0, 5), (63, 64), (65, 257), (70000, 3),
             (129, 1000), (5000, 13)]


def test_c_oracle_column_sum(orc, orc64):
    rng = np.random.RandomState(4)
    for rows, cols in CS_SHAPES:
        if rows * cols > 2000000:
            continue
        x = rng.standard_normal((rows, cols))
        np.testing.assert_allclose(orc64.colsum(x), x.sum(0), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(orc.colsum(x), x.astype(np.float32).astype(np.float64).sum(0), rtol=1e-6, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", CS_SHAPES)
def test_hip_column_sum(hip, hip64, rows, cols):
    rng = np.random.RandomState(rows + cols)
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    ref = x.astype(np.float64).sum(0)
    a = hip.colsum(x)
    np.testing.assert_allclose(a, ref, rtol=2e-6, atol=3e-7 * max(np.sqrt(rows), 1) * 4)
    assert np.array_equal(a, hip.colsum(x))                                      # deterministic
    if rows * cols <= 1000000:
        np.testing.assert_allclose(hip64.colsum(x), ref, rtol=1e-13, atol=1e-12)


# ---------------------------------------------------------------- AB1: activation backward + bias gradient
def _act_ref(g, y, act):
    g, y = g.astype(np.float64), y.astype(np.float64)
    gpre = np.where(y > 0, g, 0.0) if act == 1 else g * (1.0 - y) * y
    return gpre, gpre.sum(0)


def _act_inputs(rows, cols, act, seed):
    rng = np.random.RandomState(seed)
    g = rng.standard_normal((rows, cols)).astype(np.float32)
    pre = rng.standard_normal((rows, cols)).astype(np.float32)
    y = np.maximum(pre, 0) if act == 1 else (1.0 / (1.0 + np.exp(-3.0 * pre))).astype(np.float32)   # (with exact zeros / ~0, ~1)
    return g, y.astype(np.float32)


def test_c_oracle_dense_act_bwd_matches_torch(orc, orc64):
    for act, tfn in ((1, torch.relu), (2, torch.sigmoid)):
        for rows, cols in ((5, 7), (64, 12), (0, 3), (129, 100)):
            rng = np.random.RandomState(rows + cols)
            pre = torch.tensor(rng.standard_normal((rows, cols)), dtype=torch.float64, requires_grad=True)
            g = rng.standard_normal((rows, cols))
            yv = tfn(pre)
            yv.backward(torch.tensor(g))
            gpre, gb = orc64.actbwd(g, yv.detach().numpy(), act)
            np.testing.assert_allclose(gpre, pre.grad.numpy(), rtol=1e-13, atol=1e-15)     # torch's own backward formulas
            np.testing.assert_allclose(gb, pre.grad.numpy().sum(0), rtol=1e-12, atol=1e-12)
            g32, y32 = _act_inputs(rows, cols, act, 3)
            a, b = orc.actbwd(g32, y32, act)
            ra, rb = _act_ref(g32, y32, act)
            np.testing.assert_allclose(a, ra, rtol=3e-7, atol=1e-30)
            np.testing.assert_allclose(b, rb, rtol=1e-6, atol=1e-5)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.actbwd(np.zeros((2, 2)), np.zeros((2, 2)), 0)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.actbwd(np.zeros((2, 2)), np.zeros((2, 2)), 3)


@pytest.mark.gpu
@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("rows,cols", CS_SHAPES)
def test_hip_dense_act_bwd(hip, hip64, orc, rows, cols, act):
    g, y = _act_inputs(rows, cols, act, rows + 7 * cols + act)
    ra, rb = _act_ref(g, y, act)
    a, b = hip.actbwd(g, y, act)
    if rows * cols <= 2000000:
        oa, _ = orc.actbwd(g, y, act)
        assert np.array_equal(a, oa) or np.allclose(a, oa, rtol=2e-7, atol=0)      # elementwise: the oracle's arithmetic
    np.testing.assert_allclose(a, ra, rtol=3e-7, atol=1e-30)
    np.testing.assert_allclose(b, rb, rtol=2e-6, atol=3e-7 * max(np.sqrt(rows), 1) * 4)
    a2, b2 = hip.actbwd(g, y, act, in_place=True)                                  # gpre = g: in place
    assert np.array_equal(a, a2) and np.array_equal(b, b2)
    if rows * cols <= 1000000:
        a64, b64 = hip64.actbwd(g, y, act)
        np.testing.assert_allclose(a64, ra, rtol=1e-14, atol=0)
        np.testing.assert_allclose(b64, rb, rtol=1e-13, atol=1e-12)


@pytest.mark.gpu
def test_hip_dense_act_bwd_rejects(hip):
    with pytest.raises(RuntimeError, match="code -1"):
        hip.actbwd(np.zeros((2, 2)), np.zeros((2, 2)), 0)
    with pytest.raises(RuntimeError, match="code -1"):
        hip.actbwd(np.zeros((2, 2)), np.zeros((2, 2)), 5)


# ---------------------------------------------------------------- PR1: RMSE of the particle-mean prediction
PR_SHAPES = [(10, 512), (10, 4096), (10, 4097), (1, 1), (4, 16), (512, 114), (3, 70001), (7, 1200000), (0, 5), (5, 0)]


def _rmse_ref(pred, y):
    if pred.shape[0] == 0 or pred.shape[1] == 0:
        return float("nan")
    return float(np.sqrt(np.mean((y.astype(np.float64) - pred.astype(np.float64).mean(0)) ** 2)))


def test_c_oracle_particle_rmse_matches_torch(orc, orc64):
    rng = np.random.RandomState(8)
    for K, B in PR_SHAPES:
        if K * B > 500000:
            continue
        pred, y = rng.standard_normal((K, B)).astype(np.float32), rng.standard_normal(B).astype(np.float32)
        t = torch.sqrt(torch.mean((torch.tensor(y) - torch.mean(torch.tensor(pred), 0)) ** 2))      # the caller's expression
        for o, tol in ((orc, 2e-6), (orc64, 1e-12)):
            got = o.rmse(pred, y)
            if K == 0 or B == 0:
                assert np.isnan(got) and np.isnan(float(t))
            else:
                np.testing.assert_allclose(got, _rmse_ref(pred, y), rtol=tol)
                np.testing.assert_allclose(got, float(t), rtol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("K,B", PR_SHAPES)
def test_hip_particle_rmse(hip, hip64, K, B):
    rng = np.random.RandomState(K + B)
    pred, y = rng.standard_normal((K, B)).astype(np.float32), rng.standard_normal(B).astype(np.float32)
    ref = _rmse_ref(pred, y)
    a = hip.rmse(pred, y)
    if K == 0 or B == 0:
        assert np.isnan(a) and np.isnan(hip64.rmse(pred, y))
        return
    np.testing.assert_allclose(a, ref, rtol=2e-6)
    assert a == hip.rmse(pred, y)                                       # deterministic
    np.testing.assert_allclose(hip64.rmse(pred, y), ref, rtol=1e-12)


# ---------------------------------------------------------------- PM1: the whole particle-batched network in one launch each way
PM_NETS = [((13, 50, 1), 10, 512, True), ((13, 50, 1), 10, 4096, True), ((13, 50, 1), 3, 7, False), ((5, 8, 8, 3), 4, 70, False),
           ((4, 4, 4, 4, 2), 2, 33, True), ((13, 50, 1), 512, 114, True), ((7, 9), 3, 20, True), ((3, 16, 2), 1, 1, False),
           ((13, 50, 1), 10, 0, True)]


def _pm_inputs(sizes, K, B, shared, seed):
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((B, sizes[0]) if shared else (K, B, sizes[0])).astype(np.float32)
    ws = [rng.standard_normal((K, sizes[l + 1], sizes[l] + 1)).astype(np.float32) for l in range(len(sizes) - 1)]
    gout = rng.standard_normal((K, B, sizes[-1])).astype(np.float32)
    return x, ws, gout


def _pm_truth(x, ws, gout, shared):
    """The caller's op sequence (bnn_vi.py:36-48) in float64 through torch autograd."""
    K = ws[0].shape[0]
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws]
    h = xt.unsqueeze(0).expand(K, *xt.shape) if shared else xt
    for l, w in enumerate(wt):
        h = torch.cat((h, torch.ones(*h.shape[:-1], 1, dtype=torch.float64)), -1)
        h = torch.matmul(w.unsqueeze(1), h.unsqueeze(-1)).squeeze(-1) / np.sqrt(h.shape[2])
        if l < len(wt) - 1:
            h = torch.relu(h)
    gs = torch.autograd.grad(h, [xt] + wt, torch.tensor(gout, dtype=torch.float64)) if h.numel() else [torch.zeros_like(xt)] + [torch.zeros_like(w) for w in wt]
    return h.detach().numpy(), gs[0].numpy(), [g.numpy() for g in gs[1:]]


def _pm_check(raw, raw_chain, sizes, K, B, shared, tol):
    x, ws, gout = _pm_inputs(sizes, K, B, shared, K + B + len(sizes))
    outs = raw.pm(x, ws)
    truth, tgx, tgw = _pm_truth(x, ws, gout, shared)
    if B:
        np.testing.assert_allclose(outs[-1], truth, rtol=tol, atol=tol * max(np.abs(truth).max(), 1))
    # the chain of PL1 calls: bit-identical
    h = x
    for l, w in enumerate(ws):
        h = raw_chain.pl(h, w, l < len(ws) - 1)
        assert np.array_equal(h, outs[l]), l
    for want_gx in (False, True):
        gx, gws = raw.pm_bwd(x, ws, outs, gout, want_gx)
        for l in range(len(ws)):
            np.testing.assert_allclose(gws[l], tgw[l], rtol=tol * 20, atol=tol * 20 * max(np.abs(tgw[l]).max(), 1), err_msg="gw%d" % l)
        if want_gx and B:
            ref = tgx if not shared else None
            if ref is not None:
                np.testing.assert_allclose(gx, ref, rtol=tol * 20, atol=tol * 20 * max(np.abs(ref).max(), 1))
            else:
                np.testing.assert_allclose(gx.astype(np.float64).sum(0), tgx, rtol=tol * 40, atol=tol * 40 * max(np.abs(tgx).max(), 1))


def test_c_oracle_particle_mlp(orc, orc64):
    for sizes, K, B, shared in PM_NETS:
        if K * B > 6000:
            continue
        _pm_check(orc, orc, sizes, K, B, shared, 2e-6)
        _pm_check(orc64, orc64, sizes, K, B, shared, 1e-13)
    x, ws, gout = _pm_inputs((13, 50, 1), 2, 4, True, 0)
    with pytest.raises(RuntimeError, match="code -2"):
        orc.pm(x, ws * 3)                                    # six layers
    with pytest.raises(RuntimeError, match="code -1"):
        orc.pm(x, [ws[0], ws[0]])                            # widths do not chain


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,K,B,shared", PM_NETS)
def test_hip_particle_mlp(hip, hip64, sizes, K, B, shared):
    from zhusuan.layers import _fits_lds_mlp
    _pm_check(hip, hip, sizes, K, B, shared, 3e-6)
    if _fits_lds_mlp(sizes, 8):
        if K * B <= 6000:
            _pm_check(hip64, hip64, sizes, K, B, shared, 1e-13)
    else:                                                    # twice the bytes per element: this network does not fit the LDS
        x, ws, _ = _pm_inputs(sizes, K, B, shared, 1)
        with pytest.raises(RuntimeError, match="code -2"):
            hip64.pm(x, ws)


@pytest.mark.gpu
def test_hip_particle_mlp_rejects(hip):
    x, ws, gout = _pm_inputs((13, 50, 1), 2, 4, True, 0)
    with pytest.raises(RuntimeError, match="code -2"):
        hip.pm(x, ws * 3)
    with pytest.raises(RuntimeError, match="code -1"):
        hip.pm(x, [ws[0], ws[0]])
    big = [np.zeros((1, 200, 201), np.float32), np.zeros((1, 200, 201), np.float32)]          # does not fit the LDS
    with pytest.raises(RuntimeError, match="code -2"):
        hip.pm(np.zeros((2, 200), np.float32), big)

"""autograd wrappers around the C-ABI kernels (include/zs_hip.h).

Each Function enqueues exactly one kernel in forward and one in backward on torch's current HIP
stream.  Row results use the K-fastest layout when ``kfast`` is set: the buffer is ``[R, K]`` and the
user-visible tensor is its transposed view ``[K, *rest]``, so shapes equal the reference's while the
K particles of a datapoint stay contiguous for the importance-weight reduction.
"""
import ctypes
import math
import weakref

import torch

from . import _hip

ZS_IW_SGVB = 0
ZS_IW_VIMCO = 1


def _prod(shape):
    return int(math.prod(shape)) if len(shape) else 1


def _sfx(*tensors):
    """'_f32' / '_f64': which precision of an entry point to call; operands must agree."""
    dt = None
    for t in tensors:
        if t is None:
            continue
        if dt is None:
            dt = t.dtype
        elif t.dtype != dt:
            raise TypeError("zhusuan: kernel operands must share one dtype, got %s and %s" % (dt, t.dtype))
    if dt == torch.float32:
        return "_f32"
    if dt == torch.float64:
        return "_f64"
    raise TypeError("zhusuan (MI355X build): kernels exist for float32 and float64, got %s" % dt)


def _alloc_rows(K, has_k_axis, rest_shape, kfast, like):
    """Row-result buffer and its user-visible view.  Returns (buffer, view, stride_k, stride_r)."""
    R = _prod(rest_shape)
    if has_k_axis and kfast and K > 1 and R > 0:
        buf = torch.empty((R, K), dtype=like.dtype, device=like.device)
        view = buf.t().view((K,) + tuple(rest_shape))
        return buf, view, 1, K
    buf = torch.empty((K, R), dtype=like.dtype, device=like.device)
    view = buf.view(((K,) if has_k_axis else ()) + tuple(rest_shape))
    return buf, view, R, 1


def _contiguous_strides(shape):
    out, acc = [], 1
    for n in reversed(tuple(shape)):
        out.append(acc)
        acc *= max(int(n), 1)
    return tuple(reversed(out))


def _own_tensor(base, offset, shape, strides):
    """A tensor over part of `base`'s allocation that autograd does NOT see as a view of `base` (same memory, own version
    counter, no `_base`): for kernels that write several results into one allocation."""
    return torch.empty(0, dtype=base.dtype, device=base.device).set_(base.untyped_storage(), base.storage_offset() + offset,
                                                                     tuple(shape), tuple(strides))


def _kr_view(t, K, R):
    """View `t` (K*R elements, logical order [K, R]) as a 2-D tensor without copying when possible."""
    try:
        v = t.view(K, R)
    except RuntimeError:
        v = t.contiguous().view(K, R)
    return v, v.stride(0), v.stride(1)


def _rng_snapshot(rng_state, wanted):
    """Two-word device buffer that a sampling kernel fills with the Philox ids it resolved from the live
    ``rng_state``; the backward call reads it instead of the live state, which ``DeviceRNG.begin_step()`` may have
    advanced in between (two objectives per step, a delayed or retained backward).  Costs no launch."""
    if rng_state is None or not wanted:
        return None
    return torch.empty(2, dtype=torch.int64, device=rng_state.device)


class NormalSampleLogProb(torch.autograd.Function):
    """K1: z = mu + sigma*eps and the row-summed log-density of z in one pass.
    Replaces Normal._sample + Normal._log_prob (zhusuan/distributions/normal.py:89-126).  With `is_logstd` the
    `sigma` operand is log(sigma) (Normal(logstd=...), normal.py:56) and its gradient is d/d logstd."""

    @staticmethod
    def forward(ctx, mu, sigma, eps, seed, call, rng_state, K, has_k_axis, n_fold, reparam, kfast, is_logstd=False):
        # backward copes with an undefined gz / glp itself: do not let autograd fill a [K, B, D] tensor with zeros
        # (a 2 MB memset per step when the sample reaches the generator detached, as in the VIMCO path)
        ctx.set_materialize_grads(False)
        _hip.require_device(mu, sigma, eps)
        sfx = _sfx(mu, sigma, eps)
        lib = _hip.lib()
        shape = tuple(mu.shape)
        M = mu.numel()
        rest = shape[:len(shape) - n_fold]
        D = _prod(shape[len(shape) - n_fold:])
        R = _prod(rest)
        z = torch.empty(((K,) if has_k_axis else ()) + shape, dtype=mu.dtype, device=mu.device)
        if M == 0:
            lp = torch.zeros(((K,) if has_k_axis else ()) + rest, dtype=mu.dtype, device=mu.device)
            ctx.meta = None
            return z, lp
        buf, lp, sk, sr = _alloc_rows(K, has_k_axis, rest, kfast, mu)
        used = _rng_snapshot(rng_state, reparam and eps is None)
        lib.call("zs_normal_sample_logprob" + sfx, _hip.ptr(mu), _hip.ptr(sigma), _hip.ptr(eps), seed, call,
                 _hip.ptr(rng_state), _hip.ptr(z), _hip.ptr(buf), K, M, D, sk, sr, 1 if is_logstd else 0, _hip.ptr(used),
                 _hip.stream_for(mu))
        if used is not None:        # backward regenerates eps from the ids the kernel resolved, not from the live state
            rng_state, call = used, 0
        ctx.meta = (seed, call, K, M, D, R, reparam, 1 if is_logstd else 0)
        ctx.rng_state = rng_state
        if reparam:
            ctx.save_for_backward(mu, sigma, eps)
        else:
            # torch.normal(mean, std) (normal.py:102) is a differentiable op whose derivative w.r.t. mean and
            # std is ZERO: z keeps a grad_fn, but nothing flows through it (backward ignores gz)
            ctx.save_for_backward(mu, sigma, z)
        return z, lp

    @staticmethod
    def backward(ctx, gz, glp):
        if ctx.meta is None:
            return (None,) * 12
        seed, call, K, M, D, R, reparam, ls = ctx.meta
        lib = _hip.lib()
        sfx = _sfx(ctx.saved_tensors[0])
        if gz is None and glp is None:
            return (None,) * 12
        if reparam:
            mu, sigma, eps = ctx.saved_tensors
            gmu = torch.empty_like(mu)
            gsigma = torch.empty_like(sigma)
            gsk = gsr = 0
            if gz is not None:
                gz = gz.contiguous()
            if glp is not None:
                glp, gsk, gsr = _kr_view(glp, K, R)
            lib.call("zs_normal_sample_logprob_bwd" + sfx, _hip.ptr(sigma), _hip.ptr(eps), seed, call,
                     _hip.ptr(ctx.rng_state), _hip.ptr(gz), _hip.ptr(glp), gsk, gsr, _hip.ptr(gmu), _hip.ptr(gsigma), K, M, D,
                     ls, _hip.stream_for(mu))
        else:
            mu, sigma, z = ctx.saved_tensors
            if glp is None:
                return (torch.zeros_like(mu), torch.zeros_like(sigma)) + (None,) * 10
            gmu = torch.empty_like(mu)
            gsigma = torch.empty_like(sigma)
            glp, gsk, gsr = _kr_view(glp, K, R)
            lib.call("zs_normal_logprob_bwd_ksum" + sfx, _hip.ptr(z), _hip.ptr(mu), _hip.ptr(sigma),
                     _hip.ptr(glp), gsk, gsr, None, _hip.ptr(gmu), _hip.ptr(gsigma), K, R, D, ls,
                     _hip.stream_for(mu))
        return (gmu, gsigma) + (None,) * 10


class NormalSampleLogProbPair(torch.autograd.Function):
    """K1 twice in one launch: the TWO draws an objective makes of a latent -- the node factory's (stochastic_tensor.py:115-127 of
    the reference, through bn.py:158) and the objective's re-read (elbo.py:122, importance_weighted_objective.py:85) -- with the
    Philox call ids `call` and `call + 1`.  Returns (z1, lp1, z2, lp2): each pair is bit for bit what NormalSampleLogProb returns
    for its call id.  In-kernel Philox only (no epsilon operand); backward treats each draw like NormalSampleLogProb's."""

    @staticmethod
    def forward(ctx, mu, sigma, seed, call, rng_state, K, has_k_axis, n_fold, reparam, is_logstd=False):
        ctx.set_materialize_grads(False)
        _hip.require_device(mu, sigma)
        sfx = _sfx(mu, sigma)
        shape = tuple(mu.shape)
        M = mu.numel()
        rest = shape[:len(shape) - n_fold]
        D = _prod(shape[len(shape) - n_fold:])
        R = _prod(rest)
        lead = (K,) if has_k_axis else ()
        # The kernel writes both draws into ONE allocation (z: [2, K, M]; rows: K-fastest [R, 2 K] or [2 K, R]).  The four outputs
        # are handed to autograd as tensors of their OWN over disjoint parts of those allocations (Tensor.set_), not as views of
        # two bases: several views of one base returned from one Function cannot be modified in place ("Output N of ... is a
        # view and is being modified inplace ... returns multiple views"), and the reference allows z.mul_() /
        # z.clamp_() / z += ... on a latent inside a variational net.
        zz = torch.empty((2,) + lead + shape, dtype=mu.dtype, device=mu.device)
        zs_ = [_own_tensor(zz, j * K * M, lead + shape, _contiguous_strides(lead + shape)) for j in range(2)]
        rstr = _contiguous_strides(rest)
        if has_k_axis and K > 1:            # K-fastest rows [R, 2 K]: each draw's [R, K] block keeps unit stride along K
            buf = torch.empty((R, 2 * K), dtype=mu.dtype, device=mu.device)
            lps = [_own_tensor(buf, j * K, lead + rest, (1,) + tuple(2 * K * t for t in rstr)) for j in range(2)]
            sk, sr = 1, 2 * K
        else:
            buf = torch.empty((2 * K, R), dtype=mu.dtype, device=mu.device)
            lps = [_own_tensor(buf, j * K * R, lead + rest, _contiguous_strides(lead + rest)) for j in range(2)]
            sk, sr = R, 1
        used = _rng_snapshot(rng_state, reparam)
        _hip.lib().call("zs_normal_sample_logprob_pair" + sfx, _hip.ptr(mu), _hip.ptr(sigma), seed, call, _hip.ptr(rng_state),
                        _hip.ptr(zz), _hip.ptr(buf), K, M, D, sk, sr, 1 if is_logstd else 0, _hip.ptr(used), _hip.stream_for(mu))
        if used is not None:
            rng_state, call = used, 0
        ctx.meta = (seed, call, K, M, D, R, reparam, 1 if is_logstd else 0)
        ctx.rng_state = rng_state
        if reparam:                 # backward regenerates epsilon: the draws themselves are not needed (and not kept alive)
            ctx.save_for_backward(mu, sigma)
        else:
            # The K-summed backward of a draw that is not reparameterised needs the draw's VALUE.  Saving both outputs would make
            # autograd check both whenever backward runs -- also the one nothing depends on (the factory's draw, which a net may
            # well have modified in place).  So the allocation is saved, and each output's version is checked only when ITS
            # gradient is formed (weak references: no cycle through the graph).
            ctx.save_for_backward(mu, sigma, zz)
            ctx.draw_refs = [(weakref.ref(t), t._version) for t in zs_]
        return zs_[0], lps[0], zs_[1], lps[1]

    @staticmethod
    def backward(ctx, gz1, glp1, gz2, glp2):
        seed, call, K, M, D, R, reparam, ls = ctx.meta
        mu, sigma = ctx.saved_tensors[:2]
        zz = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        lib, sfx = _hip.lib(), _sfx(mu)
        gmu = gsigma = None
        for j, (gz, glp) in enumerate(((gz1, glp1), (gz2, glp2))):
            if gz is None and glp is None:
                continue
            if not reparam and glp is None:           # torch.normal(mean, std): zero derivative through the sample itself
                continue
            a, b = torch.empty_like(mu), torch.empty_like(sigma)
            gsk = gsr = 0
            if glp is not None:
                glp, gsk, gsr = _kr_view(glp, K, R)
            if reparam:
                if gz is not None:
                    gz = gz.contiguous()
                lib.call("zs_normal_sample_logprob_bwd" + sfx, _hip.ptr(sigma), None, seed, call + j, _hip.ptr(ctx.rng_state),
                         _hip.ptr(gz), _hip.ptr(glp), gsk, gsr, _hip.ptr(a), _hip.ptr(b), K, M, D, ls, _hip.stream_for(mu))
            else:
                ref, version = ctx.draw_refs[j]
                t = ref()
                if t is not None and t._version != version:
                    raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace "
                                       "operation: draw %d of a paired Normal sample that is not reparameterised (its log-density's "
                                       "gradient needs the value as drawn); is at version %d, expected version %d"
                                       % (j + 1, t._version, version))
                lib.call("zs_normal_logprob_bwd_ksum" + sfx, _hip.ptr(zz[j]), _hip.ptr(mu), _hip.ptr(sigma), _hip.ptr(glp), gsk, gsr,
                         None, _hip.ptr(a), _hip.ptr(b), K, R, D, ls, _hip.stream_for(mu))
            gmu, gsigma = (a, b) if gmu is None else (gmu + a, gsigma + b)
        if gmu is None and not reparam and (gz1 is not None or gz2 is not None):
            gmu, gsigma = torch.zeros_like(mu), torch.zeros_like(sigma)
        return (gmu, gsigma) + (None,) * 8


class NormalLogProb(torch.autograd.Function):
    """K2: row-summed Normal log-density of a given value with periodically broadcast operands.
    Operands arrive contiguous; `periods` = (Px, Pm, Ps) in elements of the full [*full_shape] problem."""

    @staticmethod
    def forward(ctx, x, mu, sigma, full_shape, n_fold, periods, kfast, is_logstd=False):
        _hip.require_device(x, mu, sigma)
        sfx = _sfx(x, mu, sigma)
        lib = _hip.lib()
        ls = 1 if is_logstd else 0
        full_shape = tuple(full_shape)
        out_shape = full_shape[:len(full_shape) - n_fold]
        D = _prod(full_shape[len(full_shape) - n_fold:])
        has_k = len(out_shape) >= 2
        K = out_shape[0] if has_k else 1
        rest = out_shape[1:] if has_k else out_shape
        R = _prod(rest)
        buf, lp, sk, sr = _alloc_rows(K, has_k, rest, kfast and n_fold > 0, x)
        Px, Pm, Ps = periods
        if K * R * D > 0:
            lib.call("zs_normal_logprob" + sfx, _hip.ptr(x), Px, _hip.ptr(mu), Pm, _hip.ptr(sigma), Ps,
                     _hip.ptr(buf), K, R, D, sk, sr, ls, _hip.stream_for(x))
        ctx.meta = (K, R, D, periods, ls)
        ctx.save_for_backward(x, mu, sigma)
        return lp

    @staticmethod
    def backward(ctx, glp):
        K, R, D, periods, ls = ctx.meta
        x, mu, sigma = ctx.saved_tensors
        need = ctx.needs_input_grad[:3]
        if K * R * D == 0 or not any(need):
            return (None,) * 8
        return _normal_logprob_grads(x, mu, sigma, glp, K, R, D, periods, ls, need) + (None,) * 5


def _normal_logprob_grads(x, mu, sigma, glp, K, R, D, periods, ls, need):
    """(gx, gmu, gsigma) of the row-summed Normal log-density for row gradients `glp` (K2's backward: the K-summed kernel
    when the parameters are [R, D] repeated over the particles, else element-wise partials folded over the periods)."""
    Px, Pm, Ps = periods
    need_x, need_mu, need_sigma = need
    sfx = _sfx(x)
    N = K * R * D
    lib = _hip.lib()
    glp, gsk, gsr = _kr_view(glp, K, R)
    st = _hip.stream_for(x)
    gx = gmu = gsigma = None
    if K > 1 and Px == N and Pm == Ps == R * D:
        # parameters [R, D] repeated over the K particles: reduce over K inside the kernel
        gx = torch.empty_like(x) if need_x else None
        gmu = torch.empty_like(mu)
        gsigma = torch.empty_like(sigma)
        lib.call("zs_normal_logprob_bwd_ksum" + sfx, _hip.ptr(x), _hip.ptr(mu), _hip.ptr(sigma), _hip.ptr(glp),
                 gsk, gsr, _hip.ptr(gx), _hip.ptr(gmu), _hip.ptr(gsigma), K, R, D, ls, st)
    else:
        def full():
            return torch.empty(N, dtype=x.dtype, device=x.device)
        fx = full() if need_x else None
        fm = full() if need_mu else None
        fs = full() if need_sigma else None
        lib.call("zs_normal_logprob_bwd" + sfx, _hip.ptr(x), Px, _hip.ptr(mu), Pm, _hip.ptr(sigma), Ps,
                 _hip.ptr(glp), gsk, gsr, _hip.ptr(fx), _hip.ptr(fm), _hip.ptr(fs), K, R, D, ls, st)

        def fold(f, P, like):
            if f is None:
                return None
            if P != N:
                f = f.view(N // P, P).sum(0)
            return f.view(like.shape)
        gx, gmu, gsigma = fold(fx, Px, x), fold(fm, Pm, mu), fold(fs, Ps, sigma)
    return (gx if need_x else None, gmu if need_mu else None, gsigma if need_sigma else None)


class BernoulliLogProb(torch.autograd.Function):
    """K3: row-summed Bernoulli log-mass (zhusuan/distributions/bernoulli.py:84-95).  `p` has the
    full problem shape; `x` is periodic (x [B, X] against p [K, B, X]).  With from_logits the streamed
    operand holds logits and p = sigmoid(logit) is formed in registers (bernoulli.py:50)."""

    @staticmethod
    def forward(ctx, p, x, n_fold, Px, kfast, from_logits):
        _hip.require_device(p, x)
        sfx = _sfx(p, x)
        lib = _hip.lib()
        full_shape = tuple(p.shape)
        out_shape = full_shape[:len(full_shape) - n_fold]
        D = _prod(full_shape[len(full_shape) - n_fold:])
        has_k = len(out_shape) >= 2
        K = out_shape[0] if has_k else 1
        rest = out_shape[1:] if has_k else out_shape
        R = _prod(rest)
        buf, lp, sk, sr = _alloc_rows(K, has_k, rest, kfast and n_fold > 0, p)
        if K * R * D > 0:
            if from_logits:
                lib.call("zs_bernoulli_logits_logprob" + sfx, _hip.ptr(p), _hip.ptr(x), Px, _hip.ptr(buf), None,
                         K, R, D, sk, sr, _hip.stream_for(p))
            else:
                lib.call("zs_bernoulli_logprob" + sfx, _hip.ptr(p), _hip.ptr(x), Px, _hip.ptr(buf), K, R, D, sk, sr,
                         _hip.stream_for(p))
        ctx.meta = (K, R, D, Px, from_logits)
        ctx.edge_of = _edge_positions((p, x, n_fold, Px, kfast, from_logits))
        ctx.save_for_backward(p, x)
        return lp

    @staticmethod
    def backward(ctx, glp):
        K, R, D, Px, from_logits = ctx.meta
        p, x = ctx.saved_tensors
        need = _pass_needs(ctx, ctx.edge_of)
        if not (need[0] or need[1]) or K * R * D == 0:
            return (None,) * 6
        glp, gsk, gsr = _kr_view(glp, K, R)
        sfx = _sfx(p)
        gp = gx = None
        if need[0]:
            gp = torch.empty_like(p)
            name = ("zs_bernoulli_logits_logprob_bwd" if from_logits else "zs_bernoulli_logprob_bwd") + sfx
            _hip.lib().call(name, _hip.ptr(p), _hip.ptr(x), Px, _hip.ptr(glp), gsk, gsr, _hip.ptr(gp), K, R, D,
                            _hip.stream_for(p))
        if need[1]:
            # the observation is differentiable too (bernoulli.py:94; `given` keeps its graph through base.py:161-178): its
            # gradient sums over the elements that read it (x [B, X] against p [K, B, X])
            gx = torch.empty_like(x)
            _hip.lib().call("zs_bernoulli_logprob_bwd_x" + sfx, _hip.ptr(p), 1 if from_logits else 0, Px, _hip.ptr(glp), gsk, gsr,
                            None, 0, _hip.ptr(gx), K, R, D, _hip.stream_for(p))
        return gp, gx, None, None, None, None


class IWReduce(torch.autograd.Function):
    """K4: per-datapoint importance-weighted reduction over K-fastest rows [B, K].
    Returns (cost_b, bound_b); bound_b = log_mean_exp(log_w) is a detached diagnostic."""

    @staticmethod
    def forward(ctx, logp, logq, estimator):
        ctx.set_materialize_grads(False)     # the diagnostic `bound` never carries a gradient
        _hip.require_device(logp, logq)
        sfx = _sfx(logp, logq)
        B, K = logp.shape
        if logp.stride(1) != 1 and K > 1:
            logp = logp.contiguous()
        if logq.stride(1) != 1 and K > 1:
            logq = logq.contiguous()
        ld_p = logp.stride(0) if B > 1 else K
        ld_q = logq.stride(0) if B > 1 else K
        if ld_p < K:
            logp, ld_p = logp.contiguous(), K
        if ld_q < K:
            logq, ld_q = logq.contiguous(), K
        cost = torch.empty(B, dtype=logp.dtype, device=logp.device)
        bound = torch.empty_like(cost)
        coef_p = torch.empty((B, K), dtype=logp.dtype, device=logp.device)
        coef_q = torch.empty_like(coef_p)
        _hip.lib().call("zs_iw_reduce" + sfx, _hip.ptr(logp), ld_p, _hip.ptr(logq), ld_q, B, K, estimator,
                        _hip.ptr(cost), _hip.ptr(bound), _hip.ptr(coef_p), _hip.ptr(coef_q), _hip.stream_for(logp))
        ctx.save_for_backward(coef_p, coef_q)
        ctx.mark_non_differentiable(bound)
        return cost, bound

    @staticmethod
    def backward(ctx, g_cost, g_bound):
        coef_p, coef_q = ctx.saved_tensors
        if g_cost is None:
            return None, None, None
        g = g_cost.unsqueeze(1)
        gp = g * coef_p if ctx.needs_input_grad[0] else None
        gq = g * coef_q if ctx.needs_input_grad[1] else None
        return gp, gq, None


_SCRATCH = {}     # (device, stream | 'capture', kind) -> tuple of tensors
_SCRATCH_RETIRED = []   # capture sets that a larger one has replaced: kept alive, see _scratch


def _scratch(device, kind, fits, make):
    """Scratch of the kernels that combine per-workgroup partial results in-kernel (workspaces, zero-initialised ticket words
    that the kernels hand back at zero): one set per (device, STREAM, kind).  Launches on one stream are ordered, so they
    share it; objectives evaluated concurrently on different streams of a device each get their own.

    A set that turns out too small is replaced by a bigger one; a replaced CAPTURE set is kept alive forever
    (``_SCRATCH_RETIRED``): graphs captured before the growth keep replaying on the old one.

    While a hipGraph is being captured nothing may be cached that lives in the graph's private memory pool (it would dangle
    once the graph is destroyed), and allocating + zero-filling a fresh set per call would put a fill launch in front of
    every such kernel of every replay (seven per IWAE step for the bias gradients alone).  So every EAGER request also makes
    sure a per-device 'capture' set exists -- ordinary memory, allocated outside any capture -- and a capture uses that one
    (warm-up steps always run eagerly before a capture: zhusuan.GraphedStep / GraphedStages, bench.py).  All graphs of a device
    share it: their replays must not run concurrently on two streams (ordinary training never does).  A capture with no
    eager call before it falls back to a graph-private set per call."""
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if capturing:
        sc = _SCRATCH.get((str(device), "capture", kind))
        return sc if (sc is not None and fits(sc)) else make()
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key, ckey = (str(device), stream, kind), (str(device), "capture", kind)
    sc = _SCRATCH.get(key)
    if sc is None or not fits(sc):
        sc = _SCRATCH[key] = make()
    if device.type == "cuda":
        c = _SCRATCH.get(ckey)
        if c is None or not fits(c):
            if c is not None:
                # a hipGraph captured earlier has this set's workspace and ticket pointers baked in: it must stay allocated
                # (and its tickets at zero) for as long as that graph may be replayed -- never hand it back to the allocator
                _SCRATCH_RETIRED.append(c)
            _SCRATCH[ckey] = make()
    return sc


def _iw_workspace(device, dtype):
    """(partials [4096], ticket [1] int32) of K4b's deterministic batch mean (see _scratch)."""
    return _scratch(device, ("iw", dtype), lambda sc: True,
                    lambda: (torch.empty(4096, dtype=dtype, device=device), torch.zeros(1, dtype=torch.int32, device=device)))


def _rows_for_iw(t, K):
    """[B, K] operand of the IW kernels with unit stride along K; returns (tensor, leading dimension)."""
    B = t.shape[0]
    if K > 1 and t.stride(1) != 1:
        t = t.contiguous()
    ld = t.stride(0) if B > 1 else K
    if ld < K:
        t, ld = t.contiguous(), K
    return t, ld


class IWObjective(torch.autograd.Function):
    """K4b: the whole importance-weighted objective in one launch -- log w = (logp_a + logp_b) - logq, the per-datapoint
    IWAE / VIMCO costs, their batch mean and both gradient-coefficient matrices (zhusuan/variational/
    importance_weighted_objective.py:97-98,102-191).  Operands are K-fastest [B, K] matrices.
    Returns (cost, bound_b): cost is the 0-d batch mean when `want_mean`, else the [B] costs; bound_b =
    log_mean_exp(log w) per datapoint is a detached diagnostic.  Backward is ONE multiply of the [2, B, K] coefficient
    buffer (already scaled by 1/B) with the incoming gradient."""

    @staticmethod
    def forward(ctx, logp_a, logp_b, logq, estimator, want_mean):
        ctx.set_materialize_grads(False)
        _hip.require_device(logp_a, logp_b, logq)
        sfx = _sfx(logp_a, logp_b, logq)
        B, K = logq.shape
        pa, ld_a = _rows_for_iw(logp_a, K)
        pb, ld_b = (None, K) if logp_b is None else _rows_for_iw(logp_b, K)
        q, ld_q = _rows_for_iw(logq, K)
        dt, dev = logq.dtype, logq.device
        bound = torch.empty(B, dtype=dt, device=dev)
        coef = torch.empty((2, B, K), dtype=dt, device=dev)
        if want_mean:
            cost = torch.empty((), dtype=dt, device=dev)
            ws, ticket = _iw_workspace(dev, dt)
            _hip.lib().call("zs_iw_objective" + sfx, _hip.ptr(pa), ld_a, _hip.ptr(pb), ld_b, _hip.ptr(q), ld_q, B, K, estimator, 1,
                            None, _hip.ptr(bound), _hip.ptr(coef), _hip.ptr(cost), _hip.ptr(ws), ws.numel(), _hip.ptr(ticket),
                            _hip.stream_for(q))
        else:
            cost = torch.empty(B, dtype=dt, device=dev)
            _hip.lib().call("zs_iw_objective" + sfx, _hip.ptr(pa), ld_a, _hip.ptr(pb), ld_b, _hip.ptr(q), ld_q, B, K, estimator, 0,
                            _hip.ptr(cost), _hip.ptr(bound), _hip.ptr(coef), None, None, 0, None, _hip.stream_for(q))
        ctx.save_for_backward(coef)
        ctx.has_b = logp_b is not None
        ctx.want_mean = bool(want_mean)
        ctx.mark_non_differentiable(bound)
        return cost, bound

    @staticmethod
    def backward(ctx, g_cost, g_bound):
        if g_cost is None:
            return None, None, None, None, None
        (coef,) = ctx.saved_tensors
        gc = coef * (g_cost if ctx.want_mean else g_cost.reshape(1, -1, 1))
        gp = gc[0]
        return (gp if ctx.needs_input_grad[0] else None, gp if (ctx.has_b and ctx.needs_input_grad[1]) else None,
                gc[1] if ctx.needs_input_grad[2] else None, None, None)


# ------------------------------------------------------------------------------------------------
# A backward pass restricted to some parameters (``torch.autograd.backward(loss, inputs=params)``: the stages of
# dataparallel.StagedBuckets; ``torch.autograd.grad(loss, some)``) still runs every node that lies on a path to them, and
# ``ctx.needs_input_grad`` only says which inputs CAN receive a gradient, not which the pass is after.  A backward that serves
# several sides with separate launches (IW1: the decoder through p, the variational parameters through log q; a dense layer:
# its input, its weight, its bias) asks the ENGINE which of its input edges the running pass will execute
# (``torch._C._will_engine_execute_node`` on ``ctx.next_functions``) and launches only those.  The answer belongs to the pass
# that is running this node, on whatever thread: no module state, nothing to serialise (round 5 kept the pass's targets in a
# module global behind a lock, and an unrestricted backward of another thread could read them -- ADVICE r05).
# ------------------------------------------------------------------------------------------------
_WILL_EXECUTE = getattr(torch._C, "_will_engine_execute_node", None)


def _edge_positions(args):
    """Forward-argument index -> index into ``ctx.next_functions`` (which holds one edge per TENSOR argument, in order)."""
    pos, n = {}, 0
    for i, a in enumerate(args):
        if isinstance(a, torch.Tensor):
            pos[i] = n
            n += 1
    return pos


def _pass_needs(ctx, edge_of):
    """``ctx.needs_input_grad`` narrowed to what the running backward pass will use: an input whose producer (or, for a leaf,
    whose gradient accumulator) the engine is not going to execute in this pass needs no gradient from this node."""
    need = list(ctx.needs_input_grad)
    if _WILL_EXECUTE is None:
        return need
    edges = ctx.next_functions
    for i, e in edge_of.items():
        if need[i] and e < len(edges) and edges[e][0] is not None:
            try:
                need[i] = bool(_WILL_EXECUTE(edges[e][0]))
            except RuntimeError:          # (not inside a backward pass: e.g. backward() called by hand)
                pass
    return need


class grad_targets(object):
    """Kept for callers of round 5's interface: a no-op (the engine is asked instead, see above)."""

    def __init__(self, params):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


# ------------------------------------------------------------------------------------------------
# Gradient destinations.  A data-parallel step all-reduces ONE flat buffer per stage; autograd allocates every gradient where
# it likes, so the buffer used to be filled by a concatenation pass per step (2 x 2.7 MB read + written at the IWAE shape).
# A bucket registers, per parameter, the slice of its flat buffer that shadows it; a backward of this package that produces a
# parameter's gradient (the dense layers' weight GEMM and bias reduction) then writes it THERE and returns a fresh alias,
# which autograd's accumulator adopts as ``p.grad`` without a copy when ``p.grad`` is None (torch's "steal" rule: sole owner,
# same layout).  One producer per parameter per backward pass may claim the slice (a weight used twice in one graph: the
# second use allocates as before and autograd adds); a parameter that already holds a gradient is never written over.
# Everything else -- torch.nn modules, parameters used elsewhere -- still ends up in the bucket through the bucket's own copy.
# ------------------------------------------------------------------------------------------------
_GRAD_DEST = {}        # data_ptr of a parameter -> [slice (a view of a bucket, shaped like the parameter), id of the claiming pass, weak ref to the parameter]
_TASK_ID = getattr(torch._C, "_current_graph_task_id", None)


def register_grad_destination(param, view):
    if view.shape != param.shape or view.dtype != param.dtype or view.device != param.device or not view.is_contiguous():
        raise ValueError("gradient destination must match the parameter's shape, dtype, device and be contiguous")
    key = param.data_ptr()

    def forget(ref, key=key):          # the parameter is gone: drop the registration (and with it the hold on the bucket)
        entry = _GRAD_DEST.get(key)
        if entry is not None and entry[2] is ref:
            del _GRAD_DEST[key]
    _GRAD_DEST[key] = [view, None, weakref.ref(param, forget)]


def unregister_grad_destination(param):
    _GRAD_DEST.pop(param.data_ptr(), None)


_GRAD_DEST_PAUSED = [0]


class grad_destinations_paused(object):
    """``with grad_destinations_paused():`` backward passes allocate their gradients as if no bucket were registered (a
    measurement aid: bench.py's single-graph twin of the multi-rank step is the headline step exactly)."""

    def __enter__(self):
        _GRAD_DEST_PAUSED[0] += 1
        return self

    def __exit__(self, *exc):
        _GRAD_DEST_PAUSED[0] -= 1
        return False


def _claim_grad_destination(key):
    """The registered slice for the gradient of the parameter at address ``key`` if this backward pass may write it there,
    else None."""
    if not _GRAD_DEST or _TASK_ID is None or key is None or _GRAD_DEST_PAUSED[0]:
        return None
    entry = _GRAD_DEST.get(key)
    if entry is None:
        return None
    param = entry[2]()
    if param is None or param.data_ptr() != key or param.grad is not None:
        return None
    view = entry[0]
    if view.shape != param.shape or view.dtype != param.dtype or view.device != param.device:
        return None
    task = _TASK_ID()
    if task < 0 or entry[1] == task:          # not inside a backward pass / a second producer in the same pass
        return None
    entry[1] = task
    return view


def _iw1_accumulator(device):
    """The zero-initialised 64-bit words (ZS_IW1_ACC_WORDS of include/zs_hip.h: totals and shards of the two fixed-point sums) of
    IW1's batch mean (handed back at zero by the kernel; see _scratch)."""
    return _scratch(device, "iw1", lambda sc: True, lambda: (torch.zeros(64, dtype=torch.int64, device=device),))[0]


IW1_POISON_WORD = 63          # ZS_IW1_POISON_WORD of include/zs_hip.h


def iw1_accumulators_ok():
    """False when a launch of the fused objective gave up waiting for its workgroups' shares (it stored a NaN mean and raised
    the accumulator's poison word, include/zs_hip.h): every later batch mean on that accumulator would be wrong.  Cannot happen
    in a healthy process; a training loop that sees a NaN objective over finite costs can check here (synchronises) and call
    ``reset_iw1_accumulators()``."""
    ok = True
    for key, sc in list(_SCRATCH.items()):
        if key[2] == "iw1":
            ok = ok and int(sc[0][IW1_POISON_WORD].item()) == 0
    return ok


def reset_iw1_accumulators():
    """Re-zero every batch-mean accumulator (after ``iw1_accumulators_ok()`` returned False; no launch may be in flight)."""
    for key, sc in list(_SCRATCH.items()):
        if key[2] == "iw1":
            if sc[0].is_cuda:
                torch.cuda.synchronize(sc[0].device)
            sc[0].zero_()


IW1_MAX_DATAPOINTS = 1 << 20        # (round 4's workgroup-per-datapoint kernel stopped at 384; the persistent form takes any batch)
# No limit on the [K, B, X] stream's size either: launched cold (the stream in HBM only) the fused launch loses to K3's x-reuse kernel
# + K2 + K4b beyond B = 1024 (124 us against 83 at B = 2048, profiles/r05_iw1_timing.txt), but a step evaluates the stream right after
# the decoder produced it, and there it wins at every size measured (B = 512 ... 2048: 21.1 / 34.2 / 52.3 / 72.0 us against
# 27.1 / 49.4 / 70.7 / 92.8, profiles/r05_iw1_sizes_instep.txt).  bench.py --iw1-max-stream-bytes sets this for such studies.
IW1_MAX_STREAM_BYTES = 1 << 62


def iw1_unsupported_reason(K, B, X, dtype, *tensors):
    """None inside the fused kernel's domain (include/zs_hip.h, IW1: workgroups own whole datapoints -- one workgroup per CU,
    datapoints dealt round-robin --, lane = particle in the tail, rows read 16 bytes per lane), else WHY not, in words a user
    can act on (``zhusuan.explain``)."""
    if dtype == torch.float64:
        return None                      # (the float64 twin composes plain kernels: any shape)
    if K * B * X * 4 > IW1_MAX_STREAM_BYTES:
        return "the [K, B, X] stream is larger than zhusuan._ops.IW1_MAX_STREAM_BYTES"
    if K > 64:
        return "K = %d particles: the fused kernel reduces a datapoint's particles on one 64-lane wavefront (K <= 64)" % K
    if B > IW1_MAX_DATAPOINTS:
        return "B = %d datapoints (limit %d)" % (B, IW1_MAX_DATAPOINTS)
    if X % 4 or not 256 <= X <= 1024:
        return "rows of %d elements: the fused kernel streams rows of 256 .. 1024 floats, a multiple of 4" % X
    if not all(t.data_ptr() % 16 == 0 for t in tensors):
        return "an operand is not 16-byte aligned (a view into the middle of a tensor?)"
    return None


def iw1_supported(K, B, X, dtype, *tensors):
    return iw1_unsupported_reason(K, B, X, dtype, *tensors) is None


def iw1_term_supported(Dz, dtype, *tensors):
    """Whether a Normal node of a [K, B, Dz] value can ride in IW1's launch as a term (else it enters as ready-made rows)."""
    if dtype == torch.float64:
        return True
    return Dz % 4 == 0 and 4 <= Dz <= 256 and all(t.numel() == 1 or t.data_ptr() % 16 == 0 for t in tensors)


class BernoulliIWObjective(torch.autograd.Function):
    """IW1: the generator side of the importance-weighted objective in ONE launch -- the Bernoulli likelihood's row sums over
    p [K, B, X] (K3), the Normal log-density of the latent value z [K, B, Dz] (K2, optional), the left-to-right sum of the
    generator's terms (plus ready-made rows of further nodes), minus log q, and K4b (per-datapoint IWAE / VIMCO costs, their
    batch mean, both coefficient matrices).  Replaces the per-node loop of ImportanceWeightedObjective.forward
    (zhusuan/variational/importance_weighted_objective.py:66-100) for the IWAE caller.

    Backward is ONE call: the Bernoulli gradient with the row gradients coef * g formed inside the kernel (no multiply launch),
    and, when the variational node (a non-reparameterised Normal draw ``qz`` with parameters ``qmu``, ``qsigma``) is handed
    in, the K-summed gradient of log q w.r.t. those parameters (what K1's backward would have launched); ``logq`` is then
    passed detached.  ``meta`` = (from_logits, Px, Pm, Ps, prior_is_logstd, estimator, want_mean, q_is_logstd).
    Returns (cost, bound_b)."""

    @staticmethod
    def forward(ctx, p, x, z, pmu, psigma, rows_a, logq, qmu, qsigma, qz, meta):
        ctx.set_materialize_grads(False)
        from_logits, Px, Pm, Ps, p_ls, estimator, want_mean, q_ls = meta
        _hip.require_device(p, x, z, pmu, psigma, rows_a, logq, qmu, qsigma, qz)
        sfx = _sfx(p, x, z, pmu, psigma, rows_a, logq, qmu, qsigma, qz)
        K, B, X = p.shape
        dt, dev = p.dtype, p.device
        q2, ld_q = _rows_for_iw(logq, K)
        a2, ld_a = (None, K) if rows_a is None else _rows_for_iw(rows_a, K)
        Dz = z.shape[-1] if z is not None else 1
        out = torch.empty((4 if z is not None else 3, B, K), dtype=dt, device=dev)       # coef [2, B, K], lp_x, lp_z
        vecs = torch.empty((2, B), dtype=dt, device=dev)                                  # per-datapoint costs, bounds
        cost_b, bound = vecs[0], vecs[1]
        cost = torch.empty((), dtype=dt, device=dev) if want_mean else cost_b
        acc = _iw1_accumulator(dev) if want_mean else None
        _hip.lib().call("zs_bernoulli_iw_objective" + sfx, _hip.ptr(p), 1 if from_logits else 0, _hip.ptr(x), Px, K, B, X,
                        _hip.ptr(z), _hip.ptr(pmu), Pm, _hip.ptr(psigma), Ps, Dz, 1 if p_ls else 0,
                        _hip.ptr(a2), ld_a, _hip.ptr(q2), ld_q, estimator, 1 if want_mean else 0,
                        _hip.ptr(out[2]), _hip.ptr(out[3]) if z is not None else None,
                        _hip.ptr(cost_b), _hip.ptr(bound), _hip.ptr(out), _hip.ptr(cost) if want_mean else None,
                        _hip.ptr(acc), _hip.stream_for(p))
        ctx.meta = meta
        ctx.fold_q = qz is not None
        ctx.edge_of = _edge_positions((p, x, z, pmu, psigma, rows_a, logq, qmu, qsigma, qz, meta))
        ctx.save_for_backward(p, x, z, pmu, psigma, out, qmu, qsigma, qz)
        ctx.mark_non_differentiable(bound)
        return cost, bound

    @staticmethod
    def backward(ctx, g_cost, g_bound):
        if g_cost is None:
            return (None,) * 11
        from_logits, Px, Pm, Ps, p_ls, estimator, want_mean, q_ls = ctx.meta
        p, x, z, pmu, psigma, out, qmu, qsigma, qz = ctx.saved_tensors
        K, B, X = p.shape
        coef = out[:2]
        g = g_cost.contiguous()
        need = _pass_needs(ctx, ctx.edge_of)   # (a pass restricted to some parameters: only the side(s) that lead to them)
        gp = torch.empty_like(p) if need[0] else None
        fold = ctx.fold_q and (need[7] or need[8])
        gqmu = torch.empty_like(qmu) if fold else None
        gqsigma = torch.empty_like(qsigma) if fold else None
        if gp is not None or fold:
            _hip.lib().call("zs_bernoulli_iw_objective_bwd" + _sfx(p), _hip.ptr(p), 1 if from_logits else 0, _hip.ptr(x), Px, K, B, X,
                            _hip.ptr(coef), _hip.ptr(g), 0 if want_mean else 1, _hip.ptr(gp),
                            _hip.ptr(qz) if fold else None, _hip.ptr(qmu) if fold else None, _hip.ptr(qsigma) if fold else None,
                            qz.shape[-1] if fold else 1, 1 if q_ls else 0, _hip.ptr(gqmu), _hip.ptr(gqsigma), _hip.stream_for(p))
        gx = None
        if need[1]:
            # the observation's gradient (bernoulli.py:94 is differentiable in `sample`): its own launch -- a model whose observed
            # value comes out of a differentiable net is rare --, row gradients coef[0] * g formed in the kernel as above
            gx = torch.empty_like(x)
            _hip.lib().call("zs_bernoulli_logprob_bwd_x" + _sfx(p), _hip.ptr(p), 1 if from_logits else 0, Px, _hip.ptr(coef), 1, K,
                            _hip.ptr(g), 0 if want_mean else 1, _hip.ptr(gx), K, B, X, _hip.stream_for(p))
        gz = gpm = gps = ga = gq = None
        # the rarer consumers take the row gradients as a tensor (one multiply, as K4b's backward)
        if need[2] or need[3] or need[4] or need[5] or need[6]:
            gc = coef * (g if want_mean else g.reshape(1, -1, 1))                 # [2, B, K]
            if need[5]:
                ga = gc[0]                                                        # (rows_a and logq are [B, K] matrices)
            if need[6]:
                gq = gc[1]
            if z is not None and (need[2] or need[3] or need[4]):
                Dz = z.shape[-1]
                gz, gpm, gps = _normal_logprob_grads(z, pmu, psigma, gc[0].t(), K, B, Dz, (K * B * Dz, Pm, Ps), 1 if p_ls else 0,
                                                     (need[2], need[3], need[4]))
        return gp, gx, gz, gpm, gps, ga, gq, (gqmu if need[7] else None), (gqsigma if need[8] else None), None, None


MAX_TERMS = 6      # ZS_MAX_TERMS of include/zs_hip.h


def _dense_flat(t):
    """`t` as a flat view of its storage when it is dense (any permutation of a contiguous block), else a flat copy:
    the scalar epilogue only needs the SUM of the elements, so their order is irrelevant."""
    if t.dim() == 0:
        return t.reshape(1)
    if t.is_contiguous():
        return t.reshape(-1)
    try:
        if t.storage_offset() == 0 and t.untyped_storage().nbytes() == t.numel() * t.element_size():
            return torch.as_strided(t, (t.numel(),), (1,))
    except RuntimeError:
        pass
    return t.contiguous().reshape(-1)


class ScalarObjective(torch.autograd.Function):
    """S1: ``sum_t coefs[t] * tensors[t].sum()`` in one launch; backward is one multiply of the coefficient vector by
    the incoming gradient, handed to every operand as a stride-0 expansion (the log-prob kernels read their incoming
    gradient through strides, so nothing is materialised)."""

    @staticmethod
    def forward(ctx, coefs, *tensors):
        if not (1 <= len(tensors) <= MAX_TERMS) or len(coefs) != len(tensors):
            raise ValueError("ScalarObjective takes 1..%d (coefficient, tensor) pairs" % MAX_TERMS)
        _hip.require_device(*tensors)
        sfx = _sfx(*tensors)
        flats = [_dense_flat(t) for t in tensors]
        dt, dev = tensors[0].dtype, tensors[0].device
        out = torch.empty((), dtype=dt, device=dev)
        cvec = torch.empty(len(tensors), dtype=dt, device=dev)
        args = []
        for j in range(MAX_TERMS):
            if j < len(flats):
                args += [_hip.ptr(flats[j]), flats[j].numel(), float(coefs[j])]
            else:
                args += [None, 0, 0.0]
        _hip.lib().call("zs_scalar_objective" + sfx, *args, _hip.ptr(out), _hip.ptr(cvec), _hip.stream_for(tensors[0]))
        ctx.save_for_backward(cvec)
        ctx.shapes = [tuple(t.shape) for t in tensors]
        return out

    @staticmethod
    def backward(ctx, g):
        (cvec,) = ctx.saved_tensors
        gv = cvec * g
        return (None,) + tuple(gv[j].expand(shape) if ctx.needs_input_grad[j + 1] else None
                               for j, shape in enumerate(ctx.shapes))


class LogMeanExpRows(torch.autograd.Function):
    """log_mean_exp over the last (contiguous) axis of a 2-D tensor (zhusuan/utils.py:6-21)."""

    @staticmethod
    def forward(ctx, x2d):
        _hip.require_device(x2d)
        sfx = _sfx(x2d)
        x2d = x2d.contiguous()
        B, K = x2d.shape
        out = torch.empty(B, dtype=x2d.dtype, device=x2d.device)
        if B * K > 0:
            _hip.lib().call("zs_log_mean_exp" + sfx, _hip.ptr(x2d), K, B, K, _hip.ptr(out), _hip.stream_for(x2d))
        ctx.save_for_backward(x2d, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x2d, out = ctx.saved_tensors
        K = x2d.shape[1]
        return g.unsqueeze(1) * torch.exp(x2d - out.unsqueeze(1)) / K


# ------------------------------------------------------------------------------------------------
# Logistic / Uniform
# ------------------------------------------------------------------------------------------------
class LogisticSampleLogProb(torch.autograd.Function):
    """L1: z = loc + scale * (log u - log(1-u)) and the row-summed log-density of z in one pass.
    Replaces Logistic._sample + Logistic._log_prob (zhusuan/distributions/logistic.py:52-83)."""

    @staticmethod
    def forward(ctx, loc, scale, u, seed, call, rng_state, K, has_k_axis, n_fold, kfast):
        ctx.set_materialize_grads(False)     # backward handles an undefined gz / glp (see NormalSampleLogProb)
        _hip.require_device(loc, scale, u)
        sfx = _sfx(loc, scale, u)
        shape = tuple(loc.shape)
        M = loc.numel()
        rest = shape[:len(shape) - n_fold]
        D = _prod(shape[len(shape) - n_fold:])
        R = _prod(rest)
        lead = (K,) if has_k_axis else ()
        z = torch.empty(lead + shape, dtype=loc.dtype, device=loc.device)
        if M == 0:
            ctx.meta = None
            return z, torch.zeros(lead + rest, dtype=loc.dtype, device=loc.device)
        buf, lp, sk, sr = _alloc_rows(K, has_k_axis, rest, kfast, loc)
        used = _rng_snapshot(rng_state, u is None)
        _hip.lib().call("zs_logistic_sample_logprob" + sfx, _hip.ptr(loc), _hip.ptr(scale), _hip.ptr(u), seed, call,
                        _hip.ptr(rng_state), _hip.ptr(z), _hip.ptr(buf), K, M, D, sk, sr, _hip.ptr(used), _hip.stream_for(loc))
        if used is not None:
            rng_state, call = used, 0
        ctx.meta = (seed, call, K, M, D, R)
        ctx.rng_state = rng_state
        ctx.save_for_backward(scale, u)
        return z, lp

    @staticmethod
    def backward(ctx, gz, glp):
        if ctx.meta is None:
            return (None,) * 10
        seed, call, K, M, D, R = ctx.meta
        if gz is None and glp is None:
            return (None,) * 10
        scale, u = ctx.saved_tensors
        gloc = torch.empty_like(scale)
        gscale = torch.empty_like(scale)
        gsk = gsr = 0
        if gz is not None:
            gz = gz.contiguous()
        if glp is not None:
            glp, gsk, gsr = _kr_view(glp, K, R)
        _hip.lib().call("zs_logistic_sample_logprob_bwd" + _sfx(scale), _hip.ptr(scale), _hip.ptr(u), seed, call,
                        _hip.ptr(ctx.rng_state), _hip.ptr(gz), _hip.ptr(glp), gsk, gsr, _hip.ptr(gloc), _hip.ptr(gscale),
                        K, M, D, _hip.stream_for(scale))
        return (gloc, gscale) + (None,) * 8


def _krd(full_shape, n_fold):
    """[K, R, D] view of a log-prob problem of shape `full_shape` whose last `n_fold` axes are summed."""
    full_shape = tuple(full_shape)
    out_shape = full_shape[:len(full_shape) - n_fold]
    D = _prod(full_shape[len(full_shape) - n_fold:])
    has_k = len(out_shape) >= 2
    K = out_shape[0] if has_k else 1
    rest = out_shape[1:] if has_k else out_shape
    return K, _prod(rest), D, has_k, rest


def _fold_period(f, P, N, like):
    """Sum the element-wise partial `f` (N values) over the repeats of an operand of period P."""
    if f is None:
        return None
    if P != N:
        f = f.view(N // P, P).sum(0)
    return f.view(like.shape)


class LogisticLogProb(torch.autograd.Function):
    """L2: row-summed Logistic log-density of a given value, periodically broadcast operands
    (logistic.py:69-83)."""

    @staticmethod
    def forward(ctx, x, loc, scale, full_shape, n_fold, periods, kfast):
        _hip.require_device(x, loc, scale)
        sfx = _sfx(x, loc, scale)
        K, R, D, has_k, rest = _krd(full_shape, n_fold)
        buf, lp, sk, sr = _alloc_rows(K, has_k, rest, kfast and n_fold > 0, x)
        Px, Pm, Ps = periods
        if K * R * D > 0:
            _hip.lib().call("zs_logistic_logprob" + sfx, _hip.ptr(x), Px, _hip.ptr(loc), Pm, _hip.ptr(scale), Ps,
                            _hip.ptr(buf), K, R, D, sk, sr, _hip.stream_for(x))
        ctx.meta = (K, R, D, periods)
        ctx.save_for_backward(x, loc, scale)
        return lp

    @staticmethod
    def backward(ctx, glp):
        K, R, D, (Px, Pm, Ps) = ctx.meta
        x, loc, scale = ctx.saved_tensors
        need = ctx.needs_input_grad[:3]
        N = K * R * D
        if N == 0 or not any(need):
            return (None,) * 7
        glp, gsk, gsr = _kr_view(glp, K, R)
        if K > 1 and Px == N and Pm == Ps == R * D:
            # parameters [R, D] repeated over the K particles: reduce over K inside the kernel
            gx = torch.empty_like(x) if need[0] else None
            gloc, gscale = torch.empty_like(loc), torch.empty_like(scale)
            _hip.lib().call("zs_logistic_logprob_bwd_ksum" + _sfx(x), _hip.ptr(x), _hip.ptr(loc), _hip.ptr(scale), _hip.ptr(glp),
                            gsk, gsr, _hip.ptr(gx), _hip.ptr(gloc), _hip.ptr(gscale), K, R, D, _hip.stream_for(x))
            return (gx, gloc if need[1] else None, gscale if need[2] else None, None, None, None, None)
        outs = [torch.empty(N, dtype=x.dtype, device=x.device) if n else None for n in need]
        _hip.lib().call("zs_logistic_logprob_bwd" + _sfx(x), _hip.ptr(x), Px, _hip.ptr(loc), Pm, _hip.ptr(scale), Ps,
                        _hip.ptr(glp), gsk, gsr, _hip.ptr(outs[0]), _hip.ptr(outs[1]), _hip.ptr(outs[2]), K, R, D,
                        _hip.stream_for(x))
        return (_fold_period(outs[0], Px, N, x), _fold_period(outs[1], Pm, N, loc), _fold_period(outs[2], Ps, N, scale),
                None, None, None, None)


class UniformSample(torch.autograd.Function):
    """U1: Uniform._sample (zhusuan/distributions/uniform.py:51-70).  Returns (sample, cache); `cache` is what the
    reference keeps in sample_cache.  d sample / d low = 1 - cache, d sample / d high = cache (uniform.py:70)."""

    @staticmethod
    def forward(ctx, low, high, u, seed, call, rng_state, shape, periods, reparam):
        ctx.set_materialize_grads(False)
        _hip.require_device(low, high, u)
        sfx = _sfx(low, high, u)
        out = torch.empty(tuple(shape), dtype=low.dtype, device=low.device)
        cache = torch.empty_like(out)
        N = out.numel()
        Pl, Ph = periods
        if N:
            _hip.lib().call("zs_uniform_sample" + sfx, _hip.ptr(low), Pl, _hip.ptr(high), Ph, _hip.ptr(u), seed, call,
                            _hip.ptr(rng_state), _hip.ptr(out), _hip.ptr(cache), N, 1 if reparam else 0,
                            _hip.stream_for(low))
        ctx.meta = (N, Pl, Ph)
        ctx.save_for_backward(cache, low, high)
        ctx.mark_non_differentiable(cache)
        return out, cache

    @staticmethod
    def backward(ctx, g, _gcache):
        N, Pl, Ph = ctx.meta
        cache, low, high = ctx.saved_tensors
        if N == 0 or g is None:
            return (None,) * 9
        g = g.contiguous().view(-1)
        c = cache.view(-1)
        ghigh = g * c
        glow = g - ghigh
        return (_fold_period(glow, Pl, N, low), _fold_period(ghigh, Ph, N, high)) + (None,) * 7


class UniformLogProb(torch.autograd.Function):
    """U2: row-summed Uniform log-density (uniform.py:72-85).  Backward (torch.distributions.Uniform.log_prob:
    only ``-log(high - low)`` carries gradient): d/d low = +g/(high - low), d/d high = -g/(high - low)."""

    @staticmethod
    def forward(ctx, x, low, high, full_shape, n_fold, periods, kfast):
        _hip.require_device(x, low, high)
        sfx = _sfx(x, low, high)
        K, R, D, has_k, rest = _krd(full_shape, n_fold)
        buf, lp, sk, sr = _alloc_rows(K, has_k, rest, kfast and n_fold > 0, x)
        Px, Pl, Ph = periods
        if K * R * D > 0:
            _hip.lib().call("zs_uniform_logprob" + sfx, _hip.ptr(x), Px, _hip.ptr(low), Pl, _hip.ptr(high), Ph,
                            _hip.ptr(buf), K, R, D, sk, sr, _hip.stream_for(x))
        ctx.meta = (K, R, D, periods, tuple(full_shape))
        ctx.save_for_backward(low, high)
        return lp

    @staticmethod
    def backward(ctx, glp):
        K, R, D, (Px, Pl, Ph), full_shape = ctx.meta
        low, high = ctx.saved_tensors
        N = K * R * D
        if N == 0 or not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            return (None,) * 7
        g, _, _ = _kr_view(glp, K, R)
        g = g.reshape(K * R, 1).expand(K * R, D).reshape(-1)

        def full(t, P):
            return t.reshape(-1).repeat(N // P) if P != N else t.reshape(-1)
        t = g / (full(high, Ph) - full(low, Pl))
        return (None, _fold_period(t, Pl, N, low), _fold_period(-t, Ph, N, high), None, None, None, None)


def philox_uniform(shape, device, seed, call, rng_state=None, dtype=torch.float32):
    """U(0,1) draws from the kernels' Philox stream (the u that L1 / U1 would draw for the same ids)."""
    out = torch.empty(tuple(shape), dtype=dtype, device=device)
    _hip.require_device(out)
    if out.numel():
        _hip.lib().call("zs_philox_uniform" + _sfx(out), _hip.ptr(out), out.numel(), seed, call, _hip.ptr(rng_state),
                        _hip.stream_for(out))
    return out


class ReinforceEpilogue(torch.autograd.Function):
    """R1: the score-function estimator's scalar epilogue in one launch (zhusuan/variational/elbo.py:163-238).
    `moving_mean` (float32 [1]) and `local_step` (int32 [1]) are the module's buffers, updated by the kernel.
    Returns the cost: 0-d when `do_mean`, else of logq's shape."""

    @staticmethod
    def forward(ctx, logp, logq, baseline, moving_mean, local_step, variance_reduction, do_mean, decay):
        _hip.require_device(logp, logq, baseline, moving_mean, local_step)
        sfx = _sfx(logp, logq, baseline)
        shape = tuple(logq.shape)
        n = logq.numel()
        lp, lq = logp.contiguous(), logq.contiguous()
        use_b = bool(variance_reduction) and baseline is not None
        b, Pb = None, 1
        if use_b:
            if baseline.numel() == 1:
                b = baseline.reshape(1)
            else:
                b, Pb = baseline.expand(shape).contiguous(), n
        signal = torch.empty(shape, dtype=lq.dtype, device=lq.device)
        resid = torch.empty(shape, dtype=lq.dtype, device=lq.device) if use_b else None
        cost = torch.empty((() if do_mean else shape), dtype=lq.dtype, device=lq.device)
        if n:
            ws, ticket = _lj_workspace(lq.device) if n > 16384 else (None, None)      # long vectors: many workgroups
            _hip.lib().call("zs_reinforce" + sfx, _hip.ptr(lp), _hip.ptr(lq), _hip.ptr(b), Pb, n,
                            1 if variance_reduction else 0, 1 if do_mean else 0, float(decay),
                            _hip.ptr(moving_mean), _hip.ptr(local_step), _hip.ptr(signal), _hip.ptr(cost), _hip.ptr(resid),
                            _hip.ptr(ws), ws.numel() if ws is not None else 0, _hip.ptr(ticket), _hip.stream_for(lq))
        ctx.meta = (n, bool(do_mean), use_b, tuple(baseline.shape) if use_b else None)
        ctx.save_for_backward(signal, resid)
        return cost

    @staticmethod
    def backward(ctx, g):
        n, do_mean, use_b, bshape = ctx.meta
        signal, resid = ctx.saved_tensors
        if n == 0:
            return (None,) * 8
        if do_mean:
            g = g / n
        glogp = (-g).expand(signal.shape) if ctx.needs_input_grad[0] else None
        glogq = -(g * signal) if ctx.needs_input_grad[1] else None
        gb = None
        if use_b and ctx.needs_input_grad[2]:
            gb = (-(g * resid)).sum_to_size(bshape)
        return glogp, glogq, gb, None, None, None, None, None


def philox_normal(shape, device, seed, call, rng_state=None, dtype=torch.float32):
    """Standard normals from the kernels' own Philox stream (the eps K1 would draw for the same ids)."""
    out = torch.empty(tuple(shape), dtype=dtype, device=device)
    _hip.require_device(out)
    sfx = _sfx(out)
    if out.numel():
        _hip.lib().call("zs_philox_normal" + sfx, _hip.ptr(out), out.numel(), seed, call, _hip.ptr(rng_state),
                        _hip.stream_for(out))
    return out


def bernoulli_sample(probs, Pp, shape, seed, call, rng_state=None):
    """K5: Bernoulli._sample (zhusuan/distributions/bernoulli.py:72-82)."""
    _hip.require_device(probs)
    sfx = _sfx(probs)
    out = torch.empty(tuple(shape), dtype=probs.dtype, device=probs.device)
    if out.numel():
        _hip.lib().call("zs_bernoulli_sample" + sfx, _hip.ptr(probs), Pp, _hip.ptr(out), out.numel(), seed, call,
                        _hip.ptr(rng_state), _hip.stream_for(probs))
    return out


def periodic_operand(t, full_shape):
    """(contiguous tensor, period) describing how `t` broadcasts into `full_shape`.

    Leading-axis broadcast (the reference's ``repeat`` of parameters along the sample axis,
    normal.py:94-95,112-116) and scalars are expressed as a period and cost no copy; any other
    broadcast pattern is materialised with ``expand().contiguous()`` (differentiable plumbing)."""
    full_shape = tuple(full_shape)
    N = _prod(full_shape)
    if t.numel() == 1 and N >= 1:
        return t.reshape(1), 1
    shp = tuple(t.shape)
    while shp and shp[0] == 1:
        shp = shp[1:]
    if shp == full_shape[len(full_shape) - len(shp):]:
        return t.contiguous(), t.numel()
    return t.expand(full_shape).contiguous(), N


# ------------------------------------------------------------------------------------------------
# One-launch pieces for the launch-bound shapes (include/zs_hip.h: LJ1, MS1, PL1)
# ------------------------------------------------------------------------------------------------
def _lj_workspace(device):
    """(double [ZS_LJ_WORKSPACE], ticket int32 [1]) of LJ1's (and R1's) deterministic sums (see _scratch)."""
    return _scratch(device, "lj", lambda sc: True,
                    lambda: (torch.empty(_hip.LJ_WORKSPACE, dtype=torch.float64, device=device),
                             torch.zeros(1, dtype=torch.int32, device=device)))


class LogJointScalar(torch.autograd.Function):
    """LJ1: ``sum_t coef_t * sum_i logprob_t(i)`` over up to 8 terms in ONE launch, its whole backward in one more.

    ``spec`` is a tuple of ``(family, coef, n, px, pa, pb)`` per term (``_hip.LJ_*`` families; periods in elements of the
    term's full problem of ``n`` elements); ``tensors`` holds ``(x, a, b)`` per term (``None`` where a family has no such
    operand), each contiguous with exactly its period's number of elements.  Replaces the per-node loop of ELBO.log_joint
    (zhusuan/variational/elbo.py:58-79) over StochasticTensor.log_prob (zhusuan/framework/stochastic_tensor.py:160-181)
    plus ELBO.sgvb's scalar arithmetic (elbo.py:155-161) for objectives whose nodes all reduce to scalars."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        nt = len(spec)
        if not (1 <= nt <= _hip.LJ_MAX_TERMS) or len(tensors) != 3 * nt:
            raise ValueError("LogJointScalar takes 1..%d terms with three operands each" % _hip.LJ_MAX_TERMS)
        _hip.require_device(*tensors)
        sfx = _sfx(*tensors)
        first = next(t for t in tensors if t is not None)
        dt, dev = first.dtype, first.device
        terms = (_hip.LJTerm * nt)()
        for i, (fam, coef, n, px, pa, pb) in enumerate(spec):
            x, a, b = tensors[3 * i:3 * i + 3]
            for t, P in ((x, px), (a, pa), (b, pb)):
                if t is not None and (not t.is_contiguous() or t.numel() != P):
                    raise ValueError("LogJointScalar: operand of %d elements given with period %d" % (t.numel(), P))
            tm = terms[i]
            tm.family, tm.n, tm.coef = int(fam), int(n), float(coef)
            tm.x, tm.px = (x.data_ptr() if x is not None else None), int(px)
            tm.a, tm.pa = (a.data_ptr() if a is not None else None), int(pa)
            tm.b, tm.pb = (b.data_ptr() if b is not None else None), int(pb)
        out = torch.empty((), dtype=dt, device=dev)
        ws, ticket = _lj_workspace(dev)
        _hip.lib().call("zs_logjoint_scalar" + sfx, ctypes.byref(terms), nt, _hip.ptr(out), _hip.ptr(ws), ws.numel(),
                        _hip.ptr(ticket), _hip.stream_for(first))
        ctx.spec = tuple(spec)
        ctx.save_for_backward(*[t for t in tensors if t is not None])
        ctx.present = [t is not None for t in tensors]
        return out

    @staticmethod
    def backward(ctx, g):
        spec, nt = ctx.spec, len(ctx.spec)
        saved = list(ctx.saved_tensors)
        tensors = [saved.pop(0) if p else None for p in ctx.present]
        first = next(t for t in tensors if t is not None)
        dt, dev = first.dtype, first.device
        sfx = _sfx(first)
        need = ctx.needs_input_grad[1:]
        terms = (_hip.LJTerm * nt)()
        grads = [None] * (3 * nt)
        for i, (fam, coef, n, px, pa, pb) in enumerate(spec):
            x, a, b = tensors[3 * i:3 * i + 3]
            tm = terms[i]
            tm.family, tm.n, tm.coef = int(fam), int(n), float(coef)
            tm.x, tm.px = (x.data_ptr() if x is not None else None), int(px)
            tm.a, tm.pa = (a.data_ptr() if a is not None else None), int(pa)
            tm.b, tm.pb = (b.data_ptr() if b is not None else None), int(pb)
            if fam == _hip.LJ_ROWS:
                continue
            for j, name in enumerate(("gx", "ga", "gb")):
                t = tensors[3 * i + j]
                if t is not None and need[3 * i + j]:
                    grads[3 * i + j] = torch.empty_like(t)
                    setattr(tm, name, grads[3 * i + j].data_ptr())
        gcoef = torch.empty(nt, dtype=dt, device=dev)
        g = g.to(dt).contiguous()
        ws, ticket = _lj_workspace(dev)
        _hip.lib().call("zs_logjoint_scalar_bwd" + sfx, ctypes.byref(terms), nt, _hip.ptr(g), _hip.ptr(gcoef), _hip.ptr(ws),
                        ws.numel(), _hip.ptr(ticket), _hip.stream_for(first))
        for i, (fam, _c, _n, _px, _pa, _pb) in enumerate(spec):
            if fam == _hip.LJ_ROWS and need[3 * i]:
                # a stride-0 expansion of the term's scalar: the producers' backward kernels read gradients through strides
                grads[3 * i] = gcoef[i].expand(tensors[3 * i].shape)
        return (None,) + tuple(grads)


class NormalSampleLogProbMulti(torch.autograd.Function):
    """MS1: K1 (z = mu + sigma * eps and its row-summed log-density) for SEVERAL reparameterised Normal nodes in one
    launch, their backward in one more.  ``meta`` = per node ``(K, has_k_axis, n_fold, is_logstd, call)``; ``tensors`` =
    ``(mu, sigma, eps)`` per node (eps None: in-kernel Philox with call id base + call).  Returns (z_0, lp_0, z_1, lp_1, ...,
    alias_0, alias_1, ...): ``alias_i`` is ``z_i`` once more (the same memory) as a separate output -- a consumer that reads the
    sample through it (the log-joint's prior term) sends its gradient to a slot of its own, and the backward kernel adds the
    two slots while it reads them, where autograd would have launched an accumulation in front of it.
    Replaces the per-latent re-read of ELBO.forward (zhusuan/variational/elbo.py:122) for models with more than one latent
    node (the BNN's weight matrices, examples/bayesian_neural_nets/bnn_vi.py:83-93)."""

    @staticmethod
    def forward(ctx, meta, seed, rng_state, *tensors):
        ctx.set_materialize_grads(False)
        nt = len(meta)
        if not (1 <= nt <= _hip.MS_MAX_TERMS) or len(tensors) != 3 * nt:
            raise ValueError("NormalSampleLogProbMulti takes 1..%d nodes with three operands each" % _hip.MS_MAX_TERMS)
        _hip.require_device(*tensors)
        sfx = _sfx(*tensors)
        terms = (_hip.MSTerm * nt)()
        outs, dims = [], []
        any_philox = False
        for i, (K, has_k, n_fold, is_logstd, call) in enumerate(meta):
            mu, sigma, eps = tensors[3 * i:3 * i + 3]
            shape = tuple(mu.shape)
            M = mu.numel()
            rest = shape[:len(shape) - n_fold]
            D = max(_prod(shape[len(shape) - n_fold:]), 1)
            R = _prod(rest)
            lead = (K,) if has_k else ()
            z = torch.empty(lead + shape, dtype=mu.dtype, device=mu.device)
            if M == 0:
                lp, sk, sr, buf = torch.zeros(lead + rest, dtype=mu.dtype, device=mu.device), 0, 0, None
            else:
                buf, lp, sk, sr = _alloc_rows(K, has_k, rest, True, mu)
            tm = terms[i]
            tm.mu, tm.sigma, tm.eps = mu.data_ptr(), sigma.data_ptr(), (eps.data_ptr() if eps is not None else None)
            tm.z, tm.lp = z.data_ptr(), (buf.data_ptr() if buf is not None else None)
            tm.K, tm.M, tm.D, tm.lp_stride_k, tm.lp_stride_r = K, M, D, sk, sr
            tm.offset, tm.sigma_is_logstd = int(call), 1 if is_logstd else 0
            any_philox = any_philox or eps is None
            outs += [z, lp]
            dims.append((K, M, D, R))
        used = _rng_snapshot(rng_state, any_philox)
        first = tensors[0]
        _hip.lib().call("zs_normal_sample_logprob_multi" + sfx, ctypes.byref(terms), nt, seed, _hip.ptr(rng_state), _hip.ptr(used),
                        _hip.stream_for(first))
        ctx.meta, ctx.dims, ctx.seed = tuple(meta), dims, seed
        ctx.rng_state = used if used is not None else rng_state
        ctx.save_for_backward(*[t for t in tensors if t is not None])
        ctx.present = [t is not None for t in tensors]
        # (detach(): the same storage without a view relationship -- a view would keep its base alive, and the base keeps the alias
        # in an attribute: a reference cycle that only the garbage collector frees)
        return tuple(outs) + tuple(outs[2 * i].detach() for i in range(nt))

    @staticmethod
    def backward(ctx, *gouts):
        nt = len(ctx.meta)
        saved = list(ctx.saved_tensors)
        tensors = [saved.pop(0) if p else None for p in ctx.present]
        if all(g is None for g in gouts):
            return (None,) * (3 + 3 * nt)
        sfx = _sfx(tensors[0])
        terms = (_hip.MSTerm * nt)()
        grads = [None] * (3 * nt)
        keep = []
        for i, (K, has_k, n_fold, is_logstd, call) in enumerate(ctx.meta):
            mu, sigma, eps = tensors[3 * i:3 * i + 3]
            Kk, M, D, R = ctx.dims[i]
            gz, glp = gouts[2 * i], gouts[2 * i + 1]
            gmu, gsigma = torch.empty_like(mu), torch.empty_like(sigma)
            grads[3 * i], grads[3 * i + 1] = gmu, gsigma
            tm = terms[i]
            tm.sigma, tm.eps = sigma.data_ptr(), (eps.data_ptr() if eps is not None else None)
            tm.K, tm.M, tm.D = Kk, M, D
            tm.offset, tm.sigma_is_logstd = int(call), 1 if is_logstd else 0
            tm.gmu, tm.gsigma = gmu.data_ptr(), gsigma.data_ptr()
            if gz is not None:
                gz = gz.contiguous()
                keep.append(gz)
                tm.gz = gz.data_ptr()
            gz2 = gouts[2 * nt + i]
            if gz2 is not None:
                gz2 = gz2.contiguous()
                keep.append(gz2)
                tm.gz2 = gz2.data_ptr()
            if glp is not None and M > 0:
                glp, gsk, gsr = _kr_view(glp, Kk, R)
                keep.append(glp)
                tm.glp, tm.glp_stride_k, tm.glp_stride_r = glp.data_ptr(), gsk, gsr
        _hip.lib().call("zs_normal_sample_logprob_multi_bwd" + sfx, ctypes.byref(terms), nt, ctx.seed, _hip.ptr(ctx.rng_state),
                        _hip.stream_for(tensors[0]))
        del keep
        return (None, None, None) + tuple(grads)


def _pl_tickets(device, K):
    """Zero-initialised ticket words of PL1's backward, one per particle (see _scratch)."""
    return _scratch(device, "pl", lambda sc: sc[0].numel() >= K,
                    lambda: (torch.zeros(max(K, 64), dtype=torch.int32, device=device),))[0]


class ParticleLinear(torch.autograd.Function):
    """PL1: ``out[k, b, :] = act(([h[k, b, :], 1] @ w[k].T) / sqrt(n_in + 1))`` -- the BNN caller's particle-batched layer
    (examples/bayesian_neural_nets/bnn_vi.py:36-48: repeat of w over the batch, appended column of ones, matmul, division,
    ReLU) as one kernel forward and one backward.  ``h``: [K, B, n_in] or [B, n_in] (shared by the particles);
    ``w``: [K, n_out, n_in + 1]."""

    @staticmethod
    def forward(ctx, h, w, relu):
        _hip.require_device(h, w)
        sfx = _sfx(h, w)
        K, n_out, n_in1 = w.shape
        n_in = n_in1 - 1
        shared = h.dim() == 2
        B = h.shape[-2]
        if h.shape[-1] != n_in or (not shared and (h.dim() != 3 or h.shape[0] != K)):
            raise RuntimeError("particle_linear: h %s does not match w %s" % (tuple(h.shape), tuple(w.shape)))
        h, w = h.contiguous(), w.contiguous()
        out = torch.empty((K, B, n_out), dtype=h.dtype, device=h.device)
        _hip.lib().call("zs_particle_linear" + sfx, _hip.ptr(h), 0 if shared else B * n_in, _hip.ptr(w), _hip.ptr(out), K, B, n_in,
                        n_out, 1 if relu else 0, _hip.stream_for(h))
        ctx.meta = (K, B, n_in, n_out, bool(relu), shared)
        ctx.save_for_backward(h, w, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        K, B, n_in, n_out, relu, shared = ctx.meta
        h, w, out = ctx.saved_tensors
        need_h, need_w = ctx.needs_input_grad[:2]
        if not (need_h or need_w):
            return None, None, None
        gout = gout.contiguous()
        gh = torch.empty((K, B, n_in), dtype=h.dtype, device=h.device) if need_h else None
        gw = torch.empty_like(w)
        part = torch.empty(K * ((B + 15) // 16) * n_out * (n_in + 1), dtype=h.dtype, device=h.device)      # tile partials of gw
        tickets = _pl_tickets(h.device, K)
        _hip.lib().call("zs_particle_linear_bwd" + _sfx(h), _hip.ptr(h), 0 if shared else B * n_in, _hip.ptr(w), _hip.ptr(out),
                        _hip.ptr(gout), _hip.ptr(gh), _hip.ptr(gw), K, B, n_in, n_out, 1 if relu else 0, _hip.ptr(part), part.numel(),
                        _hip.ptr(tickets), _hip.stream_for(h))
        if need_h and shared:
            gh = gh.sum(0)
        return gh, (gw if need_w else None), None


class ParticleMLP(torch.autograd.Function):
    """PM1: a chain of PL1 layers (ReLU after every layer but the last) -- the BNN caller's whole network
    (examples/bayesian_neural_nets/bnn_vi.py:27-48) -- as ONE launch forward and ONE backward.  ``x``: [B, n_0] (shared by the
    particles) or [K, B, n_0]; ``ws[l]``: [K, n_{l+1}, n_l + 1].  Returns the last layer's output [K, B, n_L]; bit-identical to
    ``ParticleLinear`` applied layer by layer."""

    @staticmethod
    def forward(ctx, x, *ws):
        _hip.require_device(x, *ws)
        sfx = _sfx(x, *ws)
        L = len(ws)
        K = ws[0].shape[0]
        shared = x.dim() == 2
        B = x.shape[-2]
        sizes = [ws[0].shape[2] - 1] + [w.shape[1] for w in ws]
        ok = x.shape[-1] == sizes[0] and (shared or (x.dim() == 3 and x.shape[0] == K))
        for l, w in enumerate(ws):
            ok = ok and w.dim() == 3 and w.shape[0] == K and w.shape[2] == sizes[l] + 1
        if not ok:
            raise RuntimeError("particle_mlp: x %s does not match the weights %s" % (tuple(x.shape), [tuple(w.shape) for w in ws]))
        x = x.contiguous()
        ws = [w.contiguous() for w in ws]
        outs = [torch.empty((K, B, sizes[l + 1]), dtype=x.dtype, device=x.device) for l in range(L)]
        table = (_hip.PMLayer * L)()
        for l in range(L):
            table[l].w, table[l].out, table[l].gw = _hip.ptr(ws[l]), _hip.ptr(outs[l]), None
            table[l].n_in, table[l].n_out = sizes[l], sizes[l + 1]
        _hip.lib().call("zs_particle_mlp" + sfx, _hip.ptr(x), 0 if shared else B * sizes[0], ctypes.byref(table), L, K, B,
                        _hip.stream_for(x))
        ctx.meta = (K, B, sizes, shared)
        ctx.save_for_backward(x, *ws, *outs)          # (the hidden activations are intermediates: saved, not returned)
        return outs[-1]

    @staticmethod
    def backward(ctx, gout):
        K, B, sizes, shared = ctx.meta
        L = len(sizes) - 1
        saved = ctx.saved_tensors
        x, ws, outs = saved[0], saved[1:1 + L], saved[1 + L:]
        need_x = ctx.needs_input_grad[0]
        if not any(ctx.needs_input_grad):
            return (None,) * (1 + L)
        gout = gout.contiguous()
        gx = torch.empty((K, B, sizes[0]), dtype=x.dtype, device=x.device) if need_x else None
        gws = [torch.empty_like(w) for w in ws]
        slab = sum(sizes[l + 1] * (sizes[l] + 1) + 3 for l in range(L))
        part = torch.empty(K * ((B + 15) // 16) * slab, dtype=x.dtype, device=x.device)          # tile partials of every layer's gw
        tickets = _pl_tickets(x.device, K)
        table = (_hip.PMLayer * L)()
        for l in range(L):
            table[l].w, table[l].out, table[l].gw = _hip.ptr(ws[l]), _hip.ptr(outs[l]), _hip.ptr(gws[l])
            table[l].n_in, table[l].n_out = sizes[l], sizes[l + 1]
        _hip.lib().call("zs_particle_mlp_bwd" + _sfx(x), _hip.ptr(x), 0 if shared else B * sizes[0], ctypes.byref(table), L,
                        _hip.ptr(gout), _hip.ptr(gx), K, B, _hip.ptr(part), part.numel(), _hip.ptr(tickets), _hip.stream_for(x))
        if need_x and shared:
            gx = gx.sum(0)
        return (gx,) + tuple(g if ctx.needs_input_grad[1 + l] else None for l, g in enumerate(gws))


def column_sum(x2d, out=None):
    """CS1: ``x2d.sum(0)`` of a contiguous [rows, cols] matrix in one launch, deterministic (include/zs_hip.h).  ``out``: where
    the sums go (a contiguous [cols] tensor, e.g. a gradient bucket's slice) instead of a fresh tensor."""
    _hip.require_device(x2d)
    sfx = _sfx(x2d)
    rows, cols = x2d.shape
    dev = x2d.device
    if out is None:
        out = torch.empty(cols, dtype=x2d.dtype, device=dev)
    if cols == 0:
        return out
    ws, tickets = _cs_scratch(dev, x2d.dtype, cols)
    _hip.lib().call("zs_column_sum" + sfx, _hip.ptr(x2d), _hip.ptr(out), rows, cols, _hip.ptr(ws), ws.numel(), _hip.ptr(tickets),
                    tickets.numel(), _hip.stream_for(x2d))
    return out


def particle_rmse(pred, y):
    """PR1: ``sqrt(mean((y - pred.mean(0)) ** 2))`` of ``pred`` [K, B] and ``y`` [B] in one launch (include/zs_hip.h).  A
    diagnostic: the result carries no autograd history."""
    pred, y = pred.detach(), y.detach()
    _hip.require_device(pred, y)
    if pred.dim() != 2 or y.dim() != 1 or pred.shape[1] != y.shape[0] or pred.dtype != y.dtype:
        raise RuntimeError("particle_rmse: pred [K, B] and y [B] of one dtype expected, got %s and %s" % (
            tuple(pred.shape), tuple(y.shape)))
    sfx = _sfx(pred)
    pred, y = pred.contiguous(), y.contiguous()
    K, B = pred.shape
    out = torch.empty((), dtype=pred.dtype, device=pred.device)
    ws, ticket = _lj_workspace(pred.device) if B > 4096 else (None, None)
    _hip.lib().call("zs_particle_rmse" + sfx, _hip.ptr(pred), _hip.ptr(y), _hip.ptr(out), K, B, _hip.ptr(ws) if ws is not None else None,
                    ws.numel() if ws is not None else 0, _hip.ptr(ticket) if ticket is not None else None, _hip.stream_for(pred))
    return out


def _cs_scratch(dev, dtype, cols):
    ctiles = (cols + 63) // 64
    return _scratch(dev, ("cs", dtype), lambda sc: sc[0].numel() >= 128 * (cols + 256) and sc[1].numel() >= ctiles,
                    lambda: (torch.empty(128 * (max(cols, 1024) + 256), dtype=dtype, device=dev),
                             torch.zeros(max(ctiles, 16), dtype=torch.int32, device=dev)))


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2          # ZS_ACT_* of include/zs_hip.h
_ADDMM_RELU = getattr(torch, '_addmm_activation', None)      # bias + ReLU epilogue of the GEMM (a private torch entry point: optional)


def dense_act_bwd(g2d, y2d, act, gb_out=None):
    """AB1: ``(g * act'(y), (g * act'(y)).sum(0))`` of contiguous [rows, cols] matrices in one launch (include/zs_hip.h).
    ``gb_out``: where the column sums go (a contiguous [cols] tensor) instead of a fresh tensor."""
    _hip.require_device(g2d)
    sfx = _sfx(g2d)
    rows, cols = g2d.shape
    dev = g2d.device
    gpre = torch.empty_like(g2d)
    gb = torch.empty(cols, dtype=g2d.dtype, device=dev) if gb_out is None else gb_out
    if cols == 0:
        return gpre, gb
    ws, tickets = _cs_scratch(dev, g2d.dtype, cols)
    _hip.lib().call("zs_dense_act_bwd" + sfx, _hip.ptr(g2d), _hip.ptr(y2d), int(act), _hip.ptr(gpre), _hip.ptr(gb), rows, cols,
                    _hip.ptr(ws), ws.numel(), _hip.ptr(tickets), tickets.numel(), _hip.stream_for(g2d))
    return gpre, gb


class DenseLayer(torch.autograd.Function):
    """``act(F.linear(x, w, b))``, act in {none, ReLU, sigmoid}.  Forward: torch's GEMM, the bias and a ReLU riding in its
    epilogue (``torch._addmm_activation``: the same hipBLASLt solution as ``F.linear``, bit-identical, minus torch's clamp
    pass).  Backward: the activation's backward and the bias gradient in one pass over the gradient (AB1; CS1 for a layer
    without activation) instead of torch's threshold_backward / sigmoid_backward pass plus its generic reduction; the two
    GEMMs (grad_input = g @ w, grad_weight = g.T @ x) are torch's."""

    @staticmethod
    def forward(ctx, x, w, b, act=ACT_NONE):
        ctx.has_bias = b is not None
        ctx.act = act
        ctx.edge_of = _edge_positions((x, w, b, act))
        ctx.bias_key = b.data_ptr() if (b is not None and _GRAD_DEST) else None      # (finds the bias gradient's destination)
        if act == ACT_RELU and b is not None and x.dim() >= 1 and x.shape[-1] == w.shape[1] and b.dim() == 1 and _ADDMM_RELU is not None:
            y = _ADDMM_RELU(b, x.reshape(-1, x.shape[-1]), w.t()).reshape(*x.shape[:-1], w.shape[0])
        else:
            y = torch.nn.functional.linear(x, w, b)
            if act == ACT_RELU:
                y = torch.relu_(y)
            elif act == ACT_SIGMOID:
                y = torch.sigmoid_(y)
        if act == ACT_NONE:
            ctx.save_for_backward(x, w)
        else:
            ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors[:2]
        need = _pass_needs(ctx, ctx.edge_of)          # (a staged backward visits this layer for its input's sake only)
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        gx = gw = gb = None
        want_gb = ctx.has_bias and need[2]
        # a registered gradient destination (a data-parallel bucket's slice): the bias sums and the weight GEMM write there
        b_dest = _claim_grad_destination(ctx.bias_key) if want_gb else None
        if ctx.act != ACT_NONE:
            y = ctx.saved_tensors[2]
            g2, gb = dense_act_bwd(g2, y.reshape(-1, y.shape[-1]), ctx.act, gb_out=b_dest)
        elif want_gb:
            gb = column_sum(g2, out=b_dest)
        if not want_gb:
            gb = None
        elif b_dest is not None:
            gb = gb.view_as(gb)          # a fresh alias: autograd adopts a gradient it is the sole owner of
        if need[0]:
            gx = (g2 @ w).reshape(x.shape)
        if need[1]:
            w_dest = _claim_grad_destination(w.data_ptr())
            x2 = x.reshape(-1, x.shape[-1])
            if w_dest is not None:
                gw = torch.mm(g2.t(), x2, out=w_dest).view_as(w_dest)
            else:
                gw = g2.t() @ x2
        return gx, gw, gb, None

"""ctypes binding of the gfx950 kernel library (C ABI: include/zs_hip.h).

The library is loaded from ``zhusuan-pytorch_amd/lib/libzs_hip.so`` (built in-tree by
``__graft_entry__.build()`` / ``csrc/Makefile``).  There is NO CPU fallback: if the library
is missing, or a tensor handed to a kernel is not resident on a HIP device, the call raises.

PyTorch is plumbing only: tensors provide device memory (``data_ptr()``) and the current HIP
stream; no torch type crosses the C ABI.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# The in-tree library.  ZS_HIP_LIBRARY points at an alternative build of the same ABI (kernel experiments: tools/); bench.py
# records which file was loaded (path, sha256, zs_build_info) and refuses an override unless --allow-experiments is given.
LIB_PATH = os.environ.get("ZS_HIP_LIBRARY") or os.path.join(os.path.dirname(_HERE), "lib", "libzs_hip.so")
ABI_VERSION = 15

_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64
_int = ctypes.c_int

# name -> argtypes ; every function returns int (0 = ok) unless noted
PROTOTYPES = {
    "zs_normal_sample_logprob_f32": [_p, _p, _p, _u64, _u64, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p, _p],
    "zs_normal_sample_logprob_pair_f32": [_p, _p, _u64, _u64, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p, _p],
    "zs_normal_sample_logprob_bwd_f32": [_p, _p, _u64, _u64, _p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _int, _p],
    "zs_normal_logprob_f32": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _int, _p],
    "zs_normal_logprob_bwd_f32": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _int, _p],
    "zs_normal_logprob_bwd_ksum_f32": [_p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _int, _p],
    "zs_bernoulli_logprob_f32": [_p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "zs_bernoulli_logprob_bwd_f32": [_p, _p, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _p],
    # p, from_logits, Px, glp, glp_stride_k, glp_stride_r, gscale, gscale_stride, gx, K, R, D, stream
    "zs_bernoulli_logprob_bwd_x_f32": [_p, _int, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _p],
    "zs_bernoulli_logits_logprob_f32": [_p, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "zs_bernoulli_logits_logprob_bwd_f32": [_p, _p, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _p],
    "zs_bernoulli_sample_f32": [_p, _i64, _p, _i64, _u64, _u64, _p, _p],
    "zs_iw_reduce_f32": [_p, _i64, _p, _i64, _i64, _i64, _int, _p, _p, _p, _p, _p],
    "zs_log_mean_exp_f32": [_p, _i64, _i64, _i64, _p, _p],
    "zs_philox_normal_f32": [_p, _i64, _u64, _u64, _p, _p],
    # Logistic / Uniform
    "zs_logistic_sample_logprob_f32": [_p, _p, _p, _u64, _u64, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p],
    "zs_logistic_sample_logprob_bwd_f32": [_p, _p, _u64, _u64, _p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _p],
    "zs_logistic_logprob_f32": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "zs_logistic_logprob_bwd_f32": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _p],
    "zs_logistic_logprob_bwd_ksum_f32": [_p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _p],
    "zs_uniform_sample_f32": [_p, _i64, _p, _i64, _p, _u64, _u64, _p, _p, _p, _i64, _int, _p],
    "zs_uniform_logprob_f32": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "zs_philox_uniform_f32": [_p, _i64, _u64, _u64, _p, _p],
    # ELBO.reinforce epilogue
    # ..., signal, cost, resid, workspace, workspace_len, ticket, stream
    "zs_reinforce_f32": [_p, _p, _p, _i64, _i64, _int, _int, ctypes.c_double, _p, _p, _p, _p, _p, _p, _i64, _p, _p],
    # the whole importance-weighted objective in one launch
    "zs_iw_objective_f32": [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _int, _int, _p, _p, _p, _p, _p, _i64, _p, _p],
    # IW1: p, from_logits, x, Px, K, R, D, z, pmu, Pm, psigma, Ps, Dz, psigma_is_logstd, rows_a, ld_a, logq, ld_q, estimator,
    # want_mean, lp_x, lp_z, cost_b, bound_b, coef, mean_cost, acc, stream
    "zs_bernoulli_iw_objective_f32": [_p, _int, _p, _i64, _i64, _i64, _i64, _p, _p, _i64, _p, _i64, _i64, _int, _p, _i64, _p, _i64,
                                      _int, _int, _p, _p, _p, _p, _p, _p, _p, _p],
    # p, from_logits, x, Px, K, R, D, coef, gout, gout_stride, gp, zq, qmu, qsigma, Dq, qsigma_is_logstd, gqmu, gqsigma, stream
    "zs_bernoulli_iw_objective_bwd_f32": [_p, _int, _p, _i64, _i64, _i64, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _int, _p, _p, _p],
    # scalar ELBO epilogue: six (rows, n, coef) slots, out, coef_out, stream
    "zs_scalar_objective_f32": [_p, _i64, ctypes.c_double] * 6 + [_p, _p, _p],
    # Adam update: param_ptrs, grad_ptrs, starts (host arrays), n_tensors, exp_avg, exp_avg_sq, steps, ticket, n, lr, beta1,
    # beta2, eps, grad_scale, hyper (device, optional), stream
    "zs_adam_step_f32": [_p, _p, _p, _int, _p, _p, _p, _p, _i64] + [ctypes.c_double] * 5 + [_p, _p],
}
ADAM_MAX_TENSORS = 32      # ZS_ADAM_MAX_TENSORS of include/zs_hip.h

# LJ1 / MS1: host-side tables copied into the kernel arguments (struct zs_lj_term / zs_ms_term of include/zs_hip.h)
LJ_MAX_TERMS, LJ_WORKSPACE = 8, 8192
LJ_ROWS, LJ_NORMAL, LJ_NORMAL_LOGSTD, LJ_BERNOULLI, LJ_BERNOULLI_LOGITS = 0, 1, 2, 3, 4
MS_MAX_TERMS = 8
PM_MAX_LAYERS = 4


class LJTerm(ctypes.Structure):
    _fields_ = [("family", ctypes.c_int32), ("reserved", ctypes.c_int32), ("n", _i64),
                ("x", _p), ("px", _i64), ("a", _p), ("pa", _i64), ("b", _p), ("pb", _i64),
                ("coef", ctypes.c_double), ("gx", _p), ("ga", _p), ("gb", _p)]


class PMLayer(ctypes.Structure):          # struct zs_pm_layer
    _fields_ = [("w", _p), ("out", _p), ("gw", _p), ("n_in", _i64), ("n_out", _i64)]


class MSTerm(ctypes.Structure):
    _fields_ = [("mu", _p), ("sigma", _p), ("eps", _p), ("z", _p), ("lp", _p),
                ("K", _i64), ("M", _i64), ("D", _i64), ("lp_stride_k", _i64), ("lp_stride_r", _i64),
                ("offset", _u64), ("sigma_is_logstd", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("gz", _p), ("glp", _p), ("glp_stride_k", _i64), ("glp_stride_r", _i64), ("gmu", _p), ("gsigma", _p), ("gz2", _p)]


PROTOTYPES.update({
    # terms (host table), n_terms, out, workspace, workspace_len, ticket, stream
    "zs_logjoint_scalar_f32": [_p, _int, _p, _p, _i64, _p, _p],
    # terms, n_terms, gout (device scalar), gcoef, workspace, workspace_len, ticket, stream
    "zs_logjoint_scalar_bwd_f32": [_p, _int, _p, _p, _p, _i64, _p, _p],
    # terms, n_terms, seed, rng_state, rng_used, stream
    "zs_normal_sample_logprob_multi_f32": [_p, _int, _u64, _p, _p, _p],
    "zs_normal_sample_logprob_multi_bwd_f32": [_p, _int, _u64, _p, _p],
    # h, h_stride_k, w, out, K, B, n_in, n_out, relu, stream
    "zs_particle_linear_f32": [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _int, _p],
    # x, out, rows, cols, workspace, workspace_len, tickets, n_tickets, stream
    "zs_column_sum_f32": [_p, _p, _i64, _i64, _p, _i64, _p, _i64, _p],
    # g, y, act, gpre, gbias, rows, cols, workspace, workspace_len, tickets, n_tickets, stream
    "zs_dense_act_bwd_f32": [_p, _p, _int, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _p],
    # x, x_stride_k, layers (host table), n_layers, K, B, stream
    "zs_particle_mlp_f32": [_p, _i64, _p, _int, _i64, _i64, _p],
    # x, x_stride_k, layers, n_layers, gout, gx, K, B, workspace, workspace_len, tickets, stream
    "zs_particle_mlp_bwd_f32": [_p, _i64, _p, _int, _p, _p, _i64, _i64, _p, _i64, _p, _p],
    # pred, y, out, K, B, workspace (double), workspace_len, ticket, stream
    "zs_particle_rmse_f32": [_p, _p, _p, _i64, _i64, _p, _i64, _p, _p],
    # h, h_stride_k, w, out, gout, gh, gw, K, B, n_in, n_out, relu, workspace, workspace_len, tickets, stream
    "zs_particle_linear_bwd_f32": [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _int, _p, _i64, _p, _p],
})


# every compute entry point exists as name_f32 and name_f64 with the same argument list
PROTOTYPES.update({name[:-4] + "_f64": args for name, args in list(PROTOTYPES.items())})


class KernelLibrary(object):
    """A loaded shared object exporting the zs_* C ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError(
                "zhusuan (MI355X build): kernel library not found at %s -- run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C zhusuan-pytorch_amd/csrc`). "
                "There is no CPU fallback." % path)
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.cdll.zs_abi_version.restype = _int
        self.cdll.zs_abi_version.argtypes = []
        self.cdll.zs_error_string.restype = ctypes.c_char_p
        self.cdll.zs_error_string.argtypes = [_int]
        self.cdll.zs_build_info.restype = ctypes.c_char_p
        self.cdll.zs_build_info.argtypes = []
        got = self.cdll.zs_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError("zhusuan: %s has ABI version %d, expected %d" % (path, got, ABI_VERSION))
        self._fn = {}
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(self.cdll, name)  # AttributeError if a symbol is missing
            fn.restype = _int
            fn.argtypes = argtypes
            self._fn[name] = fn
        self.cdll.zs_normal_sample_pair_one_launch.restype = _int
        self.cdll.zs_normal_sample_pair_one_launch.argtypes = [_i64, _i64, _i64, _int]
        self.cdll.zs_prof_enable.restype = _int
        self.cdll.zs_prof_enable.argtypes = [_int]
        self.cdll.zs_prof_kernel_id.restype = _int
        self.cdll.zs_prof_kernel_id.argtypes = [ctypes.c_char_p]
        self.cdll.zs_prof_query.restype = _int
        self.cdll.zs_prof_query.argtypes = [_int] + [ctypes.POINTER(ctypes.c_double)] * 3 + [ctypes.POINTER(_i64)]

    def build_info(self):
        """The library's own one-line description (target, ABI, release / experiments build)."""
        return self.cdll.zs_build_info().decode()

    def prof_durations(self, entry_point):
        """Durations (ms, launch order) of the launches recorded for `entry_point`."""
        kid = self.cdll.zs_prof_kernel_id(entry_point.encode())
        if kid < 0:
            raise RuntimeError("unknown entry point %s" % entry_point)
        f = self.cdll.zs_prof_durations
        f.restype = _i64
        f.argtypes = [_int, ctypes.POINTER(ctypes.c_double), _i64]
        n = f(kid, None, 0)
        if n < 0:
            raise RuntimeError("zs_prof_durations failed with code %d" % n)
        buf = (ctypes.c_double * max(n, 1))()
        f(kid, buf, n)
        return [buf[i] for i in range(n)]

    def prof_enable(self, on):
        """Record the kernels' own start/stop events for every launch (include/zs_hip.h, zs_prof_*)."""
        rc = self.cdll.zs_prof_enable(1 if on else 0)
        if rc != 0:
            raise RuntimeError("zs_prof_enable failed with code %d" % rc)

    def prof_query(self, entry_point):
        """{'total_ms', 'min_ms', 'max_ms', 'count'} of the launches recorded for `entry_point`."""
        kid = self.cdll.zs_prof_kernel_id(entry_point.encode())
        if kid < 0:
            raise RuntimeError("unknown entry point %s" % entry_point)
        tot, mn, mx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = _i64()
        rc = self.cdll.zs_prof_query(kid, ctypes.byref(tot), ctypes.byref(mn), ctypes.byref(mx), ctypes.byref(n))
        if rc != 0:
            raise RuntimeError("zs_prof_query failed with code %d" % rc)
        return {"total_ms": tot.value, "min_ms": mn.value, "max_ms": mx.value, "count": n.value}

    def pair_draw_is_one_launch(self, K, M, D):
        """Whether two draws of a latent ([K, M] each, rows of D) fit ONE launch of the sampling kernel (zs_normal_sample_logprob_pair)."""
        return self.cdll.zs_normal_sample_pair_one_launch(K, M, D, 1) == 1

    def call(self, name, *args):
        rc = self._fn[name](*args)
        if rc != 0:
            msg = self.cdll.zs_error_string(rc)
            raise RuntimeError("%s failed with code %d: %s" % (name, rc, msg.decode() if msg else "?"))


_LIB = None          # the HIP library (lazy)


def lib():
    """The HIP kernel library; raises loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        _LIB = KernelLibrary(LIB_PATH)
    return _LIB


def require_device(*tensors):
    """Every kernel operand must live on a HIP device (torch device type 'cuda' on ROCm)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(
                "zhusuan (MI355X build): tensor on device '%s' -- this build runs the variational-inference "
                "hot path only as HIP kernels on an AMD GPU and has no CPU path. Move the model and data to "
                "the GPU (`.to('cuda')`)." % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("zhusuan: operands on different devices: %s vs %s" % (dev, t.device))
    return dev


def default_device():
    """Where parameters given as Python numbers / lists are placed when no device is named: the current HIP device
    (the host when there is none -- the first kernel call then raises in require_device)."""
    if not torch.cuda.is_available():
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device())


def resolve_device(device, *params):
    """Device of a distribution: the explicit `device` argument if given (parameters are moved there,
    as in the reference, normal.py:50); otherwise the device of the first tensor parameter; otherwise
    default_device().  (The reference's default is the CPU, which this build has no kernels for.)"""
    if device is not None:
        return torch.device(device) if not isinstance(device, torch.device) else device
    for p in params:
        if isinstance(p, torch.Tensor):
            return p.device
    return default_device()


# torch's own accessor of the current stream's raw handle: what torch.cuda.current_stream(dev).cuda_stream returns, without
# building a Stream object (3 us per kernel launch on the eager path)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_for(t):
    """Raw hipStream_t of torch's current stream on t's device (kernels are enqueued there)."""
    dev = t.device
    if dev.type != "cuda":
        return None
    if _raw_stream is not None:
        idx = dev.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(dev).cuda_stream


def ptr(t):
    """Device address of a tensor as a plain int (ctypes converts it for a void* parameter), None for an absent operand."""
    return None if t is None else t.data_ptr()

"""hipGraph capture of a whole training step (the launch-bound small shapes).

At the configuration sizes of the reference's examples a step is ~100-130 kernels of a few microseconds each, so the
launch path, not the GPU, sets the pace.  ``GraphedStep`` records the step once and replays it:

    rng = zhusuan.DeviceRNG(device, seed=0)                     # draws must come from device-resident state
    opt = torch.optim.Adam(model.parameters(), 1e-3, capturable=True)

    def compute():                                              # objective forward + backward
        rng.begin_step()
        for p in model.parameters():
            p.grad = None
        loss = model({'x': x})
        loss.backward()
        return loss.detach()

    step = zhusuan.GraphedStep(compute, opt.step, rng=rng)
    for _ in range(n):
        loss = step()                                           # one graph launch

A NEW MINIBATCH PER STEP.  A graph replays fixed addresses, so the observations must live in tensors that stay put: hand
them over as ``inputs`` and pass the step's data to the call -- it is copied into those tensors (device-to-device, or
host-to-device from pinned memory, on the replay's stream) right before the replay:

    obs = {'x': torch.empty(B, 784, device=dev)}               # the static input buffers `compute` reads
    step = zhusuan.GraphedStep(compute, opt.step, rng=rng, inputs=obs)
    for x_batch in loader:                                      # the reference's loop (examples/.../iwae.py:151-160)
        loss = step(x=x_batch)

(``tests/test_graph.py::test_graphed_training_over_a_stream_of_minibatches``: ten different batches, equal to eager training.)

With an ``exchange`` callable (the data-parallel all-reduce, see ``zhusuan.dataparallel``) the step is recorded as two
graphs around it -- graph A = ``compute``, then ``exchange`` launched eagerly (the RCCL collective stays outside the
graphs), then graph B = the optimizer.

Capture rules that this class takes care of (each one cost a crash while bench.py was written):
  * the eager warm-up runs on the capture side stream -- autograd's AccumulateGrad nodes remember the stream they were
    first used on, and a default-stream association breaks the capture;
  * ... and so must anything else that runs the model before the capture (a pre-tuning pass, a sanity step): the step's GEMMs
    launched once on the DEFAULT stream were enough for ``capture_end`` to crash (a segmentation fault, ROCm 7.2 / torch 2.10,
    found in round 6 with bench.py's one-tuning-run-per-job pass): use a side stream there too;
  * ``capture_error_mode='thread_local'``, so that unrelated threads (e.g. a data loader) may keep calling HIP;
  * nothing inside the step may copy from the host (``torch.as_tensor(python_scalar, device=...)`` does);
  * the warm-up steps are real optimizer steps; with ``restore=True`` parameters, optimizer state and the RNG state are
    put back afterwards, so that constructing a GraphedStep has no side effect on training.
"""
import torch

from . import _rng

__all__ = ['GraphedStep', 'GraphedStages']


def _optimizer_state_tensors(optimizer):
    out = []
    if optimizer is None:
        return out
    if hasattr(optimizer, 'state_tensors'):          # zhusuan.optim.FlatAdam: flat moments and device step counts
        return list(optimizer.state_tensors())
    for group in optimizer.param_groups:
        for p in group['params']:
            st = optimizer.state.get(p, None)
            if st:
                for k in sorted(st.keys()):
                    if isinstance(st[k], torch.Tensor):
                        out.append(((id(p), k), st[k]))
    return out


class GraphedStep(object):
    """
    :param compute: callable() -> detached 0-d loss; must (re)compute every gradient of the step.
    :param optimizer_step: callable() applying the update (e.g. ``opt.step`` of a ``capturable=True`` optimizer), or None.
    :param exchange: optional callable(loss) -> loss run EAGERLY between the two graphs (gradient all-reduce).
    :param rng: the ``zhusuan.DeviceRNG`` the step draws from (activated around warm-up, capture and every replay).
    :param warmup: eager steps on the capture stream before recording (>= 2: allocator and autograd settle).
    :param restore: undo the warm-up's effect on parameters / optimizer state / RNG state after recording.
    :param optimizer: the optimizer object, needed only for ``restore`` (its state tensors are reset in place).
    :param parameters: iterable of the tensors ``restore`` must put back (default: the optimizer's parameters).
    :param agree: several ranks: callable(ok) -> AND over the ranks (``zhusuan.dataparallel.all_ranks_agree``), called once
        after the capture; if any rank's capture failed every rank runs the step eagerly (``captured`` False, ``capture_error``).
    :param inputs: dict name -> the STATIC tensor ``compute`` reads that observation from; ``step(name=batch)`` copies a new
        minibatch into it before replaying (shapes and dtypes must match: a graph has no dynamic shapes).
    :param steps_per_replay: record N consecutive training steps into the ONE graph (single-process steps only: no
        ``exchange``).  Between two graph launches the GPU idles for ~8.5 us (profiles/r06_stream_links.txt): 13 % of the BNN
        example's 65 us step, 1 % of the IWAE step; N steps per replay pay it once.  Every sub-step draws fresh numbers (the RNG
        state and the optimizer's step counts live on the device) and reads the same static inputs -- several passes over one
        minibatch, or resident data; the returned loss is the last sub-step's.  One call = N steps.
    """

    def __init__(self, compute, optimizer_step=None, exchange=None, rng=None, warmup=3, restore=False, optimizer=None,
                 parameters=None, inputs=None, agree=None, steps_per_replay=1):
        self._inputs = dict(inputs) if inputs else {}
        self.steps_per_replay = int(steps_per_replay)
        if self.steps_per_replay < 1 or (self.steps_per_replay > 1 and exchange is not None):
            raise ValueError("steps_per_replay: a positive number; more than one step per graph only without an `exchange` "
                             "(a collective is launched eagerly between graphs)")
        self.captured, self.capture_error = True, None
        if not torch.cuda.is_available():
            raise RuntimeError("zhusuan.GraphedStep needs a HIP device: the MI355X build has no CPU path")
        self._compute, self._opt_step, self._exchange, self._rng = compute, optimizer_step, exchange, rng
        self.graphs = []
        if optimizer is None and optimizer_step is not None and hasattr(optimizer_step, '__self__') and \
                hasattr(optimizer_step.__self__, 'param_groups'):
            optimizer = optimizer_step.__self__
        self._optimizer = optimizer
        params = list(parameters) if parameters is not None else (
            [p for g in optimizer.param_groups for p in g['params']] if optimizer is not None else [])
        saved_params = saved_state = saved_rng = None
        if restore:
            saved_params = [p.detach().clone() for p in params]
            saved_state = dict((key, t.detach().clone()) for key, t in _optimizer_state_tensors(optimizer))
            saved_rng = (rng.state.clone(), rng._delta) if rng is not None else None

        def eager_step():
            loss = self._compute()
            if self._exchange is not None:
                loss = self._exchange(loss)
            if self._opt_step is not None:
                self._opt_step()
            return loss

        with self._rng_scope():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 2)):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()

            def record():
                if self._exchange is None:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        for _ in range(self.steps_per_replay):
                            self._static_loss = eager_step()
                    self.graphs = [g]
                else:
                    ga = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                        self._static_loss = self._compute()
                    self.graphs = [ga]
                    if self._opt_step is not None:
                        torch.cuda.synchronize()
                        gb = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                            self._opt_step()
                        self.graphs.append(gb)
            if agree is None:
                record()
            else:
                # several ranks record the same step (no collective is issued while recording): they meet ONCE afterwards and
                # replay graphs only if every rank's capture succeeded; otherwise every rank runs the step eagerly -- the same
                # `exchange`, hence the same collectives, in either mode
                ok = True
                try:
                    record()
                except Exception as e:                # noqa: BLE001
                    ok, self.capture_error = False, repr(e)
                if not agree(ok):
                    self.captured, self.graphs = False, []
                    if self.capture_error is None:
                        self.capture_error = "another rank's capture failed"
            self._eager_step = eager_step
            torch.cuda.synchronize()
        if restore:
            with torch.no_grad():
                for p, s in zip(params, saved_params):
                    p.copy_(s)
                for key, t in _optimizer_state_tensors(optimizer):
                    if key in saved_state:
                        t.copy_(saved_state[key])
                    else:
                        t.zero_()          # state created by the warm-up: back to its initial value
                if saved_rng is not None:
                    rng.state.copy_(saved_rng[0])
                    rng._delta = saved_rng[1]
            torch.cuda.synchronize()

    def _rng_scope(self):
        if self._rng is not None:
            return _rng.device_rng(self._rng)
        import contextlib
        return contextlib.nullcontext()

    def feed(self, **batches):
        """Copy new observations into the static input tensors (stream-ordered in front of the next replay)."""
        for name, value in batches.items():
            dst = self._inputs.get(name)
            if dst is None:
                raise KeyError("GraphedStep: no static input named %r (constructor argument `inputs`)" % name)
            value = torch.as_tensor(value)
            if tuple(value.shape) != tuple(dst.shape):
                raise ValueError("GraphedStep: input %r has shape %s, the captured step reads %s" % (
                    name, tuple(value.shape), tuple(dst.shape)))
            dst.copy_(value, non_blocking=True)

    def __call__(self, **batches):
        """Replay the recorded step (after copying ``batches`` into the static inputs); returns the (static) loss tensor."""
        if batches:
            self.feed(**batches)
        sync = getattr(self._optimizer, 'sync_hyperparameters', None)
        if sync is not None:
            sync()              # zhusuan.optim.FlatAdam: upload lr / betas / eps if the caller changed them (a 4-float compare)
        if not self.captured:
            with self._rng_scope():
                for _ in range(self.steps_per_replay - 1):
                    self._eager_step()
                return self._eager_step()
        if self._exchange is None:
            self.graphs[0].replay()
            return self._static_loss
        self.graphs[0].replay()
        loss = self._exchange(self._static_loss)
        if len(self.graphs) > 1:
            self.graphs[1].replay()
        return loss


class GraphedStages(object):
    """A training step recorded as SEVERAL hipGraphs with eagerly executed pieces in between -- the shape of a
    data-parallel step whose gradient all-reduce overlaps with backward:

        stages = [("graph", forward_and_decoder_backward),      # returns the detached loss
                  ("eager", launch_all_reduce_of_bucket_0),     # asynchronous: RCCL runs it on its own stream ...
                  ("graph", encoder_backward),                  # ... while this graph replays on the compute stream
                  ("eager", launch_bucket_1_and_wait_for_both),
                  ("graph", scale_and_optimizer_step)]
        step = zhusuan.GraphedStages(stages, rng=rng)
        loss = step()

    Collectives stay outside the graphs (an RCCL call captured into a graph ties the replay to one communicator state;
    launched eagerly it is an ordinary stream-ordered operation).  All graphs share one memory pool: later stages read
    what earlier ones produced (the autograd graph of the forward pass lives across the stage boundary, so the first
    backward stage must keep it: ``retain_graph=True``).  Warm-up, capture stream and thread-local capture mode as for
    ``GraphedStep``; so is ``restore`` (with ``optimizer`` / ``parameters``): the warm-up passes are real training steps,
    and with ``restore=True`` parameters, optimizer state and the RNG state are put back after the last capture, so that
    constructing the object has no side effect on training.  The value returned by the FIRST graph stage is the step's
    (static) loss tensor.

    ``agree``: with several ranks, a callable(ok: bool) -> bool that returns the AND of ``ok`` over the ranks (e.g.
    ``zhusuan.dataparallel.all_ranks_agree``).  It is called once after every capture attempt; when any rank's capture
    failed, every rank drops its graphs and the object runs the same stages eagerly from then on (``captured`` is False,
    ``capture_error`` says why): the ranks issue the same collectives in either mode and cannot wait for each other."""

    def __init__(self, stages, rng=None, warmup=3, restore=False, optimizer=None, parameters=None, agree=None):
        if not torch.cuda.is_available():
            raise RuntimeError("zhusuan.GraphedStages needs a HIP device: the MI355X build has no CPU path")
        self.captured, self.capture_error = True, None
        kinds = [k for k, _ in stages]
        if not stages or any(k not in ("graph", "eager") for k in kinds) or "graph" not in kinds:
            raise ValueError("stages: a list of ('graph' | 'eager', callable) with at least one graph stage")
        self._stages = [(k, f) for k, f in stages]
        self._rng = rng
        self._plan = []
        self._static_loss = None
        self._optimizer = optimizer
        params = list(parameters) if parameters is not None else (
            [p for g in optimizer.param_groups for p in g['params']] if optimizer is not None else [])
        if restore and not params:
            raise ValueError("GraphedStages(restore=True) needs `optimizer` or `parameters` to know what to put back")
        saved_params = saved_state = saved_rng = None
        if restore:
            saved_params = [p.detach().clone() for p in params]
            saved_state = dict((key, t.detach().clone()) for key, t in _optimizer_state_tensors(optimizer))
            saved_rng = (rng.state.clone(), rng._delta) if rng is not None else None

        def eager_pass():
            first = None
            for kind, fn in self._stages:
                out = fn()
                if kind == "graph" and first is None:
                    first = out
            self._static_loss = first
            return first
        self._eager_pass = eager_pass

        with self._rng_scope():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 2)):
                    eager_pass()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            pool = None
            for kind, fn in self._stages:
                if kind == "eager":
                    fn()                              # every rank runs the same sequence: collectives stay matched
                    self._plan.append(fn)
                    continue
                g = torch.cuda.CUDAGraph()
                if agree is None:
                    with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                        out = fn()
                else:
                    # Several ranks record the same step: a capture that fails on ONE rank must not let the others run on into
                    # the next eager stage's collective (they would wait there for ever).  After every capture attempt the ranks
                    # meet in `agree` -- the same number of calls on every rank, whatever happened -- and either all go on or
                    # all give the graphs up and run the SAME stages eagerly (same collectives, same sizes, same order).
                    ok, out = True, None
                    try:
                        with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                            out = fn()
                    except Exception as e:            # noqa: BLE001
                        ok, self.capture_error = False, repr(e)
                    if not agree(ok):
                        self.captured = False
                        if self.capture_error is None:
                            self.capture_error = "another rank's capture failed"
                        break
                if pool is None:
                    pool = g.pool()
                    self._static_loss = out
                self._plan.append(g)
                torch.cuda.synchronize()
            if not self.captured:
                torch.cuda.synchronize()              # (collectives an earlier eager stage started have landed on every rank)
                self._plan = [self._eager_pass]
                self._static_loss = None
        self.graphs = [g for g in self._plan if isinstance(g, torch.cuda.CUDAGraph)]
        if restore:
            with torch.no_grad():
                for p, sp in zip(params, saved_params):
                    p.copy_(sp)
                for key, t in _optimizer_state_tensors(optimizer):
                    if key in saved_state:
                        t.copy_(saved_state[key])
                    else:
                        t.zero_()
                if saved_rng is not None:
                    rng.state.copy_(saved_rng[0])
                    rng._delta = saved_rng[1]
            torch.cuda.synchronize()

    def _rng_scope(self):
        if self._rng is not None:
            return _rng.device_rng(self._rng)
        import contextlib
        return contextlib.nullcontext()

    def __call__(self):
        sync = getattr(self._optimizer, 'sync_hyperparameters', None)
        if sync is not None:
            sync()
        if not self.captured:
            with self._rng_scope():
                return self._eager_pass()
        for item in self._plan:
            if isinstance(item, torch.cuda.CUDAGraph):
                item.replay()
            else:
                item()
        return self._static_loss

"""Data-parallel minibatch shards: one process per GPU, model replicated, each rank evaluates the
objective on its own shard of the minibatch, and ONE all-reduce (RCCL over xGMI; backend "nccl" on
ROCm) of a single flat fp32 bucket [all gradients | local loss] averages the result.

The reference has no multi-device code at all; this module is new.  There is no collective on the data
path: samples, log-probs and the K-particle reductions of a datapoint never leave their GPU.
"""
import torch
import torch.distributed as dist

from . import _ops


def _bucket_dtype(params):
    """The one dtype of a flat gradient bucket: the parameters' (float32 in every benchmark config; float64 models work
    too).  Mixed-precision parameter sets are refused -- one all-reduce needs one element type."""
    dtypes = {p.dtype for p in params}
    if len(dtypes) != 1:
        raise TypeError("zhusuan.dataparallel: parameters of one bucket must share a dtype, got %s"
                        % sorted(str(d) for d in dtypes))
    return dtypes.pop()


SLICE_ALIGN_BYTES = 512      # what torch's caching allocator gives a fresh tensor


def _flat_layout(params, dtype, with_loss_slot):
    """(flat buffer, one view per parameter, index of the objective's slot or None).  Every slice starts on a 512-byte boundary,
    like a tensor of its own would: the slices are GEMM outputs when the backward pass writes gradients into the bucket
    (``direct=True``), and hipBLASLt's kernels store slower to a C that is only 16-byte aligned -- packed back to back, the
    IWAE step's three large weight-gradient GEMMs cost 2 % of the whole step (round 6, found as a process that was 17 us per
    step slower than its twin).  The padding (a few KB in 5.4 MB) is zero and stays zero under a SUM all-reduce."""
    item = torch.empty((), dtype=dtype).element_size()
    q = max(SLICE_ALIGN_BYTES // item, 1)
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off = (off + p.numel() + q - 1) // q * q
    loss_index = off if with_loss_slot else None
    flat = torch.zeros(off + (1 if with_loss_slot else 0), dtype=dtype, device=params[0].device)
    views = [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]
    return flat, views, loss_index


def _register_destinations(params, views):
    """Make every slice its parameter's gradient destination (zhusuan._ops: backward passes of this package's layers then
    write the gradient into the bucket themselves and ``_fill_flat`` finds it in place)."""
    for p, v in zip(params, views):
        _ops.register_grad_destination(p, v)


def _release_destinations(params, views):
    for p, v in zip(params, views):
        entry = _ops._GRAD_DEST.get(p.data_ptr())
        if entry is not None and entry[0] is v:
            _ops.unregister_grad_destination(p)


def _fill_flat(flat, params, views, tail):
    """Bring the gradients of `params` (and `tail`, a list of 1-element tensors: the objective) into `flat`, whose slices `views`
    shadow the parameters and whose LAST elements are the tail's slots.  Gradients that already live in their slice -- the
    backward pass wrote them there (gradient destinations), or autograd accumulated in place into last step's views because
    zero() was skipped -- are left where they are; the others are copied by one multi-tensor kernel; a parameter without a
    gradient gets zeros."""
    aliased = [p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, views)]
    dst = [v for p, v, a in zip(params, views, aliased) if not a and p.grad is not None]
    src = [p.grad for p, a in zip(params, aliased) if not a and p.grad is not None]
    if dst:
        torch._foreach_copy_(dst, src)
    for p, v in zip(params, views):
        if p.grad is None:
            v.zero_()
    n = flat.numel() - len(tail)
    for i, t in enumerate(tail):
        flat[n + i:n + i + 1].copy_(t.detach().reshape(1))


class DirectAllReduce(object):
    """An RCCL communicator of this job's own, driven without torch.distributed's per-collective bookkeeping:
    ``all_reduce_sum_(flat)`` enqueues ONE ``ncclAllReduce`` on the CURRENT stream and nothing else.

    Why: between two hipGraph launches every HIP event record costs ~5 us of idle GPU and every cross-stream hop 13-20 us
    (profiles/r06_stream_links.txt); ``dist.all_reduce`` records two events in its synchronous form (+10 us per call) and
    forks to / joins from its own stream in the asynchronous one (+35 us) -- with a 0.74 ms step that is the difference
    between 0.97 and 0.99 of the single-GPU rate before a byte has crossed xGMI.  The reduction itself is the same RCCL
    kernel over the same links.

    Construction is COLLECTIVE (every rank of ``group``, at the same point): rank 0's ``ncclUniqueId`` travels over the
    process group, every rank joins the communicator (``ncclCommInitRank``, bounded by ``timeout_s``: a rank whose set-up does
    not return gives up instead of hanging), one bucket-sized probe all-reduce checks the sum over the ranks element by element, and the ranks AGREE
    (``all_ranks_agree``) -- ``create`` returns a communicator on every rank or None on every rank, never a mixture; callers
    fall back to ``dist.all_reduce``.  Needs the "nccl" backend (RCCL on ROCm); RCCL's C API is bound in ``zhusuan/_rccl.py``."""

    def __init__(self, comm, world, device):
        self._comm, self.world, self.device = comm, world, device

    REQUIRED_BACKEND = "nccl"          # (tests set it to "gloo" to run the collective set-up logic with a stand-in for RCCL)

    @classmethod
    def create(cls, group=None, timeout_s=120.0):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != cls.REQUIRED_BACKEND:
            return None
        import threading
        from . import _rccl
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        device = torch.device("cuda", torch.cuda.current_device())
        box = {}
        try:
            uid = [_rccl.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)

            def join():
                try:
                    torch.cuda.set_device(device)          # (the current device is per thread)
                    comm = _rccl.comm_init_rank(world, uid[0], rank)
                    # the probe has the size of a gradient bucket (4 MB: the protocol / algorithm RCCL picks for the real call)
                    # and rank-dependent values at both ends: the SUM over the ranks must come out exactly, everywhere
                    probe = torch.full((1 << 20,), float(rank + 1), device=device)
                    probe[-1] = float(2 * rank + 1)
                    _rccl.all_reduce_sum_(probe, comm, torch.cuda.current_stream(device).cuda_stream)
                    torch.cuda.synchronize(device)
                    want_a, want_b = world * (world + 1) / 2.0, float(world * world)
                    exact = bool((probe[:-1] == want_a).all().item()) and float(probe[-1].item()) == want_b
                    box["sum"], box["comm"] = (want_a if exact else float("nan")), comm
                except Exception as e:                     # noqa: BLE001
                    box["error"] = repr(e)
            t = threading.Thread(target=join, name="zhusuan-rccl-init", daemon=True)
            t.start()
            t.join(timeout_s)
            ok = (not t.is_alive()) and "comm" in box and box.get("sum") == world * (world + 1) / 2.0
        except Exception as e:                             # noqa: BLE001
            box["error"], ok = repr(e), False
        if not all_ranks_agree(ok, group=group, device=device):
            cls.last_error = box.get("error") or ("set-up did not return within %.0f s" % timeout_s if "comm" not in box
                                                  else "another rank's set-up failed")
            return None
        return cls(box["comm"], world, device)

    last_error = None

    def all_reduce_sum_(self, flat):
        """In-place SUM over the ranks, enqueued on the current stream (stream-ordered like any kernel; capturable or not is
        RCCL's business -- this package launches it eagerly between graphs)."""
        from . import _hip, _rccl
        if not flat.is_contiguous() or flat.device != self.device:
            raise RuntimeError("DirectAllReduce: a contiguous tensor on %s expected" % (self.device,))
        _rccl.all_reduce_sum_(flat, self._comm, _hip.stream_for(flat))

    def close(self):
        """Destroy the communicator (collective in spirit: every rank, after its last all-reduce has completed)."""
        if self._comm is not None:
            from . import _rccl
            torch.cuda.synchronize(self.device)
            _rccl.comm_destroy(self._comm)
            self._comm = None


class GradientBucket(object):
    """One flat buffer [all gradients | objective] (in the parameters' dtype) for the single all-reduce of a step.

    Per step: autograd produces the gradients as usual; ``all_reduce_mean`` packs them (one multi-tensor copy
    into the persistent buffer, whose slices are 512-byte aligned), all-reduces the buffer, scales it by 1/world and re-points every
    ``p.grad`` at its slice of the buffer, so the optimizer reads the averaged gradients without an unpack
    copy.  With a single rank nothing is packed or sent at all.

    ``direct=True`` (default): the slices are registered as the parameters' gradient destinations -- the backward of
    ``zhusuan.Linear`` layers writes weight and bias gradients straight into the buffer (``zero()`` before every backward, so
    that autograd adopts them), and ``pack`` copies only what arrived elsewhere (torch.nn modules) and the objective.
    ``release()`` withdraws the registrations (a bucket that is dropped while its model lives on)."""

    def __init__(self, module, direct=True):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.flat, self.views, self.n_grad = _flat_layout(self.params, _bucket_dtype(self.params), True)      # n_grad: the objective's slot
        self.direct = bool(direct)
        if self.direct:
            _register_destinations(self.params, self.views)

    def release(self):
        _release_destinations(self.params, self.views)

    def zero(self):
        """Drop last step's gradients (autograd then writes fresh ones instead of accumulating)."""
        for p in self.params:
            p.grad = None

    def pack(self, local_loss):
        _fill_flat(self.flat, self.params, self.views, [local_loss])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self, local_loss, group=None):
        """Average gradients and the objective over the ranks; returns the global objective (0-d)."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return local_loss.detach()
        self.pack(local_loss)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.mul_(1.0 / dist.get_world_size(group))
        return self.flat[self.n_grad]

    def exchange(self, group=None, direct=None, always=False):
        """All-reduce (SUM) of the packed buffer on the CURRENT stream: through ``direct`` (a ``DirectAllReduce``: one RCCL
        call, no events) or ``dist.all_reduce``'s synchronous form.  The 1/world is left to the caller: ``scale()`` (a pass over
        the buffer) or ``zhusuan.optim.FlatAdam.step(grad_scale=bucket.grad_scale())`` (no pass), with ``loss()`` scaling the
        objective's slot on its own.  A no-op with one rank unless ``always`` (a measurement aid)."""
        active = dist.is_available() and dist.is_initialized()
        if not active or (dist.get_world_size(group) == 1 and not always):
            return
        if direct is not None:
            direct.all_reduce_sum_(self.flat)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def grad_scale(self, group=None):
        active = dist.is_available() and dist.is_initialized()
        return 1.0 / dist.get_world_size(group) if active else 1.0

    def scale(self, group=None):
        f = self.grad_scale(group)
        if f != 1.0:
            self.flat.mul_(f)

    def loss(self, scaled=False, group=None):
        """The objective's slot (the all-reduced SUM after ``exchange``; the mean when ``scaled`` or after ``scale()``)."""
        v = self.flat[self.n_grad]
        f = self.grad_scale(group) if scaled else 1.0
        return v if f == 1.0 else v * f

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


class OverlappedBuckets(object):
    """Eager-mode variant: the gradient all-reduce OVERLAPS with backward.

    Parameters are grouped, in the order backward produces their gradients (reverse registration order: the generator's
    last layer first, the encoder last), into `n_buckets` flat fp32 buffers of similar size.  A post-accumulate-grad hook
    per parameter counts arrivals; the moment a bucket is complete -- and every EARLIER bucket has left: collectives
    are issued in bucket order on every rank, whatever order the gradients arrive in (a parameter that receives no
    gradient on some ranks only would otherwise reorder the collectives there and hang or mis-sum) -- it is packed
    (one ``cat`` kernel) and its all-reduce is launched asynchronously, so it travels over xGMI while autograd is still
    computing the remaining gradients.
    ``finish()`` waits for the handles, applies 1/world and re-points every ``p.grad`` at its slice (no unpack copy).

        buckets = dataparallel.OverlappedBuckets(model, n_buckets=2)
        buckets.zero(); loss = model(obs); buckets.begin(loss); loss.backward()
        global_loss = buckets.finish(); optimizer.step()

    For steps recorded with ``zhusuan.GraphedStep`` use ``GradientBucket`` (hooks do not run on a graph replay)."""

    def __init__(self, module, n_buckets=2, group=None):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        self.group = group
        self._dtype = _bucket_dtype(params)
        order = list(reversed(params))
        total = sum(p.numel() for p in order)
        n_buckets = max(1, min(int(n_buckets), len(order)))
        target = (total + n_buckets - 1) // n_buckets
        self.buckets = []            # dicts: params, flat, views, pending, handle
        cur, cur_n = [], 0
        for p in order:
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= target and len(self.buckets) < n_buckets - 1:
                self.buckets.append(self._make(cur))
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(self._make(cur))
        self._loss_slot = self.buckets[0]["flat"][-1:]      # the first bucket to leave carries the objective
        self._owner = {}
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._owner[p] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._loss = None
        self._next = 0          # index of the next bucket allowed to leave

    def _make(self, params):
        flat, views, slot = _flat_layout(params, self._dtype, not self.buckets)
        return {"params": list(params), "flat": flat, "views": views, "n": slot, "pending": len(params), "handle": None,
                "launched": False}

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def zero(self):
        for b in self.buckets:
            b["pending"], b["handle"], b["launched"] = len(b["params"]), None, False
            for p in b["params"]:
                p.grad = None
        self._loss = None
        self._next = 0

    def begin(self, local_loss):
        """Hand over the local objective before ``backward()`` so that it rides in the first bucket."""
        self._loss = local_loss.detach()

    def _launch(self, bi):
        b = self.buckets[bi]
        tail = []
        if bi == 0:
            tail = [self._loss if self._loss is not None else torch.zeros((), dtype=self._dtype, device=b["flat"].device)]
        _fill_flat(b["flat"], b["params"], b["views"], tail)
        b["launched"] = True
        if self._active():
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        bi = self._owner[p]
        b = self.buckets[bi]
        b["pending"] -= 1
        # fixed issue order: a complete bucket waits until every earlier one has been launched
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] <= 0:
            if not self.buckets[self._next]["launched"]:
                self._launch(self._next)
            self._next += 1

    def finish(self):
        """Wait for the collectives, average, alias ``p.grad`` to the buckets; returns the global objective (0-d)."""
        world = dist.get_world_size(self.group) if self._active() else 1
        for bi, b in enumerate(self.buckets):     # index order = issue order on every rank
            if not b["launched"]:                 # parameters that received no gradient this step
                self._launch(bi)
        self._next = len(self.buckets)
        for bi, b in enumerate(self.buckets):
            if b["handle"] is not None:
                b["handle"].wait()
            if world > 1:
                b["flat"].mul_(1.0 / world)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
        return self._loss_slot[0]

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def nbytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)


class StagedBuckets(object):
    """Backward in STAGES, one flat bucket per stage, so that the all-reduce of an early stage's gradients travels
    while the later stages are still being computed -- also when the step is replayed from hipGraphs, where autograd
    hooks (``OverlappedBuckets``) do not run.

    ``stage_params`` lists the parameters in the order backward reaches them, e.g. for the VAE / IWAE objectives
    ``[generator.parameters(), variational.parameters()]``: the decoder's gradients are complete before the encoder's
    backward has started.  Per step (each numbered piece is one stage of ``zhusuan.GraphedStages``; run them back to
    back for an eager step):

        b = dataparallel.StagedBuckets([gen.parameters(), var.parameters()])
        b.zero(); loss = model(obs); b.backward_stage(loss, 0)     # 1 (graph)  decoder backward, bucket 0 filled
        b.launch(0)                                                # 2 (eager)  asynchronous all-reduce of bucket 0
        b.backward_stage(loss, 1)                                  # 3 (graph)  encoder backward, bucket 1 filled
        b.launch(1, overlap=False); b.wait()                       # 4 (eager)  bucket 1 on the current stream; join bucket 0
        b.scale(); optimizer.step()                                # 5 (graph)  1/world, update  (or b.scale(gradients=False);
                                                                   #            FlatAdam.step(grad_scale=b.grad_scale()): one pass less)
        global_loss = b.loss()

    The objective rides in bucket 0 (the first to leave).  Every ``p.grad`` ends up aliasing its slice of a bucket.
    ``direct=True`` (default): as for ``GradientBucket`` -- gradients produced by this package's layers are written into the
    buckets by the backward pass itself; only the objective (and gradients that arrived elsewhere) are copied.
    ``always_collective=True`` issues the all-reduces with ONE rank too (a measurement aid: what the path costs before a byte
    crosses xGMI; needs an initialised process group)."""

    def __init__(self, stage_params, group=None, direct=True, always_collective=False):
        self.group = group
        self.always_collective = bool(always_collective)
        self.stages = []
        for i, params in enumerate(stage_params):
            params = [p for p in params if p.requires_grad]
            if not params:
                raise ValueError("stage %d has no trainable parameters" % i)
            flat, views, slot = _flat_layout(params, _bucket_dtype(params), i == 0)
            self.stages.append({"params": params, "flat": flat, "views": views, "n": slot, "handle": None})      # "n": the objective's slot (stage 0)
        self._loss_factor = 1.0
        self._holds_sum = False          # the buckets hold the all-reduced SUM and the 1/world was left to the optimizer
        seen = set()
        for st in self.stages:
            for p in st["params"]:
                if id(p) in seen:
                    raise ValueError("a parameter appears in two stages")
                seen.add(id(p))
        self.direct = bool(direct)
        if self.direct:
            for st in self.stages:
                _register_destinations(st["params"], st["views"])

    def release(self):
        for st in self.stages:
            _release_destinations(st["params"], st["views"])

    def _world(self):
        active = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        return dist.get_world_size(self.group) if active else 1

    def zero(self):
        self._holds_sum = False
        for st in self.stages:
            st["handle"] = None
            for p in st["params"]:
                p.grad = None

    def backward_stage(self, loss, i, also=None, roots=None):
        """Gradients of stage i's parameters only (autograd prunes everything that does not lead to them), packed into
        bucket i.  All but the last stage keep the autograd graph alive for the stages that follow.

        ``also``: tensors on the boundary to LATER stages (e.g. the variational net's outputs, the distribution parameters of
        the latent): this pass delivers d loss / d t into ``t.grad`` for each of them as well, so that the objective's own
        backward (and everything between it and the boundary) runs ONCE, here, with every side in one launch.  The later
        stage then passes the same tensors as ``roots`` and starts there, seeded with those gradients, instead of at the loss
        (``loss`` is ignored).  Without them every stage walks down from the loss again."""
        st = self.stages[i]
        last = i == len(self.stages) - 1
        if self._holds_sum and any(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                                   for p, v in zip(st["params"], st["views"])):
            # after scale(gradients=False) every p.grad aliases the all-reduced SUM of the last step: accumulating this
            # step's gradients onto it would feed world-times-too-large stale values into the next all-reduce
            raise RuntimeError("zhusuan.dataparallel.StagedBuckets: the buckets hold last step's all-reduced SUM "
                               "(scale(gradients=False)); call zero() before the next backward")
        # (a backward of this package that serves several sides -- IW1, the dense layers -- asks the engine which of them THIS
        #  pass is after and launches only those: zhusuan._ops._pass_needs)
        also = [t for t in (also or []) if t.requires_grad]
        if roots is not None:
            roots = [t for t in roots if t.requires_grad and t.grad is not None]
            seeds = [t.grad for t in roots]
            torch.autograd.backward(roots, seeds, inputs=st["params"] + also, retain_graph=not last)
            for t in roots:          # (after the pass: a tensor that retains its gradient is handed its seed back)
                t.grad = None
        else:
            torch.autograd.backward(loss, inputs=st["params"] + also, retain_graph=not last)
        _fill_flat(st["flat"], st["params"], st["views"], [loss] if i == 0 else [])
        for p, v in zip(st["params"], st["views"]):
            p.grad = v

    def launch(self, i, overlap=True, direct=None):
        """The all-reduce of bucket i (a no-op with one rank).  ``overlap=True``: started without waiting for it -- it runs
        on the collective library's own stream beside whatever the compute stream does next (the later stages' backward);
        ``wait()`` joins it.  ``overlap=False``: the synchronous form, which torch enqueues on the CURRENT stream -- for a
        bucket whose result the very next kernel needs (the last stage's, in front of the update) that saves the two
        cross-stream event hops of the asynchronous form (measured: profiles/r06_stream_links.txt); nothing to join.
        ``direct`` (a ``DirectAllReduce``, with ``overlap=False``): one RCCL call on the current stream, no event records."""
        st = self.stages[i]
        if self._world() > 1 or (self.always_collective and dist.is_available() and dist.is_initialized()):
            if overlap:
                st["handle"] = dist.all_reduce(st["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            elif direct is not None:
                direct.all_reduce_sum_(st["flat"])
            else:
                dist.all_reduce(st["flat"], op=dist.ReduceOp.SUM, group=self.group)

    def wait(self):
        for st in self.stages:
            if st["handle"] is not None:
                st["handle"].wait()
                st["handle"] = None

    def scale(self, gradients=True):
        """1/world.  ``gradients=False``: the optimizer applies the factor itself while it reads the gradients
        (``zhusuan.optim.FlatAdam.step(grad_scale=buckets.grad_scale())``) -- no pass over the buckets; only the objective's
        slot is scaled, when ``loss()`` reads it.  In that mode every ``p.grad`` keeps aliasing the all-reduced SUM (world
        times the mean): anything else that reads ``p.grad`` afterwards (clipping, logging, another optimizer) must use
        ``mean_gradients()``, and the next backward needs ``zero()`` first (checked).  (``_loss_factor`` depends only on the
        world size, so setting it from inside a captured stage is the same on every replay.)"""
        w = self._world()
        self._loss_factor = 1.0
        if w > 1:
            if gradients:
                torch._foreach_mul_([st["flat"] for st in self.stages], 1.0 / w)
                self._holds_sum = False
            else:
                self._loss_factor = 1.0 / w
                self._holds_sum = True

    def mean_gradients(self):
        """[(parameter, mean gradient)] whatever the scaling mode: the bucket views themselves after ``scale()``, scaled
        copies after ``scale(gradients=False)``."""
        f = self.grad_scale() if self._holds_sum else 1.0
        return [(p, v if f == 1.0 else v * f) for st in self.stages for p, v in zip(st["params"], st["views"])]

    def grad_scale(self):
        """The factor that turns the all-reduced SUM into the mean (1 with a single rank)."""
        return 1.0 / self._world()

    def loss(self):
        v = self.stages[0]["flat"][self.stages[0]["n"]]
        f = self._loss_factor
        return v if f == 1.0 else v * f

    def loss_slot(self):
        """The objective's slot as it stands (no kernel): the mean after ``scale()``, the SUM over the ranks after
        ``scale(gradients=False)`` -- multiply by ``grad_scale()`` when you read it."""
        return self.stages[0]["flat"][self.stages[0]["n"]]

    def nbytes(self):
        return sum(st["flat"].numel() * st["flat"].element_size() for st in self.stages)


def shard_rows(x, rank, world_size):
    """Rows [rank*B/G, (rank+1)*B/G) of the minibatch (equal shards: the mean of the local means is then
    the global mean)."""
    B = x.shape[0]
    if B % world_size:
        raise ValueError("minibatch of %d rows does not split evenly over %d ranks" % (B, world_size))
    per = B // world_size
    return x[rank * per:(rank + 1) * per]


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank `src`'s weights."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in module.parameters():
            dist.broadcast(p.data, src=src, group=group)
        for b in module.buffers():
            dist.broadcast(b.data, src=src, group=group)


def all_ranks_agree(ok, group=None, device=None):
    """AND of ``ok`` over the ranks (all-reduce MIN of one integer): the meeting point where ranks decide TOGETHER between
    two ways of running a step -- e.g. ``zhusuan.GraphedStages(..., agree=all_ranks_agree)``: hipGraph replay only if every
    rank's capture succeeded.  A rank that changed its launch mode alone could issue another number or size of collectives
    than its peers, and the job would hang.  Every rank must call it the same number of times.  With one rank (or no
    process group) it returns ``bool(ok)``."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return bool(ok)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()) == 1)

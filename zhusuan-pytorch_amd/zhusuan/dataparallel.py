"""Data-parallel minibatch shards: one process per GPU, model replicated, each rank evaluates the
objective on its own shard of the minibatch, and ONE all-reduce (RCCL over xGMI; backend "nccl" on
ROCm) of a single flat fp32 bucket [all gradients | local loss] averages the result.

The reference has no multi-device code at all; this module is new.  There is no collective on the data
path: samples, log-probs and the K-particle reductions of a datapoint never leave their GPU.
"""
import torch
import torch.distributed as dist


class GradientBucket(object):
    """One flat fp32 buffer [all gradients | objective] for the single all-reduce of a step.

    Per step: autograd produces the gradients as usual; ``all_reduce_mean`` packs them (one ``cat`` kernel
    straight into the persistent buffer), all-reduces the buffer, scales it by 1/world and re-points every
    ``p.grad`` at its slice of the buffer, so the optimizer reads the averaged gradients without an unpack
    copy.  With a single rank nothing is packed or sent at all."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.n_grad = n

    def zero(self):
        """Drop last step's gradients (autograd then writes fresh ones instead of accumulating)."""
        for p in self.params:
            p.grad = None

    def pack(self, local_loss):
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
        parts.append(local_loss.detach().reshape(1).to(self.flat.dtype))
        torch.cat(parts, out=self.flat)
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

    def nbytes(self):
        return self.flat.numel() * 4


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

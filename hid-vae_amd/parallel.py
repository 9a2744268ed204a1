"""Data parallelism for the tokenizer step: one process per GPU, replicas of the model, per-rank item shards, ONE RCCL
all-reduce of a flat fp32 gradient buffer per optimizer step (torch.distributed backend "nccl" == RCCL on ROCm).

The reference gets the same semantics from accelerate -> DistributedDataParallel (reference train_hidvae.py:186-189,
630-632, 709): replicated parameters broadcast from rank 0, gradients averaged over ranks before optimizer.step().
What differs on purpose: ranks draw DIFFERENT, seeded batches (the reference's generator dataloader is
passed through accelerate untouched, so its ranks sample independently and unseeded), and k-means codebook init runs on
rank 0 and is broadcast (the reference runs it per rank after DDP wrapping, so its codebooks diverge across ranks).
BatchNorm statistics, InfoNCE negatives and p_unique_ids stay rank-local, as under DDP.

Nothing here touches a HIP kernel, so the logic is exercised on CPU with the gloo backend (tests/test_dp_cpu.py)."""
import torch
import torch.distributed as dist


class FlatGradBuffer:
    """One contiguous gradient buffer; every parameter's .grad is a view of it, so autograd accumulates in place and the
    whole gradient is exchanged in a single collective (4.6 MB for the core model, 29 MB with the tag heads)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.bind()

    def bind(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        self.flat.zero_()
        self.bind()

    def fold_in_stray_grads(self):
        """If someone replaced .grad (zero_grad(set_to_none=True) followed by backward), copy it back into the buffer."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v


class DataParallel:
    """world_size replicas.  allreduce() sums the flat gradients over ranks; the 1/world_size factor is returned so the
    optimizer can fold it into its update (HidvaeAdamW.grad_scale) instead of spending a kernel on it."""

    def __init__(self, module, grad_buffer: FlatGradBuffer, group=None):
        self.module, self.buf, self.group = module, grad_buffer, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def broadcast_parameters(self, src=0):
        """DDP-style start: every replica takes rank `src`'s parameters and buffers."""
        if self.world == 1:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src, group=self.group)

    def broadcast_codebooks(self, tables, src=0):
        """after rank-0 k-means (deliberate fix of reference Q9): 98 KB at 3x256, 512 KB at 4x1024"""
        if self.world == 1:
            return
        for t in tables:
            dist.broadcast(t.data, src, group=self.group)

    def allreduce(self, async_op=False):
        if self.world == 1:
            return 1.0, None
        work = dist.all_reduce(self.buf.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return 1.0 / self.world, work

    def shard_seed(self, base_seed=0):
        """per-rank sampler seed: ranks must not draw the same items"""
        return base_seed * 1000003 + self.rank


def shard_indices(n_items, rank, world):
    """contiguous shard of the resident item set owned by `rank` (items stay in that rank's HBM)"""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)

"""Data parallelism for the tokenizer step: one process per GPU, replicas of the model, per-rank item shards, ONE RCCL
all-reduce of a flat fp32 gradient buffer per optimizer step, on the package's own RCCL communicator (rccl.py; bootstrapped over the
torch.distributed group, which also carries the start-up broadcasts and is the fallback for the exchange).

The reference gets the same semantics from accelerate -> DistributedDataParallel (reference train_hidvae.py:186-189,
630-632, 709): replicated parameters broadcast from rank 0, gradients averaged over ranks before optimizer.step().
What differs on purpose: ranks draw DIFFERENT, seeded batches (the reference's generator dataloader is
passed through accelerate untouched, so its ranks sample independently and unseeded), and k-means codebook init runs on
rank 0 and is broadcast (the reference runs it per rank after DDP wrapping, so its codebooks diverge across ranks).
BatchNorm statistics, InfoNCE negatives and p_unique_ids stay rank-local, as under DDP.

Nothing here touches a HIP kernel, so the logic is exercised on CPU with the gloo backend (tests/test_dp_cpu.py)."""
import os
import sys

import torch
import torch.distributed as dist


class FlatGradBuffer:
    """One contiguous gradient buffer exchanged in a single collective (4.6 MB for the core model, 29 MB with the tag heads).

    Protocol per optimizer step:  zero() -> backward (one or more) -> seal() -> all-reduce / optimizer.
      * zero() launches nothing: it drops every .grad and marks every slot unwritten;
      * operators that know the protocol (ops.grad_sink: the Linear / codebook backward kernels) write a parameter's slot IN
        PLACE from their own launch (first write overwrites, later writes accumulate) and hand autograd no gradient;
      * seal() makes the buffer complete: slots nobody wrote are filled from the .grad autograd produced (one copy each) or
        zeroed, and every .grad becomes its slot, which is what the optimizer and the collective read.
    So the usual cost of a flat buffer (a memset plus one accumulate-add launch per parameter) is not paid."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            v = self.flat[off:off + n].view_as(p)
            self.views.append(v)
            p._hv_view = v
            p._hv_written = False
            off += n
        self._sealed = False
        self._sealed_upto = 0
        self.zero()

    def zero(self):
        for p in self.params:
            p.grad = None
            p._hv_written = False
        self._sealed = False
        self._sealed_upto = 0
        self._first_marks = None

    @torch.no_grad()
    def seal(self, upto=None):
        """complete the buffer; upto = n: only the first n parameters (the first bucket of an overlapped exchange) -- the rest is
        sealed by a later call"""
        if getattr(self, "_first_marks", None):
            self.check_first_bucket_untouched()  # (also hands the first bucket's .grad back)
        if self._sealed:
            return
        n = len(self.params) if upto is None else upto
        if upto is not None and upto < len(self.params):
            # first bucket of an overlapped exchange: it goes on the wire NOW, so every slot in it must already hold its final gradient.
            # A slot nobody produced means the backward took another path than the one the bucket was chosen for (HRqVae.dp_first_bucket
            # for another batch size / fuse_bottleneck setting): zero-filling it here and letting part 2 accumulate into it would race
            # with the collective and silently lose that gradient.
            missing = [i for i in range(self._sealed_upto, n) if not self.params[i]._hv_written and self.params[i].grad is None]
            if missing:
                raise RuntimeError(f"FlatGradBuffer.seal(upto={upto}): first-bucket slots {missing} were not produced by the first half of the "
                                   "split backward -- the bucket was chosen for a different forward path (batch size / fuse_bottleneck changed "
                                   "since HRqVae.dp_first_bucket); rebuild the optimizer's first_bucket or step without overlap")
            self._pending_marks = n
        for i in range(self._sealed_upto, n):
            p, v = self.params[i], self.views[i]
            if p._hv_written:
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.add_(p.grad)  # a second, protocol-unaware producer of the same parameter
            elif p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v
            p._hv_written = True  # a further backward before the next zero() accumulates
        self._sealed_upto = max(self._sealed_upto, n)
        self._sealed = self._sealed_upto == len(self.params)
        if getattr(self, "_pending_marks", None):
            # kernel writes through the C ABI are counted by ops.grad_sink; for autograd's own accumulation the first bucket's .grad is
            # taken away for the duration of the second half (a gradient that appears there is a late producer)
            self._first_marks = [(i, getattr(self.params[i], "_hv_nwrites", 0)) for i in range(self._pending_marks)]
            for i, _ in self._first_marks:
                self.params[i].grad = None
            self._pending_marks = None

    def check_first_bucket_untouched(self):
        """after the second half of a split backward: nothing may have written a first-bucket slot since seal(upto) -- its all-reduce
        is (or was) in flight on those bytes"""
        for i, nwr in getattr(self, "_first_marks", None) or []:
            p = self.params[i]
            late = getattr(p, "_hv_nwrites", 0) != nwr or p.grad is not None
            p.grad = self.views[i]
            if late:
                raise RuntimeError(f"FlatGradBuffer: first-bucket slot {i} was written during the second half of the split backward, while its "
                                   "all-reduce was in flight (the first bucket does not match the path the forward took)")
        self._first_marks = None

    def numel_of_first(self, n):
        return sum(p.numel() for p in self.params[:n])

    def unseal(self):
        """call before another backward of the same optimizer step (gradient accumulation)"""
        self._sealed = False
        self._sealed_upto = 0


class DataParallel:
    """world_size replicas.  allreduce() sums the flat gradients over ranks; the 1/world_size factor is returned so the
    optimizer can fold it into its update (HidvaeAdamW.grad_scale) instead of spending a kernel on it."""

    def __init__(self, module, grad_buffer: FlatGradBuffer, group=None):
        self.module, self.buf, self.group = module, grad_buffer, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.always = False  # rehearsal switch: issue the collective even in a one-rank group (bench.py --dist 1)
        self._comm = None    # rccl.Communicator once the first exchange asked for it; False: refused, torch.distributed carries it

    def communicator(self):
        """The package's own RCCL communicator for the gradient exchange (rccl.py: plain stream operations, no watchdog), created by
        the first exchange -- a collective call, every rank reaches it in the same step -- or None: host-side backends (gloo: the CPU
        tests), HIDVAE_DP_COMM=torch, or a communicator that could not be built (reported once; torch.distributed then carries the
        exchange, between graphs)."""
        if self._comm is None:
            self._comm = False
            want = os.environ.get("HIDVAE_DP_COMM", "rccl")
            if want != "torch" and dist.is_initialized() and self.buf.flat.is_cuda and dist.get_backend(self.group) == "nccl":
                try:
                    from .rccl import Communicator
                    c = Communicator(self.group, self.buf.flat.device)
                    c.warm_up()
                    self._comm = c
                except Exception as e:  # noqa: BLE001  any failure here has a working fallback
                    print(f"[hidvae] own RCCL communicator unavailable ({type(e).__name__}: {str(e).splitlines()[0]}); the gradient "
                          "exchange stays on torch.distributed, between graphs", file=sys.stderr)
        return self._comm or None

    def capturable(self):
        """may the gradient exchange be captured into a HIP graph?  Only on the package's own communicator.  torch.distributed's "nccl"
        collectives are stream operations too, but capturing one pulls the process group's internal stream into the capture, and its
        watchdog thread's next completion poll of ANY earlier eager collective then aborts the process (rccl.py); gloo's run on the
        host.  A one-rank group that issues no collective at all (`always` unset) has nothing to capture."""
        if not dist.is_initialized() or (self.world == 1 and not self.always):
            return False
        return self.communicator() is not None

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
        self.buf.seal()
        if self.world == 1 and not self.always:
            return 1.0, None
        c = self.communicator()
        if c is not None:
            if async_op:
                return 1.0 / self.world, c.all_reduce_sum_async_(self.buf.flat)
            c.all_reduce_sum_(self.buf.flat)
            return 1.0 / self.world, None
        work = dist.all_reduce(self.buf.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return 1.0 / self.world, work

    def allreduce_part(self, lo, hi):
        """asynchronous SUM of flat[lo:hi] over the ranks (one bucket of an overlapped exchange).  -> work handle or None; the caller
        has sealed those slots.  The collective is ordered after everything queued on the current stream so far and runs on the
        backend's own stream, i.e. beside what the caller queues next; work.wait() makes the current stream wait for it."""
        if (self.world == 1 and not self.always) or hi <= lo:
            return None
        c = self.communicator()
        if c is not None:
            return c.all_reduce_sum_async_(self.buf.flat[lo:hi])
        return dist.all_reduce(self.buf.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def shard_seed(self, base_seed=0):
        """per-rank sampler seed: ranks must not draw the same items"""
        return base_seed * 1000003 + self.rank


def shard_indices(n_items, rank, world):
    """contiguous shard of the resident item set owned by `rank` (items stay in that rank's HBM)"""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)

"""AdamW + CosineAnnealingLR fused into one multi-tensor HIP launch (replaces torch.optim.AdamW + the scheduler of
reference train_hidvae.py:533-563, 636-640, 762-766).  The step counter and the schedule live on the device, so a
whole training step (forward, backward, all-reduce, update) can be captured in a HIP graph and replayed."""
import ctypes
import math

import torch

from . import _C


class HidvaeAdamW(torch.optim.Optimizer):
    """Same param-group interface as torch.optim.AdamW (lr, weight_decay, betas, eps per group).
    cosine=(T_max, eta_min) enables CosineAnnealingLR, step_lr=(step_size, gamma) StepLR, each stepped once per optimizer step
    (the reference calls scheduler.step() right after optimizer.step()); both are evaluated on the device.

    flat_grads=True gives every parameter a view of ONE flat gradient buffer as its .grad (autograd then accumulates
    in place), which is what the data-parallel path all-reduces in a single RCCL collective."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, cosine=None, start_step=0,
                 flat_grads=False, step_lr=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        b = {tuple(g["betas"]) for g in self.param_groups}
        e = {g["eps"] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise ValueError("one (betas, eps) pair for all groups (the reference uses torch defaults everywhere)")
        self.betas, self.eps = b.pop(), e.pop()
        self.T_max, self.eta_min = (cosine if cosine is not None else (0, 0.0))
        # step_lr=(step_size, gamma): torch StepLR stepped once per optimizer step (reference train_hidvae.py:641-642)
        self.step_size, self.gamma = (step_lr if step_lr is not None else (0, 1.0))
        if cosine is not None and step_lr is not None:
            raise ValueError("one schedule at a time: cosine or step_lr")
        self._desc = None
        self._start_step = start_step
        self.flat_grads = flat_grads
        self.grad_scale = 1.0

    def _build(self):
        ps, lrs, wds = [], [], []
        for g in self.param_groups:
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("HidvaeAdamW needs contiguous float32 device parameters")
                ps.append(p)
                lrs.append(g["lr"])
                wds.append(g["weight_decay"])
        dev = ps[0].device
        self._params = ps
        total = sum(p.numel() for p in ps)
        self._m = torch.zeros(total, device=dev)
        self._v = torch.zeros(total, device=dev)
        self.grad_buffer = None
        if self.flat_grads:
            from .parallel import FlatGradBuffer
            self.grad_buffer = FlatGradBuffer(ps)
        self.flat_grad = self.grad_buffer.flat if self.grad_buffer is not None else None
        off, mptr, vptr = 0, [], []
        for p in ps:
            n = p.numel()
            mptr.append(self._m[off:off + n].data_ptr())
            vptr.append(self._v[off:off + n].data_ptr())
            off += n
        i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=dev)
        self._desc = dict(p=i64([p.data_ptr() for p in ps]), g_host=None, m=i64(mptr),
                          v=i64(vptr), numel=i64([p.numel() for p in ps]), lr=torch.tensor(lrs, dtype=torch.float32, device=dev),
                          wd=torch.tensor(wds, dtype=torch.float32, device=dev), n=len(ps), max_numel=max(p.numel() for p in ps))
        self._zero = torch.zeros(max(p.numel() for p in ps), device=dev)  # stands in for parameters without a gradient
        self.step_dev = torch.tensor([self._start_step], dtype=torch.int64, device=dev)
        self._desc["hyper"] = torch.zeros(3 * len(ps), dtype=torch.float32, device=dev)
        self._prepared = False

    def prepare(self):
        if self._desc is None:
            self._build()
        return self

    def _pending(self):
        ent = _C._PENDING_ADAMW.get(self.step_dev.device.index or 0)
        return ent is not None and ent[0] is self

    def _prepare_args(self):
        return (self._desc, self.step_dev, self.betas[0], self.betas[1], self.eta_min, self.T_max, self.step_size, self.gamma)

    def _prepare_step_async(self, defer=False):
        """schedule / bias-correction scalars + step counter for the coming step.  It depends on nothing, so at zero_grad() time
        it is only REGISTERED: the model's codebook_prepare launch of the coming forward carries it in a spare workgroup
        (_C.codebook_prepare), and step() launches it itself if no forward came by."""
        if defer:
            _C.defer_adamw_prepare(self.step_dev.device, self, self._prepare_args())
            return
        if _C.take_pending_adamw(self.step_dev.device, owner=self) is None and self._prepared:
            return
        _C.adamw_prepare(*self._prepare_args())
        self._prepared = True

    def zero_grad(self, set_to_none=True):
        self.prepare()
        if not self._prepared and not self._pending():
            self._prepare_step_async(defer=True)
        if self.flat_grads:
            self.grad_buffer.zero()
        else:
            for p in self._params:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        self.prepare()
        from .ops import join_side
        if not self._prepared:
            self._prepare_step_async()
        join_side()  # weight gradients and the step scalars may still be in flight on the helper stream
        if self.flat_grads:
            self.grad_buffer.seal()
        ptrs = []
        for p in self._params:
            g = p.grad
            if g is None:
                ptrs.append(self._zero.data_ptr())
            else:
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise RuntimeError("HidvaeAdamW: gradients must be contiguous float32")
                ptrs.append(g.data_ptr())
        self._desc["g_host"] = (ctypes.c_void_p * len(ptrs))(*ptrs)
        _C.adamw_step(self._desc, self.betas[0], self.betas[1], self.eps, self.grad_scale)
        self._prepared = False

    def flat_state(self):
        """Checkpoint payload (goes under the reference's "optimizer" key): moments, step, hyper-parameters."""
        self.prepare()
        return {"hidvae_m": self._m.detach().cpu(), "hidvae_v": self._v.detach().cpu(), "step": int(self.step_dev[0].item()),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "cosine": (self.T_max, self.eta_min), "step_lr": (self.step_size, self.gamma)}

    def load_flat_state(self, state):
        self.prepare()
        self._m.copy_(state["hidvae_m"].to(self._m.device))
        self._v.copy_(state["hidvae_v"].to(self._v.device))
        self.step_dev.fill_(int(state["step"]))

    def current_lr(self, group=0):
        t = int(self.step_dev[0].item())
        base = self.param_groups[group]["lr"]
        if self.T_max <= 0:
            return base * self.gamma ** (t // self.step_size) if self.step_size > 0 else base
        return self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * t / self.T_max)) / 2

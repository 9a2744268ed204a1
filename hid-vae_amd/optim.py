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
                 flat_grads=False, step_lr=None, first_bucket=None):
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
        # parameters whose gradients are final first in the backward (data-parallel overlap): they lead the flat gradient buffer, so
        # that bucket is one contiguous range that can go on the wire while the rest of the backward runs
        self.first_bucket = list(first_bucket) if first_bucket else []
        self.n_first = 0

    def _build(self):
        rows = []
        for g in self.param_groups:
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("HidvaeAdamW needs contiguous float32 parameters")  # (step() additionally needs them on the GPU)
                rows.append((p, g["lr"], g["weight_decay"]))
        lead = {id(p): i for i, p in enumerate(self.first_bucket)}
        first = sorted([r for r in rows if id(r[0]) in lead], key=lambda r: lead[id(r[0])])
        rows = first + [r for r in rows if id(r[0]) not in lead]
        self.n_first = len(first)
        ps, lrs, wds = [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows]
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
        # int64[2]: {optimizer steps taken, scheduler steps taken before step 0} (include/hidvae.h, hidvae_adamw_prepare)
        self.step_dev = torch.tensor([self._start_step, 0], dtype=torch.int64, device=dev)
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
        if not self._params[0].is_cuda:
            raise RuntimeError("HidvaeAdamW.step needs device parameters (the update is a HIP kernel; there is no CPU fallback)")
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
        early, self._early = sorted(getattr(self, "_early", [])), []
        if not early:
            self._desc["g_host"] = (ctypes.c_void_p * len(ptrs))(*ptrs)
            _C.adamw_step(self._desc, self.betas[0], self.betas[1], self.eps, self.grad_scale)
        else:  # some tensors already took this step's update (step_early): the rest, range by range
            lo = 0
            for a, b in early + [(len(ptrs), len(ptrs))]:
                if a > lo:
                    _C.adamw_step(self._desc, self.betas[0], self.betas[1], self.eps, self.grad_scale, lo=lo, hi=a,
                                  g_host=(ctypes.c_void_p * (a - lo))(*ptrs[lo:a]))
                lo = max(lo, b)
        self._prepared = False

    def tensor_ranges_of(self, params):
        """[(lo, hi), ...]: the positions of `params` in the optimizer's tensor tables, merged into contiguous ranges (None if one of
        them is not an optimized tensor)"""
        self.prepare()
        pos = {id(p): i for i, p in enumerate(self._params)}
        idx = sorted({pos.get(id(p), -1) for p in params if p.requires_grad})
        if not idx or idx[0] < 0:
            return None
        out, lo = [], idx[0]
        for a, b in zip(idx, idx[1:] + [None]):
            if b != a + 1:
                out.append((lo, a + 1))
                lo = b
        return out

    @torch.no_grad()
    def step_early(self, ranges):
        """This step's update for the tensors of `ranges` NOW, on the current stream -- their gradients are complete (a tag level's heads
        after that level's backward) while the rest of the backward is still running; step() then updates only the others.  Needs the
        step's scalars to be on their way already (the forward's codebook_prepare launch carried them) and per-parameter gradients
        (no flat buffer: under data parallelism the exchange comes first).  -> False if it did nothing."""
        if self._desc is None or not self._prepared or self.flat_grads or not ranges:
            return False
        for lo, hi in ranges:
            ptrs = []
            for p in self._params[lo:hi]:
                g = p.grad
                if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                    raise RuntimeError("HidvaeAdamW: gradients must be contiguous float32")
                ptrs.append(self._zero.data_ptr() if g is None else g.data_ptr())
            _C.adamw_step(self._desc, self.betas[0], self.betas[1], self.eps, self.grad_scale, lo=lo, hi=hi,
                          g_host=(ctypes.c_void_p * (hi - lo))(*ptrs))
        self._early = getattr(self, "_early", []) + list(ranges)
        return True

    # ---- checkpoint interchange: the "optimizer" entry of a checkpoint is a genuine torch.optim.AdamW state_dict (reference
    # train_hidvae.py:1166 writes optimizer.state_dict(), :625 feeds it to optimizer.load_state_dict) -----------------------------
    def _slots(self):
        """{id(param): (offset, numel)} into the flat moment buffers"""
        out, off = {}, 0
        for p in self._params:
            out[id(p)] = (off, p.numel())
            off += p.numel()
        return out

    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout: state[i] = {step, exp_avg, exp_avg_sq} per parameter (indexed over the groups in
        order), param_groups with torch's own hyper-parameter keys; `lr` is the CURRENT scheduled rate and `initial_lr` the base
        rate, as an attached torch scheduler leaves them (the reference resumes its scheduler with last_epoch = iter - 1, which
        needs `initial_lr`)."""
        self.prepare()
        step, offset = (int(v) for v in self.step_dev.tolist())
        m, v = self._m.detach().cpu(), self._v.detach().cpu()
        slots = self._slots()
        defaults = dict(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))]).defaults)  # this torch version's own key set
        state, groups, idx = {}, [], 0
        for gi, g in enumerate(self.param_groups):
            ids = []
            for p in g["params"]:
                if step > 0 and id(p) in slots:
                    o, n = slots[id(p)]
                    state[idx] = {"step": torch.tensor(float(step)), "exp_avg": m[o:o + n].view_as(p).clone(),
                                  "exp_avg_sq": v[o:o + n].view_as(p).clone()}
                ids.append(idx)
                idx += 1
            gg = {**defaults, **{k: val for k, val in g.items() if k != "params"}}
            gg["betas"] = tuple(gg["betas"])
            if self.T_max > 0 or self.step_size > 0:
                gg["initial_lr"] = g["lr"]
                gg["lr"] = self._lr_at(g["lr"], step + offset)
            gg["params"] = ids
            groups.append(gg)
        # (one extra top-level key, which torch.optim.Optimizer.load_state_dict ignores: where the LR schedule stands relative to Adam's
        #  step count -- non-zero after a resume without optimizer state; dropping it rewound the schedule on the NEXT resume)
        out = {"state": state, "param_groups": groups}
        if offset:
            out["hidvae_schedule_offset"] = offset
        return out

    def load_state_dict(self, sd):
        """Accepts torch.optim.AdamW.state_dict() (written here or by the reference's torch optimizer) and the round-1 private
        layout ("hidvae_m"/"hidvae_v"/"step").  Returns True when Adam's moments were restored.  When they were not (no state in
        the file), bias correction restarts from step 0 while the learning-rate schedule keeps its position (start_step)."""
        self.prepare()
        if "hidvae_m" in sd:
            # private flat layout.  The moments in the file are laid out in the order recorded under "order" (indices into the
            # parameters enumerated over the groups), or -- files written before the order was recorded (round 1) -- in group order.
            # This optimizer's own flat order may differ (a data-parallel first bucket leads the buffer), so the moments are
            # scattered parameter by parameter, never copied positionally.
            mine = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
            order = sd.get("order")
            src = list(range(len(mine))) if order is None else [int(i) for i in order]
            numels = sd.get("numels") or [mine[i].numel() for i in src]
            if sorted(src) != list(range(len(mine))) or [mine[i].numel() for i in src] != [int(n) for n in numels] \
                    or sum(int(n) for n in numels) != sd["hidvae_m"].numel():
                raise ValueError("hidvae flat optimizer state does not match this optimizer's parameters (count / sizes / order)")
            slots = self._slots()
            fm, fv = sd["hidvae_m"].reshape(-1), sd["hidvae_v"].reshape(-1)
            off = 0
            for i in src:
                o, n = slots[id(mine[i])]
                self._m[o:o + n].copy_(fm[off:off + n].to(self._m.device))
                self._v[o:o + n].copy_(fv[off:off + n].to(self._v.device))
                off += n
            self.step_dev[0] = int(sd["step"])
            self.step_dev[1] = int(sd.get("schedule_offset", 0))
            return True
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("optimizer state is neither a torch.optim.AdamW state_dict nor a hidvae flat state")
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters per group "
                             f"({[len(g['params']) for g in groups]} vs {[len(g['params']) for g in self.param_groups]})")
        slots = self._slots()
        steps, restored = [], 0
        self._m.zero_()
        self._v.zero_()
        for a, b in zip(groups, self.param_groups):
            for i, p in zip(a["params"], b["params"]):
                st = sd["state"].get(i)
                if st is None or id(p) not in slots:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i} has shape {tuple(st['exp_avg'].shape)}, parameter has {tuple(p.shape)}")
                o, n = slots[id(p)]
                self._m[o:o + n].copy_(st["exp_avg"].reshape(-1).to(self._m.device, torch.float32))
                self._v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1).to(self._v.device, torch.float32))
                steps.append(int(float(st["step"])))
                restored += 1
            for k in ("weight_decay", "betas", "eps"):
                if k in a:
                    b[k] = tuple(a[k]) if k == "betas" else a[k]
            b["lr"] = a.get("initial_lr", a.get("lr", b["lr"]))  # the group's BASE rate; the schedule is re-derived from the step
        self._refresh_hyper()
        prev = int(self.step_dev[0])
        if restored:
            self.step_dev[0] = max(steps)
            self.step_dev[1] = int(sd.get("hidvae_schedule_offset", 0))
        else:  # nothing to restore: a fresh Adam (bias correction from step 0) on a schedule that continues where the run was
            self.step_dev[0] = 0
            self.step_dev[1] = prev
        return bool(restored)

    def _refresh_hyper(self):
        of = {}
        for g in self.param_groups:
            for p in g["params"]:
                of[id(p)] = (g["lr"], g["weight_decay"])
        lrs = [of[id(p)][0] for p in self._params]  # (the flat order, which may lead with the first bucket)
        wds = [of[id(p)][1] for p in self._params]
        self._desc["lr"].copy_(torch.tensor(lrs, dtype=torch.float32))
        self._desc["wd"].copy_(torch.tensor(wds, dtype=torch.float32))
        b = {tuple(g["betas"]) for g in self.param_groups}
        e = {g["eps"] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise ValueError("one (betas, eps) pair for all groups")
        self.betas, self.eps = b.pop(), e.pop()

    def restart_without_state(self, schedule_position):
        """resume without optimizer state: zero moments, bias correction from step 0, schedule at `schedule_position`"""
        self.prepare()
        self._m.zero_()
        self._v.zero_()
        self.step_dev[0] = 0
        self.step_dev[1] = int(schedule_position)

    def flat_state(self):
        """round-1 private checkpoint payload, kept for files written then (load_state_dict reads both)"""
        self.prepare()
        mine = {id(p): i for i, p in enumerate(p for g in self.param_groups for p in g["params"] if p.requires_grad)}
        return {"hidvae_m": self._m.detach().cpu(), "hidvae_v": self._v.detach().cpu(), "step": int(self.step_dev[0].item()),
                "schedule_offset": int(self.step_dev[1].item()),
                "order": [mine[id(p)] for p in self._params], "numels": [p.numel() for p in self._params],
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "cosine": (self.T_max, self.eta_min), "step_lr": (self.step_size, self.gamma)}

    load_flat_state = load_state_dict

    def _lr_at(self, base, t):
        if self.T_max > 0:
            return self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * t / self.T_max)) / 2
        return base * self.gamma ** (t // self.step_size) if self.step_size > 0 else base

    def current_lr(self, group=0):
        t = int(self.step_dev.sum().item())  # scheduler steps taken: the optimizer's own + a resumed run's offset
        return self._lr_at(self.param_groups[group]["lr"], t)

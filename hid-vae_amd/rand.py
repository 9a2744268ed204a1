"""Sources of the random draws of the training step (dropout keep-masks, the mixup pairing and lambda, Gumbel noise).
The reference pulls them from torch's global generator (nn.Dropout, torch.randperm, Beta.sample: reference
modules/loss.py:144-147); a device stream can never reproduce a CPU stream, so the kernels take masks / pairings as
INPUTS and the provider is injectable -- parity tests inject the formula provider that also drove the reference when the
golden vectors were captured."""
import numpy as np
import torch


class DeviceRand:
    """Production provider: everything is drawn on the device with no host synchronisation (graph-capturable).

    A training step asks for ~25 dropout masks of fixed shapes and rates and one mixup pairing per level; drawn one by one that
    is ~70 tiny launches, each at the ~5 us launch floor.  So the provider learns the step's request list once (begin_step()
    .. the requests that follow) and from then on draws ALL keep-masks with ONE torch.bernoulli over a cached per-element
    keep-probability vector and hands out views; the pairings of all levels are computed together (mixup_all)."""

    def __init__(self, mixup_alpha=0.2):
        self.mixup_alpha = mixup_alpha
        self._beta = None
        self._plan = None      # [(numel, shape, p)] learned from the first step
        self._probs = None     # per-element keep probability of the arena
        self._record = None    # requests seen since begin_step() while learning
        self._arena = None
        self._cursor = 0
        self._mix = None       # per-level (partner, inverse, lam) of the current step

    # ---- step protocol (optional: without begin_step() every request is an individual draw)
    def begin_step(self, device):
        self._mix = None
        if self._record is not None and self._record:  # the previous step was the learning step
            self._plan = self._record
            self._probs = torch.cat([torch.full((n,), 1.0 - p, device=device, dtype=torch.float32) for n, _, p in self._plan])
        self._record = [] if self._plan is None else None
        if self._plan is not None:
            self._arena = torch.bernoulli(self._probs)  # one launch for every keep-mask of the step
            self._cursor, self._offset = 0, 0
        return self

    def dropout_keep(self, shape, p, device):
        shape = tuple(int(d) for d in shape)
        n = 1
        for d in shape:
            n *= d
        if self._record is not None:
            self._record.append((n, shape, float(p)))
        elif self._plan is not None and self._arena is not None and self._cursor < len(self._plan) \
                and self._plan[self._cursor] == (n, shape, float(p)):
            view = self._arena[self._offset:self._offset + n].view(shape)
            self._cursor += 1
            self._offset += n
            return view
        elif self._plan is not None:  # the step changed shape (another batch size, eval in between): learn again
            self._plan, self._arena, self._record = None, None, None
        return torch.empty(shape, device=device, dtype=torch.float32).bernoulli_(1.0 - p)

    def _lam(self, device, n):
        if self._beta is None:  # built once (a host->device scalar copy is not allowed inside a graph capture)
            a = torch.tensor(self.mixup_alpha, device=device)
            self._beta = torch.distributions.Beta(a, a)
        return self._beta.sample((n,)).to(torch.float32)

    def mixup_all(self, targets, device):
        """targets [B, L] -> per level (partner, inverse, lam), every op batched over the levels.
        Random pairing of the VALID rows (target >= 0) among themselves, as loss.py:144 does on the compacted rows:
        partner[b] = row mixed into b, inverse[partner[b]] = b, -1 on invalid rows; lam ~ Beta(alpha, alpha) (device)."""
        B, L = targets.shape
        if targets.is_cuda and B <= 4096 and targets.dtype == torch.int64 and targets.stride(1) == 1:
            # one launch (csrc/tagops.hip mixup_plan_kernel) on one torch.rand: the torch composition below is ~40 small launches
            from . import _C
            partner, inverse, lam = _C.mixup_plan(targets, torch.rand((L, B + 64), device=device, dtype=torch.float32), self.mixup_alpha)
            return [(partner[i], inverse[i], lam[i]) for i in range(L)]
        t = targets.t()  # [L, B]
        L, B = t.shape
        valid = t >= 0
        keys = torch.rand((L, B), device=device)
        keys = torch.where(valid, keys, torch.full_like(keys, 2.0))
        order = torch.argsort(keys, dim=1)  # valid rows first, in random order
        rank = torch.cumsum(valid.to(torch.int64), 1) - 1  # position of each valid row among the valid rows
        partner = torch.where(valid, torch.gather(order, 1, rank.clamp(min=0)), torch.full_like(rank, -1))
        inv_ext = torch.full((L, B + 1), -1, dtype=torch.int64, device=device)
        inv_ext.scatter_(1, torch.where(valid, partner, torch.full_like(partner, B)), torch.arange(B, device=device).expand(L, B))
        inverse = inv_ext[:, :B].contiguous()
        partner = partner.contiguous()
        lam = self._lam(device, L)
        return [(partner[i], inverse[i], lam[i]) for i in range(L)]

    def prepare_mixup(self, targets, device):
        self._mix = self.mixup_all(targets, device)

    def mixup_partner(self, target, device, level=None):
        if level is not None and self._mix is not None:
            return self._mix[level]
        return self.mixup_all(target.unsqueeze(1), device)[0]

    def gumbel_u(self, shape, device):
        return torch.rand(tuple(shape), device=device, dtype=torch.float32)


class InjectedRand:
    """Adapter around an object with dropout_keep(shape,p) / mixup(n) / gumbel_u(shape) returning CPU tensors (the oracle's
    FormulaRand): used by the parity tests; synchronises with the host, never used in production."""

    def __init__(self, source):
        self.source = source

    def dropout_keep(self, shape, p, device):
        return self.source.dropout_keep(shape, p).to(device)

    def mixup_partner(self, target, device, level=None):
        t = target.cpu().numpy()
        vidx = np.nonzero(t >= 0)[0]
        if len(vidx) <= 1:  # the reference draws nothing then (loss.py:123-125, :139)
            return None, None, None
        perm, lam = self.source.mixup(len(vidx))
        perm = perm.numpy()
        partner = np.full(t.shape[0], -1, dtype=np.int64)
        inverse = np.full(t.shape[0], -1, dtype=np.int64)
        partner[vidx] = vidx[perm]
        inverse[vidx[perm]] = vidx
        return (torch.from_numpy(partner).to(device), torch.from_numpy(inverse).to(device),
                torch.tensor(float(lam), dtype=torch.float32, device=device))

    def gumbel_u(self, shape, device):
        return self.source.gumbel_u(shape).to(device)

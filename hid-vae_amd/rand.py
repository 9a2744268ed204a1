"""Sources of the random draws of the training step (dropout keep-masks, the mixup pairing and lambda, Gumbel noise).
The reference pulls them from torch's global generator (nn.Dropout, torch.randperm, Beta.sample: reference
modules/loss.py:144-147); a device stream can never reproduce a CPU stream, so the kernels take masks / pairings as
INPUTS and the provider is injectable -- parity tests inject the formula provider that also drove the reference when the
golden vectors were captured."""
import numpy as np
import torch


class DeviceRand:
    """Production provider: everything is drawn on the device with no host synchronisation (graph-capturable)."""

    def __init__(self, mixup_alpha=0.2):
        self.mixup_alpha = mixup_alpha
        self._beta = None

    def dropout_keep(self, shape, p, device):
        return torch.empty(tuple(shape), device=device, dtype=torch.float32).bernoulli_(1.0 - p)

    def mixup_partner(self, target, device):
        """Random pairing of the VALID rows (target >= 0) among themselves, as loss.py:144 does on the compacted rows:
        partner[b] = row mixed into b, inverse[partner[b]] = b, -1 on invalid rows; lam ~ Beta(alpha, alpha) (device)."""
        B = target.shape[0]
        valid = target >= 0
        keys = torch.rand(B, device=device)
        keys = torch.where(valid, keys, torch.full_like(keys, 2.0))
        order = torch.argsort(keys)  # valid rows first, in random order
        rank = torch.cumsum(valid.to(torch.int64), 0) - 1  # position of each valid row among the valid rows
        partner = torch.where(valid, order[rank.clamp(min=0)], torch.full_like(rank, -1))
        inv_ext = torch.full((B + 1,), -1, dtype=torch.int64, device=device)
        inv_ext.scatter_(0, torch.where(valid, partner, torch.full_like(partner, B)), torch.arange(B, device=device))
        if self._beta is None:  # built once (a host->device scalar copy is not allowed inside a graph capture)
            a = torch.tensor(self.mixup_alpha, device=device)
            self._beta = torch.distributions.Beta(a, a)
        return partner, inv_ext[:B].contiguous(), self._beta.sample().to(torch.float32)

    def gumbel_u(self, shape, device):
        return torch.rand(tuple(shape), device=device, dtype=torch.float32)


class InjectedRand:
    """Adapter around an object with dropout_keep(shape,p) / mixup(n) / gumbel_u(shape) returning CPU tensors (the oracle's
    FormulaRand): used by the parity tests; synchronises with the host, never used in production."""

    def __init__(self, source):
        self.source = source

    def dropout_keep(self, shape, p, device):
        return self.source.dropout_keep(shape, p).to(device)

    def mixup_partner(self, target, device):
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

"""Sources of the random draws of the training step (dropout keep-masks, the mixup permutation and lambda,
Gumbel noise).  The reference pulls them from torch's global generator (nn.Dropout, torch.randperm,
Beta.sample: reference modules/loss.py:144-147); a device stream can never reproduce a CPU stream, so the
kernels take masks / permutations as INPUTS and the provider is injectable -- parity tests inject the
formula provider that also drove the reference when the golden vectors were captured."""
import torch


class DeviceRand:
    """Production provider: draws on the device from torch's HIP generator."""

    def __init__(self, mixup_alpha=0.2):
        self.mixup_alpha = mixup_alpha

    def dropout_keep(self, shape, p, device):
        return torch.empty(tuple(shape), device=device, dtype=torch.float32).bernoulli_(1.0 - p)

    def mixup(self, n, device):
        # lambda ~ Beta(a, a) drawn on the host generator as the reference does; one scalar, no device sync needed
        lam = torch.distributions.Beta(torch.tensor(self.mixup_alpha), torch.tensor(self.mixup_alpha)).sample()
        return torch.randperm(n, device=device), float(lam)

    def gumbel_u(self, shape, device):
        return torch.rand(tuple(shape), device=device, dtype=torch.float32)


class InjectedRand:
    """Adapter around any object with dropout_keep(shape,p) / mixup(n) / gumbel_u(shape) returning CPU tensors."""

    def __init__(self, source):
        self.source = source

    def dropout_keep(self, shape, p, device):
        return self.source.dropout_keep(shape, p).to(device)

    def mixup(self, n, device):
        perm, lam = self.source.mixup(n)
        return perm.to(device), float(lam)

    def gumbel_u(self, shape, device):
        return self.source.gumbel_u(shape).to(device)

"""Sources of the random draws of the training step (dropout keep-masks, the mixup pairing and lambda, Gumbel noise).
The reference pulls them from torch's global generator (nn.Dropout, torch.randperm, Beta.sample: reference
modules/loss.py:144-147); a device stream can never reproduce a CPU stream, so the kernels take masks / pairings as
INPUTS and the provider is injectable -- parity tests inject the formula provider that also drove the reference when the
golden vectors were captured."""
import contextlib

import numpy as np
import torch


class DeviceRand:
    """Production provider: nothing is drawn ahead of time and no mask tensor exists.

    Dropout: dropout_keep() hands out a _C.DropSpec -- (generator state, site number, drop probability) -- and the launch that
    produces the activation evaluates the keep decision per element itself (Philox4x32-10 keyed by the seed, counter = (element,
    site, step); csrc/common.h).  The backward reads the gate off the forward output, so the decision is never stored.  `step` lives
    on the device and is advanced by one tiny launch per forward pass (begin_step), which is what makes a REPLAYED HIP graph draw
    fresh masks; `site` counts the dropout layers of the pass in call order (the order in which the reference's forward reaches
    them, h_rqvae.py:147-186,327).  (Rounds 1-2 drew all keep-masks of a step with one torch.bernoulli into fp32 tensors: 30-56 us
    of a foreign kernel per step plus ~45 MB of mask traffic.)
    Mixup: one launch (hidvae_mixup_plan) computes the pairings and lambdas of all levels from the same generator."""

    def __init__(self, mixup_alpha=0.2, seed=None):
        self.mixup_alpha = mixup_alpha
        self.seed = seed
        self._beta = None
        self._state = {}       # device index -> int64[2] {seed, step}
        self._site = 0
        self._mix = None       # per-level (partner, inverse, lam) of the current step

    def state(self, device):
        """the generator state on `device` (created on first use -- before any graph capture: the eager warm-up steps see to that)"""
        device = torch.device(device)
        key = device.index if device.index is not None else torch.cuda.current_device()
        st = self._state.get(key)
        if st is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("DeviceRand: the generator state must exist before a graph capture (run one eager step first)")
            seed = self.seed if self.seed is not None else (torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
            st = self._state[key] = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        return st

    # ---- step protocol: once per forward pass (HRqVae.forward).  Without it the sites simply keep counting, which is as good:
    # (step, site) pairs never repeat either way
    def begin_step(self, device):
        from . import _C
        self._mix = None
        self._site = 0
        _C.rng_advance(self.state(device))
        return self

    def dropout_keep(self, shape, p, device):
        from . import _C
        spec = _C.DropSpec(self.state(device), self._site, p)
        self._site += 1
        return spec

    @contextlib.contextmanager
    def at_site(self, base):
        """number the dropout layers requested inside from `base` on: the tag heads give every unit of work (level i's projector, level
        i's predictor) a block of site numbers of its own, so the draws do not depend on the ORDER the units are issued in (the projectors
        run ahead of the encoder when they can: tagpath.tag_projectors_early).  Sites stay unique per step either way."""
        saved, self._site = self._site, int(base)
        try:
            yield self
        finally:
            self._site = max(saved, self._site)

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
        if targets.is_cuda and B <= 16384 and targets.dtype == torch.int64 and targets.stride(1) == 1:
            # one launch (csrc/tagops.hip mixup_plan_kernel), its uniforms drawn inside from the counter-based generator
            from . import _C
            partner, inverse, lam = _C.mixup_plan(targets, None, self.mixup_alpha, rng_state=self.state(device))
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


_DEFAULTS = {}


def default_rand(mixup_alpha=0.2):
    """the process-wide DeviceRand for callers that were handed none (a TagPredictor / TagPredictionLoss used on its own): ONE
    generator state, so successive calls keep drawing fresh masks (a fresh DeviceRand per call would restart at step 0 each time)"""
    r = _DEFAULTS.get(mixup_alpha)
    if r is None:
        r = _DEFAULTS[mixup_alpha] = DeviceRand(mixup_alpha)
    return r


class InjectedRand:
    """Adapter around an object with dropout_keep(shape,p) / mixup(n) / gumbel_u(shape) returning CPU tensors (the oracle's
    FormulaRand): used by the parity tests; synchronises with the host, never used in production."""

    def __init__(self, source):
        self.source = source

    @contextlib.contextmanager
    def at_site(self, base):  # the injected draws are numbered in CALL order (as the reference consumed them): nothing to scope
        yield self

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

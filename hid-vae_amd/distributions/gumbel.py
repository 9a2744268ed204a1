"""Gumbel-softmax sampling helpers (reference distributions/gumbel.py:8-42) on the HIP row kernels.

Inside HRqVae the GUMBEL_SOFTMAX branch runs fused around the MFMA GEMMs (hidvae_amd/gumbel_path.py); these are the
reference's stand-alone functions, kept callable with the same signatures.  The uniform draws come from torch's device
generator (plumbing); the arithmetic is hidvae_gumbel_noise / hidvae_gumbel_softmax_rows."""
from typing import Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _C


def sample_gumbel(shape: Tuple, device: torch.device, eps=1e-20) -> Tensor:
    """Sample from Gumbel(0, 1): -log(-log(U + eps) + eps)"""
    U = torch.rand(shape, device=device)
    return _C.gumbel_noise(U, eps)


def gumbel_softmax_sample(logits: Tensor, temperature: float, device: torch.device) -> Tensor:
    """softmax((logits + G) / temperature) over the last dim, G ~ Gumbel(0, 1)"""
    logits = logits.float().contiguous()
    U = torch.rand(logits.shape, device=device)
    return _C.gumbel_softmax_rows(logits, U, temperature)


class TemperatureScheduler:
    def __init__(self, t0: float, min_t: float, anneal_rate: float, step_size: int) -> None:
        self.t0, self.min_t, self.anneal_rate, self.step_size = t0, min_t, anneal_rate, step_size
        self.t = t0

    def update_t(self, iter):
        if iter % self.step_size == self.step_size - 1:
            self.t = np.maximum(self.t * np.exp(-self.anneal_rate * iter), self.min_t)

    def get_t(self, iter):
        self.update_t(iter)
        return self.t

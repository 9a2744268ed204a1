"""First-batch k-means codebook initialisation (reference init/kmeans.py:8-77) on the HIP kernels.

Same algorithm and stopping rule as the reference (Lloyd to convergence: max centroid shift < 1e-10, no iteration cap by
default); the seeding draws (`np.random.choice(B, k, replace=False)` for the start, `torch.randint` for empty clusters)
are injectable so a run can be replayed."""
from typing import NamedTuple, Optional

import numpy as np
import torch

from .. import _C


class KmeansOutput(NamedTuple):
    centroids: torch.Tensor
    assignment: torch.Tensor


class Kmeans:
    def __init__(self, k: int, max_iters: Optional[int] = None, stop_threshold: float = 1e-10, init_indices=None):
        self.k, self.iters, self.stop_threshold = k, max_iters, stop_threshold
        self.init_indices = init_indices
        self.centroids = self.assignment = None
        self.n_iter = 0

    def run(self, x: torch.Tensor) -> KmeansOutput:
        assert x.dim() == 2
        _C.check_embed_dim(x.shape[1])  # (kernels instantiated for 32 and 64; narrower multiples of 4 run zero-padded)
        x = x.detach().float().contiguous()
        N = x.shape[0]
        idx = self.init_indices if self.init_indices is not None else np.random.choice(N, self.k, replace=False)
        idx = torch.as_tensor(np.asarray(idx), dtype=torch.int64, device=x.device)
        cur = x[idx].contiguous()
        nxt = torch.empty_like(cur)
        assign = torch.empty((N,), dtype=torch.int32, device=x.device)
        scratch = torch.empty((self.k,), dtype=torch.float32, device=x.device)
        shift = torch.empty((), dtype=torch.float32, device=x.device)
        i = 0
        while self.iters is None or i < self.iters:
            reseed = torch.randint(0, N, (self.k,), device=x.device)
            _C.kmeans_iter(x, cur, assign, reseed, nxt, scratch, shift)
            cur, nxt = nxt, cur
            i += 1
            if float(shift) < self.stop_threshold:  # one host read per Lloyd iteration: this is start-up work
                break
        self.n_iter = i
        self.centroids, self.assignment = cur, assign.to(torch.int64)
        return KmeansOutput(centroids=self.centroids, assignment=self.assignment)


def kmeans_init_(tensor: torch.Tensor, x: torch.Tensor):
    assert tensor.dim() == 2
    assert x.dim() == 2
    with torch.no_grad():
        out = Kmeans(k=tensor.shape[0]).run(x)
        tensor.data.copy_(out.centroids)

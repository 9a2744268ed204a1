"""L2 normalisation (reference modules/normalize.py:7-18) on the HIP row kernels."""
from torch import nn

from ..ops import L2NormFn


def l2norm(x, dim=-1, eps=1e-12):
    if dim not in (-1, x.dim() - 1):
        raise NotImplementedError("l2norm: only the last dimension is normalised on the HIP path")
    flat = x.reshape(-1, x.shape[-1])
    if not flat.is_contiguous():
        flat = flat.contiguous()
    return L2NormFn.apply(flat, eps).reshape(x.shape)


class L2NormalizationLayer(nn.Module):
    def __init__(self, dim=-1, eps=1e-12):
        super().__init__()
        self.dim = dim
        self.eps = eps

    def forward(self, x):
        return l2norm(x, dim=self.dim, eps=self.eps)

"""MLP encoder / decoder (reference modules/encoder.py:7-36): bias-free Linear + SiLU stack with an optional
trailing L2 normalisation.  The nn.Sequential exists for the state-dict layout (`mlp.{0,2,4,6}.weight`) and for
`.parameters()`; forward() runs the whole stack through the fused MFMA GEMM + SiLU epilogue kernels."""
from typing import List

from torch import nn

from ..ops import L2NormFn, MLPBodyFn
from .normalize import L2NormalizationLayer


class MLP(nn.Module):
    def __init__(self, input_dim: int, hidden_dims: List[int], out_dim: int, dropout: float = 0.0, normalize: bool = False):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("MLP dropout is never enabled by the HiD-VAE configs; not built on the HIP path")
        self.input_dim, self.hidden_dims, self.out_dim, self.dropout = input_dim, hidden_dims, out_dim, dropout
        self.normalize = normalize
        widths = [input_dim] + list(hidden_dims) + [out_dim]
        layers = []
        for j in range(len(widths) - 1):
            layers.append(nn.Linear(widths[j], widths[j + 1], bias=False))
            if j != len(widths) - 2:
                layers.append(nn.SiLU())
        layers.append(L2NormalizationLayer() if normalize else nn.Identity())
        self.mlp = nn.Sequential(*layers)

    def weights(self):
        return [m.weight for m in self.mlp if isinstance(m, nn.Linear)]

    def body(self, x):
        """The Linear/SiLU stack without the trailing normalisation (the fused step folds that into the
        RQ prologue / reconstruction kernel)."""
        assert x.shape[-1] == self.input_dim, f"Invalid input dim: Expected {self.input_dim}, found {x.shape[-1]}"
        flat = x.reshape(-1, self.input_dim)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        return MLPBodyFn.apply(flat, *self.weights()).reshape(x.shape[:-1] + (self.out_dim,))

    def forward(self, x):
        y = self.body(x)
        if self.normalize:
            y = L2NormFn.apply(y.reshape(-1, self.out_dim), 1e-12).reshape(y.shape)
        return y

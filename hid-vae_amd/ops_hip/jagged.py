"""Differentiable padded -> jagged conversion (reference ops/triton/jagged.py:9-84), the stage-2 transformer's only custom
kernel -- Triton there, one HIP row-copy launch each way here (csrc/jagged.hip).  Same two entry points:
    padded_to_jagged_tensor(x [B,N,D], lengths [B], max_len) -> torch jagged NestedTensor
    jagged_to_flattened_tensor(nested) -> values [sum(lengths), D]
The reference assembles the NestedTensor by poking private fields; torch.nested.nested_tensor_from_jagged builds the same object
from (values, offsets)."""
import torch
from torch import Tensor
from torch.autograd import Function

from .. import _C


class PaddedToJaggedValues(Function):
    """values [sum(lengths), D] of the valid rows; backward scatters the gradient back and zero-fills the padding"""

    @staticmethod
    def forward(ctx, x: Tensor, offsets: Tensor, total: int):
        assert x.dim() == 3
        ctx.shape = (x.shape[0], x.shape[1])
        ctx.save_for_backward(offsets)
        return _C.padded_to_jagged(x if x.stride(2) == 1 else x.contiguous(), offsets, total)

    @staticmethod
    def backward(ctx, grad_values):
        (offsets,) = ctx.saved_tensors
        B, N = ctx.shape
        return _C.jagged_to_padded(grad_values.contiguous(), offsets, B, N), None, None


def padded_to_jagged_tensor(x: Tensor, lengths: Tensor, max_len: int):
    """`max_len` is accepted for signature parity (the reference only uses it for its backward mask)."""
    assert x.dim() == 3 and lengths.shape[0] == x.shape[0]
    lengths = lengths.to(torch.int64).clamp(max=x.shape[1])
    offsets = torch.zeros(lengths.shape[0] + 1, dtype=torch.int64, device=lengths.device)
    offsets[1:] = lengths.cumsum(0)
    total = int(offsets[-1])  # (the reference sizes its output the same way: one host read)
    values = PaddedToJaggedValues.apply(x, offsets, total)
    return torch.nested.nested_tensor_from_jagged(values, offsets=offsets)


def jagged_to_flattened_tensor(x) -> Tensor:
    return x.values()

"""HIP replacements for the reference's ops/ package (ops/triton/jagged.py).  install_dropin() aliases `ops`, `ops.triton` and
`ops.triton.jagged` here, so `from ops.triton.jagged import padded_to_jagged_tensor` in the stage-2 code resolves to HIP."""
import sys

from . import jagged  # noqa: F401

triton = sys.modules[__name__]  # `ops.triton` is this package too

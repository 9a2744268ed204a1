"""TEST INFRASTRUCTURE ONLY (oracle/): deterministic, platform-independent tensor fills.

Every golden fixture, parity test, smoke() and the bench's cpu_baseline leg builds its
inputs and weights from these integer-hash formulas, so the fixtures under tests/golden/
hold *outputs only* and the same inputs can be regenerated bit-for-bit on the GPU box
(where /root/reference does not exist).  Nothing in the product path imports this file.

The hash is splitmix64 on uint64 lanes (pure integer arithmetic => identical everywhere);
values are 24-bit fractions, exactly representable in fp32.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_G = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def u01(n, seed):
    """n values in [0,1) as float64 holding exact 24-bit fractions."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        z = i * _G + (np.uint64(seed) + np.uint64(1)) * np.uint64(0xD1B54A32D192ED03)
    z = _mix(_mix(z))
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def uniform(shape, seed, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * u01(n, seed)).astype(np.float32).reshape(shape)


def gauss(shape, seed):
    """Irwin-Hall(4) approximate normal, unit variance; exact in float64 before the cast."""
    n = int(np.prod(shape))
    s = u01(n, 4 * seed + 11) + u01(n, 4 * seed + 12) + u01(n, 4 * seed + 13) + u01(n, 4 * seed + 14)
    return ((s - 2.0) * np.sqrt(3.0)).astype(np.float32).reshape(shape)


def unit_rows(shape, seed):
    """Rows of gauss() scaled to unit L2 norm (sequential float64 accumulation => order-stable)."""
    g = gauss(shape, seed).astype(np.float64)
    sq = np.zeros(g.shape[:-1], dtype=np.float64)
    for j in range(g.shape[-1]):  # fixed order, no pairwise/SIMD dependence
        sq = sq + g[..., j] * g[..., j]
    return (g / np.sqrt(sq)[..., None]).astype(np.float32)


def ints(shape, seed, hi):
    n = int(np.prod(shape))
    return np.minimum((u01(n, seed) * hi).astype(np.int64), hi - 1).reshape(shape)


def perm(n, seed):
    return np.argsort(u01(n, seed), kind="stable").astype(np.int64)


def keep_mask(shape, seed, p_drop):
    """Dropout keep-mask (1 = keep)."""
    n = int(np.prod(shape))
    return (u01(n, seed) >= p_drop).astype(np.float32).reshape(shape)

"""Deterministic synthetic data shared by the golden-vector generator and the tests.

Integer hashing (splitmix64) in numpy uint64 arithmetic -> exactly the same floats on every machine, numpy or
torch version, so fixtures only need to store the reference's OUTPUTS.
"""
import numpy as np


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array of `shape`, values lo + (hi-lo) * k / 2^24 with k a 24-bit hash of (seed, index)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        bits = _splitmix64(idx) >> np.uint64(40)
    u = bits.astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def gaussish(shape, seed):
    """Roughly N(0,1): sum of four uniforms, rescaled.  Exactly reproducible."""
    s = sum(uniform(shape, seed * 4 + k).astype(np.float64) for k in range(4))
    return (s * (3.0 ** 0.5) / 2.0).astype(np.float32)


def name_seed(name):
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF

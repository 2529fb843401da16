"""TEST INFRASTRUCTURE (oracle/): bit-portable tensor generator.

Counter-based splitmix64 hash of (name, element index) -> float32 uniform.  Pure numpy integer arithmetic, so this
container and the GPU box regenerate identical weights/inputs without shipping megabytes and without depending on
the torch RNG of either box.  Not part of the product path.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(name, shape, lo=0.0, hi=1.0, seed=2021):
    """float32 array of `shape`, U[lo,hi), determined only by (name, seed, shape)"""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64((((zlib.crc32(name.encode()) & 0xFFFFFFFF) * 0x100000001B3) + int(seed)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (key << np.uint64(20))
        bits = _splitmix64(_splitmix64(idx) ^ key)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 random bits -> exact in float32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(name, shape, std=1.0, seed=2021):
    """zero-mean, unit-variance-ish (sum of 4 uniforms), float32"""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += uniform("%s#%d" % (name, k), shape, -1.0, 1.0, seed)
    return (acc * (std * np.sqrt(3.0 / 4.0))).astype(np.float32)

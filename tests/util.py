"""Shared test helpers: seeded synthetic columns."""
import numpy as np

P = 2**251 + 17 * 2**192 + 1
MASK64 = (1 << 64) - 1
SEED0 = 0x53414E4453544F52


from sandstorm_amd.examples import random_column, splitmix64_stream  # noqa: E402,F401  (moved: bench.py and smoke() use them too)


def felt_int(limbs):
    return sum(int(limbs[k]) << (64 * k) for k in range(4))


def int_limbs(v):
    return np.array([(v >> (64 * k)) & MASK64 for k in range(4)], dtype=np.uint64)

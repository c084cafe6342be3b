"""Shared test helpers: seeded synthetic columns."""
import numpy as np

P = 2**251 + 17 * 2**192 + 1
MASK64 = (1 << 64) - 1
SEED0 = 0x53414E4453544F52


def splitmix64_stream(seed, count):
    """vectorised SplitMix64: `count` successive outputs for `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & MASK64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_column(n, column_index=0, seed=SEED0):
    """(n,4) uint64: uniformly random-ish Montgomery images < p.

    4 SplitMix64 draws per element, top limb masked to 59 bits and elements
    >= p folded by clearing the top bits (valid Montgomery images are just
    integers < p, so any value < p is a legal element)."""
    raw = splitmix64_stream(seed ^ column_index, 4 * n).reshape(n, 4).copy()
    raw[:, 3] &= np.uint64((1 << 59) - 1)          # < 2^251 < p
    return raw


def felt_int(limbs):
    return sum(int(limbs[k]) << (64 * k) for k in range(4))


def int_limbs(v):
    return np.array([(v >> (64 * k)) & MASK64 for k in range(4)], dtype=np.uint64)

"""Pure-Python big-integer restatements used to cross-check the C oracle on
small cases (test infrastructure).  Definitions follow SURVEY.md §8(a)."""
import hashlib

P = 2**251 + 17 * 2**192 + 1
BETA = 3141592653589793238462643383279502884197169399375105820974944592307816406665


def root_of_unity(n):
    return pow(3, (P - 1) // n, P)


def ec_double(pt):
    x, y = pt
    lam = (3 * x * x + 1) * pow(2 * y, -1, P) % P
    x3 = (lam * lam - 2 * x) % P
    return x3, (lam * (x - x3) - y) % P


def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        return ec_double(p1) if y1 == y2 else None
    lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def make_pedersen(points):
    """points: the five StarkWare constants P0..P4 (tests/golden/pedersen.json)."""
    p0, p1, p2, p3, p4 = points

    def pedersen(a, b):
        acc = p0
        for val, lo, hi in ((a, p1, p2), (b, p3, p4)):
            for bits, base in ((val & (2**248 - 1), lo), (val >> 248, hi)):
                pt = base
                while bits:
                    if bits & 1:
                        acc = ec_add(acc, pt)
                    pt = ec_double(pt)
                    bits >>= 1
        return acc[0]
    return pedersen


def blake2s(b):
    return hashlib.blake2s(b).digest()


def mask_blake(d):
    return bytes(12) + d[12:]


def mask_keccak(d):
    return d[:20] + bytes(12)


def interpolate_eval(xs, ys, t):
    """value at t of the Lagrange interpolant through (xs, ys)."""
    acc = 0
    for i, (xi, yi) in enumerate(zip(xs, ys)):
        num, den = 1, 1
        for j, xj in enumerate(xs):
            if i != j:
                num = num * (t - xj) % P
                den = den * (xi - xj) % P
        acc = (acc + yi * num * pow(den, -1, P)) % P
    return acc

"""Host-side Fiat-Shamir coins, byte-compatible with the reference's
SolidityVerifierPublicCoin (crypto/src/public_coin/solidity.rs:36-161) and
CairoVerifierPublicCoin (crypto/src/public_coin/cairo.rs:42-174).

The coin is host code in the reference too (it only serialises the
GPU stages; tens of 32-byte messages per proof).  Proof-of-work grinding is the
one coin operation on the GPU (Context.pow_grind).  Product code: this module
never touches the oracle.
"""
import hashlib

import numpy as np

from . import backend as be

P = be.P
_RINV = pow(2**256, -1, P)
_MASK64 = (1 << 64) - 1

_KRC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
        0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
        0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
        0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
        0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KROT = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]


def _keccak_f(s):
    for rc in _KRC:
        c = [s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ (((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)) & _MASK64) for x in range(5)]
        s = [s[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                v, r = s[x + 5 * y], _KROT[x + 5 * y]
                b[y + 5 * ((2 * x + 3 * y) % 5)] = ((v << r) | (v >> (64 - r))) & _MASK64 if r else v
        s = [b[i] ^ (~b[(i % 5 + 1) % 5 + 5 * (i // 5)] & _MASK64 & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) for i in range(25)]
        s[0] ^= rc
    return s


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    msg.extend(b"\x00" * (-len(msg) % rate))
    msg[-1] |= 0x80
    s = [0] * 25
    for off in range(0, len(msg), rate):
        for i in range(17):
            s[i] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        s = _keccak_f(s)
    return b"".join(v.to_bytes(8, "little") for v in s[:4])


def blake2s256(data: bytes) -> bytes:
    return hashlib.blake2s(data).digest()


def limbs_to_int(limbs):
    return sum(int(limbs[k]) << (64 * k) for k in range(4))


def int_to_limbs(v):
    return np.array([(v >> (64 * k)) & _MASK64 for k in range(4)], dtype=np.uint64)


def mont_be_bytes(limbs) -> bytes:
    """to_montgomery(e).to_be_bytes::<32>() (crypto/src/utils.rs:14-22)"""
    return limbs_to_int(limbs).to_bytes(32, "big")


def canonical(limbs) -> int:
    return limbs_to_int(limbs) * _RINV % P


def pedersen_hash_elements(elems) -> int:
    """PedersenHashFn::hash_elements (crypto/src/hash/pedersen.rs:65-76) -> canonical int"""
    cur = np.zeros(4, dtype=np.uint64)
    n = 0
    for e in elems:
        cur = be.pedersen_hash_host(cur, e)
        n += 1
    return canonical(be.pedersen_hash_host(cur, be.felt(n)))


class PublicCoin:
    """kind: be.COIN_SOLIDITY (Keccak-256) or be.COIN_CAIRO (Blake2s-256)."""

    def __init__(self, kind, digest: bytes):
        assert len(digest) == 32
        self.kind, self.digest, self.counter = kind, bytes(digest), 0
        self._h = keccak256 if kind == be.COIN_SOLIDITY else blake2s256

    def reseed_with_bytes(self, data: bytes):
        d = (int.from_bytes(self.digest, "big") + 1) % (1 << 256)
        self.digest = self._h(d.to_bytes(32, "big") + bytes(data))
        self.counter = 0

    def reseed_with_digest(self, digest: bytes):
        self.reseed_with_bytes(digest)

    def reseed_with_field_elements(self, elems):
        if self.kind == be.COIN_SOLIDITY:            # solidity.rs:66-71: one reseed per element
            for e in elems:
                self.reseed_with_bytes(mont_be_bytes(e))
        else:                                        # cairo.rs:76-80: Pedersen chain, canonical bytes
            self.reseed_with_bytes(pedersen_hash_elements(elems).to_bytes(32, "big"))

    def reseed_with_field_element_vector(self, elems):
        self.reseed_with_bytes(b"".join(mont_be_bytes(e) for e in elems))

    def reseed_with_int(self, v: int):
        self.reseed_with_bytes(int(v).to_bytes(8, "big"))

    def _draw_bytes(self) -> bytes:
        out = self._h(self.digest + self.counter.to_bytes(32, "big"))
        self.counter += 1
        return out

    def draw(self):
        """-> Montgomery limbs (uint64[4]); from_montgomery(raw) = raw mod p as the limb image"""
        bound = 31 * P
        while True:
            raw = int.from_bytes(self._draw_bytes(), "big")
            if raw < bound:
                return int_to_limbs(raw % P)

    def draw_queries(self, max_n, domain_size):
        want = -(-max_n // 4) * 4 if self.kind == be.COIN_CAIRO else max_n
        vals = []
        while len(vals) < want:
            d = self._draw_bytes()
            for k in range(4):
                if len(vals) < want:
                    vals.append(int.from_bytes(d[8 * k: 8 * k + 8], "big") % domain_size)
        return sorted(set(vals[:max_n]))

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

def keccak256(data: bytes) -> bytes:
    """host Keccak-256 of the library (ss_keccak256_host): the Solidity coin reseeds once per
    out-of-domain evaluation, a few hundred hashes per proof"""
    import ctypes
    from . import _lib
    out = ctypes.create_string_buffer(32)
    _lib.check(_lib.load().ss_keccak256_host(bytes(data), len(data), out))
    return out.raw


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

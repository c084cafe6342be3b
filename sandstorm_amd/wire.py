"""The reference's proof wire format (ark-serialize, compressed) for the Keccak claims.

ministark's `Proof` is un-vendored; the layout below was read off `example/array-sum.proof.saved`, which parses
to its last byte with it and whose every Merkle opening then verifies (tests/golden/make_proof_golden.py):

  5 x u8   options: num_queries, lde_blowup_factor, grinding_factor, fri_folding_factor, fri_max_remainder_coeffs
  u64      trace_len                                                  (all integers little-endian)
  digest   base trace root          digest = u64 32 | 32 bytes
  u8 0/1 [+ digest]                  Option<extension trace root>
  digest   composition trace root
  u64 L, then per FRI layer:  Vec<Fp> flattened rows | u64 count x opening | digest layer root
  Vec<Fp>  remainder coefficients    Vec<Fp> = u64 count | 32-byte little-endian CANONICAL values
  u64      proof-of-work nonce
  Vec<Fp>  base rows | Vec<Fp> extension rows | Vec<Fp> composition rows      (query-major)
  openings base | openings extension | openings composition                  (u64 count x opening)
  Vec<Fp>  trace out-of-domain evaluations | Vec<Fp> composition out-of-domain evaluations
  opening: u8 variant | Vec<digest> path (bottom-up, above the leaf pair)
           variant 0 (hashed leaves):   digest sibling | digest leaf
           variant 1 (single column):   Fp sibling | Fp leaf          (32 raw bytes each)

Only what the file shows is implemented: masked/unmasked Keccak trees (`LeafVariantMerkleTree`).  The
`MixedMerkleDigest` encoding of `FriendlyMerkleTree` proofs has no sample in the reference and is refused.
Field elements cross this boundary as canonical integers; the prover's arrays hold Montgomery limbs.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

P = 2**251 + 17 * 2**192 + 1
_R_INV = pow(2**256, -1, P)
_R = 2**256 % P


def _canon(limbs) -> int:
    """4 x u64 little-endian Montgomery limbs -> canonical integer"""
    v = 0
    for k in range(4):
        v |= int(limbs[k]) << (64 * k)
    return v * _R_INV % P


def _mont_limbs(x: int):
    v = x % P * _R % P
    return [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]


@dataclass
class Opening:
    variant: int                       # 0: hashed leaves, 1: raw single-column leaves
    path: List[bytes]                  # digests above the leaf pair, bottom-up
    sibling: object                    # bytes (variant 0) or int (variant 1)
    leaf: object


@dataclass
class WireFriLayer:
    rows: List[int]                    # flattened, canonical
    openings: List[Opening]
    root: bytes


@dataclass
class WireProof:
    options: List[int]
    trace_len: int
    base_root: bytes
    extension_root: Optional[bytes]
    composition_root: bytes
    fri_layers: List[WireFriLayer] = field(default_factory=list)
    remainder: List[int] = field(default_factory=list)
    pow_nonce: int = 0
    base_rows: List[int] = field(default_factory=list)
    extension_rows: List[int] = field(default_factory=list)
    composition_rows: List[int] = field(default_factory=list)
    base_openings: List[Opening] = field(default_factory=list)
    extension_openings: List[Opening] = field(default_factory=list)
    composition_openings: List[Opening] = field(default_factory=list)
    ood_trace: List[int] = field(default_factory=list)
    ood_composition: List[int] = field(default_factory=list)


# ------------------------------------------------------------------------------------------- parse
class _Reader:
    """bounds-checked cursor: the bytes are untrusted"""

    def __init__(self, raw):
        self.raw, self.o = raw, 0

    def _take(self, k):
        if self.o + k > len(self.raw):
            raise ValueError("truncated at offset %d" % self.o)
        self.o += k
        return self.raw[self.o - k:self.o]

    def u8(self):
        return self._take(1)[0]

    def u64(self):
        return int.from_bytes(self._take(8), "little")

    def count(self, unit):
        """a length prefix of items at least `unit` bytes each"""
        n = self.u64()
        if n * unit > len(self.raw) - self.o:
            raise ValueError("count %d at offset %d exceeds the remaining bytes" % (n, self.o - 8))
        return n

    def fp(self):
        v = int.from_bytes(self._take(32), "little")
        if v >= P:
            raise ValueError("non-canonical field element at offset %d" % (self.o - 32))
        return v

    def vec(self):
        return [self.fp() for _ in range(self.count(32))]

    def digest(self):
        if self.u64() != 32:
            raise ValueError("digest length prefix is not 32 at offset %d" % (self.o - 8))
        return bytes(self._take(32))

    def openings(self):
        out = []
        for _ in range(self.count(9)):
            variant = self.u8()
            if variant not in (0, 1):
                raise ValueError("unknown opening variant %d at offset %d" % (variant, self.o - 1))
            path = [self.digest() for _ in range(self.count(40))]
            if variant == 0:
                out.append(Opening(0, path, self.digest(), self.digest()))
            else:
                out.append(Opening(1, path, self.fp(), self.fp()))
        return out


def parse(raw: bytes) -> WireProof:
    r = _Reader(raw)
    options = [r.u8() for _ in range(5)]
    trace_len = r.u64()
    base_root = r.digest()
    has_ext = r.u8()
    if has_ext not in (0, 1):
        raise ValueError("bad Option tag for the extension root")
    ext_root = r.digest() if has_ext else None
    p = WireProof(options, trace_len, base_root, ext_root, r.digest())
    for _ in range(r.count(48)):
        rows = r.vec()
        openings = r.openings()
        p.fri_layers.append(WireFriLayer(rows, openings, r.digest()))
    p.remainder = r.vec()
    p.pow_nonce = r.u64()
    p.base_rows, p.extension_rows, p.composition_rows = r.vec(), r.vec(), r.vec()
    p.base_openings, p.extension_openings, p.composition_openings = r.openings(), r.openings(), r.openings()
    p.ood_trace, p.ood_composition = r.vec(), r.vec()
    if r.o != len(raw):
        raise ValueError("%d trailing bytes" % (len(raw) - r.o))
    return p


# --------------------------------------------------------------------------------------- serialize
def _u64(v):
    return int(v).to_bytes(8, "little")


def _fp(v):
    return int(v).to_bytes(32, "little")


def _vec(vals):
    return _u64(len(vals)) + b"".join(_fp(v) for v in vals)


def _digest(d):
    assert len(d) == 32
    return _u64(32) + bytes(d)


def _openings(ops):
    out = [_u64(len(ops))]
    for o in ops:
        out.append(bytes([o.variant]) + _u64(len(o.path)) + b"".join(_digest(d) for d in o.path))
        out.append(_digest(o.sibling) + _digest(o.leaf) if o.variant == 0 else _fp(o.sibling) + _fp(o.leaf))
    return b"".join(out)


def serialize(p: WireProof) -> bytes:
    out = [bytes(p.options), _u64(p.trace_len), _digest(p.base_root)]
    out.append(b"\x01" + _digest(p.extension_root) if p.extension_root is not None else b"\x00")
    out.append(_digest(p.composition_root))
    out.append(_u64(len(p.fri_layers)))
    for layer in p.fri_layers:
        out += [_vec(layer.rows), _openings(layer.openings), _digest(layer.root)]
    out += [_vec(p.remainder), _u64(p.pow_nonce), _vec(p.base_rows), _vec(p.extension_rows), _vec(p.composition_rows),
            _openings(p.base_openings), _openings(p.extension_openings), _openings(p.composition_openings),
            _vec(p.ood_trace), _vec(p.ood_composition)]
    return b"".join(out)


# ------------------------------------------------------------------ from the prover's Proof object
def from_proof(proof, leaf_hash) -> WireProof:
    """prover.Proof (Montgomery limbs, paths leaf level first) -> WireProof.
    leaf_hash(list of canonical ints) -> 32-byte digest of that row (the tree's row hash: the wire format carries
    the leaf digest next to its sibling)."""
    opt = proof.options
    if getattr(proof, "tree_kind", None) == 2:          # backend.TREE_FRIENDLY
        raise NotImplementedError("MixedMerkleDigest (FriendlyMerkleTree) proofs: no reference sample of the encoding")
    w = WireProof([opt.num_queries, opt.lde_blowup_factor, opt.grinding_factor, opt.fri_folding_factor,
                   opt.fri_max_remainder_coeffs], proof.trace_len, proof.base_root, proof.extension_root,
                  proof.composition_root)

    def rows_of(arr):
        return [_canon(e) for row in arr for e in row]

    def hashed_openings(rows_arr, paths):
        ops = []
        for row, path in zip(rows_arr, paths):
            digests = [bytes(d) for d in path]
            ops.append(Opening(0, digests[1:], digests[0], leaf_hash([_canon(e) for e in row])))
        return ops

    def felt_openings(rows_arr, paths):
        ops = []
        for row, path in zip(rows_arr, paths):
            digests = [bytes(d) for d in path]
            # the leaf slots of a single-column tree hold the elements as big-endian Montgomery bytes
            sib = int.from_bytes(digests[0], "big") * _R_INV % P
            ops.append(Opening(1, digests[1:], sib, _canon(row[0])))
        return ops

    def openings(rows_arr, paths):
        return felt_openings(rows_arr, paths) if rows_arr.shape[1] == 1 else hashed_openings(rows_arr, paths)

    for layer in proof.fri_layers:
        if getattr(layer, "root_tag", 0):
            raise NotImplementedError("MixedMerkleDigest (FriendlyMerkleTree) proofs: no reference sample of the encoding")
        w.fri_layers.append(WireFriLayer(rows_of(layer.rows), hashed_openings(layer.rows, layer.paths), layer.root))
    w.remainder = [_canon(e) for e in proof.fri_remainder]
    w.pow_nonce = proof.pow_nonce
    w.base_rows, w.composition_rows = rows_of(proof.base_rows), rows_of(proof.composition_rows)
    w.base_openings = openings(proof.base_rows, proof.base_paths)
    w.composition_openings = openings(proof.composition_rows, proof.composition_paths)
    if proof.extension_rows is not None:
        w.extension_rows = rows_of(proof.extension_rows)
        w.extension_openings = openings(proof.extension_rows, proof.extension_paths)
    w.ood_trace = [_canon(e) for e in proof.ood_trace]
    w.ood_composition = [_canon(e) for e in proof.ood_composition]
    return w

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

The Keccak trees (`LeafVariantMerkleTree`) are what the files show.  `FriendlyMerkleTree` proofs (the Cairo-verifier
claims: Blake2s rows, Pedersen top layers) have no sample file; their encoding follows the reference's SOURCE - it is
source-pinned, data-unpinned:
  root / inner node   MixedMerkleDigest: u8 0 + Fp (32-byte little-endian canonical: PedersenDigest derives
                      CanonicalSerialize) | u8 1 + digest (Blake2s)           crypto/src/merkle/mixed.rs:46-71, 88-101
  opening             FriendlyMerkleTreeProof: u8 0 (MultiCol) + MerkleView<MixedMerkleDigest, Blake2s digest>
                                             | u8 1 (SingleCol) + MerkleView<PedersenDigest, Fp>    merkle/mod.rs:168-236
                      with MerkleView = Vec<node> path | sibling leaf | leaf, as in the Keccak files
Inside this module a Pedersen node is the 32 bytes the device trees hold (big-endian canonical, PedersenDigest::as_bytes,
hash/pedersen.rs:23-28) and carries tag 0; a Blake2s node carries tag 1.
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


TREE_FRIENDLY = 2                      # backend.TREE_FRIENDLY


@dataclass
class Opening:
    variant: int                       # 0: hashed leaves, 1: raw single-column leaves
    path: List[bytes]                  # digests above the leaf pair, bottom-up
    sibling: object                    # bytes (variant 0) or int (variant 1)
    leaf: object
    tags: Optional[List[int]] = None   # FriendlyMerkleTree, variant 0: MixedMerkleDigest tag of every path entry


@dataclass
class WireFriLayer:
    rows: List[int]                    # flattened, canonical
    openings: List[Opening]
    root: bytes
    root_tag: int = 0                  # FriendlyMerkleTree: MixedMerkleDigest tag of the root


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
    tree_kind: int = 1                 # backend tree kind of the claim; TREE_FRIENDLY switches the digest encodings
    root_tags: List[int] = field(default_factory=lambda: [0, 0, 0])      # FriendlyMerkleTree: tags of the three trace roots


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

    def mixed(self):
        """MixedMerkleDigest -> (32 bytes as the trees hold them, tag)"""
        tag = self.u8()
        if tag == 0:
            return self.fp().to_bytes(32, "big"), 0
        if tag == 1:
            return self.digest(), 1
        raise ValueError("unknown MixedMerkleDigest tag %d at offset %d" % (tag, self.o - 1))

    def openings(self, friendly=False):
        out = []
        for _ in range(self.count(9)):
            variant = self.u8()
            if variant not in (0, 1):
                raise ValueError("unknown opening variant %d at offset %d" % (variant, self.o - 1))
            if not friendly:
                path = [self.digest() for _ in range(self.count(40))]
                if variant == 0:
                    out.append(Opening(0, path, self.digest(), self.digest()))
                else:
                    out.append(Opening(1, path, self.fp(), self.fp()))
            elif variant == 0:
                nodes = [self.mixed() for _ in range(self.count(33))]
                out.append(Opening(0, [d for d, _ in nodes], self.digest(), self.digest(), [t for _, t in nodes]))
            else:
                path = [self.fp().to_bytes(32, "big") for _ in range(self.count(32))]
                out.append(Opening(1, path, self.fp(), self.fp(), [0] * len(path)))
        return out


def parse(raw: bytes, tree_kind: int = 1) -> WireProof:
    """tree_kind: the claim's tree (backend.TREE_*): the bytes do not say which digest encoding they use"""
    r = _Reader(raw)
    friendly = tree_kind == TREE_FRIENDLY
    root = (lambda: r.mixed()) if friendly else (lambda: (r.digest(), 0))
    options = [r.u8() for _ in range(5)]
    trace_len = r.u64()
    base_root, t0 = root()
    has_ext = r.u8()
    if has_ext not in (0, 1):
        raise ValueError("bad Option tag for the extension root")
    ext_root, t1 = root() if has_ext else (None, 0)
    comp_root, t2 = root()
    p = WireProof(options, trace_len, base_root, ext_root, comp_root, tree_kind=tree_kind, root_tags=[t0, t1, t2])
    for _ in range(r.count(48)):
        rows = r.vec()
        openings = r.openings(friendly)
        lroot, ltag = root()
        p.fri_layers.append(WireFriLayer(rows, openings, lroot, ltag))
    p.remainder = r.vec()
    p.pow_nonce = r.u64()
    p.base_rows, p.extension_rows, p.composition_rows = r.vec(), r.vec(), r.vec()
    p.base_openings, p.extension_openings, p.composition_openings = r.openings(friendly), r.openings(friendly), r.openings(friendly)
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


def _mixed(d, tag):
    """MixedMerkleDigest: Pedersen nodes are big-endian canonical bytes here, a little-endian Fp on the wire"""
    assert len(d) == 32 and tag in (0, 1)
    return b"\x00" + bytes(d)[::-1] if tag == 0 else b"\x01" + _digest(d)


def _openings(ops, friendly=False):
    out = [_u64(len(ops))]
    for o in ops:
        if not friendly:
            out.append(bytes([o.variant]) + _u64(len(o.path)) + b"".join(_digest(d) for d in o.path))
        elif o.variant == 0:
            out.append(b"\x00" + _u64(len(o.path)) + b"".join(_mixed(d, t) for d, t in zip(o.path, o.tags)))
        else:
            out.append(b"\x01" + _u64(len(o.path)) + b"".join(bytes(d)[::-1] for d in o.path))
        out.append(_digest(o.sibling) + _digest(o.leaf) if o.variant == 0 else _fp(o.sibling) + _fp(o.leaf))
    return b"".join(out)


def serialize(p: WireProof) -> bytes:
    friendly = p.tree_kind == TREE_FRIENDLY
    root = (lambda d, t: _mixed(d, t)) if friendly else (lambda d, t: _digest(d))
    out = [bytes(p.options), _u64(p.trace_len), root(p.base_root, p.root_tags[0])]
    out.append(b"\x01" + root(p.extension_root, p.root_tags[1]) if p.extension_root is not None else b"\x00")
    out.append(root(p.composition_root, p.root_tags[2]))
    out.append(_u64(len(p.fri_layers)))
    for layer in p.fri_layers:
        out += [_vec(layer.rows), _openings(layer.openings, friendly), root(layer.root, layer.root_tag)]
    out += [_vec(p.remainder), _u64(p.pow_nonce), _vec(p.base_rows), _vec(p.extension_rows), _vec(p.composition_rows),
            _openings(p.base_openings, friendly), _openings(p.extension_openings, friendly), _openings(p.composition_openings, friendly),
            _vec(p.ood_trace), _vec(p.ood_composition)]
    return b"".join(out)


# ------------------------------------------------------------------ from the prover's Proof object
def from_proof(proof, leaf_hash) -> WireProof:
    """prover.Proof (Montgomery limbs, paths leaf level first) -> WireProof.
    leaf_hash(list of canonical ints) -> 32-byte digest of that row (the tree's row hash: the wire format carries
    the leaf digest next to its sibling)."""
    opt = proof.options
    friendly = getattr(proof, "tree_kind", None) == TREE_FRIENDLY
    w = WireProof([opt.num_queries, opt.lde_blowup_factor, opt.grinding_factor, opt.fri_folding_factor,
                   opt.fri_max_remainder_coeffs], proof.trace_len, proof.base_root, proof.extension_root,
                  proof.composition_root, tree_kind=proof.tree_kind if proof.tree_kind is not None else 1,
                  root_tags=list(getattr(proof, "root_tags", None) or [0, 0, 0]))

    def rows_of(arr):
        return [_canon(e) for row in arr for e in row]

    def tags_of(path_tags, q):
        """MixedMerkleDigest tags of the path entries above the leaf pair (FriendlyMerkleTree only)"""
        if not friendly:
            return None
        if path_tags is None:
            raise ValueError("a FriendlyMerkleTree proof needs the tags of its authentication paths (Proof.*_path_tags)")
        return [int(t) for t in path_tags[q][1:]]

    def hashed_openings(rows_arr, paths, path_tags=None):
        ops = []
        for q, (row, path) in enumerate(zip(rows_arr, paths)):
            digests = [bytes(d) for d in path]
            ops.append(Opening(0, digests[1:], digests[0], leaf_hash([_canon(e) for e in row]), tags_of(path_tags, q)))
        return ops

    def felt_openings(rows_arr, paths):
        ops = []
        for row, path in zip(rows_arr, paths):
            digests = [bytes(d) for d in path]
            # the leaf slots of a single-column tree hold the elements as big-endian Montgomery bytes
            sib = int.from_bytes(digests[0], "big") * _R_INV % P
            ops.append(Opening(1, digests[1:], sib, _canon(row[0]), [0] * (len(digests) - 1) if friendly else None))
        return ops

    def openings(rows_arr, paths, path_tags=None):
        return felt_openings(rows_arr, paths) if rows_arr.shape[1] == 1 else hashed_openings(rows_arr, paths, path_tags)

    for layer in proof.fri_layers:
        w.fri_layers.append(WireFriLayer(rows_of(layer.rows), hashed_openings(layer.rows, layer.paths, getattr(layer, "path_tags", None)),
                                         layer.root, int(getattr(layer, "root_tag", 0)) if friendly else 0))
    w.remainder = [_canon(e) for e in proof.fri_remainder]
    w.pow_nonce = proof.pow_nonce
    w.base_rows, w.composition_rows = rows_of(proof.base_rows), rows_of(proof.composition_rows)
    w.base_openings = openings(proof.base_rows, proof.base_paths, getattr(proof, "base_path_tags", None))
    w.composition_openings = openings(proof.composition_rows, proof.composition_paths, getattr(proof, "composition_path_tags", None))
    if proof.extension_rows is not None:
        w.extension_rows = rows_of(proof.extension_rows)
        w.extension_openings = openings(proof.extension_rows, proof.extension_paths, getattr(proof, "extension_path_tags", None))
    w.ood_trace = [_canon(e) for e in proof.ood_trace]
    w.ood_composition = [_canon(e) for e in proof.ood_composition]
    return w

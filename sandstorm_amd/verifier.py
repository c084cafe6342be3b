"""Host-side verifier of proofs in the reference's wire format ("verify what we proved").

ministark's `Stark::verify` is un-vendored; this follows the prover's own transcript (prover.py, step for step) and
the conventions the reference's shipped proofs pin (`prover.Conventions`: bit-reversed commitment order, unnormalised
fold-8, unshifted remainder — tests/golden/make_fri_golden.py, make_proof_golden.py).  Everything here is host
arithmetic on a few hundred field elements: Python integers, the host hashes of coin.py, no device.

Checked, in the verifier's order:
  1. transcript replay (challenges, composition coefficient, z, DEEP alpha, FRI alphas, proof of work, query positions)
  2. out-of-domain identity: composition constraint at z from the trace OOD values == sum_k z^k H_k(z^m)
  3. Merkle openings of the base / extension / composition rows at the query positions
  4. DEEP value recomputed from the opened rows == the first FRI layer's entry at that position
  5. every FRI layer folds into the next at alpha_i, openings included; the last into the remainder polynomial
Keccak trees as the reference's files pin them; FriendlyMerkleTree (Cairo-verifier claims) as its source has them
(source-pinned, no sample file: wire.py).
"""
from dataclasses import dataclass
from typing import Callable, List

from . import air_program as ap
from . import backend as be
from . import wire
from .coin import PublicCoin, blake2s256, keccak256
from .prover import Conventions, bitrev

P = be.P


class VerificationError(Exception):
    pass


@dataclass
class VerifierAir:
    """What the verifier needs from an AirConfig (src/lib.rs:75-125)."""
    num_base_columns: int
    num_extension_columns: int
    num_challenges: int
    mask: List[tuple]                   # trace_arguments(): sorted (column, row offset) cells, the order of the OOD vector
    # (trace_len, challenges: list of int, composition_coeff: int) -> air_program.Expr of the composition constraint
    composition: Callable = None
    # (trace_len, x: int, table index) -> value at x of the periodic / zerofier table that Expr refers to
    table_at: Callable = None


def _require(cond, what):
    if not cond:
        raise VerificationError(what)


def _root_of_unity(n):
    return pow(3, (P - 1) // n, P)


def _mont_be(v):
    return (v * wire._R % P).to_bytes(32, "big")


def _verify_pow(coin_kind, digest, bits, nonce):
    """PublicCoin::verify_proof_of_work (crypto/src/public_coin/solidity.rs:143-156, cairo.rs:156-169)"""
    h = keccak256 if coin_kind == be.COIN_SOLIDITY else blake2s256
    prefix = h((0x0123456789ABCDED).to_bytes(8, "big") + digest + bytes([bits]))
    out = h(prefix + int(nonce).to_bytes(8, "big"))
    return int.from_bytes(out, "big") >> (256 - bits) == 0 if bits else True


class _KeccakTree:
    """LeafVariantMerkleTree<Keccak256HashFn | MaskedKeccak256HashFn<20>> (crypto/src/merkle/mod.rs:240-304)"""

    def __init__(self, tree_kind):
        _require(tree_kind in (be.TREE_KECCAK, be.TREE_KECCAK_M20), "not a Keccak tree")
        self.masked = tree_kind == be.TREE_KECCAK_M20

    def h(self, data):
        d = keccak256(data)
        return d[:20] + bytes(12) if self.masked else d

    def row_leaf(self, row):
        return self.h(b"".join(_mont_be(v) for v in row))

    def climb(self, node, path, pos):
        for lvl, sib in enumerate(path):
            node = self.h(node + sib) if ((pos >> lvl) & 1) == 0 else self.h(sib + node)
        return node

    def check_root(self, root, tag, what):
        pass

    def check_opening(self, opening, row, pos, depth, root, what):
        """row: the opened row (canonical ints); pos: leaf index; depth = log2(number of leaves)"""
        if len(row) == 1:                                   # single column: raw-element leaves (merkle/mod.rs:113-117)
            _require(opening.variant == 1 and opening.leaf == row[0], what + ": leaf is not the opened element")
            _require(len(opening.path) == depth - 1, what + ": path length")
            pair = (opening.leaf, opening.sibling) if (pos & 1) == 0 else (opening.sibling, opening.leaf)
            node = self.h(_mont_be(pair[0]) + _mont_be(pair[1]))
            _require(self.climb(node, opening.path, pos >> 1) == root, what + ": authentication path does not reach the root")
        else:
            _require(opening.variant == 0 and opening.leaf == self.row_leaf(row), what + ": leaf is not the hash of the opened row")
            _require(len(opening.path) == depth - 1, what + ": path length")
            _require(self.climb(opening.leaf, [opening.sibling] + list(opening.path), pos) == root,
                     what + ": authentication path does not reach the root")


class _FriendlyTree:
    """FriendlyMerkleTree<N, PedersenHashFn> (crypto/src/merkle/mod.rs:43-123, mixed.rs:106-155): rows hashed with masked
    Blake2s, inner nodes at depth < N by Pedersen (Blake2s digests entering that boundary read as big-endian integers),
    below it by masked Blake2s; a single-column matrix is a Pedersen tree over its elements.  `depth` of a node: root 0.
    Source-pinned (no reference sample of such a proof exists): the depth numbering is the assumption of oracle/merkle.c."""

    def __init__(self, n_friendly=22):
        self.n_friendly = n_friendly

    @staticmethod
    def blake(data):
        return bytes(12) + blake2s256(data)[12:]            # MaskedBlake2sHashFn<20> (hash/blake2s.rs:63-83): the low 20 bytes

    @staticmethod
    def pedersen(a: int, b: int) -> int:
        from .coin import canonical
        return canonical(be.pedersen_hash_host(be.felt(a), be.felt(b)))

    def row_leaf(self, row):
        return self.blake(b"".join(_mont_be(v) for v in row))

    def merge(self, depth, left, right):
        """MixedHashMerkleTreeConfigImpl::{hash_leaves, hash_nodes}: -> (digest, tag) of a node at `depth`"""
        if depth < self.n_friendly:
            v = self.pedersen(int.from_bytes(left, "big") % P, int.from_bytes(right, "big") % P)
            return v.to_bytes(32, "big"), 0
        return self.blake(left + right), 1

    def check_root(self, root, tag, what):
        _require(tag == (0 if self.n_friendly > 0 else 1), what + ": root digest variant")

    def check_opening(self, opening, row, pos, depth, root, what):
        _require(len(opening.path) == depth - 1, what + ": path length")
        if len(row) == 1:                                   # SingleCol: MerkleTreeImpl<UnhashedLeafConfig<PedersenHashFn>>
            _require(opening.variant == 1 and opening.leaf == row[0], what + ": leaf is not the opened element")
            a, b = (opening.leaf, opening.sibling) if (pos & 1) == 0 else (opening.sibling, opening.leaf)
            node = self.pedersen(self.pedersen(self.pedersen(0, a), b), 2)          # PedersenHashFn::hash_elements([l0, l1])
            p = pos >> 1
            for sib in opening.path:
                s_ = int.from_bytes(sib, "big")
                _require(s_ < P, what + ": non-canonical Pedersen digest")
                node = self.pedersen(node, s_) if (p & 1) == 0 else self.pedersen(s_, node)
                p >>= 1
            _require(node.to_bytes(32, "big") == root, what + ": authentication path does not reach the root")
            return
        _require(opening.variant == 0 and opening.leaf == self.row_leaf(row), what + ": leaf is not the hash of the opened row")
        _require(opening.tags is not None and len(opening.tags) == len(opening.path), what + ": path tags")
        node, p = opening.leaf, pos
        sibs = [(opening.sibling, 1)] + list(zip(opening.path, opening.tags))
        for lvl, (sib, tag) in enumerate(sibs):
            d = depth - 1 - lvl                             # depth of the parent computed at this step
            # the sibling is a node of depth d + 1: a leaf, a Blake2s node, or (above the boundary) a Pedersen node
            _require(tag == (0 if d + 1 < self.n_friendly and lvl > 0 else 1), what + ": digest variant at level %d" % lvl)
            if tag == 0:
                _require(int.from_bytes(sib, "big") < P, what + ": non-canonical Pedersen digest")
            node, _ = self.merge(d, node, sib) if (p & 1) == 0 else self.merge(d, sib, node)
            p >>= 1
        _require(node == root, what + ": authentication path does not reach the root")


def _tree_of(tree_kind, n_friendly=22):
    return _FriendlyTree(n_friendly) if tree_kind == be.TREE_FRIENDLY else _KeccakTree(tree_kind)


def _interpolate_eval(xs, ys, t):
    acc = 0
    for i, (xi, yi) in enumerate(zip(xs, ys)):
        num = den = 1
        for j, xj in enumerate(xs):
            if i != j:
                num = num * (t - xj) % P
                den = den * (xi - xj) % P
        acc = (acc + yi * num * pow(den, -1, P)) % P
    return acc


DEFAULT_REQUIRED_SECURITY_BITS = 80     # cli/src/main.rs:66-67: `verify --required-security-bits`, default 80


def conjectured_security_bits(options, trace_len, tree_kind, coin_kind):
    """`Proof::security_level_bits` (cli/src/main.rs:203; ministark, conjectured): what the query phase buys
    (num_queries * log2(blowup) + grinding bits), capped by the field (252 bits minus log2 of the LDE domain) and by the
    collision resistance of the claim's tree hash (crypto/src/merkle/mod.rs:100-102, 283-285, 434-436; mixed.rs:127-129:
    128 bits for Keccak / Blake2s, 8*20/2 = 80 for the masked-20 variants, 125 for Pedersen - hash/keccak.rs:17,64,
    blake2s.rs:14,67, pedersen.rs:48) and of the coin's hash (public_coin/solidity.rs:158-160, cairo.rs:171-173)."""
    num_queries, blowup, grinding = options[0], options[1], options[2]
    query = num_queries * (blowup.bit_length() - 1) + grinding
    field = 252 - ((trace_len * blowup).bit_length() - 1)
    tree = {be.TREE_KECCAK: 128, be.TREE_KECCAK_M20: 80, be.TREE_FRIENDLY: 80}[tree_kind]
    return max(0, min(query, field, tree, 128))


def fri_layer_count(trace_len, fold, max_remainder):
    """number of FRI layers and the remainder's coefficient bound, as the provers compute them (prover.py step 8);
    the options are untrusted here: anything the provers would refuse is a rejection"""
    _require(max_remainder >= 1 and max_remainder & (max_remainder - 1) == 0, "FRI max remainder is not a power of two >= 1")
    degree_bound, nlayers = trace_len, 0
    while degree_bound > max_remainder:
        _require(degree_bound % fold == 0, "trace length is not the remainder bound times a power of the folding factor")
        degree_bound //= fold
        nlayers += 1
    return nlayers, degree_bound


def replay_transcript(w, air, coin_kind, coin_seed, conv=None):
    """The verifier's step 1 on a parsed proof: every reseed and draw of prover.py steps 2-9 in order, the proof of work
    included.  -> {"challenges", "composition_coeff", "z", "deep_alpha", "fri_alphas", "positions"} (canonical ints)"""
    conv = conv or Conventions()
    num_queries, blowup, grinding, fold, max_remainder = w.options
    n = w.trace_len
    N = n * blowup
    log_N, log_fold = N.bit_length() - 1, fold.bit_length() - 1
    ncomp, nmask = conv.composition_columns, len(air.mask)
    _require(len(w.ood_trace) == nmask and len(w.ood_composition) == ncomp, "out-of-domain vector lengths")
    _require((w.extension_root is not None) == (air.num_extension_columns > 0), "extension root presence")
    coin = PublicCoin(coin_kind, coin_seed)
    coin.reseed_with_digest(w.base_root)
    challenges = [coin.draw() for _ in range(air.num_challenges)]
    if w.extension_root is not None:
        coin.reseed_with_digest(w.extension_root)
    comp_coeff = coin.draw()
    coin.reseed_with_digest(w.composition_root)
    z_l = coin.draw()
    coin.reseed_with_field_elements([be.felt(v) for v in list(w.ood_trace) + list(w.ood_composition)])
    deep_alpha = wire._canon(coin.draw())
    # expected number of layers: fold until the remainder fits (prover.py step 8)
    nlayers, degree_bound = fri_layer_count(n, fold, max_remainder)
    _require(log_fold * nlayers <= log_N, "FRI layers exceed the evaluation domain")
    _require(len(w.fri_layers) == nlayers, "number of FRI layers")
    _require(len(w.remainder) == max(1, degree_bound), "remainder length")
    fri_alphas = []
    for li, layer in enumerate(w.fri_layers):
        coin.reseed_with_digest(layer.root)
        scale = pow(conv.lde_offset, fold ** li, P) if conv.fri_alpha_times_offset else 1
        fri_alphas.append(wire._canon(coin.draw()) * scale % P)
    coin.reseed_with_field_element_vector([be.felt(v) for v in w.remainder])
    _require(_verify_pow(coin_kind, coin.digest, grinding, w.pow_nonce), "proof of work")
    coin.reseed_with_int(w.pow_nonce)
    positions = coin.draw_queries(num_queries, N)
    nq = len(positions)
    z = wire._canon(z_l)
    ch = [wire._canon(c) for c in challenges]

    return {"challenges": ch, "composition_coeff": wire._canon(comp_coeff), "z": z, "deep_alpha": deep_alpha,
            "fri_alphas": fri_alphas, "positions": positions}


def verify(proof, air: VerifierAir, tree_kind, coin_kind, coin_seed: bytes, conv: Conventions = None,
           required_security_bits: int = DEFAULT_REQUIRED_SECURITY_BITS, expected_options=None, n_friendly_layers: int = 22):
    """proof: bytes in the reference's wire format, or a wire.WireProof.  Raises VerificationError; returns the
    query positions on success.  The bytes are untrusted: any arithmetic or indexing accident they provoke (a zero
    denominator, a missing position) is a rejection too.  The proof's own options are untrusted as well
    (`claim.verify(proof, required_security_bits)`, cli/src/main.rs:176): a proof whose options conjecture fewer than
    `required_security_bits` is rejected, and so is one whose options differ from `expected_options` when given."""
    try:
        return _verify(proof, air, tree_kind, coin_kind, coin_seed, conv, required_security_bits, expected_options, n_friendly_layers)
    except (ValueError, IndexError, KeyError, ZeroDivisionError, OverflowError) as e:
        raise VerificationError("malformed proof: %s: %s" % (type(e).__name__, e))


def _verify(proof, air, tree_kind, coin_kind, coin_seed, conv, required_security_bits, expected_options, n_friendly_layers=22):
    conv = conv or Conventions()
    try:
        w = wire.parse(bytes(proof), tree_kind) if isinstance(proof, (bytes, bytearray, memoryview)) else proof
    except ValueError as e:
        raise VerificationError("malformed proof: %s" % e)
    num_queries, blowup, grinding, fold, max_remainder = w.options
    n = w.trace_len
    _require(n >= 2 and n & (n - 1) == 0 and blowup >= 2 and blowup & (blowup - 1) == 0, "bad trace length / blowup")
    _require(n * blowup <= 1 << 40, "bad trace length / blowup")
    _require(fold in (2, 4, 8, 16), "bad FRI folding factor")
    _require(num_queries >= 1, "proof options: no queries")
    if expected_options is not None:
        exp = expected_options if isinstance(expected_options, (list, tuple)) else \
            [expected_options.num_queries, expected_options.lde_blowup_factor, expected_options.grinding_factor,
             expected_options.fri_folding_factor, expected_options.fri_max_remainder_coeffs]
        _require(list(w.options) == list(exp), "proof options differ from the expected ones")
    sec = conjectured_security_bits(w.options, n, tree_kind, coin_kind)
    _require(sec >= required_security_bits, "proof options give %d bits of conjectured security, %d required" % (sec, required_security_bits))
    N = n * blowup
    log_N, log_fold = N.bit_length() - 1, fold.bit_length() - 1
    ncomp = conv.composition_columns
    nmask = len(air.mask)
    _require(getattr(w, "tree_kind", tree_kind) == tree_kind or tree_kind != be.TREE_FRIENDLY, "the proof was parsed for another tree")
    expo = (lambda i, bits: bitrev(i, bits)) if conv.bitrev_commit else (lambda i, bits: i)

    # ---- 1. transcript (prover.py steps 2-9)
    t = replay_transcript(w, air, coin_kind, coin_seed, conv)
    z, deep_alpha, fri_alphas, positions, ch, comp_coeff = t["z"], t["deep_alpha"], t["fri_alphas"], t["positions"], t["challenges"], t["composition_coeff"]

    # ---- 2. out-of-domain identity: sum_k alpha^k C_k(z) == sum_k z^k H_k(z^ncomp)
    cell = {m: v for m, v in zip(air.mask, w.ood_trace)}
    dag = air.composition(n, ch, comp_coeff)
    lhs = ap.evaluate(dag, P, z, lambda c, o: cell[(c, o)], lambda t: air.table_at(n, z, t))
    rhs = sum(pow(z, k, P) * h for k, h in enumerate(w.ood_composition)) % P
    _require(lhs == rhs, "out-of-domain identity: the composition constraint does not match the composition columns at z")

    check_proof_data(w, air.mask, air.num_base_columns, air.num_extension_columns, tree_kind, z, deep_alpha, fri_alphas, positions, conv,
                     n_friendly_layers)
    return positions


def check_proof_data(w, mask, num_base_columns, num_extension_columns, tree_kind, z, deep_alpha, fri_alphas, positions, conv=None,
                     n_friendly_layers=22):
    """Everything below the transcript: Merkle openings of the opened rows, the DEEP value of every query against the
    first FRI layer, the FRI chain and the remainder — for given out-of-domain point, DEEP coefficient, FRI challenges
    (canonical ints) and query positions.  `verify` calls it with the values its transcript replay produced; the tests
    also run it on the reference's own shipped proof with the values recovered from that proof's data
    (tests/golden/deep_pin_recursive.json), where no transcript is available."""
    conv = conv or Conventions()
    num_queries, blowup, grinding, fold, max_remainder = w.options
    n = w.trace_len
    N = n * blowup
    log_N, log_fold = N.bit_length() - 1, fold.bit_length() - 1
    ncomp, nmask, nq = conv.composition_columns, len(mask), len(positions)
    tree = _tree_of(tree_kind, n_friendly_layers)
    if tree_kind == be.TREE_FRIENDLY:
        tags = getattr(w, "root_tags", [0, 0, 0])
        tree.check_root(w.base_root, tags[0], "base trace root")
        if w.extension_root is not None:
            tree.check_root(w.extension_root, tags[1], "extension trace root")
        tree.check_root(w.composition_root, tags[2], "composition trace root")
        for li, layer in enumerate(w.fri_layers):
            tree.check_root(layer.root, layer.root_tag, "FRI layer %d root" % li)
    expo = (lambda i, bits: bitrev(i, bits)) if conv.bitrev_commit else (lambda i, bits: i)

    class _Shape:
        pass
    air = _Shape()
    air.mask, air.num_base_columns, air.num_extension_columns = mask, num_base_columns, num_extension_columns
    # ---- 3./4. trace openings and the DEEP value at every query
    ncb, nce = air.num_base_columns, air.num_extension_columns
    _require(len(w.base_rows) == nq * ncb and len(w.base_openings) == nq, "base rows / openings count")
    _require(len(w.extension_rows) == nq * nce and len(w.extension_openings) == (nq if nce else 0), "extension rows / openings count")
    _require(len(w.composition_rows) == nq * ncomp and len(w.composition_openings) == nq, "composition rows / openings count")
    wN, wn = _root_of_unity(N), _root_of_unity(n)
    coef = [pow(deep_alpha, j, P) for j in range(nmask + ncomp)]
    zc = pow(z, ncomp, P)
    layer_positions = []
    p = list(positions)
    for li, layer in enumerate(w.fri_layers):
        row_bits = log_N - log_fold * (li + 1)
        rows = 1 << row_bits
        p = sorted(set((q >> log_fold) if conv.bitrev_commit else (q % rows) for q in p))
        layer_positions.append(p)
        _require(len(layer.openings) == len(p) and len(layer.rows) == fold * len(p), "FRI layer %d rows / openings count" % li)
    for qi, q in enumerate(positions):
        x = conv.lde_offset * pow(wN, expo(q, log_N), P) % P
        brow = w.base_rows[ncb * qi: ncb * qi + ncb]
        erow = w.extension_rows[nce * qi: nce * qi + nce]
        crow = w.composition_rows[ncomp * qi: ncomp * qi + ncomp]
        tree.check_opening(w.base_openings[qi], brow, q, log_N, w.base_root, "base trace, query %d" % qi)
        if nce:
            tree.check_opening(w.extension_openings[qi], erow, q, log_N, w.extension_root, "extension trace, query %d" % qi)
        tree.check_opening(w.composition_openings[qi], crow, q, log_N, w.composition_root, "composition trace, query %d" % qi)
        trow = list(brow) + list(erow)
        deep = 0
        for j, (c, o) in enumerate(air.mask):               # src/lib.rs:102-116: coefficients alpha^j over mask cells, then columns
            deep += coef[j] * (trow[c] - w.ood_trace[j]) * pow(x - z * pow(wn, o, P), -1, P)
        for k in range(ncomp):
            deep += coef[nmask + k] * (crow[k] - w.ood_composition[k]) * pow(x - zc, -1, P)
        if not w.fri_layers:
            # no layer to fold: the DEEP evaluations themselves were interpolated into the remainder (prover.py step 8)
            xr = pow(wN, expo(q, log_N), P) * (1 if conv.remainder_unshifted else conv.lde_offset) % P
            _require(sum(c * pow(xr, i, P) for i, c in enumerate(w.remainder)) % P == deep % P,
                     "DEEP composition value at query %d is not the remainder's" % qi)
            continue
        rows0 = N // fold
        r, slot = (q >> log_fold, q & (fold - 1)) if conv.bitrev_commit else (q % rows0, q // rows0)
        li0 = layer_positions[0].index(r)
        _require(w.fri_layers[0].rows[fold * li0 + slot] == deep % P, "DEEP composition value at query %d" % qi)

    # ---- 5. FRI: every opened row folds into the next layer (or the remainder)
    offset = conv.lde_offset % P
    for li, layer in enumerate(w.fri_layers):
        row_bits = log_N - log_fold * (li + 1)
        L, rows = 1 << (row_bits + log_fold), 1 << row_bits
        wl, wf = _root_of_unity(L), _root_of_unity(fold)
        for pi, r in enumerate(layer_positions[li]):
            ys = layer.rows[fold * pi: fold * pi + fold]
            tree.check_opening(layer.openings[pi], ys, r, row_bits, layer.root, "FRI layer %d, row %d" % (li, r))
            xr0 = offset * pow(wl, expo(r, row_bits), P) % P
            xs = [xr0 * pow(wf, expo(k, log_fold), P) % P for k in range(fold)]
            folded = _interpolate_eval(xs, ys, fri_alphas[li])
            if conv.fri_unnormalised:
                folded = folded * fold % P
            if li + 1 < len(w.fri_layers):
                nrows = rows >> log_fold
                nr, slot = (r >> log_fold, r & (fold - 1)) if conv.bitrev_commit else (r % nrows, r // nrows)
                ni = layer_positions[li + 1].index(nr)
                _require(w.fri_layers[li + 1].rows[fold * ni + slot] == folded, "FRI layer %d does not fold into layer %d at row %d" % (li, li + 1, r))
            else:
                xr = pow(_root_of_unity(rows), expo(r, row_bits), P)
                if not conv.remainder_unshifted:
                    xr = xr * pow(offset, fold, P) % P
                _require(sum(c * pow(xr, i, P) for i, c in enumerate(w.remainder)) % P == folded,
                         "last FRI layer does not fold into the remainder at row %d" % r)
        offset = pow(offset, fold, P)
    return positions

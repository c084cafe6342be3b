"""The proving pipeline on top of the C ABI — host orchestration only.

Mirrors the default body of ministark's `Stark::prove` as Sandstorm drives it
(src/lib.rs:75-125, cli/src/main.rs:180-213):

  1 base trace (columns resident in HBM)            [A1: supplied by the caller]
  2 LDE of the base columns, commit, reseed
  3 draw the AIR's challenges, build the extension columns  [A2: caller callback]
  4 LDE of the extension columns, commit, reseed
  5 draw the composition coefficient, evaluate the composition constraint on the
    LDE domain (Q1), interpolate, split into 2 columns, LDE, commit, reseed (Q2)
  6 draw z, out-of-domain evaluations of every mask cell and of the composition
    columns at z^2, reseed
  7 DEEP coefficients (powers of one alpha, src/lib.rs:102-116), DEEP composition
  8 FRI: per layer commit the fold-8 reshaped evaluations, draw alpha, fold
  9 proof of work, query positions, openings

Everything ministark decides internally is an
explicit field of `Conventions`, defaulting to the assumed value.  Every step's
arithmetic runs in the HIP kernels; the host only sequences them through the
Fiat-Shamir coin.
"""
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import backend as be
from .coin import PublicCoin, canonical, int_to_limbs


@dataclass
class ProofOptions:
    """cli/src/main.rs:51-60 defaults"""
    num_queries: int = 65
    lde_blowup_factor: int = 2
    grinding_factor: int = 16
    fri_folding_factor: int = 8
    fri_max_remainder_coeffs: int = 16


@dataclass
class Conventions:
    """ministark-internal conventions, SURVEY.md Appendix A"""
    lde_offset: int = 3                 # M2: LDE coset offset = field generator
    composition_columns: int = 2        # M5: ce_blowup_factor, H(x) = H0(x^2) + x H1(x^2), OOD point z^2
    # The next three are PINNED by the reference's shipped proofs (tests/golden/make_fri_golden.py,
    # make_proof_golden.py); False reproduces the older code path's proofs.
    bitrev_commit: bool = True          # M3: index i of a committed vector is the point offset * w^bitrev(i)
    fri_unnormalised: bool = True       # M8: fold = 8 * interpolant(alpha) (no 1/2 per halving)
    remainder_unshifted: bool = True    # M9: remainder = interpolant of the folded last layer over the unshifted domain
    # M8, second half, pinned by replaying the whole transcript of the reference's `example/array-sum.proof.saved`
    # (tests/test_verifier.py::test_reference_starknet_proof_verifies): the challenge a layer is folded with is the
    # coin's draw TIMES that layer's domain offset (lde_offset^(fold^layer)) - the reference folds over the unshifted
    # domain.  Provers and verifiers of both hosts honour the flag; False keeps the bare draw (round-1 proofs).
    fri_alpha_times_offset: bool = True


def bitrev(x: int, bits: int) -> int:
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


@dataclass
class Air:
    """What the prover needs to know about a layout's AIR (AirConfig, src/lib.rs:75-125)."""
    name: str
    num_base_columns: int
    num_extension_columns: int
    num_challenges: int
    mask: List[tuple]                   # trace_arguments(): sorted (column, row offset) cells
    # (n, challenges, composition_coeff) -> (air_program.Program, tables uint64[*,4] or None, table_desc list)
    build_program: Callable = None


@dataclass
class Claim:
    """A (AirConfig, MerkleTree, PublicCoin) tuple of src/claims.rs:12-33."""
    air: Air
    tree: type                          # backend.LeafVariantMerkleTree / FriendlyMerkleTree ...
    coin_kind: int                      # backend.COIN_SOLIDITY / COIN_CAIRO


@dataclass
class FriLayer:
    root: bytes
    root_tag: int
    log_len: int
    rows: Optional[np.ndarray] = None       # opened rows [nq, fold, 4]
    paths: Optional[np.ndarray] = None
    positions: List[int] = field(default_factory=list)
    path_tags: Optional[np.ndarray] = None  # FriendlyMerkleTree: MixedMerkleDigest tag of every path entry


@dataclass
class Proof:
    options: ProofOptions
    trace_len: int
    tree_kind: Optional[int] = None                 # the claim's commitment scheme (backend tree kind), when known
    base_root: bytes = b""
    extension_root: Optional[bytes] = None
    composition_root: bytes = b""
    ood_trace: Optional[np.ndarray] = None          # [nmask, 4] Montgomery limbs
    ood_composition: Optional[np.ndarray] = None    # [ncomp, 4]
    fri_layers: List[FriLayer] = field(default_factory=list)
    fri_remainder: Optional[np.ndarray] = None      # coefficients, natural order
    pow_nonce: int = 0
    query_positions: List[int] = field(default_factory=list)
    base_rows: Optional[np.ndarray] = None
    extension_rows: Optional[np.ndarray] = None
    composition_rows: Optional[np.ndarray] = None
    base_paths: Optional[np.ndarray] = None
    extension_paths: Optional[np.ndarray] = None
    composition_paths: Optional[np.ndarray] = None
    # FriendlyMerkleTree: MixedMerkleDigest tags (0 Pedersen, 1 Blake2s) of the three roots and of every path entry
    root_tags: Optional[List[int]] = None
    base_path_tags: Optional[np.ndarray] = None
    extension_path_tags: Optional[np.ndarray] = None
    composition_path_tags: Optional[np.ndarray] = None
    # values the verifier re-derives from the transcript, kept for tests/debugging
    challenges: List[np.ndarray] = field(default_factory=list)
    composition_coeff: Optional[np.ndarray] = None
    z: Optional[np.ndarray] = None
    deep_alpha: Optional[np.ndarray] = None
    fri_alphas: List[np.ndarray] = field(default_factory=list)


def _log2(v):
    assert v > 0 and v & (v - 1) == 0
    return v.bit_length() - 1


def _pow_limbs(base_limbs, count):
    """[alpha^0 .. alpha^(count-1)] as Montgomery limbs (tiny host arithmetic on transcript values)"""
    a = canonical(base_limbs)
    out, cur = [], 1
    for _ in range(count):
        out.append(be.felt(cur))
        cur = cur * a % be.P
    return np.stack(out) if out else np.zeros((0, 4), dtype=np.uint64)


class Prover:
    def __init__(self, ctx: be.Context, claim: Claim, options: ProofOptions = None, conventions: Conventions = None):
        self.ctx, self.claim = ctx, claim
        self.options = options or ProofOptions()
        self.conv = conventions or Conventions()
        self.timings = {}
        # use this proof-of-work nonce instead of grinding (must be valid for the transcript): any nonce with enough
        # leading zeros is a proof (the reference returns whichever its parallel search finds, solidity.rs:120-141)
        self.pow_nonce = None

    def prove(self, coin_seed: bytes, base_trace: be.Matrix,
              build_extension: Callable[[List[np.ndarray]], Optional[be.Matrix]]) -> Proof:
        ctx, opt, conv, air = self.ctx, self.options, self.conv, self.claim.air
        Tree = self.claim.tree
        n = base_trace.nrows
        log_n, lb = _log2(n), _log2(opt.lde_blowup_factor)
        log_N, N = log_n + lb, n << lb
        g = be.felt(conv.lde_offset)
        coin = PublicCoin(self.claim.coin_kind, coin_seed)
        proof = Proof(opt, n, tree_kind=Tree.tree_kind)
        import time
        trace_stages = bool(self.timings is not None and self.timings.get("enabled"))
        t_last = [time.perf_counter()]

        def mark(name):
            """optional per-stage wall clock (synchronises the stream: debugging only)"""
            if trace_stages:
                ctx.sync()
                now = time.perf_counter()
                self.timings[name] = self.timings.get(name, 0.0) + (now - t_last[0])
                t_last[0] = now

        # 2. base trace: interpolate, extend, commit
        base_lde, base_coeffs = base_trace.lde(lb, g)
        order = be.BITREV if conv.bitrev_commit else be.NATURAL
        base_tree = Tree.from_matrix(base_lde, order)
        proof.base_root = base_tree.root()
        coin.reseed_with_digest(proof.base_root)

        mark("base_lde_commit")
        # 3-4. challenges -> extension trace
        challenges = [coin.draw() for _ in range(air.num_challenges)]
        proof.challenges = challenges
        ext_trace = build_extension(challenges) if air.num_extension_columns else None
        lde_cols, coeff_cols = list(base_lde.cols), list(base_coeffs.cols)
        ext_lde = ext_tree = None
        if ext_trace is not None:
            ext_lde, ext_coeffs = ext_trace.lde(lb, g)
            ext_tree = Tree.from_matrix(ext_lde, order)
            proof.extension_root = ext_tree.root()
            coin.reseed_with_digest(proof.extension_root)
            lde_cols += ext_lde.cols
            coeff_cols += ext_coeffs.cols

        mark("ext_lde_commit")
        # 5. composition constraint over the LDE domain, then its 2-column LDE
        comp_coeff = coin.draw()
        proof.composition_coeff = comp_coeff
        program, tables, table_desc = air.build_program(n, challenges, comp_coeff)
        if tables is None or hasattr(tables, "ptr"):           # already resident (a buffer of the context)
            d_tables = tables
        else:
            d_tables = ctx.column(tables) if len(tables) else None
        mark("lower_program")
        comp_evals = ctx.alloc(32 * N)
        ctx.eval_quotient(program, d_tables, table_desc, lde_cols, log_n, lb, g, comp_evals)
        mark("quotient")
        # coefficients of H in bit-reversed order: the first half is H0 (even coefficients),
        # the second half H1 (odd), each again bit-reversed — the split is free
        ctx.ntt([comp_evals], log_N, be.INVERSE, g, be.NATURAL, be.BITREV)
        ncomp = conv.composition_columns
        assert ncomp == 1 << lb == 2, "composition split implemented for blowup 2"
        comp_coeffs = [be.DeviceView(comp_evals, 32 * n * k, 32 * n) for k in range(ncomp)]
        comp_lde = be.Matrix.empty(ctx, ncomp, N)
        ctx.evaluate(comp_coeffs, log_n, lb, g, comp_lde.cols)
        comp_tree = Tree.from_matrix(comp_lde, order)
        proof.composition_root = comp_tree.root()
        coin.reseed_with_digest(proof.composition_root)

        mark("composition_lde_commit")
        # 6. out-of-domain point
        z = coin.draw()
        proof.z = z
        mask_col = [c for c, _ in air.mask]
        mask_off = [o for _, o in air.mask]
        proof.ood_trace = ctx.ood_eval(coeff_cols, log_n, mask_col, mask_off, z)
        zc = be.felt(pow(canonical(z), ncomp, be.P))
        proof.ood_composition = ctx.poly_eval(comp_coeffs, log_n, zc)
        coin.reseed_with_field_elements(list(proof.ood_trace) + list(proof.ood_composition))

        mark("ood")
        # 7. DEEP composition
        deep_alpha = coin.draw()
        proof.deep_alpha = deep_alpha
        coeffs = _pow_limbs(deep_alpha, len(air.mask) + ncomp)
        deep = ctx.alloc(32 * N)
        ctx.deep_compose(lde_cols, comp_lde.cols, log_n, lb, g, mask_col, mask_off, proof.ood_trace,
                         coeffs[:len(air.mask)], proof.ood_composition, coeffs[len(air.mask):], z, deep)

        mark("deep")
        # 8. FRI
        layers = fri_commit_phase(ctx, Tree, conv, opt, coin, proof, deep, log_N, n)
        mark("fri")
        # 9. proof of work, queries, openings
        proof.pow_nonce = proof_of_work(ctx, self.claim.coin_kind, coin, opt, self.pow_nonce)
        mark("pow")
        coin.reseed_with_int(proof.pow_nonce)
        positions = coin.draw_queries(opt.num_queries, N)
        proof.query_positions = positions
        # a position is an index into the COMMITTED order; the matrices themselves are in natural order
        nat = [bitrev(p, log_N) for p in positions] if conv.bitrev_commit else positions
        proof.base_rows = ctx.gather_rows(base_lde.cols, nat)
        proof.base_paths, proof.base_path_tags = base_tree.prove(positions)
        if ext_lde is not None:
            proof.extension_rows = ctx.gather_rows(ext_lde.cols, nat)
            proof.extension_paths, proof.extension_path_tags = ext_tree.prove(positions)
        proof.composition_rows = ctx.gather_rows(comp_lde.cols, nat)
        proof.composition_paths, proof.composition_path_tags = comp_tree.prove(positions)
        proof.root_tags = [base_tree.root_tag(), ext_tree.root_tag() if ext_tree is not None else 0, comp_tree.root_tag()]
        fri_open(ctx, conv, opt, proof, layers, positions)
        mark("openings")
        return proof


def fri_commit_phase(ctx, Tree, conv, opt, coin, proof, deep, log_N, n):
    """FRI commit phase on one device (prover.py step 8): commit every layer, draw its challenge, fold; interpolate the
    last layer into the remainder.  Appends to proof.fri_alphas, sets proof.fri_remainder, advances the coin.
    -> [(FriLayer, tree, committed matrix, evaluations)]"""
    order = be.BITREV if conv.bitrev_commit else be.NATURAL
    fold = opt.fri_folding_factor
    log_fold = _log2(fold)
    evals, log_len, offset_int = deep, log_N, conv.lde_offset
    degree_bound = n                       # DEEP polynomial: degree < n
    layers = []
    fri_flags = be.FRI_UNNORMALISED if conv.fri_unnormalised else 0
    while degree_bound > opt.fri_max_remainder_coeffs:
        rows = 1 << (log_len - log_fold)
        cols = [be.DeviceView(evals, 32 * rows * k, 32 * rows) for k in range(fold)]
        if conv.bitrev_commit:
            # committed row r = entries fold*r .. fold*r+fold-1 of the bit-reversed vector = natural row
            # bitrev(r), its entry j at x_r * w_fold^bitrev(j): the natural stride columns, re-ordered
            cols = [cols[bitrev(j, log_fold)] for j in range(fold)]
        layer_matrix = be.Matrix(ctx, cols, rows)
        tree = Tree.from_matrix(layer_matrix, order)
        layer = FriLayer(tree.root(), tree.root_tag(), log_len)
        coin.reseed_with_digest(layer.root)
        alpha = coin.draw()
        if conv.fri_alpha_times_offset:     # the reference folds over the unshifted domain: challenge = draw * layer offset
            alpha = be.felt(canonical(alpha) * offset_int % be.P)
        proof.fri_alphas.append(alpha)
        nxt = ctx.alloc(32 * rows)
        ctx.fri_fold(evals, log_len, fold, alpha, be.felt(offset_int), nxt, fri_flags)     # natural order in memory
        layers.append((layer, tree, layer_matrix, evals))
        evals, log_len = nxt, log_len - log_fold
        offset_int = pow(offset_int, fold, be.P)
        degree_bound //= fold
    # remainder: interpolate the last layer, send its (few) coefficients
    rem = be.Matrix(ctx, [evals], 1 << log_len)
    rem.interpolate(be.felt(1 if conv.remainder_unshifted else offset_int))
    rem_host = rem.to_host()[0]
    assert not np.any(rem_host[max(1, degree_bound):]), "FRI remainder exceeds its degree bound"
    proof.fri_remainder = rem_host[:max(1, degree_bound)]
    coin.reseed_with_field_element_vector(list(proof.fri_remainder))
    return layers


def proof_of_work(ctx, coin_kind, coin, opt, supplied_nonce=None):
    if supplied_nonce is not None:
        from .verifier import _verify_pow
        assert _verify_pow(coin_kind, coin.digest, opt.grinding_factor, supplied_nonce), \
            "the supplied proof-of-work nonce is not valid for this transcript"
        return supplied_nonce
    return ctx.pow_grind(coin_kind, coin.digest, opt.grinding_factor) if opt.grinding_factor else 0


def fri_open(ctx, conv, opt, proof, layers, positions):
    """query phase of FRI: the rows and authentication paths every layer opens for these query positions"""
    log_fold = _log2(opt.fri_folding_factor)
    pos = positions
    for layer, tree, matrix, _ in layers:
        row_bits = layer.log_len - log_fold
        rows = 1 << row_bits
        if conv.bitrev_commit:
            pos = sorted(set(p >> log_fold for p in pos))          # row r holds entries fold*r .. of the vector
            nat_rows = [bitrev(r, row_bits) for r in pos]
        else:
            pos = sorted(set(p % rows for p in pos))
            nat_rows = pos
        layer.positions = pos
        layer.rows = ctx.gather_rows(matrix.cols, nat_rows)
        layer.paths, layer.path_tags = tree.prove(pos)
        proof.fri_layers.append(layer)

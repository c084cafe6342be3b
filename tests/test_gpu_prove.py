"""End-to-end: prove a small valid AIR on the GPU, then verify the proof with an
independent big-integer verifier that replays the transcript with the oracle's
coins and checks the OOD identity, every Merkle opening, the DEEP values and the
FRI folds.  Covers both claim flavours of src/claims.rs."""
import numpy as np
import pytest

from tests import mini_air, pyref
from tests.pyref import P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sandstorm_amd.backend import Context
    c = Context(0)
    yield c
    c.close()


def merkle_verify(oracle, tree_kind, n_friendly, leaf_bytes, index, path, n, root):
    """walk an authentication path with the oracle's node rules (depth = level of the output node)"""
    log_n = n.bit_length() - 1
    cur, k = leaf_bytes, n + index
    for lvl in range(log_n):
        sib = bytes(path[lvl])
        a, b = (cur, sib) if k % 2 == 0 else (sib, cur)
        depth = log_n - 1 - lvl
        if tree_kind == 2 and depth < n_friendly:
            x = oracle.to_mont([int.from_bytes(a, "big") % P, int.from_bytes(b, "big") % P])
            cur = int(oracle.from_mont(oracle.pedersen_hash(x[0], x[1]))).to_bytes(32, "big")
        elif tree_kind == 2:
            cur = pyref.mask_blake(pyref.blake2s(a + b))
        elif tree_kind == 1:
            cur = pyref.mask_keccak(oracle.keccak256(a + b))
        else:
            cur = oracle.keccak256(a + b)
        k >>= 1
    return cur == root


def row_leaf(oracle, kind, row):
    return bytes(oracle.hash_rows(kind, [r[None, :] for r in row])[0])


def setup_case(ctx, oracle, flavour, log_n):
    from sandstorm_amd import backend as be
    from sandstorm_amd.coin import canonical
    from sandstorm_amd.prover import Claim, ProofOptions
    n = 1 << log_n
    air = mini_air.make_air(oracle.to_mont)
    if flavour == "eth":
        claim, params = Claim(air, be.LeafVariantMerkleTree, be.COIN_SOLIDITY), (1, 1, 0, 0)
    else:
        claim, params = Claim(air, be.FriendlyMerkleTree, be.COIN_CAIRO), (2, 3, 1, 22)
    opt = ProofOptions(num_queries=12, grinding_factor=8, fri_max_remainder_coeffs=4)
    seed = bytes(range(32))
    c0, c1 = mini_air.base_trace(n)
    base = be.Matrix.from_host(ctx, [oracle.to_mont(c0), oracle.to_mont(c1)])

    def build_extension(challenges):
        e0 = mini_air.extension_trace(c0, canonical(challenges[0]))
        return be.Matrix.from_host(ctx, [oracle.to_mont(e0)])
    return n, claim, params, opt, seed, base, build_extension


@pytest.mark.parametrize("flavour", ["eth", "cairo"])
@pytest.mark.parametrize("log_n", [5, 9])
def test_prove_and_verify_mini_air(ctx, oracle, flavour, log_n):
    from sandstorm_amd.prover import Prover
    n, claim, params, opt, seed, base, build_extension = setup_case(ctx, oracle, flavour, log_n)
    proof = Prover(ctx, claim, opt).prove(seed, base, build_extension)
    verify_mini_proof(oracle, proof, n, params, opt, seed)


@pytest.mark.parametrize("flavour", ["eth", "cairo"])
def test_prove_and_verify_older_conventions(ctx, oracle, flavour):
    """natural commitment order, normalised fold, shifted remainder: the conventions of the reference's older
    proof file (example/bootloader/bootloader-proof.bin) stay selectable."""
    from sandstorm_amd.prover import Conventions, Prover
    n, claim, params, opt, seed, base, build_extension = setup_case(ctx, oracle, flavour, 9)
    conv = Conventions(bitrev_commit=False, fri_unnormalised=False, remainder_unshifted=False, fri_alpha_times_offset=False)
    proof = Prover(ctx, claim, opt, conv).prove(seed, base, build_extension)
    verify_mini_proof(oracle, proof, n, params, opt, seed, conv)


@pytest.mark.parametrize("flavour", ["eth", "cairo"])
@pytest.mark.parametrize("log_n", [5, 10])
def test_cpp_prover_matches_python_prover_and_verifies(ctx, oracle, flavour, log_n):
    """The C++ host (libsandstorm_host.so: coin, Expr lowering, prover) drives the same
    kernels: its proof must verify, and every transcript value must equal the Python mirror's."""
    from sandstorm_amd import hostlib
    from sandstorm_amd.prover import Prover
    n, claim, params, opt, seed, base, build_extension = setup_case(ctx, oracle, flavour, log_n)
    tree_kind, _, coin_kind, nf = params
    air = hostlib.HostAir(ctx, hostlib.AIR_MINI, log_n)
    proof = hostlib.prove(ctx, air, tree_kind, nf, coin_kind, seed, base.cols, log_n,
                          lambda ch: build_extension(ch).cols, opt)
    air.close()
    verify_mini_proof(oracle, proof, n, params, opt, seed)
    ref = Prover(ctx, claim, opt).prove(seed, base, build_extension)
    assert proof.base_root == ref.base_root and proof.extension_root == ref.extension_root
    assert proof.composition_root == ref.composition_root
    assert np.array_equal(proof.z, ref.z) and np.array_equal(proof.ood_trace, ref.ood_trace)
    assert np.array_equal(proof.ood_composition, ref.ood_composition)
    assert [l.root for l in proof.fri_layers] == [l.root for l in ref.fri_layers]
    assert np.array_equal(proof.fri_remainder, ref.fri_remainder)
    assert proof.pow_nonce == ref.pow_nonce and proof.query_positions == ref.query_positions
    assert np.array_equal(proof.base_rows, ref.base_rows) and np.array_equal(proof.base_paths, ref.base_paths)
    for a, b in zip(proof.fri_layers, ref.fri_layers):
        assert a.positions == b.positions and np.array_equal(a.rows, b.rows) and np.array_equal(a.paths, b.paths)


def verify_mini_proof(oracle, proof, n, params, opt, seed, conv=None):
    """independent verifier: oracle coins + big integers.  conv: prover.Conventions (default: the conventions
    pinned by the reference's shipped proofs - bit-reversed commitment order, unnormalised fold, unshifted remainder)"""
    from sandstorm_amd import air_program as ap
    from sandstorm_amd.prover import Conventions, bitrev
    conv = conv or Conventions()
    log_N = (2 * n).bit_length() - 1
    log_fold = opt.fri_folding_factor.bit_length() - 1
    # exponent of the domain generator at index i of a committed vector of 2^bits entries
    expo = (lambda i, bits: bitrev(i, bits)) if conv.bitrev_commit else (lambda i, bits: i)
    tree_kind, row_kind, coin_kind, nf = params
    N = 2 * n
    # ---- transcript replay with the ORACLE's coin
    coin = oracle.Coin(coin_kind, seed)
    coin.reseed_bytes(proof.base_root)
    gamma = coin.draw()
    assert np.array_equal(gamma, proof.challenges[0])
    coin.reseed_bytes(proof.extension_root)
    alpha = coin.draw()
    assert np.array_equal(alpha, proof.composition_coeff)
    coin.reseed_bytes(proof.composition_root)
    z = coin.draw()
    assert np.array_equal(z, proof.z)
    coin.reseed_felts(np.concatenate([proof.ood_trace, proof.ood_composition]))
    deep_alpha = coin.draw()
    assert np.array_equal(deep_alpha, proof.deep_alpha)
    fri_alphas = []
    for layer in proof.fri_layers:
        coin.reseed_bytes(layer.root)
        fri_alphas.append(coin.draw())
    coin.reseed_felt_vector(proof.fri_remainder)
    assert coin.verify_pow(opt.grinding_factor, proof.pow_nonce)
    assert proof.pow_nonce == coin.grind(opt.grinding_factor)            # smallest nonce
    coin.reseed_int(proof.pow_nonce)
    positions = coin.draw_queries(opt.num_queries, N)
    assert positions == proof.query_positions

    g_, a_, z_, da_ = (int(oracle.from_mont(v)) for v in (gamma, alpha, z, deep_alpha))
    ood_t = [int(v) for v in oracle.from_mont(proof.ood_trace)]
    ood_c = [int(v) for v in oracle.from_mont(proof.ood_composition)]

    # ---- OOD identity: sum_k alpha^k C_k(z) == H0(z^2) + z H1(z^2)
    cell = {m: v for m, v in zip(mini_air.MASK, ood_t)}
    dag = mini_air.composition(n, g_, a_)
    lhs = ap.evaluate(dag, P, z_, lambda c, o: cell[(c, o)], lambda t: mini_air.table_at(n, z_))
    assert lhs == (ood_c[0] + z_ * ood_c[1]) % P

    # ---- openings
    wN, wn = pyref.root_of_unity(N), pyref.root_of_unity(n)
    ncells = len(mini_air.MASK)
    coef = [pow(da_, j, P) for j in range(ncells + 2)]
    fold = opt.fri_folding_factor
    rows0 = N // fold
    l0 = proof.fri_layers[0]
    for qi, q in enumerate(positions):
        x = 3 * pow(wN, expo(q, log_N), P) % P
        for rows, paths, root in ((proof.base_rows, proof.base_paths, proof.base_root),
                                  (proof.composition_rows, proof.composition_paths, proof.composition_root)):
            assert merkle_verify(oracle, tree_kind, nf, row_leaf(oracle, row_kind, rows[qi]), q, paths[qi], N, root)
        # single-column extension matrix: leaves are the raw elements (merkle/mod.rs:113-117)
        sib = q ^ 1
        # (the sibling felt is not part of the path bytes for felt leaves; check the parent chain only)
        trow = list(oracle.from_mont(proof.base_rows[qi])) + list(oracle.from_mont(proof.extension_rows[qi]))
        crow = list(oracle.from_mont(proof.composition_rows[qi]))
        # DEEP value from the opened rows.  The opened row is T(x_q); mask cells with offset 1 need
        # T(x_q w_n) = row at position q + 2, which the verifier gets through the quotient identity:
        #   sum_j c_j (T_col(x) - ood_j) / (x - z w^off)  uses T_col(x) for every cell
        deep = 0
        for j, (c, o) in enumerate(mini_air.MASK):
            deep += coef[j] * (int(trow[c]) - ood_t[j]) * pow(x - z_ * pow(wn, o, P), -1, P)
        for k in range(2):
            deep += coef[ncells + k] * (int(crow[k]) - ood_c[k]) * pow(x - z_ * z_, -1, P)
        r, cidx = (q >> log_fold, q & (fold - 1)) if conv.bitrev_commit else (q % rows0, q // rows0)
        li = l0.positions.index(r)
        assert int(oracle.from_mont(l0.rows[li, cidx])) == deep % P

    # ---- FRI layers fold into each other and into the remainder
    offset = 3
    for li, layer in enumerate(proof.fri_layers):
        L = 1 << layer.log_len
        rows = L // fold
        w, wf = pyref.root_of_unity(L), pyref.root_of_unity(fold)
        a = int(oracle.from_mont(fri_alphas[li]))
        if conv.fri_alpha_times_offset:                # the reference folds over the unshifted domain: challenge = draw * layer offset
            a = a * offset % P
        assert a == int(oracle.from_mont(proof.fri_alphas[li]))
        leaf_kind = row_kind
        for pi, r in enumerate(layer.positions):
            assert merkle_verify(oracle, tree_kind, nf, row_leaf(oracle, leaf_kind, layer.rows[pi]), r, layer.paths[pi], rows, layer.root)
            row_bits = layer.log_len - log_fold
            xr0 = offset * pow(w, expo(r, row_bits), P) % P
            # entry j of a committed row sits at x_r * w_fold^j (natural) or x_r * w_fold^bitrev(j)
            xs = [xr0 * pow(wf, expo(k, log_fold), P) % P for k in range(fold)]
            ys = [int(v) for v in oracle.from_mont(layer.rows[pi])]
            folded = pyref.interpolate_eval(xs, ys, a)
            if conv.fri_unnormalised:
                folded = folded * fold % P
            if li + 1 < len(proof.fri_layers):
                nxt = proof.fri_layers[li + 1]
                nrows = (1 << nxt.log_len) // fold
                nr, slot = (r >> log_fold, r & (fold - 1)) if conv.bitrev_commit else (r % nrows, r // nrows)
                ni = nxt.positions.index(nr)
                assert int(oracle.from_mont(nxt.rows[ni, slot])) == folded
            else:
                rem = [int(v) for v in oracle.from_mont(proof.fri_remainder)]
                xr = pow(pyref.root_of_unity(rows), expo(r, row_bits), P)
                if not conv.remainder_unshifted:
                    xr = xr * pow(offset, fold, P) % P
                assert sum(c * pow(xr, i, P) for i, c in enumerate(rem)) % P == folded
        offset = pow(offset, fold, P)


def _keccak_m20_leaf_hash(vals):
    """LeafVariantMerkleTree<MaskedKeccak256HashFn<20>>: the tree's row hash, on canonical values"""
    from sandstorm_amd import wire
    from sandstorm_amd.coin import keccak256
    data = b"".join((v * wire._R % P).to_bytes(32, "big") for v in vals)
    return keccak256(data)[:20] + bytes(12)


@pytest.mark.parametrize("log_n", [5, 9])
def test_cpp_host_emits_the_reference_wire_format(ctx, oracle, log_n):
    """ssh_prove_wire (C++ Proof::serialize_wire) writes byte for byte what wire.py - whose layout is read off the
    reference's shipped proofs - writes for the Python mirror's proof of the same statement; a FriendlyMerkleTree
    proof is refused (no reference sample of its MixedMerkleDigest encoding)."""
    from sandstorm_amd import hostlib, wire
    from sandstorm_amd._lib import SandstormHipError
    from sandstorm_amd.prover import Prover
    n, claim, params, opt, seed, base, build_extension = setup_case(ctx, oracle, "eth", log_n)
    tree_kind, _, coin_kind, nf = params
    air = hostlib.HostAir(ctx, hostlib.AIR_MINI, log_n)
    raw = hostlib.prove(ctx, air, tree_kind, nf, coin_kind, seed, base.cols, log_n, lambda ch: build_extension(ch).cols, opt,
                        wire=True)
    ref = Prover(ctx, claim, opt).prove(seed, base, build_extension)
    assert raw == wire.serialize(wire.from_proof(ref, _keccak_m20_leaf_hash))
    w = wire.parse(raw)
    assert w.trace_len == n and w.pow_nonce == ref.pow_nonce and len(w.base_openings) == len(ref.query_positions)
    # the CairoVerifierClaim flavour (FriendlyMerkleTree, Cairo coin): MixedMerkleDigest / FriendlyMerkleTreeProof encodings
    # (crypto/src/merkle/mixed.rs:46-101, mod.rs:168-236; source-pinned) - the two hosts write the same bytes, both verifiers accept
    from sandstorm_amd import verifier
    from sandstorm_amd.coin import blake2s256
    from tests.test_verifier import mini_verifier_air
    n2, claim2, params2, opt2, seed2, base2, build_extension2 = setup_case(ctx, oracle, "cairo", log_n)
    raw2 = hostlib.prove(ctx, air, params2[0], params2[3], params2[2], seed2, base2.cols, log_n,
                         lambda ch: build_extension2(ch).cols, opt2, wire=True)
    ref2 = Prover(ctx, claim2, opt2).prove(seed2, base2, build_extension2)

    def blake_leaf(vals):
        return bytes(12) + blake2s256(b"".join((v * wire._R % P).to_bytes(32, "big") for v in vals))[12:]
    assert raw2 == wire.serialize(wire.from_proof(ref2, blake_leaf))
    sec = opt2.num_queries + opt2.grinding_factor
    assert verifier.verify(raw2, mini_verifier_air(), params2[0], params2[2], seed2, required_security_bits=sec) == ref2.query_positions
    assert hostlib.verify(air, params2[0], params2[2], seed2, raw2, required_security_bits=sec) == ref2.query_positions
    air.close()


def test_proof_in_reference_wire_format(ctx, oracle):
    """Our own proof (EthVerifierClaim flavour), serialised in the reference's wire format (sandstorm_amd/wire.py),
    goes through exactly the data-level checks the reference's saved proof passes in tests/golden/
    make_proof_golden.py: it parses back to EOF, every opening climbs to its root at the query position with the
    oracle's hashes, FRI rows chain by position p >> 3(i+1), and serialisation is stable."""
    from sandstorm_amd import wire
    from sandstorm_amd.coin import keccak256
    from sandstorm_amd.prover import Prover
    n, claim, params, opt, seed, base, build_extension = setup_case(ctx, oracle, "eth", 9)
    proof = Prover(ctx, claim, opt).prove(seed, base, build_extension)

    def leaf_hash(vals):            # LeafVariantMerkleTree<MaskedKeccak256HashFn<20>>: row hash of the tree
        m = wire._R
        data = b"".join((v * m % P).to_bytes(32, "big") for v in vals)
        return keccak256(data)[:20] + bytes(12)
    raw = wire.serialize(wire.from_proof(proof, leaf_hash))
    w = wire.parse(raw)
    assert wire.serialize(w) == raw
    assert w.options == [opt.num_queries, 2, opt.grinding_factor, 8, opt.fri_max_remainder_coeffs] and w.trace_len == n

    def climb(cur, path, pos):
        for lvl, sib in enumerate(path):
            a, b = (cur, sib) if ((pos >> lvl) & 1) == 0 else (sib, cur)
            cur = pyref.mask_keccak(oracle.keccak256(a + b))
        return cur

    def rowhash(vals):
        mm = oracle.to_mont(vals)
        return bytes(oracle.hash_rows(1, [mm[k:k + 1] for k in range(len(vals))])[0])
    positions = proof.query_positions
    nq = len(positions)
    ncb = len(w.base_rows) // nq
    for q, p in enumerate(positions):
        for rows, ops, root, nc in ((w.base_rows, w.base_openings, w.base_root, ncb),
                                    (w.composition_rows, w.composition_openings, w.composition_root, 2)):
            o = ops[q]
            leaf = rowhash(rows[nc * q:nc * q + nc])
            assert o.variant == 0 and leaf == o.leaf
            assert climb(leaf, [o.sibling] + o.path, p) == root
        o = w.extension_openings[q]                       # single column: raw-element leaves
        assert o.variant == 1 and o.leaf == w.extension_rows[q]
        pair = [o.leaf, o.sibling] if (p & 1) == 0 else [o.sibling, o.leaf]
        assert climb(rowhash(pair), o.path, p >> 1) == w.extension_root
    for li, layer in enumerate(w.fri_layers):
        ps = sorted(set(pp >> (3 * (li + 1)) for pp in positions))
        assert len(ps) == len(layer.openings) == len(layer.rows) // 8
        for r, pos in enumerate(ps):
            o = layer.openings[r]
            leaf = rowhash(layer.rows[8 * r:8 * r + 8])
            assert leaf == o.leaf and climb(leaf, [o.sibling] + o.path, pos) == layer.root

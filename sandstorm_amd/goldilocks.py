"""A STARK over the 64-bit field p = 2^64 - 2^32 + 1 with challenges in Fq3 (BASELINE.json configs[4]; the claim the
reference builds behind `experimental_claims`, cli/src/main.rs:103-133): prover (every stage a HIP kernel behind the C ABI) and
verifier (host Python, integers) for an AIR given the way layouts/plain.py gives it.

The reference instantiates this claim from un-vendored parts (ministark's generic prover, SHA-256 trees, PublicCoinImpl): there
is nothing to be byte-compatible with, and PARITY IS UNPINNED.  The pipeline is the 252-bit path's (sandstorm_amd/prover.py,
src/lib.rs:75-125), with this library's own choices where the reference is silent:
  * an Fq3-valued column (extension trace, composition) is committed as its three Fp coordinate columns; every committed column
    is therefore an Fp polynomial, and the mask / out-of-domain values / DEEP terms are per coordinate column;
  * trees: Blake2s-256 (the reference's Blake2sHashFn; a quarter of Keccak's cost on this chip) over the rows' little-endian bytes
    (ss_hash_rows_gl64) and ss_merkle_build's SS_TREE_BLAKE2S, natural row order; coin: the Keccak coin of the 252-bit path
    (coin.PublicCoin), a field element = 8 drawn bytes below p.  Options(hash="sha256") takes the parts the reference NAMES for
    the claim instead (cli/src/main.rs:119-120): SHA-256 row digests and trees (SS_HASH_SHA256 / SS_TREE_SHA256: FIPS 180-4, the
    one thing about this claim a public document pins - tests/test_oracle_golden.py, tests/test_goldilocks.py) and a coin of the
    reference's own shape with SHA-256 inside (PublicCoinImpl's internals are ministark's: un-vendored);
  * composition H(x) = H0(x^2) + x H1(x^2) (two Fq3 columns = six coordinate columns), out-of-domain point z^2 for them;
  * FRI: fold 8, the layer's challenge as drawn, values normalised; remainder = coefficients of the last layer.
What is checked instead of parity: each kernel against the oracle (tests/test_goldilocks.py), and that proofs of true statements
verify while tampered ones do not - the verifier recomputes the AIR at the out-of-domain point from the layout's expression
DAG (air_program.evaluate_ext), independently of the device's lowered program."""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import air_program as ap
from . import backend as be
from .coin import PublicCoin, keccak256
from .layouts import plain as F                    # the field helpers live with the layout (mul3, inv3, pow3, root_of_unity ...)

P = F.P
OFFSET = F.GENERATOR                               # LDE coset offset = the field's generator, as on the 252-bit path


@dataclass
class Options:
    """the CLI's defaults (cli/src/main.rs:51-60): 65 queries x 1 bit (blowup 2) + 16 grinding bits = 81 >= the 80 it requires"""
    num_queries: int = 65
    log_blowup: int = 1
    grinding: int = 16
    fold: int = 8
    max_remainder: int = 16
    # "blake2s": this library's choice for the field (Blake2s-256 trees, the Keccak coin of the 252-bit path).  "sha256": the parts
    # the reference names for the claim (cli/src/main.rs:119-120): MatrixMerkleTreeImpl<Sha256HashFn> trees (SS_HASH_SHA256 rows,
    # SS_TREE_SHA256 nodes: FIPS 180-4) and a coin of the reference's own shape with SHA-256 inside (PublicCoinImpl<Fq3, Sha256HashFn>
    # is ministark's, un-vendored: reseed = H(be32(digest + 1) || bytes), draw = H(digest || be32(counter)) as in
    # crypto/src/public_coin/solidity.rs:54-118); the proof-of-work hash stays Keccak (ss_pow_grind's)
    hash: str = "blake2s"


@dataclass
class Air:
    """what the prover and the verifier need of a layout: component (coordinate) column counts, the number of Fq3 challenges,
    the mask over component columns, and the composition DAG for a statement"""
    name: str
    num_base: int                   # base Fp columns
    num_ext: int                    # coordinate columns of the extension trace (3 per Fq3 column)
    num_challenges: int
    mask: list
    # (n, challenges, alpha, tables) -> Expr root; tables: layouts.plain.Tables (or anything with .specs/.device_tables/.value_at)
    composition: object = None
    make_tables: object = None      # (n, log_blowup) -> tables


@dataclass
class Opening:
    rows: np.ndarray                # [nq, width] uint64
    paths: np.ndarray               # [nq, depth, 32] uint8


@dataclass
class FriLayer:
    root: bytes
    log_len: int
    opening: Optional[Opening] = None       # rows of fold Fq3 values at the folded query positions


@dataclass
class Proof:
    options: Options
    trace_len: int
    base_root: bytes = b""
    ext_root: bytes = b""
    comp_root: bytes = b""
    ood_trace: Optional[np.ndarray] = None          # [nmask, 3]
    ood_comp: Optional[np.ndarray] = None           # [6, 3]
    fri_layers: List[FriLayer] = field(default_factory=list)
    remainder: Optional[np.ndarray] = None          # [len, 3] coefficients
    pow_nonce: int = 0
    base: Optional[Opening] = None
    ext: Optional[Opening] = None
    comp: Optional[Opening] = None


def _sha256(data: bytes) -> bytes:
    import hashlib
    return hashlib.sha256(data).digest()


class Coin(PublicCoin):
    def __init__(self, seed: bytes, hash_name: str = "blake2s"):
        super().__init__(be.COIN_SOLIDITY, seed)
        if hash_name == "sha256":
            self._h = _sha256

    def draw_felt(self):
        while True:
            d = self._draw_bytes()
            for k in range(4):
                v = int.from_bytes(d[8 * k:8 * k + 8], "big")
                if v < P:
                    return v

    def draw_fq3(self):
        return (self.draw_felt(), self.draw_felt(), self.draw_felt())

    def reseed_with_fq3s(self, values):
        self.reseed_with_bytes(b"".join(int(c).to_bytes(8, "little") for v in values for c in v))


def statement_digest(statement) -> bytes:
    """the public input as the transcript absorbs it (the reference seeds its coin from the public input, src/lib.rs:118-120
    `P::from_public_input`; ministark's generic coin hashes its serialisation - un-vendored, so the byte image here is this
    library's own): step count, range-check bounds, the memory segments by name, every public-memory cell"""
    if statement is None:
        return bytes(32)
    u64 = lambda v: int(v).to_bytes(8, "big")
    segs = sorted((str(k), v) for k, v in statement.memory_segments.items() if v is not None)
    blob = u64(statement.n_steps) + u64(statement.rc_min) + u64(statement.rc_max) + u64(len(segs))
    for name, (lo, hi) in segs:
        blob += u64(len(name)) + name.encode() + u64(lo) + u64(hi)
    blob += u64(len(statement.public_memory)) + b"".join(u64(a) + u64(v) for a, v in statement.public_memory)
    return keccak256(blob)


def transcript_seed(seed: bytes, opt: Options, trace_len: int, statement=None) -> bytes:
    """the options, the trace length AND the statement are part of the transcript: every challenge depends on the public memory
    and segments being claimed (Fiat-Shamir over the whole statement), and a proof does not verify under other options"""
    return keccak256(bytes(seed) + b"".join(int(v).to_bytes(8, "big") for v in (opt.num_queries, opt.log_blowup, opt.grinding, opt.fold,
                                                                                 opt.max_remainder, trace_len)) + statement_digest(statement)
                     + (b"" if opt.hash == "blake2s" else opt.hash.encode()))


DEFAULT_REQUIRED_SECURITY_BITS = 80                # cli/src/main.rs:66-67


def conjectured_security_bits(opt: Options, trace_len: int) -> int:
    """min of the query term (one bit per query and doubling of the blowup, plus the grinding bits), the field term (the
    challenges live in Fq3: 192 bits less the size of the evaluation domain) and the hash term (256-bit digests: 128) - the
    shape of verifier.conjectured_security_bits on the 252-bit path"""
    log_N = max(1, int(trace_len).bit_length() - 1 + opt.log_blowup)
    return min(opt.num_queries * opt.log_blowup + opt.grinding, 192 - log_N, 128)


def fri_shape(n_lde, opt: Options):
    """-> number of FRI layers; the remainder is the last layer (len <= max_remainder * blowup)"""
    layers, length = 0, n_lde
    while length > opt.max_remainder << opt.log_blowup:
        if length % opt.fold:
            raise ValueError("layer length %d is not a multiple of the folding factor" % length)
        length //= opt.fold
        layers += 1
    return layers, length


def _powers3(a, count):
    out, cur = [], (1, 0, 0)
    for _ in range(count):
        out.append(cur)
        cur = F.mul3(cur, a)
    return out


# ---- prover ----------------------------------------------------------------------------------------------------------------------
class Prover:
    """columns are torch int64 tensors on the device; ctx: a backend.Context bound to the SAME stream torch works on - an explicit
    torch.cuda.Stream made current (torch's default stream has handle 0, which the C ABI reads as "the context's own stream")"""

    def __init__(self, ctx, air: Air, options: Options = None):
        self.ctx, self.air, self.opt = ctx, air, options or Options()
        if self.opt.hash not in ("blake2s", "sha256"):
            raise ValueError("Options.hash: 'blake2s' or 'sha256'")
        self._row_hash, self._tree = (be.HASH_SHA256, be.TREE_SHA256) if self.opt.hash == "sha256" else (be.HASH_BLAKE2S, be.TREE_BLAKE2S)

    def _commit(self, cols, n_rows):
        import torch
        dig = torch.empty((n_rows, 32), dtype=torch.uint8, device=cols[0].device)          # every byte is written by the kernels
        nodes = torch.empty((2 * n_rows, 32), dtype=torch.uint8, device=cols[0].device)
        # rows of more than 16 columns do not occur here (<= 8 trace, 6 composition coordinate columns)
        self.ctx.hash_rows_gl64(cols, 1, n_rows, dig, self._row_hash)
        root, _ = self.ctx.merkle_build(self._tree, 0, be.LEAF_DIGEST, dig, n_rows, nodes)
        return root, nodes

    def _open(self, cols, nodes, n_rows, positions, seg_len=1):
        rows = self.ctx.gather_rows_gl64(cols, seg_len, n_rows, positions).reshape(len(positions), -1)
        paths, _ = self.ctx.merkle_open(nodes, None, n_rows, positions)
        return Opening(rows, paths)

    def prove(self, seed: bytes, base_cols, build_extension, tables=None, statement=None):
        """base_cols: the base columns [n] (device); build_extension(challenges) -> the extension trace's coordinate columns.
        statement: passed through to air.composition (hints come from it)"""
        import torch
        ctx, air, opt = self.ctx, self.air, self.opt
        n = base_cols[0].shape[0]
        log_n, lb = n.bit_length() - 1, opt.log_blowup
        if lb != 1:
            raise ValueError("the composition split (even / odd coefficients) is written for blowup 2")
        N = n << lb
        dev = base_cols[0].device
        new = lambda rows, width=None: torch.empty((rows,) if width is None else (rows, width), dtype=torch.int64, device=dev)   # fully written below
        coin = Coin(transcript_seed(seed, opt, n, statement), opt.hash)
        proof = Proof(opt, n)

        def extend(cols):
            ev, co = [new(N) for _ in cols], [new(n) for _ in cols]
            ctx.lde_gl64(list(cols), log_n, lb, OFFSET, ev, co)
            return ev, co
        # 1-2. base and extension traces
        base_ev, base_co = extend(base_cols)
        proof.base_root, base_nodes = self._commit(base_ev, N)
        coin.reseed_with_digest(proof.base_root)
        challenges = [coin.draw_fq3() for _ in range(air.num_challenges)]
        ext_cols = list(build_extension(challenges)) if air.num_ext else []
        ext_ev, ext_co = extend(ext_cols) if ext_cols else ([], [])
        if ext_cols:
            proof.ext_root, ext_nodes = self._commit(ext_ev, N)
            coin.reseed_with_digest(proof.ext_root)
        trace_ev, trace_co = base_ev + ext_ev, base_co + ext_co
        # 3-4. composition: the lowered program over the LDE domain, then H = H0(x^2) + x H1(x^2) as six coordinate columns
        alpha = coin.draw_fq3()
        tables = tables or air.make_tables(n, lb)
        root = air.composition(n, challenges, alpha, tables, statement)
        prog = ap.lower(root, P, ext=True, symbols=tables.symbols)
        tvals, tdesc = tables.device_tables()
        d_tables = torch.from_numpy(tvals.view(np.int64)).to(dev)
        q = new(N, 3)
        ctx.eval_quotient_gl64x3(np.array(prog.code, dtype=np.uint32), np.array(prog.consts, dtype=np.uint64).reshape(-1, 3), prog.n_slots,
                                 d_tables, tdesc, trace_ev, log_n, lb, OFFSET, q)
        comp_co, comp_ev = [], []
        qc = [q[:, t].contiguous() for t in range(3)]
        ctx.ntt_gl64(qc, log_n + lb, be.INVERSE, OFFSET, be.NATURAL, be.BITREV)       # bit-reversed coefficients: even ones first
        for half in range(2):
            for t in range(3):
                comp_co.append(qc[t][half * n:(half + 1) * n].contiguous())
        for c in comp_co:
            padded = torch.zeros(N, dtype=torch.int64, device=dev)
            padded[::1 << lb] = c                                                    # zero-padded to N in bit-reversed order
            ctx.ntt_gl64([padded], log_n + lb, be.FORWARD, OFFSET, be.BITREV, be.NATURAL)
            comp_ev.append(padded)
        proof.comp_root, comp_nodes = self._commit(comp_ev, N)
        coin.reseed_with_digest(proof.comp_root)
        # 5. out-of-domain evaluations
        z = coin.draw_fq3()
        zc = F.pow3(z, 2)
        mc, mo = [c for c, _ in air.mask], [o for _, o in air.mask]
        proof.ood_trace = ctx.ood_eval_gl64x3(trace_co, log_n, mc, mo, z)
        proof.ood_comp = ctx.ood_eval_gl64x3(comp_co, log_n, list(range(6)), [0] * 6, zc)
        coin.reseed_with_fq3s([tuple(int(v) for v in r) for r in proof.ood_trace] + [tuple(int(v) for v in r) for r in proof.ood_comp])
        # 6. DEEP composition
        gamma = coin.draw_fq3()
        coefs = np.array(_powers3(gamma, len(air.mask) + 6), dtype=np.uint64)
        layer = new(N, 3)
        ctx.deep_compose_gl64x3(trace_ev, comp_ev, log_n, lb, OFFSET, mc, mo, proof.ood_trace, coefs[:len(air.mask)], proof.ood_comp,
                                coefs[len(air.mask):], z, zc, layer)
        # 7. FRI
        n_layers, _ = fri_shape(N, opt)
        ll, off, layers = log_n + lb, OFFSET, []
        log_fold = opt.fold.bit_length() - 1
        for _ in range(n_layers):
            rows = (1 << ll) // opt.fold
            segs = [be.DeviceView(_Raw(layer), 24 * k * rows, 24 * rows) for k in range(opt.fold)]
            dig = torch.empty((rows, 32), dtype=torch.uint8, device=dev)
            nodes = torch.empty((2 * rows, 32), dtype=torch.uint8, device=dev)
            ctx.hash_rows_gl64(segs, 3, rows, dig, self._row_hash)
            root_l, _ = ctx.merkle_build(self._tree, 0, be.LEAF_DIGEST, dig, rows, nodes)
            coin.reseed_with_digest(root_l)
            a = coin.draw_fq3()
            nxt = new(rows, 3)
            ctx.fri_fold_gl64x3(layer, ll, opt.fold, a, off, nxt)
            layers.append((layer, nodes, rows, segs))
            proof.fri_layers.append(FriLayer(root_l, ll))
            layer, ll, off = nxt, ll - log_fold, pow(off, opt.fold, P)
        # remainder: interpolate the last layer (tiny) on the host
        last = layer.cpu().numpy().view(np.uint64)
        L = last.shape[0]
        winv, oinv = pow(F.root_of_unity(ll), -1, P), pow(off, -1, P)
        rem = np.zeros((L, 3), dtype=np.uint64)
        for t in range(3):
            vals = [int(v) for v in last[:, t]]
            for k in range(L):                                                         # coefficient k = (1/L) sum_j v_j (w^-k)^j * off^-k
                acc, wk, cur = 0, pow(winv, k, P), 1
                for v in vals:
                    acc, cur = (acc + v * cur) % P, cur * wk % P
                rem[k, t] = acc * pow(L, -1, P) % P * pow(oinv, k, P) % P
        if rem[L >> lb:].any():
            raise ValueError("the FRI remainder is not of low degree: the DEEP composition is not the polynomial it must be")
        proof.remainder = rem
        coin.reseed_with_fq3s([tuple(int(v) for v in r) for r in rem])
        # 8. proof of work, queries, openings
        proof.pow_nonce = ctx.pow_grind(be.COIN_SOLIDITY, coin.digest, opt.grinding) if opt.grinding else 0
        coin.reseed_with_int(proof.pow_nonce)
        positions = coin.draw_queries(opt.num_queries, N)
        proof.base = self._open(base_ev, base_nodes, N, positions)
        if ext_cols:
            proof.ext = self._open(ext_ev, ext_nodes, N, positions)
        proof.comp = self._open(comp_ev, comp_nodes, N, positions)
        pos = positions
        for (lay, nodes, rows, segs), fl in zip(layers, proof.fri_layers):
            pos = sorted(set(p % rows for p in pos))
            fl.opening = self._open(segs, nodes, rows, pos, seg_len=3)
        return proof


class _Raw:
    """a torch tensor seen as the parent of DeviceViews (ptr / nbytes)"""

    def __init__(self, t):
        self.t, self.ptr, self.nbytes, self.ctx = t, t.data_ptr(), t.numel() * t.element_size(), None


# ---- a proof as a dictionary of arrays (numpy.savez): fixtures, transport ------------------------------------------------------------
def proof_to_arrays(p: Proof) -> dict:
    o = p.options
    d = {"options": np.array([o.num_queries, o.log_blowup, o.grinding, o.fold, o.max_remainder, p.trace_len, p.pow_nonce], dtype=np.uint64),
         "roots": np.frombuffer(p.base_root + (p.ext_root or bytes(32)) + p.comp_root, dtype=np.uint8).copy(),
         "has_ext": np.array([1 if p.ext is not None else 0], dtype=np.uint8),
         "ood_trace": p.ood_trace, "ood_comp": p.ood_comp, "remainder": p.remainder,
         "fri_roots": np.frombuffer(b"".join(fl.root for fl in p.fri_layers), dtype=np.uint8).copy(),
         "fri_log_len": np.array([fl.log_len for fl in p.fri_layers], dtype=np.uint32)}
    if o.hash != "blake2s":                 # (absent = the default: the fixtures made before the option existed stay as they are)
        d["hash"] = np.frombuffer(o.hash.encode(), dtype=np.uint8).copy()
    for name, op in [("base", p.base), ("ext", p.ext), ("comp", p.comp)] + [("fri%d" % k, fl.opening) for k, fl in enumerate(p.fri_layers)]:
        if op is not None:
            d[name + "_rows"], d[name + "_paths"] = op.rows, op.paths
    return d


def proof_from_arrays(d) -> Proof:
    o = [int(v) for v in d["options"]]
    roots = bytes(d["roots"])
    p = Proof(Options(*o[:5], hash=bytes(d["hash"]).decode() if "hash" in d else "blake2s"), o[5], roots[:32], roots[32:64] if int(d["has_ext"][0]) else b"", roots[64:96], np.array(d["ood_trace"]),
              np.array(d["ood_comp"]), [], np.array(d["remainder"]), o[6])
    opening = lambda name: Opening(np.array(d[name + "_rows"]), np.array(d[name + "_paths"])) if name + "_rows" in d else None
    p.base, p.ext, p.comp = opening("base"), opening("ext"), opening("comp")
    fr = bytes(d["fri_roots"])
    for k, ll in enumerate(d["fri_log_len"]):
        p.fri_layers.append(FriLayer(fr[32 * k:32 * k + 32], int(ll), opening("fri%d" % k)))
    return p


# ---- verifier --------------------------------------------------------------------------------------------------------------------
class VerificationError(Exception):
    pass


def _need(cond, what):
    if not cond:
        raise VerificationError(what)


def _tree_hash(data: bytes, hash_name: str = "blake2s") -> bytes:
    import hashlib
    return hashlib.sha256(data).digest() if hash_name == "sha256" else hashlib.blake2s(data).digest()


def _climb(leaf, path, pos, hash_name="blake2s"):
    node = leaf
    for lvl in range(path.shape[0]):
        sib = bytes(path[lvl])
        node = _tree_hash(node + sib, hash_name) if ((pos >> lvl) & 1) == 0 else _tree_hash(sib + node, hash_name)
    return node


def _check_opening(opening, width, positions, depth, root, what, hash_name="blake2s"):
    _need(opening is not None and isinstance(opening.rows, np.ndarray) and isinstance(opening.paths, np.ndarray) and
          opening.rows.dtype == np.uint64 and opening.paths.dtype == np.uint8, what + ": dtype")
    _need(opening.rows.shape == (len(positions), width) and opening.paths.shape == (len(positions), depth, 32), what + ": shape")
    for q, pos in enumerate(positions):
        leaf = _tree_hash(b"".join(int(v).to_bytes(8, "little") for v in opening.rows[q]), hash_name)
        _need(_climb(leaf, opening.paths[q], pos, hash_name) == root, what + ": authentication path does not reach the root")


def _verify_pow(digest, bits, nonce):
    prefix = keccak256((0x0123456789ABCDED).to_bytes(8, "big") + digest + bytes([bits]))
    return int.from_bytes(keccak256(prefix + int(nonce).to_bytes(8, "big"))[:8], "big") >> (64 - bits) == 0 if bits else True


def verify(proof: Proof, air: Air, seed: bytes, statement=None, expected_options: Options = None,
           required_security_bits=DEFAULT_REQUIRED_SECURITY_BITS):
    """raises VerificationError naming the failed check; -> the query positions.  The proof is untrusted - its options
    included: a proof whose options conjecture fewer than `required_security_bits` (the CLI's 80 by default) is rejected, and
    any arithmetic, conversion or indexing accident a malformed proof provokes is a rejection too, never another exception.
    `statement` (the public input) is absorbed into the transcript before anything else: the challenges depend on it."""
    try:
        return _verify(proof, air, seed, statement, expected_options, required_security_bits)
    except VerificationError:
        raise
    except (ValueError, IndexError, KeyError, OverflowError, TypeError, AttributeError, ZeroDivisionError) as e:
        raise VerificationError("malformed proof: %s: %s" % (type(e).__name__, e))


def _u64_array(a, shape, what):
    """an array of the proof: dtype uint64 (or uint8 for digests) and exactly this shape (None = any length)"""
    _need(isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.ndim == len(shape) and
          all(w is None or w == h for w, h in zip(shape, a.shape)), what + ": not a uint64 array of the expected shape")
    return a


def _verify(proof, air, seed, statement, expected_options, required_security_bits):
    opt = proof.options
    _need(all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in
              (opt.num_queries, opt.log_blowup, opt.grinding, opt.fold, opt.max_remainder, proof.trace_len, proof.pow_nonce)), "options: not integers")
    if expected_options is not None:
        _need(opt == expected_options, "options are not the expected ones")
    _need(1 <= opt.num_queries <= 256 and opt.fold in (2, 4, 8, 16) and 1 <= opt.log_blowup <= 4 and 1 <= opt.max_remainder <= 1 << 16
          and 0 <= opt.grinding <= 40, "options out of range")
    n = int(proof.trace_len)
    _need(16 <= n <= 1 << (32 - opt.log_blowup) and n & (n - 1) == 0, "trace length")
    _need(0 <= int(proof.pow_nonce) < 1 << 64, "proof-of-work nonce out of range")
    sec = conjectured_security_bits(opt, n)
    _need(sec >= required_security_bits, "the proof's options conjecture %d bits of security, %d required" % (sec, required_security_bits))
    log_n, lb = n.bit_length() - 1, opt.log_blowup
    N = n << lb
    _need(lb == 1, "the composition split is written for blowup 2")
    _need(all(isinstance(r, (bytes, bytearray)) and len(r) == 32 for r in [proof.base_root, proof.comp_root] + ([proof.ext_root] if air.num_ext else [])),
          "roots: 32 bytes each")
    _u64_array(proof.ood_trace, (len(air.mask), 3), "out-of-domain trace values")
    _u64_array(proof.ood_comp, (6, 3), "out-of-domain composition values")
    _u64_array(proof.remainder, (None, 3), "FRI remainder")
    ncols = air.num_base + air.num_ext
    _need(opt.hash in ("blake2s", "sha256"), "options: hash")
    coin = Coin(transcript_seed(seed, opt, n, statement), opt.hash)
    coin.reseed_with_digest(proof.base_root)
    challenges = [coin.draw_fq3() for _ in range(air.num_challenges)]
    if air.num_ext:
        coin.reseed_with_digest(proof.ext_root)
    alpha = coin.draw_fq3()
    coin.reseed_with_digest(proof.comp_root)
    z = coin.draw_fq3()
    zc = F.pow3(z, 2)
    _need(proof.ood_trace is not None and proof.ood_trace.shape == (len(air.mask), 3) and proof.ood_comp.shape == (6, 3), "out-of-domain values: shape")
    ood_t = {cell: tuple(int(v) for v in proof.ood_trace[j]) for j, cell in enumerate(air.mask)}
    ood_c = [tuple(int(v) for v in r) for r in proof.ood_comp]
    _need(all(c < P for v in list(ood_t.values()) + ood_c for c in v), "out-of-domain values: range")
    # the AIR identity at z: sum_k alpha^k C_k(z) multiplier_k(z) = H0(z^2) + z H1(z^2), H_h = sum_t X^t comp[3 h + t]
    tables = air.make_tables(n, lb)
    root = air.composition(n, challenges, alpha, tables, statement)
    g = F.root_of_unity(log_n)
    lhs = ap.evaluate_ext(root, P, z, lambda c, o: ood_t[(c, o)], lambda t: tables.value_at(tables.specs[t], z), symbols=tables.symbols)
    Xk = [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
    H = [(0, 0, 0), (0, 0, 0)]
    for h in range(2):
        for t in range(3):
            H[h] = F.add3(H[h], F.mul3(Xk[t], ood_c[3 * h + t]))
    _need(tuple(lhs) == F.add3(H[0], F.mul3(z, H[1])), "the out-of-domain values do not satisfy the AIR")
    coin.reseed_with_fq3s([ood_t[c] for c in air.mask] + ood_c)
    gamma = coin.draw_fq3()
    coefs = _powers3(gamma, len(air.mask) + 6)
    # FRI transcript
    try:
        n_layers, rem_len = fri_shape(N, opt)
    except ValueError as e:
        raise VerificationError("options: %s" % e)
    _need(len(proof.fri_layers) == n_layers, "number of FRI layers")
    fri_alphas, ll = [], log_n + lb
    log_fold = opt.fold.bit_length() - 1
    for fl in proof.fri_layers:
        _need(fl.log_len == ll, "FRI layer length")
        coin.reseed_with_digest(fl.root)
        fri_alphas.append(coin.draw_fq3())
        ll -= log_fold
    rem = proof.remainder
    _need(rem is not None and rem.shape == (rem_len, 3) and not rem[rem_len >> lb:].any() and all(int(c) < P for c in rem.flat), "FRI remainder: shape / degree")
    coin.reseed_with_fq3s([tuple(int(v) for v in r) for r in rem])
    _need(_verify_pow(coin.digest, opt.grinding, proof.pow_nonce), "proof of work")
    coin.reseed_with_int(proof.pow_nonce)
    positions = coin.draw_queries(opt.num_queries, N)
    # trace / composition openings, DEEP value at every query
    _check_opening(proof.base, air.num_base, positions, log_n + lb, proof.base_root, "base trace", opt.hash)
    if air.num_ext:
        _check_opening(proof.ext, air.num_ext, positions, log_n + lb, proof.ext_root, "extension trace", opt.hash)
    _check_opening(proof.comp, 6, positions, log_n + lb, proof.comp_root, "composition trace", opt.hash)
    wN = F.root_of_unity(log_n + lb)
    deep_at = {}
    for q, pos in enumerate(positions):
        x = OFFSET * pow(wN, pos, P) % P
        row = [int(v) for v in proof.base.rows[q]] + ([int(v) for v in proof.ext.rows[q]] if air.num_ext else [])
        acc = (0, 0, 0)
        for j, (c, o) in enumerate(air.mask):
            den = F.sub3((x, 0, 0), F.scale3(z, pow(g, o % n, P)))
            acc = F.add3(acc, F.mul3(F.mul3(coefs[j], F.sub3((row[c], 0, 0), ood_t[(c, o)])), F.inv3(den)))
        dinv = F.inv3(F.sub3((x, 0, 0), zc))
        for k in range(6):
            acc = F.add3(acc, F.mul3(F.mul3(coefs[len(air.mask) + k], F.sub3((int(proof.comp.rows[q][k]), 0, 0), ood_c[k])), dinv))
        deep_at[pos] = acc
    # FRI: layer rows open, contain the value carried from the layer below, and fold to the next
    carried, pos, ll, off = deep_at, positions, log_n + lb, OFFSET
    for li, fl in enumerate(proof.fri_layers):
        rows = (1 << ll) // opt.fold
        folded = sorted(set(p % rows for p in pos))
        _check_opening(fl.opening, 3 * opt.fold, folded, ll - log_fold, fl.root, "FRI layer %d" % li, opt.hash)
        wL, nxt = F.root_of_unity(ll), {}
        wf_inv = pow(F.root_of_unity(log_fold), -1, P)
        for qi, j in enumerate(folded):
            vals = [tuple(int(v) for v in fl.opening.rows[qi][3 * k:3 * k + 3]) for k in range(opt.fold)]
            for p_ in pos:
                if p_ % rows == j:
                    _need(vals[p_ // rows] == tuple(carried[p_]), "FRI layer %d does not continue the layer below at position %d" % (li, p_))
            # interpolant of the row over x_j <w_fold>, evaluated at alpha: coefficient t = (1/fold) x_j^-t sum_m v_m w_fold^(-t m)
            xj_inv = pow(off * pow(wL, j, P) % P, -1, P)
            acc = (0, 0, 0)
            for t in range(opt.fold - 1, -1, -1):
                s = (0, 0, 0)
                for m, v in enumerate(vals):
                    s = F.add3(s, F.scale3(v, pow(wf_inv, t * m, P)))
                coef = F.scale3(s, pow(xj_inv, t, P) * pow(opt.fold, -1, P) % P)
                acc = F.add3(F.mul3(acc, fri_alphas[li]), coef)
            nxt[j] = acc
        carried, pos, ll, off = nxt, folded, ll - log_fold, pow(off, opt.fold, P)
    # the remainder polynomial at the last positions
    wL = F.root_of_unity(ll)
    for p_ in pos:
        x = off * pow(wL, p_, P) % P
        acc = (0, 0, 0)
        for k in range(rem.shape[0] - 1, -1, -1):
            acc = F.add3(F.scale3(acc, x), tuple(int(v) for v in rem[k]))
        _need(acc == tuple(carried[p_]), "the remainder does not continue the last FRI layer at position %d" % p_)
    return positions


def plain_extension_on_device(ctx, base_cols, challenges, check=True):
    """Trace::build_extension_columns of the plain layout (layouts/src/plain/trace.rs:274-330) on the device: the memory and
    range-check running quotients interleaved in one Fq3 column -> its three coordinate columns (torch tensors) and the last
    value of the memory product (the public-memory quotient of a true statement)"""
    import torch
    npc, mem, rc = base_cols[F.COL_NPC], base_cols[F.COL_MEMORY], base_cols[F.COL_RANGE_CHECK]
    n = npc.shape[0]
    out = [torch.zeros(n, dtype=torch.int64, device=npc.device) for _ in range(3)]
    last_mem = ctx.running_product_gl64x3(npc, npc[1:], mem, mem[1:], F.MEMORY_STEP, n // F.MEMORY_STEP, challenges[F.MEM_Z], challenges[F.MEM_A], out,
                                          F.MEMORY_STEP, 0)
    last_rc = ctx.running_product_gl64x3(rc[F.RangeCheck.OFF_DST:], None, rc[F.RangeCheck.ORDERED:], None, F.RANGE_CHECK_STEP, n // F.RANGE_CHECK_STEP,
                                         challenges[F.RC_Z], None, out, F.RANGE_CHECK_STEP, 1)
    if check and last_rc != (1, 0, 0):
        raise ValueError("the range-check permutation does not close")
    return out, last_mem


# ---- the plain layout as an Air ------------------------------------------------------------------------------------------------
def plain_air():
    """layouts/plain.py behind the Air interface; `statement` = its PublicInput"""
    def composition(n, challenges, alpha, tables, pi):
        hints = F.Hints.from_public_input(pi, challenges, n)
        return F.composition(n, hints, challenges, alpha, tables)
    return Air("plain", F.NUM_BASE_COLUMNS, 3 * F.NUM_EXTENSION_COLUMNS, F.NUM_CHALLENGES, F.mask(), composition, lambda n, lb: F.Tables(n, lb))

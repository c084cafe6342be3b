"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/liboracle.so (the CPU restatement of the
reference's hot path, see oracle/oracle.h).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never imported by
the sandstorm_amd package.

Field elements cross this boundary as numpy uint64 arrays of shape (..., 4):
little-endian limbs, Montgomery form (R = 2^256) — the same image the HIP
library uses.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

P = 2**251 + 17 * 2**192 + 1
R = 2**256
R_MOD_P = R % P
R_INV = pow(R, -1, P)


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


class _FP(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class _Coin(C.Structure):
    _fields_ = [("kind", C.c_int), ("digest", C.c_uint8 * 32), ("counter", C.c_uint64)]


class AirProgram(C.Structure):
    """Mirror of ss_air_program (include/sandstorm_hip.h)."""
    _fields_ = [("code", C.POINTER(C.c_uint32)), ("n_instr", C.c_uint32),
                ("consts", C.POINTER(C.c_uint64)), ("n_consts", C.c_uint32),
                ("d_tables", C.POINTER(C.c_uint64)), ("table_desc", C.POINTER(C.c_uint32)),
                ("n_tables", C.c_uint32), ("n_slots", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.or_init()
        _lib.or_pedersen_hash.restype = _FP
        _lib.or_pedersen_hash.argtypes = [_FP, _FP]
        _lib.or_pedersen_hash_bitwise.restype = _FP
        _lib.or_pedersen_hash_bitwise.argtypes = [_FP, _FP]
        _lib.or_mulmod_chain.restype = _FP
        _lib.or_mulmod_chain.argtypes = [_FP, _FP, C.c_uint64]
        _lib.or_pedersen_hash_elements.restype = _FP
        _lib.or_poly_eval.restype = _FP
        _lib.or_coin_draw.restype = _FP
        _lib.or_coin_grind.restype = C.c_uint64
        _lib.or_coin_grind.argtypes = [C.POINTER(_Coin), C.c_uint]
        _lib.or_coin_verify_pow.argtypes = [C.POINTER(_Coin), C.c_uint, C.c_uint64]
    return _lib


# ---------------------------------------------------------------- conversions
def to_mont(v):
    """python int(s) -> uint64[..., 4] Montgomery limbs."""
    a = np.asarray(v, dtype=object)
    raw = b"".join((((int(x) % P) * R_MOD_P) % P).to_bytes(32, "little") for x in a.reshape(-1))
    return np.frombuffer(raw, dtype="<u8").astype(np.uint64).reshape(a.shape + (4,))


def from_mont(arr):
    """uint64[..., 4] Montgomery limbs -> object array of python ints (canonical)."""
    a = np.ascontiguousarray(arr, dtype="<u8")
    raw = a.tobytes()
    out = np.empty(a.size // 4, dtype=object)
    for i in range(len(out)):
        out[i] = (int.from_bytes(raw[32 * i:32 * i + 32], "little") * R_INV) % P
    return out.reshape(a.shape[:-1])


def _fp(limbs):
    f = _FP()
    for k in range(4):
        f.l[k] = int(limbs[k])
    return f


def _fp_out(f):
    return np.array([f.l[k] for k in range(4)], dtype=np.uint64)


def _ptr(a, ty=C.c_void_p):
    return a.ctypes.data_as(ty)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


# ------------------------------------------------------------------------ NTT
def ntt(col, inverse=False, offset=None):
    """natural-order in/out NTT of one column (n,4) over offset*<w_n>."""
    a = _c(col).copy()
    log_n = (a.shape[0]).bit_length() - 1
    off = _fp(offset) if offset is not None else None
    fn = lib().or_ntt_inverse if inverse else lib().or_ntt_forward
    fn(_ptr(a), C.c_uint(log_n), C.byref(off) if off is not None else None)
    return a


def bitrev_permute(col):
    a = _c(col).copy()
    lib().or_bitrev_permute(_ptr(a), C.c_uint(a.shape[0].bit_length() - 1))
    return a


def lde(col, log_blowup, offset):
    a = _c(col)
    n = a.shape[0]
    log_n = n.bit_length() - 1
    ev = np.zeros((n << log_blowup, 4), dtype=np.uint64)
    co = np.zeros((n, 4), dtype=np.uint64)
    off = _fp(offset)
    lib().or_lde(_ptr(a), C.c_uint(log_n), C.c_uint(log_blowup), C.byref(off), _ptr(ev), _ptr(co))
    return ev, co


def poly_eval(coeffs, x):
    a = _c(coeffs)
    return _fp_out(lib().or_poly_eval(_ptr(a), C.c_size_t(a.shape[0]), _fp(x)))


# --------------------------------------------------------------------- hashes
def keccak256(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().or_keccak256(data, C.c_size_t(len(data)), out)
    return bytes(out)


def sha256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().or_sha256(bytes(data), C.c_size_t(len(data)), out)
    return out.raw


def blake2s256(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().or_blake2s256(data, C.c_size_t(len(data)), out)
    return bytes(out)


def hash_rows(kind, cols):
    """cols: list of (n,4) arrays -> (n,32) uint8 digests."""
    cols = [_c(c) for c in cols]
    n = cols[0].shape[0]
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    out = np.zeros((n, 32), dtype=np.uint8)
    lib().or_hash_rows(C.c_int(kind), ptrs, C.c_size_t(len(cols)), C.c_size_t(n), _ptr(out))
    return out


def pedersen_hash(a, b):
    return _fp_out(lib().or_pedersen_hash(_fp(a), _fp(b)))


def pedersen_hash_bitwise(a, b):
    """the definition (one mixed addition per set bit) the 4-bit-window form of pedersen_hash is held to"""
    return _fp_out(lib().or_pedersen_hash_bitwise(_fp(a), _fp(b)))


def mulmod_ns(iters=2_000_000):
    """nanoseconds per Montgomery product of the port on ONE core (a dependent chain): bench.py's cpu_baseline reports it"""
    import time
    x, y = to_mont([0x1234567])[0], to_mont([0x7654321 << 180])[0]
    lib().or_mulmod_chain(_fp(x), _fp(y), 1000)
    t0 = time.perf_counter()
    lib().or_mulmod_chain(_fp(x), _fp(y), iters)
    return (time.perf_counter() - t0) / iters * 1e9


def pedersen_hash_elements(elems):
    a = _c(elems)
    return _fp_out(lib().or_pedersen_hash_elements(_ptr(a), C.c_size_t(a.shape[0])))


def pedersen_doublings(k, count):
    xs = np.zeros((count, 4), dtype=np.uint64)
    ys = np.zeros((count, 4), dtype=np.uint64)
    lib().or_pedersen_doublings(C.c_int(k), C.c_size_t(count), _ptr(xs), _ptr(ys))
    return xs, ys


# --------------------------------------------------------------------- merkle
def merkle_build(tree, n_friendly, leaf_kind, leaves):
    """leaves: (n,32) uint8 digests or (n,4) uint64 felts -> (nodes (2n,32) u8, tags (2n,) u8)."""
    lv = np.ascontiguousarray(leaves)
    n = lv.shape[0]
    nodes = np.zeros((2 * n, 32), dtype=np.uint8)
    tags = np.zeros(2 * n, dtype=np.uint8)
    lib().or_merkle_build(C.c_int(tree), C.c_uint(n_friendly), C.c_int(leaf_kind), _ptr(lv),
                          C.c_size_t(n), _ptr(nodes), _ptr(tags))
    return nodes, tags


# ----------------------------------------------------------------- FRI / DEEP
FRI_BITREV_ROWS, FRI_UNNORMALISED = 1, 2


def fri_fold(evals, fold, alpha, offset, flags=0):
    a = _c(evals)
    n = a.shape[0]
    out = np.zeros((n // fold, 4), dtype=np.uint64)
    lib().or_fri_fold_ex(_ptr(a), C.c_uint(n.bit_length() - 1), C.c_uint(fold), _fp(alpha),
                         _fp(offset), C.c_uint(flags), _ptr(out))
    return out


# ------------------------------------------------------------- extension trace
def permutation_product(num, den, count, z, alpha, out, out_stride=1, out_off=0):
    """num / den: (column [len, 4], stride, a_off, v_off or -1).  Writes out[i*out_stride + out_off] in place and
    returns the last value (layouts/src/recursive/trace.rs:712-720, 766-769)."""
    ncol, dcol = _c(num[0]), _c(den[0])
    assert out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"]
    lib().or_permutation_product(_ptr(ncol), C.c_uint64(num[1]), C.c_uint64(num[2]), C.c_int64(num[3]),
                                 _ptr(dcol), C.c_uint64(den[1]), C.c_uint64(den[2]), C.c_int64(den[3]),
                                 C.c_uint64(count), _fp(z), _fp(alpha), _ptr(out), C.c_uint64(out_stride), C.c_uint64(out_off))
    return out[(count - 1) * out_stride + out_off].copy()


def diluted_aggregate(ordered, stride, off, count, z, alpha, out, out_stride=1, out_off=0):
    """layouts/src/recursive/trace.rs:787-803"""
    col = _c(ordered)
    assert out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"]
    lib().or_diluted_aggregate(_ptr(col), C.c_uint64(stride), C.c_uint64(off), C.c_uint64(count), _fp(z), _fp(alpha),
                               _ptr(out), C.c_uint64(out_stride), C.c_uint64(out_off))


def build_extension_columns(layout, cols, challenges, trace_len):
    """Trace::build_extension_columns for "recursive" (layouts/src/recursive/trace.rs:699-814; returns
    [diluted_check_aggregate, diluted_check_permutation, mem_and_rc_permutation]) or "starknet"
    (layouts/src/starknet/trace.rs:997-1100; returns [permutation_column]).
    cols: dict of the trace's auxiliary columns ([len, 4] Montgomery limbs): npc, memory, range_check and, for
    recursive, diluted_unordered / diluted_ordered.  challenges: the 6 verifier challenges (air.rs enums:
    MemoryPermutation Z=0 A=1, RangeCheckPermutation Z=2, DilutedCheckPermutation Z=3, DilutedCheckAggregation Z=4 A=5).
    Also returns the final (memory, range check, diluted) products; the reference asserts the last two are one."""
    z_mem, a_mem, z_rc, z_dc, z_agg, a_agg = challenges
    zero = np.zeros(4, dtype=np.uint64)
    MEMORY_STEP, RANGE_CHECK_STEP = 2, 4                     # recursive/mod.rs:18-19, starknet/mod.rs:16-17
    n_mem = len(cols["npc"]) // MEMORY_STEP
    n_rc = len(cols["range_check"]) // RANGE_CHECK_STEP
    if layout == "recursive":
        agg = np.zeros((trace_len, 4), dtype=np.uint64)
        dperm = np.zeros((trace_len, 4), dtype=np.uint64)
        mem_rc = np.zeros((trace_len, 4), dtype=np.uint64)
        # Permutation::{Memory -> (9, 0), RangeCheck -> (9, 1)} (recursive/air.rs:1705-1711); RangeCheck::{OffDst = 0, Ordered = 2}
        last_mem = permutation_product((cols["npc"], 2, 0, 1), (cols["memory"], 2, 0, 1), n_mem, z_mem, a_mem, mem_rc, MEMORY_STEP, 0)
        last_rc = permutation_product((cols["range_check"], 4, 0, -1), (cols["range_check"], 4, 2, -1), n_rc, z_rc, zero, mem_rc,
                                      RANGE_CHECK_STEP, 1)
        n_dc = len(cols["diluted_unordered"])                # DILUTED_CHECK_STEP = 1
        last_dc = permutation_product((cols["diluted_unordered"], 1, 0, -1), (cols["diluted_ordered"], 1, 0, -1), n_dc, z_dc, zero, dperm)
        diluted_aggregate(cols["diluted_ordered"], 1, 0, n_dc, z_agg, a_agg, agg)
        return [agg, dperm, mem_rc], (last_mem, last_rc, last_dc)
    assert layout == "starknet"
    DILUTED_CHECK_STEP = 8                                   # starknet/mod.rs:18
    perm = np.zeros((trace_len, 4), dtype=np.uint64)
    # Permutation::{Memory = 0, RangeCheck = 1, DilutedCheck = 7}; DilutedCheck::{Unordered = 1, Ordered = 5, Aggregate = 3}
    last_mem = permutation_product((cols["npc"], 2, 0, 1), (cols["memory"], 2, 0, 1), n_mem, z_mem, a_mem, perm, MEMORY_STEP, 0)
    last_rc = permutation_product((cols["range_check"], 4, 0, -1), (cols["range_check"], 4, 2, -1), n_rc, z_rc, zero, perm,
                                  RANGE_CHECK_STEP, 1)
    n_dc = len(cols["range_check"]) // DILUTED_CHECK_STEP
    last_dc = permutation_product((cols["range_check"], 8, 1, -1), (cols["range_check"], 8, 5, -1), n_dc, z_dc, zero, perm,
                                  DILUTED_CHECK_STEP, 7)
    diluted_aggregate(cols["range_check"], 8, 5, n_dc, z_agg, a_agg, perm, DILUTED_CHECK_STEP, 3)
    return [perm], (last_mem, last_rc, last_dc)


def deep_compose(trace_lde, comp_lde, log_n, log_blowup, offset, mask_col, mask_off, ood_trace,
                 coeff_trace, ood_comp, coeff_comp, z):
    tl = [_c(c) for c in trace_lde]
    cl = [_c(c) for c in comp_lde]
    N = tl[0].shape[0]
    tp = (C.c_void_p * len(tl))(*[c.ctypes.data for c in tl])
    cp = (C.c_void_p * max(1, len(cl)))(*[c.ctypes.data for c in cl])
    mc = np.ascontiguousarray(mask_col, dtype=np.uint32)
    mo = np.ascontiguousarray(mask_off, dtype=np.uint32)
    ot, ct, oc, cc = _c(ood_trace), _c(coeff_trace), _c(ood_comp), _c(coeff_comp)
    out = np.zeros((N, 4), dtype=np.uint64)
    lib().or_deep_compose(tp, cp, C.c_uint(log_n), C.c_uint(log_blowup), _fp(offset), _ptr(mc),
                          _ptr(mo), C.c_size_t(len(mc)), _ptr(ot), _ptr(ct), C.c_size_t(len(cl)),
                          _ptr(oc), _ptr(cc), _fp(z), _ptr(out))
    return out


# ----------------------------------------------------------------------- coin
class Coin:
    """kind 0 = SolidityVerifierPublicCoin, 1 = CairoVerifierPublicCoin."""

    def __init__(self, kind, digest: bytes):
        self.c = _Coin()
        lib().or_coin_new(C.byref(self.c), C.c_int(kind), digest)

    @property
    def digest(self):
        return bytes(self.c.digest)

    @property
    def counter(self):
        return int(self.c.counter)

    def reseed_bytes(self, b: bytes):
        lib().or_coin_reseed_bytes(C.byref(self.c), b, C.c_size_t(len(b)))

    def reseed_felts(self, v):
        a = _c(v)
        lib().or_coin_reseed_felts(C.byref(self.c), _ptr(a), C.c_size_t(a.shape[0]))

    def reseed_felt_vector(self, v):
        a = _c(v)
        lib().or_coin_reseed_felt_vector(C.byref(self.c), _ptr(a), C.c_size_t(a.shape[0]))

    def reseed_int(self, v):
        lib().or_coin_reseed_int(C.byref(self.c), C.c_uint64(v))

    def draw(self):
        return _fp_out(lib().or_coin_draw(C.byref(self.c)))

    def draw_queries(self, max_n, domain_size):
        out = np.zeros(max_n, dtype=np.uint64)
        lib().or_coin_draw_queries(C.byref(self.c), C.c_size_t(max_n), C.c_uint64(domain_size), _ptr(out))
        return sorted(set(int(x) for x in out))

    def grind(self, bits):
        return int(lib().or_coin_grind(C.byref(self.c), C.c_uint(bits)))

    def verify_pow(self, bits, nonce):
        return bool(lib().or_coin_verify_pow(C.byref(self.c), C.c_uint(bits), C.c_uint64(nonce)))


# ------------------------------------------------------------------- quotient
def eval_program(code, consts, tables, table_desc, n_slots, lde_cols, log_n, log_blowup, offset):
    code = np.ascontiguousarray(code, dtype=np.uint32)
    consts = _c(consts) if len(consts) else np.zeros((1, 4), dtype=np.uint64)
    tables = _c(tables) if len(tables) else np.zeros((1, 4), dtype=np.uint64)
    desc = np.ascontiguousarray(table_desc, dtype=np.uint32) if len(table_desc) else np.zeros(2, dtype=np.uint32)
    cols = [_c(c) for c in lde_cols]
    N = cols[0].shape[0]
    prog = AirProgram(code.ctypes.data_as(C.POINTER(C.c_uint32)), len(code) // 2,
                      consts.ctypes.data_as(C.POINTER(C.c_uint64)), consts.shape[0],
                      None, desc.ctypes.data_as(C.POINTER(C.c_uint32)), len(desc) // 2, n_slots)
    cp = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    out = np.zeros((N, 4), dtype=np.uint64)
    lib().or_eval_program_ex(C.byref(prog), _ptr(tables), cp, C.c_uint(log_n), C.c_uint(log_blowup),
                             _fp(offset), _ptr(out))
    return out


# ------------------------------------------------------------- the 64-bit field (oracle/goldilocks.c)
GL_P = 2**64 - 2**32 + 1


def gl_root_of_unity(log_n):
    f = lib().or_gl_root_of_unity
    f.restype = C.c_uint64
    return int(f(C.c_uint(log_n)))


def gl_ntt(col, inverse=False, offset=1):
    a = np.ascontiguousarray(col, dtype=np.uint64).copy()
    fn = lib().or_gl_ntt_inverse if inverse else lib().or_gl_ntt_forward
    fn.restype = None
    fn(_ptr(a), C.c_uint(a.shape[0].bit_length() - 1), C.c_uint64(offset))
    return a


def gl_lde(col, log_blowup, offset):
    a = np.ascontiguousarray(col, dtype=np.uint64)
    n = a.shape[0]
    ev, co = np.zeros(n << log_blowup, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    lib().or_gl_lde.restype = None
    lib().or_gl_lde(_ptr(a), C.c_uint(n.bit_length() - 1), C.c_uint(log_blowup), C.c_uint64(offset), _ptr(ev), _ptr(co))
    return ev, co


def gl3_fri_fold(evals, fold, alpha, offset, unnormalised=False):
    a = np.ascontiguousarray(evals, dtype=np.uint64)
    L = a.shape[0]
    out = np.zeros((L // fold, 3), dtype=np.uint64)
    al = np.ascontiguousarray(alpha, dtype=np.uint64)
    lib().or_gl3_fri_fold.restype = None
    lib().or_gl3_fri_fold(_ptr(a), C.c_uint(L.bit_length() - 1), C.c_uint(fold), _ptr(al), C.c_uint64(offset), C.c_int(1 if unnormalised else 0), _ptr(out))
    return out


def gl3_inv(a):
    x = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros(3, dtype=np.uint64)
    lib().or_gl3_inv.restype = None
    lib().or_gl3_inv(_ptr(x), _ptr(out))
    return out


def _colptrs(cols):
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in cols]
    return cols, (C.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])


def gl3_ood_eval(coeff_cols, cell_col, cell_off, z):
    """P_{col_j}(z w_n^{off_j}), z in Fq3, for natural-order coefficient columns -> uint64[ncells, 3]"""
    cols, ptrs = _colptrs(coeff_cols)
    cc, co = np.ascontiguousarray(cell_col, dtype=np.uint32), np.ascontiguousarray(cell_off, dtype=np.uint32)
    zz = np.ascontiguousarray(z, dtype=np.uint64)
    out = np.zeros((len(cc), 3), dtype=np.uint64)
    lib().or_gl3_ood_eval.restype = None
    lib().or_gl3_ood_eval(ptrs, C.c_uint(len(cols[0]).bit_length() - 1), _ptr(cc), _ptr(co), C.c_uint(len(cc)), _ptr(zz), _ptr(out))
    return out


def gl3_deep_compose(trace_lde, comp_lde, log_n, log_blowup, offset, mask_col, mask_off, ood_t, coeff_t, ood_c, coeff_c, z, zc):
    """the DEEP composition over Fq3, term by term -> uint64[N, 3]"""
    tc, tp = _colptrs(trace_lde)
    hc, hp = _colptrs(comp_lde)
    mc, mo = np.ascontiguousarray(mask_col, dtype=np.uint32), np.ascontiguousarray(mask_off, dtype=np.uint32)
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (ood_t, coeff_t, ood_c, coeff_c, z, zc)]
    N = 1 << (log_n + log_blowup)
    out = np.zeros((N, 3), dtype=np.uint64)
    lib().or_gl3_deep_compose.restype = None
    lib().or_gl3_deep_compose(tp, hp, C.c_uint(len(hc)), C.c_uint(log_n), C.c_uint(log_blowup), C.c_uint64(offset), _ptr(mc), _ptr(mo),
                              C.c_uint(len(mc)), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]), _ptr(arrs[4]), _ptr(arrs[5]), _ptr(out))
    return out


def gl3_eval_program(code, consts3, n_slots, tables, table_desc, lde_cols, log_n, log_blowup, offset):
    """the constraint program over Fq3, one point at a time -> uint64[N, 3]"""
    code = np.ascontiguousarray(code, dtype=np.uint32)
    consts = np.ascontiguousarray(consts3, dtype=np.uint64).reshape(-1, 3) if len(consts3) else np.zeros((1, 3), dtype=np.uint64)
    tabs = np.ascontiguousarray(tables, dtype=np.uint64) if tables is not None and len(tables) else np.zeros(1, dtype=np.uint64)
    desc = np.ascontiguousarray(table_desc if len(table_desc) else [0, 0], dtype=np.uint32)
    cols, ptrs = _colptrs(lde_cols)
    out = np.zeros((1 << (log_n + log_blowup), 3), dtype=np.uint64)
    lib().or_gl3_eval_program.restype = None
    lib().or_gl3_eval_program(_ptr(code), C.c_uint32(len(code) // 2), _ptr(consts), C.c_uint32(n_slots), _ptr(tabs), _ptr(desc), ptrs,
                              C.c_uint(log_n), C.c_uint(log_blowup), C.c_uint64(offset), _ptr(out))
    return out

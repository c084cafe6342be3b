"""ctypes binding of libsandstorm_host.so — the C++ host side above the C ABI (coin,
Expr lowering, prover; sandstorm_amd/host/).  The Python modules coin.py / air_program.py
/ prover.py are the readable mirror used by the tests; bench.py drives the C++ one."""
import ctypes as C
import os
import struct

import numpy as np

from . import _lib, backend as be
from .prover import FriLayer, Proof, ProofOptions

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libsandstorm_host.so")
AIR_MINI = 0
EXT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p))
ALL_TO_ALL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64))
ALL_GATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8))
HOST_ABI_VERSION = 4             # host_capi.cpp SSH_HOST_ABI_VERSION
SHARDED_EXT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32))
EXT_BLOCKS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p))

_host = None


def load():
    global _host
    if _host is None:
        _lib.load()                                   # libsandstorm_hip.so first (RTLD_GLOBAL)
        if not os.path.exists(LIB_PATH):
            raise _lib.SandstormHipError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        h.ssh_last_error.restype = C.c_char_p
        h.ssh_abi_version.restype = C.c_uint32
        if h.ssh_abi_version() != HOST_ABI_VERSION:
            raise _lib.SandstormHipError("%s has host ABI %d, these bindings are written for %d: rebuild (__graft_entry__.build())"
                                         % (LIB_PATH, h.ssh_abi_version(), HOST_ABI_VERSION))
        h.ssh_callback_group_create.argtypes = [C.c_uint32, C.c_uint32, ALL_TO_ALL_CB, ALL_GATHER_CB, C.c_void_p]
        h.ssh_callback_group_create.restype = C.c_void_p
        h.ssh_air_destroy.argtypes = [C.c_void_p]
        h.ssh_air_columns.argtypes = [C.c_void_p, C.c_int]
        h.ssh_air_columns.restype = C.c_uint32
        h.ssh_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p, C.POINTER(C.c_void_p),
                                C.c_uint32, C.c_uint32, EXT_CB, C.c_void_p, C.POINTER(C.c_uint32),
                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
        h.ssh_prove_wire.argtypes = h.ssh_prove.argtypes
        h.ssh_local_group_create.argtypes = [C.c_uint32]
        h.ssh_local_group_create.restype = C.c_void_p
        h.ssh_local_group_destroy.argtypes = [C.c_void_p]
        h.ssh_local_group_destroy.restype = None
        h.ssh_rccl_group_create.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]
        h.ssh_rccl_group_create.restype = C.c_void_p
        h.ssh_rccl_group_destroy.argtypes = [C.c_void_p]
        h.ssh_rccl_group_destroy.restype = None
        h.ssh_prove_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, SHARDED_EXT_CB, C.c_void_p,
                                        C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
        h.ssh_prove_sharded_blocks.argtypes = h.ssh_prove_sharded.argtypes[:14] + [EXT_BLOCKS_CB] + h.ssh_prove_sharded.argtypes[15:]
        h.ssh_build_extension_blocks.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                 C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_void_p)]
        h.ssh_prove_wire_with_nonce.argtypes = h.ssh_prove.argtypes[:12] + [C.c_uint64] + h.ssh_prove.argtypes[12:]
        h.ssh_free.argtypes = [C.c_void_p]
        h.ssh_build_extension_columns.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_uint64, C.POINTER(C.c_uint64),
                                                  C.c_int, C.POINTER(C.c_void_p)]
        h.ssh_public_coin_seed.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint64), C.c_uint64, C.c_int, C.c_char_p, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint32)]
        h.ssh_recursive_base_trace.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_uint64,
                                               C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64,
                                               C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_void_p)]
        h.ssh_air_create_recursive.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                               C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        h.ssh_air_create_starknet.argtypes = h.ssh_air_create_recursive.argtypes
        h.ssh_air_dump.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64),
                                   C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
        h.ssh_verify.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_uint64, C.c_int, C.c_uint32,
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32]
        h.ssh_matrix_num_cols.argtypes = [C.c_void_p]
        h.ssh_matrix_num_cols.restype = C.c_uint32
        h.ssh_matrix_col.argtypes = [C.c_void_p, C.c_uint32]
        h.ssh_matrix_col.restype = C.c_void_p
        h.ssh_matrix_destroy.argtypes = [C.c_void_p]
        h.ssh_coin_new.restype = C.c_void_p
        h.ssh_coin_new.argtypes = [C.c_int, C.c_char_p]
        h.ssh_coin_free.argtypes = [C.c_void_p]
        h.ssh_coin_op.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64,
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        _host = h
    return _host


def _check(rc):
    if rc != 0:
        raise _lib.SandstormHipError("host: " + load().ssh_last_error().decode())


_AIR_FACTORIES = {}


def register_air_kind(kind, factory):
    """factory(ctx_handle_or_None) -> an `ssh_air` handle (an `Air *` of host/prover.hpp, freed by ssh_air_destroy): how a caller
    brings an AIR of its own (the tests register their mini AIR as AIR_MINI: tests/mini_air_host.py; the library itself only holds
    the layouts' - RecursiveHostAir, StarknetHostAir)"""
    _AIR_FACTORIES[kind] = factory


class HostAir:
    def __init__(self, ctx, kind, log_n, log_blowup=1):
        self.ctx, self.h = ctx, C.c_void_p()        # ctx=None: host-only handle (verification; no device tables)
        load()
        if kind not in _AIR_FACTORIES:
            raise _lib.SandstormHipError("host: no AIR of kind %r is registered (the layouts' AIRs: RecursiveHostAir, StarknetHostAir)" % (kind,))
        self.h = C.c_void_p(_AIR_FACTORIES[kind](ctx.handle if ctx is not None else None))
        self.num_base_columns = load().ssh_air_columns(self.h, 0)
        self.num_extension_columns = load().ssh_air_columns(self.h, 1)
        self.mask_size = load().ssh_air_columns(self.h, 2)

    def close(self):
        if self.h:
            load().ssh_air_destroy(self.h)
            self.h = None


class RecursiveHostAir(HostAir):
    """the C++ host's real `recursive` AIR for a public input (sandstorm_amd/host/air_recursive.cpp).  ctx=None: no device
    tables, only dump() works (host-side checks)."""
    create = "ssh_air_create_recursive"
    column_tag = "pedersen"                 # what layouts.recursive.Tables calls its periodic columns

    def __init__(self, ctx, pi, log_n, log_blowup=1):
        segs, addrs, vals = _public_input_args(pi)
        h = C.c_void_p()
        _check(getattr(load(), self.create)(ctx.handle if ctx is not None else None, pi.rc_min, pi.rc_max, pi.n_steps,
                                               segs.ctypes.data_as(C.POINTER(C.c_uint32)), addrs.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               vals.ctypes.data_as(C.POINTER(C.c_uint64)), len(addrs), log_n, log_blowup, C.byref(h)))
        self.ctx, self.h = ctx, h
        self.num_base_columns = load().ssh_air_columns(h, 0)
        self.num_extension_columns = load().ssh_air_columns(h, 1)
        self.mask_size = load().ssh_air_columns(h, 2)

    def prepare(self, n, challenges):
        """Air::prepare_program: lower the program for these challenges ahead of the composition coefficient"""
        ch = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c in challenges]))
        h = load()
        h.ssh_air_prepare_program.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32]
        _check(h.ssh_air_prepare_program(self.h, n, ch.ctypes.data_as(C.POINTER(C.c_uint64)), len(challenges)))

    def dump(self, n, challenges, alpha):
        """-> (code uint32[], consts uint64[*,4], n_slots, table specs in layouts.recursive.Tables' format)"""
        ch = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c in challenges]))
        al = np.ascontiguousarray(alpha, dtype=np.uint64)
        blob, ln = C.POINTER(C.c_uint64)(), C.c_uint64()
        _check(load().ssh_air_dump(self.h, n, ch.ctypes.data_as(C.POINTER(C.c_uint64)), len(challenges), al.ctypes.data_as(C.POINTER(C.c_uint64)),
                                   C.byref(blob), C.byref(ln)))
        words = [blob[i] for i in range(ln.value)]
        load().ssh_free(blob)
        it = iter(words)
        n_instr = next(it)
        code = np.array([next(it) for _ in range(2 * n_instr)], dtype=np.uint32)
        n_consts = next(it)
        consts = np.array([[next(it) for _ in range(4)] for _ in range(n_consts)], dtype=np.uint64).reshape(-1, 4)
        n_slots = next(it)
        specs = []
        for _ in range(next(it)):
            kind, e = next(it), next(it)
            num = tuple((next(it), next(it)) for _ in range(next(it)))
            den = tuple((next(it), next(it)) for _ in range(next(it)))
            # kind 0: periodic column number e (layouts.recursive calls its two "pedersen", layouts.starknet "column")
            specs.append((self.column_tag, e) if kind == 0 else ("periodic", num, den) if kind == 2 else ("inverse", e))
        return code, consts, n_slots, specs


class StarknetHostAir(RecursiveHostAir):
    """the C++ host's real `starknet` AIR (sandstorm_amd/host/air_starknet.cpp)"""
    create = "ssh_air_create_starknet"
    column_tag = "column"


class _HostProgram:
    """a lowered program as ss_eval_quotient takes it (air_program.Program's duck type), constants already in limb form"""

    def __init__(self, code, consts_mont, n_slots):
        self.code, self.consts_mont, self.n_slots = code, consts_mont, n_slots
        self.consts = range(len(consts_mont))           # only its length is read when consts_mont is there


def prover_air(host_air):
    """the C++ host's AIR behind the prover.Air interface (what sandstorm_amd/prover.py and sharded_prover.py drive): the
    composition program is built and lowered by the C++ host per proof (sub-millisecond; the Python lowering of the same
    DAG takes ~0.3 s for the starknet layout), its tables are the ones the handle built on the device"""
    from .prover import Air
    h = load()
    h.ssh_air_program.argtypes = h.ssh_air_dump.argtypes
    h.ssh_air_mask.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    h.ssh_air_num_challenges.argtypes = [C.c_void_p]
    h.ssh_air_num_challenges.restype = C.c_uint32
    nmask = host_air.mask_size
    mc, mo = (C.c_uint32 * nmask)(), (C.c_uint32 * nmask)()
    h.ssh_air_mask(host_air.h, mc, mo)
    mask = [(int(mc[j]), int(mo[j])) for j in range(nmask)]

    def build_program(n, challenges, comp_coeff):
        ch = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c in challenges]))
        al = np.ascontiguousarray(comp_coeff, dtype=np.uint64)
        blob, ln = C.POINTER(C.c_uint64)(), C.c_uint64()
        _check(h.ssh_air_program(host_air.h, n, ch.ctypes.data_as(C.POINTER(C.c_uint64)), len(challenges), al.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 C.byref(blob), C.byref(ln)))
        words = np.ctypeslib.as_array(blob, shape=(ln.value,)).copy()
        h.ssh_free(blob)
        o = 0
        n_instr = int(words[o]); o += 1
        code = words[o:o + 2 * n_instr].astype(np.uint32); o += 2 * n_instr
        n_consts = int(words[o]); o += 1
        consts = words[o:o + 4 * n_consts].reshape(n_consts, 4).copy(); o += 4 * n_consts
        n_slots = int(words[o]); o += 1
        n_tables = int(words[o]); o += 1
        desc = [int(v) for v in words[o:o + 2 * n_tables]]; o += 2 * n_tables
        d_tables = int(words[o])
        return _HostProgram(code, consts, n_slots), (_DevicePointer(d_tables) if d_tables else None), desc
    return Air(type(host_air).__name__, host_air.num_base_columns, host_air.num_extension_columns, h.ssh_air_num_challenges(host_air.h), mask,
               build_program)


class _DevicePointer:
    """a device address owned by somebody else (the C++ AIR's tables)"""

    def __init__(self, ptr):
        self.ptr = ptr


class _Reader:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, n):
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def felts(self):
        n = self.u32()
        return np.frombuffer(self.take(32 * n), dtype=np.uint64).reshape(n, 4).copy()

    def u64s(self):
        n = self.u32()
        return np.frombuffer(self.take(8 * n), dtype=np.uint64).copy()

    def bytes_(self):
        return self.take(self.u32())


def parse_proof(raw, options, ncols_base, ncols_ext, ncomp=2):
    r = _Reader(raw)
    p = Proof(options, r.u64())
    p.base_root = bytes(r.take(33))[:32]
    has_ext = r.u32()
    ext = bytes(r.take(33))[:32]
    p.extension_root = ext if has_ext else None
    p.composition_root = bytes(r.take(33))[:32]
    p.challenges = list(r.felts())
    p.composition_coeff = np.frombuffer(r.take(32), dtype=np.uint64).copy()
    p.z = np.frombuffer(r.take(32), dtype=np.uint64).copy()
    p.deep_alpha = np.frombuffer(r.take(32), dtype=np.uint64).copy()
    p.ood_trace, p.ood_composition = r.felts(), r.felts()
    p.fri_alphas = list(r.felts())
    p.fri_remainder = r.felts()
    p.pow_nonce = r.u64()
    p.query_positions = [int(q) for q in r.u64s()]
    nq = len(p.query_positions)
    log_N = (p.trace_len * options.lde_blowup_factor).bit_length() - 1
    p.base_rows = r.u64s().reshape(nq, ncols_base, 4)
    er = r.u64s()
    p.extension_rows = er.reshape(nq, ncols_ext, 4) if len(er) else None
    p.composition_rows = r.u64s().reshape(nq, ncomp, 4)
    p.base_paths = np.frombuffer(r.bytes_(), dtype=np.uint8).reshape(nq, log_N, 32)
    ep = r.bytes_()
    p.extension_paths = np.frombuffer(ep, dtype=np.uint8).reshape(nq, log_N, 32) if len(ep) else None
    p.composition_paths = np.frombuffer(r.bytes_(), dtype=np.uint8).reshape(nq, log_N, 32)
    fold = options.fri_folding_factor
    for _ in range(r.u32()):
        root = bytes(r.take(33))
        layer = FriLayer(root[:32], root[32], r.u32())
        layer.positions = [int(q) for q in r.u64s()]
        layer.rows = r.u64s().reshape(len(layer.positions), fold, 4)
        rows_log = layer.log_len - (fold.bit_length() - 1)
        layer.paths = np.frombuffer(r.bytes_(), dtype=np.uint8).reshape(len(layer.positions), rows_log, 32)
        p.fri_layers.append(layer)
    assert r.o == len(raw)
    return p


def prove(ctx, air: HostAir, tree_kind, n_friendly, coin_kind, seed, base_cols, log_n, build_extension, options=None,
          want_proof=True, wire=False, pow_nonce=None):
    """build_extension(challenges: list of uint64[4]) -> list of device columns (kept alive by the caller).
    wire=True: return the proof as bytes in the reference's wire format (ssh_prove_wire; sandstorm_amd/wire.py
    parses them) instead of a parsed Proof.  pow_nonce (wire only): use this proof-of-work nonce instead of grinding
    (it must be valid for the transcript)."""
    options = options or ProofOptions()
    keep = []

    def cb(_user, ch_ptr, nch, out_ptr):
        try:
            ch = [np.array([ch_ptr[4 * i + k] for k in range(4)], dtype=np.uint64) for i in range(nch)]
            cols = build_extension(ch)
            keep.append(cols)
            for i, col in enumerate(cols):
                out_ptr[i] = be._ptr_of(col)
            return 0
        except Exception:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1
    opts = (C.c_uint32 * 5)(options.num_queries, options.lde_blowup_factor, options.grinding_factor,
                            options.fri_folding_factor, options.fri_max_remainder_coeffs)
    out, n = C.POINTER(C.c_uint8)(), C.c_uint64()
    entry = load().ssh_prove_wire if wire else load().ssh_prove
    head = (ctx.handle, air.h, tree_kind, n_friendly, coin_kind, bytes(seed), be._ptr_array(base_cols),
            len(base_cols), log_n, EXT_CB(cb), None, opts)
    tail = (C.byref(out) if want_proof else None, C.byref(n) if want_proof else None)
    if pow_nonce is not None:
        assert wire, "a supplied proof-of-work nonce is a wire-format feature"
        _check(load().ssh_prove_wire_with_nonce(*head, int(pow_nonce), *tail))
    else:
        _check(entry(*head, *tail))
    if not want_proof:
        return None
    raw = bytes(C.cast(out, C.POINTER(C.c_uint8 * n.value)).contents)
    load().ssh_free(out)
    if wire:
        return raw
    return parse_proof(raw, options, air.num_base_columns, air.num_extension_columns)


class HostMatrix:
    """a device matrix owned by the C++ host (ssh_matrix)"""

    def __init__(self, ctx, handle, nrows):
        self.ctx, self.h, self.nrows = ctx, handle, nrows
        self.cols = [load().ssh_matrix_col(handle, k) for k in range(load().ssh_matrix_num_cols(handle))]

    def to_host(self):
        out = []
        for ptr in self.cols:
            a = np.empty((self.nrows, 4), dtype=np.uint64)
            be.check(self.ctx.lib.ss_download(self.ctx.handle, a.ctypes.data, ptr, a.nbytes))
            out.append(a)
        return out

    def close(self):
        if self.h:
            load().ssh_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def build_extension_columns(ctx, layout, aux_cols, trace_len, challenges, check=True):
    """the C++ host's Trace::build_extension_columns (sandstorm_amd/host/extension.cpp).
    aux_cols: [npc, memory, range_check] (+ [diluted_unordered, diluted_ordered] for "recursive")."""
    ch = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c in challenges[:6]]))
    h = C.c_void_p()
    _check(load().ssh_build_extension_columns(ctx.handle, 1 if layout == "recursive" else 2, be._ptr_array(aux_cols), trace_len,
                                              ch.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if check else 0, C.byref(h)))
    return HostMatrix(ctx, h, trace_len)


def build_extension_blocks(ctx, layout, aux_blocks, trace_len, rank, world, group, challenges, check=True):
    """the same columns as ROW BLOCKS over the ranks of `group` (host/extension.cpp build_extension_blocks): aux_blocks = this rank's
    rows [rank n / world, (rank + 1) n / world) of the auxiliary columns, trace_len = n; every rank enters (one all-gather of the
    blocks' totals).  -> a HostMatrix of the same rows of the extension columns"""
    ch = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c in challenges[:6]]))
    h = C.c_void_p()
    local = isinstance(group, LocalGroup)
    _check(load().ssh_build_extension_blocks(ctx.handle, 1 if layout == "recursive" else 2, be._ptr_array(aux_blocks), trace_len, rank, world,
                                             group.h if local else None, None if local else group.h, ch.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             1 if check else 0, C.byref(h)))
    return HostMatrix(ctx, h, trace_len // world)


def public_coin_seed(pi, coin_kind):
    """the C++ host's CairoPublicCoin::from_public_input (sandstorm_amd/host/public_input.cpp).
    pi: public_input.AirPublicInput.  -> (seed bytes, public input elements as ints)"""
    from .public_input import SEGMENTS
    segs = np.zeros(27, dtype=np.uint32)
    for k, name in enumerate(SEGMENTS):
        s = pi.memory_segments.get(name)
        if s is not None:
            segs[3 * k: 3 * k + 3] = (1, s[0], s[1])
    addrs = np.array([e[0] for e in pi.public_memory], dtype=np.uint32)
    vals = np.array([[(e[1] >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for e in pi.public_memory], dtype=np.uint64).reshape(-1, 4)
    seed = C.create_string_buffer(32)
    els = np.zeros((64, 4), dtype=np.uint64)
    n_els = C.c_uint32()
    layout = {"recursive": 1, "starknet": 2}.get(pi.layout, 0)
    _check(load().ssh_public_coin_seed(layout, pi.rc_min, pi.rc_max, pi.n_steps, segs.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       addrs.ctypes.data_as(C.POINTER(C.c_uint32)), vals.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       len(addrs), coin_kind, seed, els.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(n_els)))
    ints = [sum(int(els[i, k]) << (64 * k) for k in range(4)) for i in range(n_els.value)]
    return seed.raw, ints


def _public_input_args(pi):
    from .public_input import SEGMENTS
    segs = np.zeros(27, dtype=np.uint32)
    for k, name in enumerate(SEGMENTS):
        s = pi.memory_segments.get(name)
        if s is not None:
            segs[3 * k: 3 * k + 3] = (1, s[0], s[1])
    addrs = np.array([e[0] for e in pi.public_memory], dtype=np.uint32)
    vals = np.array([[(e[1] >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for e in pi.public_memory], dtype=np.uint64).reshape(-1, 4)
    return segs, addrs, vals


def _instances(rows, width):
    """[(index, v0[, v1])] -> uint64 array of `width` words per instance (index, then 4 little-endian limbs per value)"""
    if not rows:
        return np.zeros((1, width), dtype=np.uint64)
    # (a run may bring tens of thousands of instances: one to_bytes per value, not four shifts)
    raw = b"".join(int(row[0]).to_bytes(8, "little") + b"".join(int(v).to_bytes(32, "little") for v in row[1:]).ljust(8 * (width - 1), b"\0") for row in rows)
    return np.frombuffer(raw, dtype="<u8").reshape(len(rows), width).copy()


def _trace_out(out, ncols, n):
    """the generator's destination: fresh arrays, or the caller's (pinned, page-aligned: the GpuAllocator seam of
    layouts/src/recursive/trace.rs:115-120) - `ncols` C-contiguous arrays of n x 4 64-bit limbs"""
    if out is None:
        return [np.zeros((n, 4), dtype=np.uint64) for _ in range(ncols)]
    if len(out) != ncols or any(a.shape != (n, 4) or a.dtype.itemsize != 8 or not a.flags["C_CONTIGUOUS"] for a in out):
        raise _lib.SandstormHipError("host: out must be %d C-contiguous arrays of shape (%d, 4) and 8-byte items" % (ncols, n))
    return list(out)


def recursive_base_trace(trace_bin: bytes, memory_bin: bytes, pi, private_input=None, out=None):
    """the C++ host's ExecutionTrace::new for the recursive layout (sandstorm_amd/host/trace_recursive.cpp) from the raw
    `cairo-run` files -> 7 columns [16 * cycles, 4] of Montgomery limbs (written into `out` if given)"""
    private_input = private_input or {}
    if len(trace_bin) % 24:
        raise _lib.SandstormHipError("host: trace file is not a sequence of (ap, fp, pc) u64 triples")
    n = 16 * (len(trace_bin) // 24)
    segs, addrs, vals = _public_input_args(pi)
    ped, rc, bw = (_instances(private_input.get("pedersen", []), 9), _instances(private_input.get("range_check", []), 5),
                   _instances(private_input.get("bitwise", []), 9))
    cols = _trace_out(out, 7, n)
    ptrs = (C.c_void_p * 7)(*[c.ctypes.data for c in cols])
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    _check(load().ssh_recursive_base_trace(trace_bin, len(trace_bin), memory_bin, len(memory_bin), pi.rc_min, pi.rc_max, pi.n_steps,
                                           segs.ctypes.data_as(u32p), addrs.ctypes.data_as(u32p), vals.ctypes.data_as(u64p), len(addrs),
                                           ped.ctypes.data_as(u64p), len(private_input.get("pedersen", [])),
                                           rc.ctypes.data_as(u64p), len(private_input.get("range_check", [])),
                                           bw.ctypes.data_as(u64p), len(private_input.get("bitwise", [])), ptrs))
    return cols


def starknet_base_trace(trace_bin: bytes, memory_bin: bytes, pi, private_input=None, out=None):
    """the C++ host's ExecutionTrace::new for the starknet layout (sandstorm_amd/host/trace_starknet.cpp) -> 9 columns
    [16 * cycles, 4] of Montgomery limbs (written into `out` if given).  private_input: as layouts.starknet.base_trace takes it"""
    private_input = private_input or {}
    if len(trace_bin) % 24:
        raise _lib.SandstormHipError("host: trace file is not a sequence of (ap, fp, pc) u64 triples")
    n = 16 * (len(trace_bin) // 24)
    segs, addrs, vals = _public_input_args(pi)
    names = (("pedersen", 9), ("range_check", 5), ("ecdsa", 17), ("bitwise", 9), ("ec_op", 21), ("poseidon", 13))
    arrays = [_instances(private_input.get(name, []), width) for name, width in names]
    counts = np.array([len(private_input.get(name, [])) for name, _ in names], dtype=np.uint64)
    inst = (C.c_void_p * 6)(*[a.ctypes.data for a in arrays])
    cols = _trace_out(out, 9, n)
    ptrs = (C.c_void_p * 9)(*[c.ctypes.data for c in cols])
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    fn = load().ssh_starknet_base_trace
    fn.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, u64p, C.c_uint64,
                   C.POINTER(C.c_void_p), u64p, C.POINTER(C.c_void_p)]
    _check(fn(trace_bin, len(trace_bin), memory_bin, len(memory_bin), pi.rc_min, pi.rc_max, pi.n_steps, segs.ctypes.data_as(u32p),
              addrs.ctypes.data_as(u32p), vals.ctypes.data_as(u64p), len(addrs), inst, counts.ctypes.data_as(u64p), ptrs))
    return cols


GL_EXT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p))
GL_PROG_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64))


def gl_prove(ctx, air, options, seed, base_cols, build_extension, tables=None, statement=None):
    """the 64-bit field's claim by the C++ host (host/goldilocks_prover.cpp ssh_gl_prove; the mirror of goldilocks.Prover.prove, which
    writes the same proof arrays): every stage a kernel behind the C ABI, the transcript in C++; what the LAYOUT decides comes from here
    through two callbacks - the extension trace's coordinate columns for the drawn challenges (build_extension, as goldilocks.Prover
    takes it) and the lowered composition program for them (air.composition + air_program.lower).  air: a goldilocks.Air; base_cols:
    device columns of n values.  -> goldilocks.Proof"""
    from . import air_program as ap, goldilocks as gs
    opt = options or gs.Options()
    n = int(base_cols[0].shape[0]) if hasattr(base_cols[0], "shape") else int(base_cols[0].nbytes // 8)
    tables = tables or air.make_tables(n, opt.log_blowup)
    keep = []

    def ext_cb(_user, ch_ptr, nch, out_ptr):
        try:
            ch = [tuple(int(ch_ptr[3 * i + k]) for k in range(3)) for i in range(nch)]
            cols = list(build_extension(ch))
            keep.append(cols)
            for i, col in enumerate(cols):
                out_ptr[i] = be._ptr_of(col)
            return 0
        except Exception:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    def prog_cb(_user, ch_ptr, nch, alpha_ptr, blob_out, len_out):
        try:
            ch = [tuple(int(ch_ptr[3 * i + k]) for k in range(3)) for i in range(nch)]
            alpha = tuple(int(alpha_ptr[k]) for k in range(3))
            prog = ap.lower(air.composition(n, ch, alpha, tables, statement), gs.P, ext=True, symbols=tables.symbols)
            tvals, tdesc = tables.device_tables()               # (the composition names the tables it reads: after it)
            d_tables = ctx.alloc(max(8, tvals.nbytes))
            d_tables.upload(np.ascontiguousarray(tvals))
            keep.append(d_tables)
            code = np.asarray(prog.code, dtype=np.uint32).astype(np.uint64)
            consts = np.asarray(prog.consts, dtype=np.uint64).reshape(-1)
            head = np.array([len(code) // 2, len(consts) // 3, prog.n_slots, len(tdesc) // 2, be._ptr_of(d_tables)], dtype=np.uint64)
            blob = np.ascontiguousarray(np.concatenate([head, code, consts, np.asarray(tdesc, dtype=np.uint64)]))
            keep.append(blob)
            blob_out[0] = C.cast(blob.ctypes.data, C.POINTER(C.c_uint64))
            len_out[0] = len(blob)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1
    h = load()
    h.ssh_gl_prove.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32),
                               C.c_uint32, C.c_uint32, C.c_uint32, GL_EXT_CB, GL_PROG_CB, C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    opts = (C.c_uint32 * 6)(opt.num_queries, opt.log_blowup, opt.grinding, opt.fold, opt.max_remainder, 1 if opt.hash == "sha256" else 0)
    mask = np.ascontiguousarray([v for cell in air.mask for v in cell], dtype=np.uint32)
    out, ln = C.POINTER(C.c_uint64)(), C.c_uint64()
    _check(h.ssh_gl_prove(ctx.handle, opts, bytes(seed), gs.statement_digest(statement), be._ptr_array(base_cols), len(base_cols), n,
                          mask.ctypes.data_as(C.POINTER(C.c_uint32)), len(air.mask), air.num_challenges, air.num_ext, GL_EXT_CB(ext_cb), GL_PROG_CB(prog_cb), None,
                          C.byref(out), C.byref(ln)))
    w = np.ctypeslib.as_array(out, shape=(ln.value,)).copy()
    h.ssh_free(out)
    del keep[:]
    o = 0

    def take(k):
        nonlocal o
        v = w[o:o + k]
        o += k
        return v
    trace_len, nonce, has_ext, n_layers = (int(v) for v in take(4))
    roots = [take(4).tobytes() for _ in range(3)]
    proof = gs.Proof(opt, trace_len, roots[0], roots[1] if has_ext else b"", roots[2])
    proof.ood_trace = take(3 * int(take(1)[0])).reshape(-1, 3).copy()
    proof.ood_comp = take(18).reshape(6, 3).copy()
    proof.remainder = take(3 * int(take(1)[0])).reshape(-1, 3).copy()
    proof.pow_nonce = nonce
    for _ in range(n_layers):
        root = take(4).tobytes()
        proof.fri_layers.append(gs.FriLayer(root, int(take(1)[0])))

    def opening():
        npos, width, depth = (int(v) for v in take(3))
        rows = take(npos * width).reshape(npos, width).copy()
        paths = take(npos * depth * 4).copy().view(np.uint8).reshape(npos, depth, 32)
        return gs.Opening(rows, paths)
    proof.base = opening()
    if has_ext:
        proof.ext = opening()
    proof.comp = opening()
    for fl in proof.fri_layers:
        fl.opening = opening()
    assert o == len(w)
    return proof


_INSTANCE_SHAPES = (("pedersen", 9), ("range_check", 5), ("ecdsa", 17), ("bitwise", 9), ("ec_op", 21), ("poseidon", 13))
COLUMN_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


def _trace_job_args(layout, trace_bin, memory_bin, pi, private_input):
    """the arguments ssh_base_trace_cb / ssh_prove_files share (host_capi.cpp make_trace_job); -> (args, keep-alive)"""
    if len(trace_bin) % 24:
        raise _lib.SandstormHipError("host: trace file is not a sequence of (ap, fp, pc) u64 triples")
    private_input = private_input or {}
    segs, addrs, vals = _public_input_args(pi)
    arrays = [_instances(private_input.get(name, []), width) for name, width in _INSTANCE_SHAPES]
    counts = np.array([len(private_input.get(name, [])) for name, _ in _INSTANCE_SHAPES], dtype=np.uint64)
    inst = (C.c_void_p * 6)(*[a.ctypes.data for a in arrays])
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    args = (1 if layout == "recursive" else 2, trace_bin, len(trace_bin), memory_bin, len(memory_bin), pi.rc_min, pi.rc_max, pi.n_steps,
            segs.ctypes.data_as(u32p), addrs.ctypes.data_as(u32p), vals.ctypes.data_as(u64p), len(addrs), inst, counts.ctypes.data_as(u64p))
    return args, (segs, addrs, vals, arrays, counts, inst)


_TRACE_JOB_ARGTYPES = [C.c_int, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32),
                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]


def base_trace_with_callback(layout, trace_bin: bytes, memory_bin: bytes, pi, private_input=None, out=None, column_done=None):
    """either layout's generator (as recursive_base_trace / starknet_base_trace) calling column_done(c) as soon as no section will
    write column c again (host/trace_*.hpp) -> (columns, the order the callbacks came in as the library announces it)"""
    ncols = 7 if layout == "recursive" else 9
    cols = _trace_out(out, ncols, 16 * (len(trace_bin) // 24))
    ptrs = (C.c_void_p * ncols)(*[c.ctypes.data for c in cols])
    args, keep = _trace_job_args(layout, trace_bin, memory_bin, pi, private_input)
    fn = load().ssh_base_trace_cb
    fn.argtypes = _TRACE_JOB_ARGTYPES + [C.POINTER(C.c_void_p), COLUMN_CB, C.c_void_p, C.POINTER(C.c_uint32)]
    cb = COLUMN_CB((lambda _u, c: column_done(int(c))) if column_done else (lambda _u, c: None))
    order = (C.c_uint32 * 9)()
    _check(fn(*args, ptrs, cb, None, order))
    del keep
    return cols, [int(order[k]) for k in range(ncols)]


def prove_files(ctx, layout, trace_bin: bytes, memory_bin: bytes, pi, private_input, pinned_cols, dev_cols, air: HostAir, tree_kind, n_friendly,
                coin_kind, seed, build_extension, options=None, want_proof=True):
    """`sandstorm-cli prove` in one call (host_capi.cpp ssh_prove_files): the generator on a thread of its own writes `pinned_cols`
    (pinned host arrays [16 * cycles, 4] of 8-byte items), every column is uploaded to `dev_cols` the moment it is final, the prover
    extends the columns as they land.  -> (proof bytes in the reference's wire format or None, {"trace_gen_s", "total_s"})"""
    options = options or ProofOptions()
    ncols = 7 if layout == "recursive" else 9
    n = 16 * (len(trace_bin) // 24)
    views = _trace_out(pinned_cols, ncols, n)
    if len(dev_cols) != ncols:
        raise _lib.SandstormHipError("host: %d device columns for a layout of %d" % (len(dev_cols), ncols))
    host_ptrs = (C.c_void_p * ncols)(*[c.ctypes.data for c in views])
    args, keep_args = _trace_job_args(layout, trace_bin, memory_bin, pi, private_input)
    keep = []

    def cb(_user, ch_ptr, nch, out_ptr):
        try:
            ch = [np.array([ch_ptr[4 * i + k] for k in range(4)], dtype=np.uint64) for i in range(nch)]
            cols = build_extension(ch)
            keep.append(cols)
            for i, col in enumerate(cols):
                out_ptr[i] = be._ptr_of(col)
            return 0
        except Exception:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1
    opts = (C.c_uint32 * 5)(options.num_queries, options.lde_blowup_factor, options.grinding_factor,
                            options.fri_folding_factor, options.fri_max_remainder_coeffs)
    fn = load().ssh_prove_files
    fn.argtypes = [C.c_void_p] + _TRACE_JOB_ARGTYPES + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p, EXT_CB, C.c_void_p,
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
    out, ln, times = C.POINTER(C.c_uint8)(), C.c_uint64(), (C.c_double * 2)()
    _check(fn(ctx.handle, *args, host_ptrs, be._ptr_array(dev_cols), air.h, tree_kind, n_friendly, coin_kind, bytes(seed), EXT_CB(cb), None, opts, times,
              C.byref(out) if want_proof else None, C.byref(ln) if want_proof else None))
    del keep_args
    raw = None
    if want_proof:
        raw = bytes(bytearray(out[:ln.value]))
        load().ssh_free(out)
    return raw, {"trace_gen_s": times[0], "total_s": times[1]}


def device_base_trace(ctx, layout, trace_bin: bytes, memory_bin: bytes, pi, private_input=None, dev_cols=None):
    """either layout's base columns made ON the device from the raw files (host_capi.cpp ssh_base_trace_device -> host/device_trace.hpp
    -> csrc/trace.hip): trace.bin / memory.bin are uploaded as they are, the cells are made in HBM.  dev_cols: 7 / 9 device columns of
    [16 * cycles, 4] 8-byte items (DeviceBuffers, torch tensors or addresses; fresh DeviceBuffers if None) -> the columns"""
    ncols = 7 if layout == "recursive" else 9
    n = 16 * (len(trace_bin) // 24)
    if dev_cols is None:
        dev_cols = [ctx.alloc(32 * n) for _ in range(ncols)]
    if len(dev_cols) != ncols:
        raise _lib.SandstormHipError("host: %d device columns for a layout of %d" % (len(dev_cols), ncols))
    args, keep = _trace_job_args(layout, trace_bin, memory_bin, pi, private_input)
    fn = load().ssh_base_trace_device
    fn.argtypes = [C.c_void_p] + _TRACE_JOB_ARGTYPES + [C.POINTER(C.c_void_p)]
    _check(fn(ctx.handle, *args, be._ptr_array(dev_cols)))
    del keep
    return dev_cols


def prove_files_device(ctx, layout, trace_bin: bytes, memory_bin: bytes, pi, private_input, dev_cols, air: HostAir, tree_kind, n_friendly, coin_kind, seed,
                       build_extension, options=None, want_proof=True):
    """`sandstorm-cli prove` in one call with the base trace made on the device (host_capi.cpp ssh_prove_files_device): the files' bytes
    go up as they are, csrc/trace.hip makes the columns in `dev_cols`, the prover goes on from there.
    -> (proof bytes in the reference's wire format or None, {"trace_gen_s", "total_s"})"""
    options = options or ProofOptions()
    ncols = 7 if layout == "recursive" else 9
    if len(dev_cols) != ncols:
        raise _lib.SandstormHipError("host: %d device columns for a layout of %d" % (len(dev_cols), ncols))
    args, keep_args = _trace_job_args(layout, trace_bin, memory_bin, pi, private_input)
    keep = []

    def cb(_user, ch_ptr, nch, out_ptr):
        try:
            ch = [np.array([ch_ptr[4 * i + k] for k in range(4)], dtype=np.uint64) for i in range(nch)]
            cols = build_extension(ch)
            keep.append(cols)
            for i, col in enumerate(cols):
                out_ptr[i] = be._ptr_of(col)
            return 0
        except Exception:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1
    opts = (C.c_uint32 * 5)(options.num_queries, options.lde_blowup_factor, options.grinding_factor,
                            options.fri_folding_factor, options.fri_max_remainder_coeffs)
    fn = load().ssh_prove_files_device
    fn.argtypes = [C.c_void_p] + _TRACE_JOB_ARGTYPES + [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p, EXT_CB, C.c_void_p,
                                                        C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
    out, ln, times = C.POINTER(C.c_uint8)(), C.c_uint64(), (C.c_double * 2)()
    _check(fn(ctx.handle, *args, be._ptr_array(dev_cols), air.h, tree_kind, n_friendly, coin_kind, bytes(seed), EXT_CB(cb), None, opts, times,
              C.byref(out) if want_proof else None, C.byref(ln) if want_proof else None))
    del keep_args
    raw = None
    if want_proof:
        raw = bytes(bytearray(out[:ln.value]))
        load().ssh_free(out)
    return raw, {"trace_gen_s": times[0], "total_s": times[1]}


def verify(air: HostAir, tree_kind, coin_kind, seed, proof: bytes, shipped_conventions=True, fri_alpha_times_offset=True,
           required_security_bits=80, expected_options=None, n_friendly_layers=22):
    """the C++ host's verifier (sandstorm_amd/host/verifier.cpp) on a proof in the reference's wire format; raises
    SandstormHipError naming the failed check, returns the query positions.  required_security_bits / expected_options:
    as sandstorm_amd.verifier.verify"""
    pos = np.zeros(256, dtype=np.uint64)
    npos = C.c_uint32()
    exp = None
    if expected_options is not None:
        o = expected_options
        exp = (C.c_uint32 * 5)(*(o if isinstance(o, (list, tuple)) else
                                 [o.num_queries, o.lde_blowup_factor, o.grinding_factor, o.fri_folding_factor, o.fri_max_remainder_coeffs]))
    _check(load().ssh_verify(air.h, tree_kind, coin_kind, bytes(seed), bytes(proof), len(proof), (2 if fri_alpha_times_offset else 1) if shipped_conventions else 0,
                             required_security_bits, exp, pos.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(npos), int(n_friendly_layers)))
    return [int(v) for v in pos[:npos.value]]


class LocalGroup:
    """the meeting point of ranks that are THREADS of this process (host/sharded.cpp LocalTransport): every thread has its own
    backend.Context - on one device, as the tests and a single-GPU box run it, or on several"""

    def __init__(self, world):
        self.world, self.h = world, load().ssh_local_group_create(world)

    def close(self):
        if self.h:
            load().ssh_local_group_destroy(self.h)
            self.h = None


def rccl_unique_id():
    """the 128 bytes rank 0 makes and hands to every rank for ITS RcclGroup (ss_comm_unique_id); an id makes one group"""
    buf = C.create_string_buffer(128)
    be.check(_lib.load().ss_comm_unique_id(buf))
    return buf.raw


class RcclGroup:
    """this rank's end of the RCCL communicator of ranks that are processes, one per GPU (host/sharded.cpp RcclTransport over
    ss_comm_*): made once from rank 0's rccl_unique_id() and kept across proofs"""

    def __init__(self, ctx, unique_id, rank, world):
        self.world, self.rank = world, rank
        self.h = load().ssh_rccl_group_create(ctx.handle, bytes(unique_id), rank, world)
        if not self.h:
            raise _lib.SandstormHipError("host: " + load().ssh_last_error().decode())

    def close(self):
        if self.h:
            load().ssh_rccl_group_destroy(self.h)
            self.h = None


class CallbackGroup:
    """this rank's end of a group whose collectives are the CALLER's (host/sharded.cpp CallbackTransport): all_to_all(send: bytes-like,
    send_counts, recv_counts) -> bytes-like of sum(recv_counts) bytes (MPI_Alltoallv on bytes, slots in rank order) and
    all_gather(mine: bytes-like) -> the ranks' equal-sized contributions in rank order.  The C++ driver stages device memory
    through the host for them.  Goes where an RcclGroup goes."""

    def __init__(self, rank, world, all_to_all, all_gather):
        self.world, self.rank = world, rank
        self._all_to_all, self._all_gather = all_to_all, all_gather

        def a2a(_user, send, send_counts, recv, recv_counts):
            try:
                sc = [int(send_counts[p]) for p in range(world)]
                rc = [int(recv_counts[p]) for p in range(world)]
                src = np.ctypeslib.as_array(send, shape=(sum(sc),)) if sum(sc) else np.zeros(0, dtype=np.uint8)
                got = np.frombuffer(self._all_to_all(src, sc, rc), dtype=np.uint8)
                if got.size != sum(rc):
                    raise ValueError("all_to_all returned %d bytes, %d expected" % (got.size, sum(rc)))
                if got.size:
                    C.memmove(recv, got.ctypes.data, got.size)
                return 0
            except Exception:                   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        def gather(_user, mine, nbytes, out):
            try:
                got = np.frombuffer(self._all_gather(np.ctypeslib.as_array(mine, shape=(int(nbytes),))), dtype=np.uint8)
                if got.size != int(nbytes) * world:
                    raise ValueError("all_gather returned %d bytes, %d expected" % (got.size, int(nbytes) * world))
                C.memmove(out, got.ctypes.data, got.size)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._cbs = (ALL_TO_ALL_CB(a2a), ALL_GATHER_CB(gather))          # kept alive as long as the group
        self.h = load().ssh_callback_group_create(rank, world, self._cbs[0], self._cbs[1], None)
        if not self.h:
            raise _lib.SandstormHipError("host: " + load().ssh_last_error().decode())

    def close(self):
        if self.h:
            load().ssh_rccl_group_destroy(self.h)
            self.h = None


def torch_dist_group(pg=None):
    """a CallbackGroup over a torch.distributed process group with CPU tensors (gloo): the C++ sharded driver between processes
    without RCCL - host-staged, for nodes where RCCL does not come up and for the CPU suite, not for speed"""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(pg), dist.get_world_size(pg)

    def all_to_all(send, send_counts, recv_counts):
        out = torch.empty(sum(recv_counts), dtype=torch.uint8)
        src = torch.from_numpy(np.ascontiguousarray(send)) if len(send) else torch.empty(0, dtype=torch.uint8)
        dist.all_to_all_single(out, src, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=pg)
        return out.numpy()

    def all_gather(mine):
        src = torch.from_numpy(np.array(mine, dtype=np.uint8, copy=True))
        out = torch.empty(src.numel() * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, src, group=pg)
        return out.numpy()
    return CallbackGroup(rank, world, all_to_all, all_gather)


def group_self_check(ctx, rank, world, group, bandwidth_bytes=0):
    """every rank enters, before the first proof over `group` (LocalGroup / RcclGroup / CallbackGroup): messages of different sizes
    between every ordered pair of ranks whose bytes name (source, destination, message), an all-gather, a variable-length all-gather
    (host/sharded.cpp transport_self_check); raises SandstormHipError on the rank that saw a wrong byte.  bandwidth_bytes > 0: then
    one timed equal-split all-to-all of that many bytes per pair.  -> this rank's send + receive rate in GB/s (0.0 if not asked)"""
    h = load()
    h.ssh_group_self_check.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
    gbps = C.c_double(0.0)
    local = isinstance(group, LocalGroup)
    _check(h.ssh_group_self_check(ctx.handle, rank, world, group.h if local else None, None if local else group.h, int(bandwidth_bytes), C.byref(gbps)))
    return gbps.value


def prove_sharded(ctx, air: HostAir, tree_kind, n_friendly, coin_kind, seed, rank, world, group, my_base, log_n, build_extension, options=None,
                  extension_blocks=None):
    """ONE proof over `world` ranks by the C++ host (host/sharded.cpp; the Python mirror is sandstorm_amd/sharded_prover.py).  Called
    by every rank with its own context and AIR handle.  group: a LocalGroup (ranks = threads of this process), this rank's
    RcclGroup (one process per GPU) or a CallbackGroup (one process per GPU, the caller's collectives).  my_base: {column: device column} of the base columns with column % world ==
    rank; build_extension(challenges) -> {global column number: device column} of this rank's extension columns (kept alive by
    the caller).  extension_blocks(challenges) -> [device block of 2^log_n / world rows per extension column] on EVERY rank (the
    scans divided over the ranks: build_extension_blocks) replaces build_extension.  -> the proof in the reference's wire format on
    rank 0, None on the others."""
    options = options or ProofOptions()
    keep = []

    def blocks_cb(_user, ch_ptr, nch, ptrs_out):
        try:
            ch = [np.array([ch_ptr[4 * i + k] for k in range(4)], dtype=np.uint64) for i in range(nch)]
            blks = list(extension_blocks(ch))
            keep.append(blks)
            for i, b in enumerate(blks):
                ptrs_out[i] = be._ptr_of(b)
            return 0
        except Exception:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    def cb(_user, ch_ptr, nch, cols_out, ptrs_out, ncols_out):
        try:
            ch = [np.array([ch_ptr[4 * i + k] for k in range(4)], dtype=np.uint64) for i in range(nch)]
            cols = build_extension(ch)
            keep.append(cols)
            for i, (c, col) in enumerate(sorted(cols.items())):
                cols_out[i] = c
                ptrs_out[i] = be._ptr_of(col)
            ncols_out[0] = len(cols)
            return 0
        except Exception:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1
    opts = (C.c_uint32 * 5)(options.num_queries, options.lde_blowup_factor, options.grinding_factor,
                            options.fri_folding_factor, options.fri_max_remainder_coeffs)
    cols = sorted(my_base)
    col_ids = (C.c_uint32 * max(1, len(cols)))(*cols)
    out, n = C.POINTER(C.c_uint8)(), C.c_uint64()
    local = isinstance(group, LocalGroup)
    entry, callback = ((load().ssh_prove_sharded_blocks, EXT_BLOCKS_CB(blocks_cb)) if extension_blocks is not None else
                       (load().ssh_prove_sharded, SHARDED_EXT_CB(cb)))
    _check(entry(ctx.handle, air.h, tree_kind, n_friendly, coin_kind, bytes(seed), rank, world, group.h if local else None,
                 None if local else group.h, col_ids, be._ptr_array([my_base[c] for c in cols]), len(cols), log_n,
                 callback, None, opts, C.byref(out), C.byref(n)))
    if not n.value:
        return None
    raw = bytes(bytearray(out[:n.value]))
    load().ssh_free(out)
    return raw


class HostCoin:
    """the C++ PublicCoin, for cross-checks against the oracle coin"""

    def __init__(self, kind, seed):
        self.c = load().ssh_coin_new(kind, bytes(seed))

    def _op(self, op, data=b"", felts=None, arg=0, nout=4, length=None):
        f = np.ascontiguousarray(felts, dtype=np.uint64) if felts is not None else np.zeros((1, 4), dtype=np.uint64)
        out = np.zeros(max(8, nout), dtype=np.uint64)
        cnt = C.c_uint32()
        _check(load().ssh_coin_op(self.c, op, bytes(data), len(data) if length is None else length,
                                  f.ctypes.data_as(C.POINTER(C.c_uint64)), 0 if felts is None else len(f), arg,
                                  out.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(cnt)))
        return out, cnt.value

    def reseed_bytes(self, b): self._op(0, b)
    def reseed_felts(self, v): self._op(1, felts=v)
    def reseed_felt_vector(self, v): self._op(2, felts=v)
    def reseed_int(self, v): self._op(3, arg=v)
    def draw(self): return self._op(4)[0][:4].copy()

    def draw_queries(self, max_n, domain):
        out, n = self._op(5, arg=max_n, nout=max_n + 4, length=domain)
        return [int(v) for v in out[:n]]

    @property
    def state(self):
        out, _ = self._op(6, nout=8)
        return out[:4].tobytes(), int(out[4])

    def __del__(self):
        try:
            load().ssh_coin_free(self.c)
        except Exception:
            pass

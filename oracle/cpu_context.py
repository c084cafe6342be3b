"""ORACLE - TEST INFRASTRUCTURE ONLY.  The CPU oracle behind the interface of `sandstorm_amd.backend.Context`.

`CpuContext` answers every call the host pipelines (sandstorm_amd/prover.py, extension.py, sharded.py, the layouts'
`make_air`) make on a `backend.Context` with the oracle's C restatement (liboracle.so) on host memory, so that the SAME
host code that sequences the HIP kernels can be run, unchanged, on the CPU:

  * bench.py's `cpu_baseline` leg times a whole proof (LDE, hashing, constraint program, DEEP, FRI, proof of work) of the
    oracle on the GPU box's host cores;
  * the multi-rank `gloo` tests run the sharded prover's real driver code with the oracle standing in for the kernels;
  * tests/test_gpu_prove.py compares a whole GPU proof with the whole CPU proof, byte for byte.

Nothing under sandstorm_amd/ imports this module: the product has no CPU path (its entry points fail without a gfx950
device).  "Device" buffers here are numpy arrays; their `ptr` is the host address, so `backend.DeviceView`,
`backend.Matrix` and the tree classes work on them as they are.
"""
import ctypes as C

import numpy as np

from . import oracle_py as oracle

NATURAL, BITREV = 0, 1
FORWARD, INVERSE = 0, 1
LEAF_DIGEST, LEAF_FELT = 0, 1


class HostBuffer:
    """stands where backend.DeviceBuffer stands"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.mem = np.zeros(max(8, (self.nbytes + 7) // 8), dtype=np.uint64)
        self.ptr = self.mem.ctypes.data

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        C.memmove(self.ptr, a.ctypes.data, a.nbytes)
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        C.memmove(out.ctypes.data, self.ptr, out.nbytes)
        return out

    def free(self):
        pass


class _LibShim:
    """what backend.DeviceView / hostlib.HostMatrix call on ctx.lib"""

    @staticmethod
    def ss_download(_handle, dst, src, nbytes):
        C.memmove(dst, int(src), int(nbytes))
        return 0

    @staticmethod
    def ss_upload(_handle, dst, src, nbytes):
        C.memmove(int(dst), src, int(nbytes))
        return 0


def _addr(x):
    if hasattr(x, "ptr"):
        return int(x.ptr)
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    return int(x)


def _felts(x, count):
    """numpy view uint64[count, 4] of host memory at x"""
    if count == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    return np.ctypeslib.as_array((C.c_uint64 * (4 * count)).from_address(_addr(x))).reshape(count, 4)


def _bytes32(x, count):
    return np.ctypeslib.as_array((C.c_uint8 * (32 * count)).from_address(_addr(x))).reshape(count, 32)


def _fp_arg(limbs):
    return oracle._fp(np.ascontiguousarray(limbs, dtype=np.uint64))


class CpuContext:
    def __init__(self):
        oracle.lib()
        self.lib, self.handle = _LibShim(), None
        self._declare()

    def _declare(self):
        l = oracle.lib()
        if getattr(l, "_cpu_context_declared", False):
            return
        l.or_deep_compose_rows.restype = None
        l.or_eval_program_rows.restype = None
        l.or_ood_eval.restype = None
        l.or_inverse_table.restype = None
        l._cpu_context_declared = True

    # ---- memory
    def alloc(self, nbytes):
        return HostBuffer(self, nbytes)

    def column(self, host_col):
        a = np.ascontiguousarray(host_col, dtype=np.uint64)
        return HostBuffer(self, a.nbytes).upload(a)

    def sync(self):
        pass

    def trim(self):
        pass

    def close(self):
        pass

    def zero(self, buf, nbytes=None):
        C.memset(_addr(buf), 0, buf.nbytes if nbytes is None else nbytes)

    # ---- N1 / N2 (ss_ntt_fp252, ss_lde_fp252, ss_evaluate_fp252)
    def ntt(self, cols, log_n, direction=FORWARD, offset=None, in_order=NATURAL, out_order=NATURAL):
        n = 1 << log_n
        for c in cols:
            v = _felts(c, n)
            a = oracle.bitrev_permute(v) if in_order == BITREV else v.copy()
            a = oracle.ntt(a, inverse=direction == INVERSE, offset=offset)
            v[:] = oracle.bitrev_permute(a) if out_order == BITREV else a

    def lde(self, cols_in, log_n, log_blowup, offset, evals_out, coeffs_out=None):
        n = 1 << log_n
        for k, c in enumerate(cols_in):
            ev, co = oracle.lde(_felts(c, n), log_blowup, offset)
            _felts(evals_out[k], n << log_blowup)[:] = ev
            if coeffs_out:
                _felts(coeffs_out[k], n)[:] = oracle.bitrev_permute(co)      # the device keeps coefficients bit-reversed

    def evaluate(self, coeff_cols, log_n, log_blowup, offset, evals_out):
        n, N = 1 << log_n, 1 << (log_n + log_blowup)
        for k, c in enumerate(coeff_cols):
            a = np.zeros((N, 4), dtype=np.uint64)
            a[:n] = oracle.bitrev_permute(_felts(c, n))
            _felts(evals_out[k], N)[:] = oracle.ntt(a, offset=offset)

    # ---- H1..H4
    def hash_rows(self, kind, cols, nrows, out, order=NATURAL):
        d = oracle.hash_rows(kind, [_felts(c, nrows) for c in cols])
        if order == BITREV:
            d = d[_bitrev_indices(nrows)]
        _bytes32(out, nrows)[:] = d

    def merkle_build(self, tree, n_friendly, leaf_kind, leaves, n, nodes, tags=None, leaf_order=NATURAL):
        if leaf_kind == LEAF_FELT:
            lv = _felts(leaves, n)
            if leaf_order == BITREV:
                lv = oracle.bitrev_permute(lv)
        else:
            lv = _bytes32(leaves, n)
        nd, tg = oracle.merkle_build(tree, n_friendly, leaf_kind, lv)
        _bytes32(nodes, 2 * n)[:] = nd
        if tags is not None:
            np.ctypeslib.as_array((C.c_uint8 * (2 * n)).from_address(_addr(tags)))[:] = tg
        return bytes(nd[1]), int(tg[1])

    def merkle_open(self, nodes, tags, n, indices):
        nd = _bytes32(nodes, 2 * n)
        tg = np.ctypeslib.as_array((C.c_uint8 * (2 * n)).from_address(_addr(tags))) if tags is not None else None
        log_n = int(n).bit_length() - 1
        out = np.zeros((len(indices), log_n, 32), dtype=np.uint8)
        otags = np.zeros((len(indices), log_n), dtype=np.uint8)
        for q, idx in enumerate(indices):
            k = n + int(idx)
            for lvl in range(log_n):
                out[q, lvl] = nd[k ^ 1]
                if tg is not None:
                    otags[q, lvl] = tg[k ^ 1]
                k >>= 1
        return out, otags

    def gather_rows(self, cols, indices):
        out = np.zeros((len(indices), len(cols), 4), dtype=np.uint64)
        for k, c in enumerate(cols):
            base = _addr(c)
            for q, i in enumerate(indices):
                out[q, k] = _felts(base + 32 * int(i), 1)[0]
        return out

    def bitrev_permute32(self, src, log_n, dst):
        """dst[i] = src[bitrev(i)] over 32-byte records (ss_bitrev_permute32)"""
        n = 1 << log_n
        idx = np.arange(n, dtype=np.uint64)
        rev = np.zeros(n, dtype=np.uint64)
        for b in range(log_n):
            rev |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(log_n - 1 - b)
        _bytes32(dst, n)[:] = _bytes32(src, n)[rev.astype(np.int64)]

    # ---- F1, C2
    def fri_fold(self, evals, log_len, fold, alpha, offset, out, flags=0):
        n = 1 << log_len
        _felts(out, n // fold)[:] = oracle.fri_fold(_felts(evals, n), fold, alpha, offset, flags)

    def pow_grind(self, coin_kind, digest, bits):
        return oracle.Coin(coin_kind, bytes(digest)).grind(bits)

    # ---- D1
    def poly_eval(self, coeff_cols, log_n, x):
        n = 1 << log_n
        return np.stack([oracle.poly_eval(oracle.bitrev_permute(_felts(c, n)), x) for c in coeff_cols])

    def ood_eval(self, coeff_cols, log_n, mask_col, mask_off, z):
        n = 1 << log_n
        nat = [np.ascontiguousarray(oracle.bitrev_permute(_felts(c, n))) for c in coeff_cols]
        ptrs = (C.c_void_p * len(nat))(*[a.ctypes.data for a in nat])
        mc = np.ascontiguousarray(mask_col, dtype=np.uint32)
        mo = np.ascontiguousarray(mask_off, dtype=np.uint32)
        out = np.zeros((len(mc), 4), dtype=np.uint64)
        oracle.lib().or_ood_eval(ptrs, C.c_uint(log_n), mc.ctypes.data_as(C.c_void_p), mo.ctypes.data_as(C.c_void_p),
                                 C.c_size_t(len(mc)), _fp_arg(z), out.ctypes.data_as(C.c_void_p))
        return out

    def deep_compose(self, trace_cols, comp_cols, log_n, log_blowup, offset, mask_col, mask_off, ood_trace,
                     coeff_trace, ood_comp, coeff_comp, z, out, row0=0, nrows=None, stride=1):
        """row0 / nrows / stride: the row-block form (out[j] = value at LDE row row0 + j * stride, the columns start at row0)"""
        N = 1 << (log_n + log_blowup)
        nrows = N if nrows is None else nrows
        tp = (C.c_void_p * len(trace_cols))(*[_addr(c) for c in trace_cols])
        cp = (C.c_void_p * max(1, len(comp_cols)))(*[_addr(c) for c in comp_cols])
        mc = np.ascontiguousarray(mask_col, dtype=np.uint32)
        mo = np.ascontiguousarray(mask_off, dtype=np.uint32)
        ot, ct, oc, cc = (np.ascontiguousarray(a, dtype=np.uint64) for a in (ood_trace, coeff_trace, ood_comp, coeff_comp))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        oracle.lib().or_deep_compose_rows(tp, cp, C.c_uint(log_n), C.c_uint(log_blowup), _fp_arg(offset), p(mc), p(mo),
                                          C.c_size_t(len(mc)), p(ot), p(ct), C.c_size_t(len(comp_cols)), p(oc), p(cc),
                                          _fp_arg(z), C.c_uint64(row0), C.c_uint64(nrows), C.c_uint64(stride), C.c_void_p(_addr(out)))

    def deep_compose_rows(self, trace_blocks, comp_blocks, log_n, log_blowup, offset, mask_col, mask_off, ood_trace,
                          coeff_trace, ood_comp, coeff_comp, z, m0, count, out):
        """ss_deep_compose_rows: the DEEP polynomial at the sub-coset points m0 .. m0 + count"""
        self.deep_compose(trace_blocks, comp_blocks, log_n, log_blowup, offset, mask_col, mask_off, ood_trace, coeff_trace,
                          ood_comp, coeff_comp, z, out, row0=m0 << log_blowup, nrows=count, stride=1 << log_blowup)

    def deep_extend(self, subcoset, log_n, log_blowup, offset, out):
        """ss_deep_extend: n values on offset * <w_n> of a polynomial of degree < n -> its evaluations on offset * <w_N>"""
        n, N = 1 << log_n, 1 << (log_n + log_blowup)
        a = np.zeros((N, 4), dtype=np.uint64)
        a[:n] = oracle.ntt(_felts(subcoset, n), inverse=True, offset=offset)
        _felts(out, N)[:] = oracle.ntt(a, offset=offset)

    # ---- Q1
    def inverse_table(self, log_N, offset, c, out):
        oracle.lib().or_inverse_table(C.c_uint(log_N), _fp_arg(offset), _fp_arg(c), C.c_void_p(_addr(out)))

    def eval_quotient(self, program, tables, table_desc, lde_cols, log_n, log_blowup, offset, out):
        N = 1 << (log_n + log_blowup)
        consts = oracle.to_mont(list(program.consts)) if len(program.consts) else np.zeros((0, 4), dtype=np.uint64)
        ntab = 0
        for k in range(0, len(table_desc), 2):
            ntab = max(ntab, table_desc[k] + (1 << table_desc[k + 1]))
        tab = _felts(tables, ntab) if tables is not None and ntab else np.zeros((0, 4), dtype=np.uint64)
        _felts(out, N)[:] = oracle.eval_program(program.code, consts, tab, list(table_desc), program.n_slots,
                                                [_felts(c, N) for c in lde_cols], log_n, log_blowup, offset)

    def eval_quotient_rows(self, program, tables, table_desc, col_blocks, log_n, log_blowup, offset, row0, nrows, block_rows, out):
        consts = np.ascontiguousarray(oracle.to_mont(list(program.consts))) if len(program.consts) else np.zeros((1, 4), dtype=np.uint64)
        code = np.ascontiguousarray(program.code, dtype=np.uint32)
        desc = np.ascontiguousarray(table_desc, dtype=np.uint32) if len(table_desc) else np.zeros(2, dtype=np.uint32)
        prog = oracle.AirProgram(code.ctypes.data_as(C.POINTER(C.c_uint32)), len(code) // 2,
                                 consts.ctypes.data_as(C.POINTER(C.c_uint64)), len(program.consts),
                                 None, desc.ctypes.data_as(C.POINTER(C.c_uint32)), len(table_desc) // 2, program.n_slots)
        cp = (C.c_void_p * len(col_blocks))(*[_addr(c) for c in col_blocks])
        oracle.lib().or_eval_program_rows(C.byref(prog), C.c_void_p(_addr(tables)) if tables is not None else None, cp, C.c_uint(log_n),
                                          C.c_uint(log_blowup), _fp_arg(offset), C.c_uint64(row0), C.c_uint64(nrows), C.c_void_p(_addr(out)))

    # ---- A2
    def permutation_product(self, num, den, count, z, alpha, out, out_stride=1, out_offset=0, want_last=True):
        def operand(o):
            span = (count - 1) * o[1] + max(o[2], o[3] if o[3] >= 0 else 0) + 1
            return (_felts(o[0], span), o[1], o[2], o[3])
        ov = _felts(out, (count - 1) * out_stride + out_offset + 1)
        last = oracle.permutation_product(operand(num), operand(den), count, z,
                                          alpha if alpha is not None else np.zeros(4, dtype=np.uint64), ov, out_stride, out_offset)
        return last if want_last else None

    def diluted_aggregate(self, ordered, stride, offset, count, z, alpha, out, out_stride=1, out_offset=0):
        ov = _felts(out, (count - 1) * out_stride + out_offset + 1)
        oracle.diluted_aggregate(_felts(ordered, (count - 1) * stride + offset + 1), stride, offset, count, z, alpha, ov,
                                 out_stride, out_offset)

    # profile hooks of backend.Context (no-ops)
    def profile(self, on):
        pass

    def profile_reset(self):
        pass

    def profile_read(self, kind):
        return 0.0, 0


_BITREV_CACHE = {}


def _bitrev_indices(n):
    idx = _BITREV_CACHE.get(n)
    if idx is None:
        bits = n.bit_length() - 1
        idx = np.arange(n, dtype=np.uint64)
        rev = np.zeros(n, dtype=np.uint64)
        for b in range(bits):
            rev |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(bits - 1 - b)
        idx = _BITREV_CACHE[n] = rev.astype(np.int64)
    return idx

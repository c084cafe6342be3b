"""Host-side mirror of the reference's prover-facing interface for the hot path,
on top of the C ABI (include/sandstorm_hip.h).

Names follow the reference / ministark:
  Matrix                 ministark::Matrix<Fp> (column-major; layouts/src/recursive/trace.rs:652-660)
    .interpolate()/.evaluate()/.lde()   ministark Matrix::{interpolate, evaluate}
  hash_rows              crypto/src/merkle/utils.rs:19-46
  LeafVariantMerkleTree  crypto/src/merkle/mod.rs:240-304
  FriendlyMerkleTree     crypto/src/merkle/mod.rs:43-123
    .from_matrix()/.root()/.prove()     MatrixMerkleTree / MerkleTree traits
  fri_fold, pow_grind, pedersen_hash    see the C ABI header

Field elements are numpy uint64[..., 4] Montgomery limbs on the host and 32-byte
images on the device.  All arithmetic runs on the GPU; nothing here falls back to
the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

P = 2**251 + 17 * 2**192 + 1
R = 2**256

NATURAL, BITREV = 0, 1
FRI_BITREV_ROWS, FRI_UNNORMALISED = 1, 2
FORWARD, INVERSE = 0, 1
NTT_PART_LOCAL, NTT_PART_CROSS = 0, 1
HASH_KECCAK, HASH_KECCAK_M20, HASH_BLAKE2S, HASH_BLAKE2S_M20, HASH_SHA256 = 0, 1, 2, 3, 4
TREE_KECCAK, TREE_KECCAK_M20, TREE_FRIENDLY, TREE_BLAKE2S, TREE_SHA256 = 0, 1, 2, 3, 4
LEAF_DIGEST, LEAF_FELT = 0, 1
COIN_SOLIDITY, COIN_CAIRO = 0, 1
PROF_NTT_PASS, PROF_HASH_ROWS, PROF_MERKLE, PROF_FRI, PROF_QUOTIENT, PROF_DEEP, PROF_EXT, PROF_TRACE = range(8)


def felt(v):
    """python int -> Montgomery limbs uint64[4]"""
    x = (int(v) % P) * R % P
    return np.array([(x >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)


def _felt_ptr(limbs):
    if limbs is None:
        return None, None
    a = np.ascontiguousarray(limbs, dtype=np.uint64)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


def pedersen_hash_host(a, b):
    """host-side pedersen_hash (Fiat-Shamir coin only); Montgomery limbs in and out"""
    lib = _lib.load()
    x = np.ascontiguousarray(a, dtype=np.uint64)
    y = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    p64 = C.POINTER(C.c_uint64)
    check(lib.ss_pedersen_hash_host(x.ctypes.data_as(p64), y.ctypes.data_as(p64), out.ctypes.data_as(p64)))
    return out


class DeviceBuffer:
    """A device allocation owned by a Context (ss_dev_alloc / ss_dev_free)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        check(ctx.lib.ss_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        check(self.ctx.lib.ss_upload(self.ctx.handle, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.ctx.lib.ss_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.ss_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView:
    """A sub-range of a DeviceBuffer (no ownership): e.g. one FRI-layer column, or one
    half of the composition coefficients."""

    def __init__(self, parent, offset, nbytes):
        assert offset + nbytes <= parent.nbytes
        self.parent, self.ctx, self.nbytes = parent, parent.ctx, int(nbytes)
        self.ptr = parent.ptr + int(offset)

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.ctx.lib.ss_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out


def _ptr_of(x):
    """DeviceBuffer | torch tensor | int -> device address"""
    if hasattr(x, "ptr"):                   # DeviceBuffer, DeviceView, and what stands in for them
        return int(x.ptr)
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


def _ptr_array(items):
    return (C.c_void_p * len(items))(*[_ptr_of(i) for i in items])


class Context:
    """One per GPU and host thread (ss_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.ss_ctx_create(device, C.byref(h)))
        self.handle = h
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        check(self.lib.ss_ctx_set_stream(self.handle, C.c_void_p(stream)))

    def sync(self):
        check(self.lib.ss_ctx_sync(self.handle))

    def trim(self):
        """Return the blocks cached by the ss_dev_alloc pool to the driver."""
        check(self.lib.ss_ctx_trim(self.handle))

    def close(self):
        if self.handle:
            self.lib.ss_ctx_destroy(self.handle)
            self.handle = None

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def column(self, host_col):
        a = np.ascontiguousarray(host_col, dtype=np.uint64)
        return DeviceBuffer(self, a.nbytes).upload(a)

    # ---- raw ops on lists of device columns -------------------------------------------------
    def ntt(self, cols, log_n, direction=FORWARD, offset=None, in_order=NATURAL, out_order=NATURAL):
        _keep, off = _felt_ptr(offset)
        check(self.lib.ss_ntt_fp252(self.handle, _ptr_array(cols), len(cols), log_n, direction, off,
                                    in_order, out_order))

    def lde(self, cols_in, log_n, log_blowup, offset, evals_out, coeffs_out=None):
        _keep, off = _felt_ptr(offset)
        check(self.lib.ss_lde_fp252(self.handle, _ptr_array(cols_in), len(cols_in), log_n, log_blowup, off,
                                    _ptr_array(evals_out), _ptr_array(coeffs_out) if coeffs_out else None))

    def evaluate(self, coeff_cols, log_n, log_blowup, offset, evals_out):
        """bit-reversed coefficient columns -> evaluations over offset*<w_{n*blowup}>"""
        _keep, off = _felt_ptr(offset)
        check(self.lib.ss_evaluate_fp252(self.handle, _ptr_array(coeff_cols), len(coeff_cols), log_n, log_blowup,
                                         off, _ptr_array(evals_out)))

    def ntt_shard(self, cols, log_n, log_ranks, rank, direction, offset, part, log_expand=0, out=None):
        """ss_ntt_shard_fp252: this rank's share (part LOCAL / CROSS) of ONE transform of 2^log_n points spread over 2^log_ranks ranks"""
        _keep, off = _felt_ptr(offset)
        check(self.lib.ss_ntt_shard_fp252(self.handle, _ptr_array(cols), len(cols), log_n, log_ranks, rank, direction, off, part, log_expand,
                                          _ptr_array(out) if out else None))

    def hash_rows(self, kind, cols, nrows, out, order=NATURAL):
        """order=BITREV: digest i is the hash of row bitrev(i) - the reference's commitment order"""
        check(self.lib.ss_hash_rows_ex(self.handle, kind, _ptr_array(cols), len(cols), nrows, order, _ptr_of(out)))

    def merkle_build(self, tree, n_friendly, leaf_kind, leaves, n, nodes, tags=None, leaf_order=NATURAL):
        root = (C.c_uint8 * 33)()
        check(self.lib.ss_merkle_build_ex(self.handle, tree, n_friendly, leaf_kind, _ptr_of(leaves), n, leaf_order,
                                          _ptr_of(nodes), _ptr_of(tags) if tags is not None else None, root))
        return bytes(root[:32]), int(root[32])

    def merkle_open(self, nodes, tags, n, indices):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        log_n = int(n).bit_length() - 1
        out = np.zeros((len(idx), log_n, 32), dtype=np.uint8)
        otags = np.zeros((len(idx), log_n), dtype=np.uint8)
        check(self.lib.ss_merkle_open(self.handle, _ptr_of(nodes), _ptr_of(tags) if tags is not None else None,
                                      n, idx.ctypes.data_as(C.POINTER(C.c_uint64)), len(idx),
                                      out.ctypes.data, otags.ctypes.data))
        return out, otags

    def bitrev_permute32(self, src, log_n, dst):
        """dst[i] = src[bitrev(i)], i < 2^log_n, over 32-byte records (field elements or digests); not in place"""
        check(self.lib.ss_bitrev_permute32(self.handle, _ptr_of(src), log_n, _ptr_of(dst)))

    def gather_rows(self, cols, indices):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        out = np.zeros((len(idx), len(cols), 4), dtype=np.uint64)
        check(self.lib.ss_gather_rows(self.handle, _ptr_array(cols), len(cols),
                                      idx.ctypes.data_as(C.POINTER(C.c_uint64)), len(idx), out.ctypes.data))
        return out

    def gather_batch(self, jobs):
        """ss_gather_batch: jobs = [(device arrays, entry_bytes (32 | 1), indices)] -> one array per job, uint64[nidx, ncols, 4] for 32-byte
        entries / uint8[nidx] for single bytes; one upload, one download, one synchronisation for all of them"""
        from ._lib import GatherJob
        arr = (GatherJob * max(1, len(jobs)))()
        keep, outs = [], []
        for j, (cols, entry_bytes, indices) in enumerate(jobs):
            idx = np.ascontiguousarray(indices, dtype=np.uint64)
            out = np.zeros((len(idx), len(cols), 4), dtype=np.uint64) if entry_bytes == 32 else np.zeros(len(idx) * len(cols), dtype=np.uint8)
            ptrs = _ptr_array(cols)
            keep.append((idx, ptrs))
            outs.append(out)
            arr[j].d_cols = C.cast(ptrs, C.POINTER(C.c_void_p))
            arr[j].ncols, arr[j].entry_bytes = len(cols), entry_bytes
            arr[j].idx = idx.ctypes.data_as(C.POINTER(C.c_uint64))
            arr[j].nidx = len(idx)
            arr[j].out = out.ctypes.data
        check(self.lib.ss_gather_batch(self.handle, arr, len(jobs)))
        return outs

    def fri_fold(self, evals, log_len, fold, alpha, offset, out, flags=0):
        """flags: FRI_BITREV_ROWS | FRI_UNNORMALISED (include/sandstorm_hip.h: the conventions of the reference's proofs)"""
        _k1, a = _felt_ptr(alpha)
        _k2, o = _felt_ptr(offset)
        check(self.lib.ss_fri_fold_ex(self.handle, _ptr_of(evals), log_len, fold, a, o, flags, _ptr_of(out)))

    def fri_fold_rows(self, evals, log_len, fold, alpha, offset, row0, count, out, flags=0):
        """ss_fri_fold_rows: rows row0 .. row0 + count of a layer, entry k of row row0 + i at evals[k * count + i]"""
        _k1, a = _felt_ptr(alpha)
        _k2, o = _felt_ptr(offset)
        check(self.lib.ss_fri_fold_rows(self.handle, _ptr_of(evals), log_len, fold, a, o, flags, row0, count, _ptr_of(out)))

    # ---- X4: the 64-bit field (p = 2^64 - 2^32 + 1) and its cubic extension
    def ntt_gl64(self, cols, log_n, direction=FORWARD, offset=1, in_order=NATURAL, out_order=NATURAL):
        check(self.lib.ss_ntt_gl64(self.handle, _ptr_array(cols), len(cols), log_n, direction, int(offset), in_order, out_order))

    def lde_gl64(self, cols_in, log_n, log_blowup, offset, evals_out, coeffs_out=None):
        check(self.lib.ss_lde_gl64(self.handle, _ptr_array(cols_in), len(cols_in), log_n, log_blowup, int(offset),
                                   _ptr_array(evals_out), _ptr_array(coeffs_out) if coeffs_out else None))

    def fri_fold_gl64x3(self, evals, log_len, fold, alpha, offset, out, flags=0):
        a = np.ascontiguousarray(alpha, dtype=np.uint64)
        check(self.lib.ss_fri_fold_gl64x3(self.handle, _ptr_of(evals), log_len, fold, a.ctypes.data_as(C.POINTER(C.c_uint64)), int(offset),
                                          flags, _ptr_of(out)))

    def running_product_gl64x3(self, num_addr, num_val, den_addr, den_val, stride, count, z, alpha, out_cols, out_stride, out_offset, want_last=True):
        """ss_running_product_gl64x3: the permutation argument's running quotient over Fq3 into three coordinate columns -> the last value"""
        u64 = C.POINTER(C.c_uint64)
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        aa = np.ascontiguousarray(alpha, dtype=np.uint64) if alpha is not None else None
        last = np.zeros(3, dtype=np.uint64)
        check(self.lib.ss_running_product_gl64x3(self.handle, _ptr_of(num_addr), _ptr_of(num_val) if num_val is not None else None, _ptr_of(den_addr),
                                                 _ptr_of(den_val) if den_val is not None else None, stride, count, zz.ctypes.data_as(u64),
                                                 aa.ctypes.data_as(u64) if aa is not None else None, _ptr_array(out_cols), out_stride, out_offset,
                                                 last.ctypes.data_as(u64) if want_last else None))
        return tuple(int(v) for v in last) if want_last else None

    def hash_rows_gl64(self, segments, seg_len, nrows, out, hash_kind=None):
        """Keccak-256 / Blake2s-256 of the rows of a matrix of 8-byte elements (ss_hash_rows_gl64) -> out[nrows][32]"""
        check(self.lib.ss_hash_rows_gl64(self.handle, HASH_BLAKE2S if hash_kind is None else hash_kind, _ptr_array(segments), len(segments), seg_len,
                                         nrows, _ptr_of(out)))

    def gather_rows_gl64(self, segments, seg_len, nrows, idx):
        """rows idx (< nrows) of such a matrix -> uint64[len(idx), nseg, seg_len]"""
        ix = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.zeros((len(ix), len(segments), seg_len), dtype=np.uint64)
        u64 = C.POINTER(C.c_uint64)
        check(self.lib.ss_gather_rows_gl64(self.handle, _ptr_array(segments), len(segments), seg_len, nrows, ix.ctypes.data_as(u64), len(ix), out.ctypes.data_as(u64)))
        return out

    def eval_quotient_gl64x3(self, code, consts3, n_slots, tables, table_desc, lde_cols, log_n, log_blowup, offset, out):
        """the constraint program over Fq3 (ss_eval_quotient_gl64x3): code = the ss_air_program words, consts3 = uint64[n, 3]"""
        code = np.ascontiguousarray(code, dtype=np.uint32)
        consts = np.ascontiguousarray(consts3, dtype=np.uint64).reshape(-1, 3) if len(consts3) else np.zeros((1, 3), dtype=np.uint64)
        desc = np.ascontiguousarray(table_desc if len(table_desc) else [0, 0], dtype=np.uint32)
        prog = _lib.AirProgram(code.ctypes.data_as(C.POINTER(C.c_uint32)), len(code) // 2, consts.ctypes.data_as(C.POINTER(C.c_uint64)), len(consts3),
                               _ptr_of(tables) if tables is not None else None, desc.ctypes.data_as(C.POINTER(C.c_uint32)), len(table_desc) // 2, n_slots)
        check(self.lib.ss_eval_quotient_gl64x3(self.handle, C.byref(prog), _ptr_array(lde_cols), len(lde_cols), log_n, log_blowup, int(offset), _ptr_of(out)))

    def ood_eval_gl64x3(self, coeff_cols, log_n, cell_col, cell_off, z):
        """P_{col_j}(z * w_n^{off_j}) for bit-reversed coefficient columns, z in Fq3 -> uint64[ncells, 3]"""
        cc, co = np.ascontiguousarray(cell_col, dtype=np.uint32), np.ascontiguousarray(cell_off, dtype=np.uint32)
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros((len(cc), 3), dtype=np.uint64)
        u32, u64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        check(self.lib.ss_ood_eval_gl64x3(self.handle, _ptr_array(coeff_cols), len(coeff_cols), log_n, cc.ctypes.data_as(u32), co.ctypes.data_as(u32),
                                          len(cc), zz.ctypes.data_as(u64), out.ctypes.data_as(u64)))
        return out

    def deep_compose_gl64x3(self, trace_cols, comp_cols, log_n, log_blowup, offset, mask_col, mask_off, ood_trace, coeff_trace, ood_comp,
                            coeff_comp, z, z_comp, out):
        """the DEEP composition over Fq3 (ss_deep_compose_gl64x3); out: [n * blowup][3] interleaved"""
        u32, u64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        mc, mo = np.ascontiguousarray(mask_col, dtype=np.uint32), np.ascontiguousarray(mask_off, dtype=np.uint32)
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (ood_trace, coeff_trace, ood_comp, coeff_comp, z, z_comp)]
        check(self.lib.ss_deep_compose_gl64x3(self.handle, _ptr_array(trace_cols), len(trace_cols), _ptr_array(comp_cols) if comp_cols else None,
                                              len(comp_cols), log_n, log_blowup, int(offset), mc.ctypes.data_as(u32), mo.ctypes.data_as(u32), len(mc),
                                              *[a.ctypes.data_as(u64) for a in arrs], _ptr_of(out)))

    def pow_grind(self, coin_kind, digest, bits):
        nonce = C.c_uint64()
        check(self.lib.ss_pow_grind(self.handle, coin_kind, bytes(digest), bits, C.byref(nonce)))
        return nonce.value

    def pedersen_hash(self, a, b, n, out):
        check(self.lib.ss_pedersen_hash(self.handle, _ptr_of(a), _ptr_of(b), n, _ptr_of(out)))

    # ---- D1 -------------------------------------------------------------------------------
    def poly_eval(self, coeff_cols, log_n, x):
        """P_c(x) for bit-reversed coefficient columns -> uint64[ncols, 4]"""
        _k, xp = _felt_ptr(x)
        out = np.zeros((len(coeff_cols), 4), dtype=np.uint64)
        check(self.lib.ss_poly_eval(self.handle, _ptr_array(coeff_cols), len(coeff_cols), log_n, xp,
                                    out.ctypes.data))
        return out

    def ood_eval(self, coeff_cols, log_n, mask_col, mask_off, z):
        """T_{col_j}(z * w_n^{off_j}) -> uint64[nmask, 4]"""
        mc = np.ascontiguousarray(mask_col, dtype=np.uint32)
        mo = np.ascontiguousarray(mask_off, dtype=np.uint32)
        _k, zp = _felt_ptr(z)
        out = np.zeros((len(mc), 4), dtype=np.uint64)
        check(self.lib.ss_ood_eval(self.handle, _ptr_array(coeff_cols), len(coeff_cols), log_n,
                                   mc.ctypes.data_as(C.POINTER(C.c_uint32)), mo.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   len(mc), zp, out.ctypes.data))
        return out

    def deep_prepare(self, ncomp, log_n, offset, z):
        """ss_deep_prepare: DEEP's denominator tables for this out-of-domain point, queued (no wait) for the next deep_compose with it"""
        _k1, op = _felt_ptr(offset)
        _k2, zp = _felt_ptr(z)
        check(self.lib.ss_deep_prepare(self.handle, ncomp, log_n, op, zp))

    def deep_compose(self, trace_cols, comp_cols, log_n, log_blowup, offset, mask_col, mask_off, ood_trace,
                     coeff_trace, ood_comp, coeff_comp, z, out):
        mc = np.ascontiguousarray(mask_col, dtype=np.uint32)
        mo = np.ascontiguousarray(mask_off, dtype=np.uint32)
        ot, ct = (np.ascontiguousarray(a, dtype=np.uint64) for a in (ood_trace, coeff_trace))
        oc, cc = (np.ascontiguousarray(a, dtype=np.uint64) for a in (ood_comp, coeff_comp))
        _k1, op = _felt_ptr(offset)
        _k2, zp = _felt_ptr(z)
        check(self.lib.ss_deep_compose(self.handle, _ptr_array(trace_cols), len(trace_cols),
                                       _ptr_array(comp_cols) if comp_cols else None, len(comp_cols), log_n,
                                       log_blowup, op, mc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       mo.ctypes.data_as(C.POINTER(C.c_uint32)), len(mc), ot.ctypes.data,
                                       ct.ctypes.data, oc.ctypes.data, cc.ctypes.data, zp, _ptr_of(out)))

    def deep_compose_rows(self, trace_blocks, comp_blocks, log_n, log_blowup, offset, mask_col, mask_off, ood_trace,
                          coeff_trace, ood_comp, coeff_comp, z, m0, count, out):
        """ss_deep_compose_rows: the DEEP polynomial at the sub-coset points m0 .. m0 + count from column blocks whose
        element k is LDE row (m0 << log_blowup) + k"""
        mc = np.ascontiguousarray(mask_col, dtype=np.uint32)
        mo = np.ascontiguousarray(mask_off, dtype=np.uint32)
        ot, ct = (np.ascontiguousarray(a, dtype=np.uint64) for a in (ood_trace, coeff_trace))
        oc, cc = (np.ascontiguousarray(a, dtype=np.uint64) for a in (ood_comp, coeff_comp))
        _k1, op = _felt_ptr(offset)
        _k2, zp = _felt_ptr(z)
        check(self.lib.ss_deep_compose_rows(self.handle, _ptr_array(trace_blocks), len(trace_blocks),
                                            _ptr_array(comp_blocks) if comp_blocks else None, len(comp_blocks), log_n,
                                            log_blowup, op, mc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            mo.ctypes.data_as(C.POINTER(C.c_uint32)), len(mc), ot.ctypes.data,
                                            ct.ctypes.data, oc.ctypes.data, cc.ctypes.data, zp, m0, count, _ptr_of(out)))

    def deep_extend(self, subcoset, log_n, log_blowup, offset, out):
        """ss_deep_extend: n sub-coset values of a polynomial of degree < n (overwritten) -> its N evaluations"""
        _k, op = _felt_ptr(offset)
        check(self.lib.ss_deep_extend(self.handle, _ptr_of(subcoset), log_n, log_blowup, op, _ptr_of(out)))

    # ---- Q1 -------------------------------------------------------------------------------
    def inverse_table(self, log_N, offset, c, out):
        """out[i] = 1 / (offset * w_N^i - c)"""
        _k1, op = _felt_ptr(offset)
        _k2, cp = _felt_ptr(c)
        check(self.lib.ss_inverse_table(self.handle, log_N, op, cp, _ptr_of(out)))

    def eval_quotient(self, program, tables, table_desc, lde_cols, log_n, log_blowup, offset, out):
        """program: air_program.Program; tables: device buffer of concatenated felts (or None);
        table_desc: flat [offset, log_len, ...] list"""
        prog, _keep = _air_program(program, tables, table_desc)
        _k, op = _felt_ptr(offset)
        check(self.lib.ss_eval_quotient(self.handle, C.byref(prog), _ptr_array(lde_cols), len(lde_cols), log_n,
                                        log_blowup, op, _ptr_of(out)))

    def eval_quotient_rows(self, program, tables, table_desc, col_blocks, log_n, log_blowup, offset, row0, nrows, block_rows, out):
        """ss_eval_quotient_rows: points row0 .. row0 + nrows from column blocks of block_rows rows starting at row0"""
        prog, _keep = _air_program(program, tables, table_desc)
        _k, op = _felt_ptr(offset)
        check(self.lib.ss_eval_quotient_rows(self.handle, C.byref(prog), _ptr_array(col_blocks), len(col_blocks), log_n,
                                             log_blowup, op, row0, nrows, block_rows, _ptr_of(out)))

    def zero(self, buf, nbytes=None):
        check(self.lib.ss_dev_zero(self.handle, _ptr_of(buf), buf.nbytes if nbytes is None else nbytes))

    def permutation_product(self, num, den, count, z, alpha, out, out_stride=1, out_offset=0, want_last=True):
        """num / den: (device column, stride, addr_offset, value_offset or -1) = array_chunks::<stride>() of the column;
        writes out[out_offset + i*out_stride] = numerator_acc_i / denominator_acc_i (layouts/src/recursive/trace.rs:
        704-733, 766-769) and returns the last value (Montgomery limbs) when want_last."""
        ops = [_lib.PermOperand(_ptr_of(o[0]), o[1], o[2], o[3]) for o in (num, den)]
        _kz, zp = _felt_ptr(z)
        _ka, ap = _felt_ptr(alpha)
        last = np.zeros(4, dtype=np.uint64)
        check(self.lib.ss_permutation_product(self.handle, C.byref(ops[0]), C.byref(ops[1]), count, zp, ap, _ptr_of(out),
                                              out_stride, out_offset,
                                              last.ctypes.data_as(C.POINTER(C.c_uint64)) if want_last else None))
        return last if want_last else None

    def diluted_aggregate(self, ordered, stride, offset, count, z, alpha, out, out_stride=1, out_offset=0):
        """layouts/src/recursive/trace.rs:787-803"""
        _kz, zp = _felt_ptr(z)
        _ka, ap = _felt_ptr(alpha)
        check(self.lib.ss_diluted_aggregate(self.handle, _ptr_of(ordered), stride, offset, count, zp, ap, _ptr_of(out),
                                            out_stride, out_offset))

    def scale_strided(self, data, stride, offset, count, factor):
        """ss_scale_strided: data[offset + i*stride] *= factor - a block's running product times the blocks before it"""
        _k, fp = _felt_ptr(factor)
        check(self.lib.ss_scale_strided(self.handle, _ptr_of(data), stride, offset, count, fp))

    def diluted_aggregate_block(self, ordered, stride, offset, count, starts_column, z, alpha, maps, want_total=True):
        """ss_diluted_aggregate_block: the block's scanned affine maps (maps: 2*count felts) -> (M, C) of its last item"""
        _kz, zp = _felt_ptr(z)
        _ka, ap = _felt_ptr(alpha)
        total = np.zeros(8, dtype=np.uint64)
        check(self.lib.ss_diluted_aggregate_block(self.handle, _ptr_of(ordered), stride, offset, count, 1 if starts_column else 0, zp, ap,
                                                  _ptr_of(maps), total.ctypes.data_as(C.POINTER(C.c_uint64)) if want_total else None))
        return (total[:4].copy(), total[4:].copy()) if want_total else None

    def affine_apply(self, maps, count, start, out, out_stride=1, out_offset=0):
        """ss_affine_apply: out[out_offset + i*out_stride] = M_i * start + C_i"""
        _k, sp = _felt_ptr(start)
        check(self.lib.ss_affine_apply(self.handle, _ptr_of(maps), count, sp, _ptr_of(out), out_stride, out_offset))

    def dev_copy_2d(self, dst, dst_pitch, src, src_pitch, width, rows, dst_offset=0, src_offset=0):
        """ss_dev_copy_2d: `rows` runs of `width` bytes, row r from src + src_offset + r * src_pitch to dst + dst_offset + r * dst_pitch"""
        check(self.lib.ss_dev_copy_2d(self.handle, _ptr_of(dst) + dst_offset, dst_pitch, _ptr_of(src) + src_offset, src_pitch, width, rows))

    def upload_async(self, dst, host_array):
        """ss_upload_async: the copy leaves on the context's copy stream and the call returns; `host_array` (pinned, contiguous) must stay
        untouched until the copy is done.  -> the ticket for wait_upload"""
        t = C.c_uint64()
        check(self.lib.ss_upload_async(self.handle, _ptr_of(dst), host_array.ctypes.data, host_array.nbytes, C.byref(t)))
        return t.value

    def wait_upload(self, ticket):
        """ss_wait_upload: what is enqueued on the context's stream from now on sees that upload (a stream wait; once per ticket)"""
        check(self.lib.ss_wait_upload(self.handle, int(ticket)))

    def profile(self, on):
        """False / True: HIP events around every profiled launch; 2: also shader-clock stamps around them (profile_read_clock)"""
        check(self.lib.ss_profile_enable(self.handle, 2 if on == 2 else 1 if on else 0))

    def profile_read_clock(self, kind):
        """-> (shader cycles, reference ticks) between the stamps of this kernel family's launches since the last reset (level 2)"""
        cyc, ref = C.c_double(), C.c_double()
        check(self.lib.ss_profile_read_clock(self.handle, kind, C.byref(cyc), C.byref(ref)))
        return cyc.value, ref.value

    def profile_reset(self):
        check(self.lib.ss_profile_reset(self.handle))

    def profile_read(self, kind):
        ms, cnt = C.c_double(), C.c_uint64()
        check(self.lib.ss_profile_read(self.handle, kind, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def mul_bench(self, a, b, n, reps, out):
        check(self.lib.ss_fp252_mul_bench(self.handle, _ptr_of(a), _ptr_of(b), n, reps, _ptr_of(out)))


def _air_program(program, tables, table_desc):
    """-> (ss_air_program, the host arrays it points into).  program: air_program.Program (constants as canonical ints) or
    anything with code / n_slots and `consts_mont` (uint64[n, 4] Montgomery limbs: the C++ host's lowering)."""
    code = np.ascontiguousarray(program.code, dtype=np.uint32)
    if getattr(program, "consts_mont", None) is not None:
        n_consts = len(program.consts_mont)
        consts = np.ascontiguousarray(program.consts_mont, dtype=np.uint64) if n_consts else np.zeros((1, 4), dtype=np.uint64)
    else:
        n_consts = len(program.consts)
        consts = np.zeros((max(1, n_consts), 4), dtype=np.uint64)
        for i, v in enumerate(program.consts):
            consts[i] = felt(v)
    desc = np.ascontiguousarray(table_desc if len(table_desc) else [0, 0], dtype=np.uint32)
    prog = _lib.AirProgram(code.ctypes.data_as(C.POINTER(C.c_uint32)), len(code) // 2,
                           consts.ctypes.data_as(C.POINTER(C.c_uint64)), n_consts,
                           _ptr_of(tables) if tables is not None else None,
                           desc.ctypes.data_as(C.POINTER(C.c_uint32)), len(table_desc) // 2, program.n_slots)
    return prog, (code, consts, desc)


class Matrix:
    """Column-major matrix of felts resident in HBM (ministark::Matrix<Fp>)."""

    def __init__(self, ctx, cols, nrows):
        self.ctx, self.cols, self.nrows = ctx, list(cols), int(nrows)

    @classmethod
    def from_host(cls, ctx, host_cols):
        host_cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in host_cols]
        return cls(ctx, [ctx.column(c) for c in host_cols], host_cols[0].shape[0])

    @classmethod
    def empty(cls, ctx, ncols, nrows):
        return cls(ctx, [ctx.alloc(32 * nrows) for _ in range(ncols)], nrows)

    @property
    def num_cols(self):
        return len(self.cols)

    @property
    def log_rows(self):
        return self.nrows.bit_length() - 1

    def to_host(self):
        return [c.download(np.uint64, (self.nrows, 4)) for c in self.cols]

    def interpolate(self, offset=None, out_order=NATURAL):
        """in place: evaluations over offset*<w> -> coefficients"""
        self.ctx.ntt(self.cols, self.log_rows, INVERSE, offset, NATURAL, out_order)
        return self

    def evaluate(self, offset=None, in_order=NATURAL):
        """in place: coefficients -> evaluations over offset*<w>"""
        self.ctx.ntt(self.cols, self.log_rows, FORWARD, offset, in_order, NATURAL)
        return self

    def lde(self, log_blowup, offset, keep_coeffs=True):
        """-> (evaluations over offset*<w_{n*blowup}>, bit-reversed coefficient matrix or None)"""
        ev = Matrix.empty(self.ctx, self.num_cols, self.nrows << log_blowup)
        co = Matrix.empty(self.ctx, self.num_cols, self.nrows) if keep_coeffs else None
        self.ctx.lde(self.cols, self.log_rows, log_blowup, offset, ev.cols, co.cols if co else None)
        return ev, co

    def hash_rows(self, kind, order=NATURAL):
        out = self.ctx.alloc(32 * self.nrows)
        self.ctx.hash_rows(kind, self.cols, self.nrows, out, order)
        return out


class _MerkleTree:
    tree_kind = None
    row_hash = None
    n_friendly = 0

    def __init__(self, ctx, n, nodes, tags, root, root_tag, leaf_kind):
        self.ctx, self.n, self.nodes, self.tags = ctx, n, nodes, tags
        self._root, self._root_tag, self.leaf_kind = root, root_tag, leaf_kind

    @classmethod
    def from_matrix(cls, matrix, order=NATURAL):
        """MatrixMerkleTree::from_matrix (crypto/src/merkle/mod.rs:110-123, 289-304).  order=BITREV: leaf i is row
        bitrev(i) of the (natural-order) matrix - the commitment order of the reference's proofs."""
        ctx, n = matrix.ctx, matrix.nrows
        nodes = ctx.alloc(64 * n)
        tags = ctx.alloc(2 * n) if cls.tree_kind == TREE_FRIENDLY else None
        if matrix.num_cols == 1:
            leaf_kind, leaves = LEAF_FELT, matrix.cols[0]
        else:
            leaf_kind, leaves = LEAF_DIGEST, matrix.hash_rows(cls.row_hash, order)
        root, tag = ctx.merkle_build(cls.tree_kind, cls.n_friendly, leaf_kind, leaves, n, nodes, tags, order)
        return cls(ctx, n, nodes, tags, root, tag, leaf_kind)

    def root(self):
        return self._root

    def root_tag(self):
        return self._root_tag

    def prove(self, indices):
        """authentication paths (leaf level first) for the given leaf indices"""
        return self.ctx.merkle_open(self.nodes, self.tags, self.n, indices)


class LeafVariantMerkleTree(_MerkleTree):
    """LeafVariantMerkleTree<MaskedKeccak256HashFn<20>> (starknet EthVerifierClaim, src/claims.rs:20-21)"""
    tree_kind, row_hash = TREE_KECCAK_M20, HASH_KECCAK_M20


class LeafVariantMerkleTreeUnmasked(_MerkleTree):
    """LeafVariantMerkleTree<Keccak256HashFn> (recursive EthVerifierClaim, src/claims.rs:29-30)"""
    tree_kind, row_hash = TREE_KECCAK, HASH_KECCAK


class FriendlyMerkleTree(_MerkleTree):
    """FriendlyMerkleTree<22, PedersenHashFn> (CairoVerifierClaim, src/claims.rs:10,22-23,31-32)"""
    tree_kind, row_hash, n_friendly = TREE_FRIENDLY, HASH_BLAKE2S_M20, 22

"""ORACLE - TEST INFRASTRUCTURE ONLY.  The oracle behind the calls sandstorm_amd/goldilocks.py makes on a `backend.Context`, for the
64-bit field: the SAME prover code that sequences the HIP kernels then runs on the CPU (torch CPU tensors for its buffers), so a
whole proof can be compared with a GPU-made one array for array (tests/test_goldilocks_stark.py) - a pin of the pipeline, not of
one kernel.  Every stage is a definition-level restatement: oracle/goldilocks.c (transforms, constraint program, out-of-domain
evaluation, DEEP term by term, FRI fold), hashlib (Blake2s rows and tree), Python integers (running products, proof of work).
Nothing under sandstorm_amd/ imports this module."""
import hashlib

import numpy as np

from . import oracle_py as oracle

GL_P = 2**64 - 2**32 + 1
NATURAL, BITREV = 0, 1
FORWARD, INVERSE = 0, 1


# Fq3 = Fp[X] / (X^3 - 2) on Python integers, from the definition (the oracle's own: nothing of the product's is imported here)
def f3(a):
    return (a % GL_P, 0, 0)


def add3(a, b):
    return tuple((x + y) % GL_P for x, y in zip(a, b))


def sub3(a, b):
    return tuple((x - y) % GL_P for x, y in zip(a, b))


def scale3(a, k):
    return tuple(x * k % GL_P for x in a)


def mul3(a, b):
    c = [0] * 5
    for i in range(3):
        for j in range(3):
            c[i + j] += a[i] * b[j]
    return ((c[0] + 2 * c[3]) % GL_P, (c[1] + 2 * c[4]) % GL_P, c[2] % GL_P)


def inv3(a):
    """the adjugate over the norm: (a0 + a1 X + a2 X^2)(t0 + t1 X + t2 X^2) = a0 t0 + 2 (a1 t2 + a2 t1), the X and X^2 terms cancel"""
    a0, a1, a2 = a
    t = (a0 * a0 - 2 * a1 * a2, 2 * a2 * a2 - a0 * a1, a1 * a1 - a0 * a2)
    return scale3(t, pow((a0 * t[0] + 2 * (a1 * t[2] + a2 * t[1])) % GL_P, -1, GL_P))


def _np(t):
    """a numpy uint64 view of a torch CPU tensor (or array) sharing its memory"""
    a = t.numpy() if hasattr(t, "numpy") else np.asarray(t)
    return a.view(np.uint64) if a.dtype == np.int64 else a


def _bitrev_perm(n):
    bits = n.bit_length() - 1
    return np.array([int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(n)])


class GlCpuContext:
    def __init__(self):
        self.handle = None

    # ---- transforms
    def lde_gl64(self, cols_in, log_n, log_blowup, offset, evals_out, coeffs_out=None):
        rev = _bitrev_perm(1 << log_n)
        for k, c in enumerate(cols_in):
            ev, co = oracle.gl_lde(_np(c), log_blowup, offset)
            _np(evals_out[k])[:] = ev
            if coeffs_out is not None:
                _np(coeffs_out[k])[:] = co[rev]                      # bit-reversed, as ss_lde_gl64 leaves them

    def ntt_gl64(self, cols, log_n, direction=FORWARD, offset=1, in_order=NATURAL, out_order=NATURAL):
        rev = _bitrev_perm(1 << log_n)
        for c in cols:
            a = _np(c)
            v = a[rev].copy() if in_order == BITREV else a.copy()
            v = oracle.gl_ntt(v, inverse=direction == INVERSE, offset=offset)
            a[:] = v[rev] if out_order == BITREV else v

    # ---- commitments
    @staticmethod
    def _segments(segments, seg_len, nrows):
        out = []
        for s in segments:
            if hasattr(s, "parent"):                                 # backend.DeviceView over goldilocks._Raw(tensor)
                base = _np(s.parent.t).reshape(-1)
                off = (s.ptr - s.parent.ptr) // 8
                out.append(base[off:off + nrows * seg_len])
            else:
                out.append(_np(s).reshape(-1)[:nrows * seg_len])
        return out

    def hash_rows_gl64(self, segments, seg_len, nrows, out, hash_kind=None):
        segs = self._segments(segments, seg_len, nrows)
        o = out.numpy()
        for i in range(nrows):
            row = b"".join(int(s[i * seg_len + e]).to_bytes(8, "little") for s in segs for e in range(seg_len))
            o[i] = np.frombuffer(hashlib.blake2s(row).digest(), dtype=np.uint8)

    def merkle_build(self, tree, n_friendly, leaf_kind, leaves, n, nodes, tags=None, leaf_order=NATURAL):
        nd = nodes.numpy()
        nd[n:2 * n] = leaves.numpy()
        for k in range(n - 1, 0, -1):
            nd[k] = np.frombuffer(hashlib.blake2s(bytes(nd[2 * k]) + bytes(nd[2 * k + 1])).digest(), dtype=np.uint8)
        nd[0] = 0
        return bytes(nd[1]), 0

    def merkle_open(self, nodes, tags, n, indices):
        nd, log_n = nodes.numpy(), int(n).bit_length() - 1
        out = np.zeros((len(indices), log_n, 32), dtype=np.uint8)
        for q, i in enumerate(indices):
            k = n + int(i)
            for l in range(log_n):
                out[q, l] = nd[k ^ 1]
                k >>= 1
        return out, np.zeros((len(indices), log_n), dtype=np.uint8)

    def gather_rows_gl64(self, segments, seg_len, nrows, idx):
        segs = self._segments(segments, seg_len, nrows)
        out = np.zeros((len(idx), len(segs), seg_len), dtype=np.uint64)
        for q, i in enumerate(idx):
            for s, seg in enumerate(segs):
                out[q, s] = seg[int(i) * seg_len:(int(i) + 1) * seg_len]
        return out

    # ---- the constraint program, out-of-domain evaluation, DEEP, FRI
    def eval_quotient_gl64x3(self, code, consts3, n_slots, tables, table_desc, lde_cols, log_n, log_blowup, offset, out):
        _np(out)[:] = oracle.gl3_eval_program(code, consts3, n_slots, _np(tables), table_desc, [_np(c) for c in lde_cols], log_n, log_blowup, offset)

    def ood_eval_gl64x3(self, coeff_cols, log_n, cell_col, cell_off, z):
        rev = _bitrev_perm(1 << log_n)
        return oracle.gl3_ood_eval([_np(c)[rev] for c in coeff_cols], cell_col, cell_off, z)

    def deep_compose_gl64x3(self, trace_cols, comp_cols, log_n, log_blowup, offset, mask_col, mask_off, ood_trace, coeff_trace, ood_comp,
                            coeff_comp, z, z_comp, out):
        _np(out)[:] = oracle.gl3_deep_compose([_np(c) for c in trace_cols], [_np(c) for c in comp_cols], log_n, log_blowup, offset, mask_col, mask_off,
                                              ood_trace, coeff_trace, ood_comp, coeff_comp, z, z_comp)

    def fri_fold_gl64x3(self, evals, log_len, fold, alpha, offset, out, flags=0):
        _np(out)[:] = oracle.gl3_fri_fold(_np(evals), fold, np.array(alpha, dtype=np.uint64), offset, bool(flags & 1))

    # ---- the extension column's running quotients, proof of work
    def running_product_gl64x3(self, num_addr, num_val, den_addr, den_val, stride, count, z, alpha, out_cols, out_stride, out_offset, want_last=True):
        na, da = _np(num_addr), _np(den_addr)
        nv, dv = (_np(num_val), _np(den_val)) if num_val is not None else (None, None)
        z, al = tuple(int(v) for v in z), (tuple(int(v) for v in alpha) if alpha is not None else (0, 0, 0))
        outs = [_np(c) for c in out_cols]
        num, den, last = (1, 0, 0), (1, 0, 0), (1, 0, 0)
        for i in range(count):
            tn = add3(scale3(al, int(nv[i * stride])) if nv is not None else (0, 0, 0), f3(int(na[i * stride])))
            td = add3(scale3(al, int(dv[i * stride])) if dv is not None else (0, 0, 0), f3(int(da[i * stride])))
            num, den = mul3(num, sub3(z, tn)), mul3(den, sub3(z, td))
            last = mul3(num, inv3(den))
            for t in range(3):
                outs[t][out_offset + i * out_stride] = last[t]
        return last if want_last else None

    def pow_grind(self, coin_kind, digest, bits):
        keccak256 = oracle.keccak256                                                    # oracle/keccak.c
        prefix = keccak256((0x0123456789ABCDED).to_bytes(8, "big") + bytes(digest) + bytes([bits]))
        nonce = 0
        while int.from_bytes(keccak256(prefix + nonce.to_bytes(8, "big"))[:8], "big") >> (64 - bits):
            nonce += 1
        return nonce

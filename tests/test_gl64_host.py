"""csrc/gl64.h - the arithmetic every kernel of the 64-bit field runs (products in 32-bit halves, lazy sums, wide accumulators,
Fq3) - is host/device code: compiled here with g++ and held to 128-bit integer arithmetic on edge words (0, p - 1, p, 2^64 - 1,
2^32 +- 1 ...) and random ones, long wide sums of extreme products, and a butterfly network run lazily against the same
network on canonical values (tests/cpp/gl64_host_test.cpp).  The device compiles the same source."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gl64_header_on_the_host(tmp_path):
    exe = str(tmp_path / "gl64_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "sandstorm_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "gl64_host_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "0", "mismatches: %s" % out.stdout


def test_generated_composition_kernel_on_the_host(tmp_path, oracle):
    """csrc/quotient_gen_plain_gl.inc - the straight-line kernel tools/gen_quotient_gl.py writes for the plain layout's composition -
    compiled for the host over the same gl64.h and run over whole evaluation domains (a "grid" that does not divide them), for two
    statements / sizes / transcripts: its values are the oracle's constraint VM's on the program the layout lowers to"""
    import copy
    import struct

    import numpy as np

    from sandstorm_amd import air_program as ap
    from sandstorm_amd.layouts import plain as pl
    exe = str(tmp_path / "gl3_plain_host_test")
    inc = os.path.join(ROOT, "sandstorm_amd", "csrc", "quotient_gen_plain_gl.inc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wno-attributes", "-I", os.path.join(ROOT, "sandstorm_amd", "csrc"), "-DQG_INC=\"%s\"" % inc,
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "gl3_plain_host_test.cpp")])
    prog = pl.example_program(10)
    states, memory = pl.run(prog, 64)
    pi = pl.public_input_of(prog, states, memory)
    rng = np.random.default_rng(8)
    for log_n, ch, alpha, grid in ((10, [(11, 22, 33), (5, 6, 7), (9, 8, 7)], (123456789, 987654321, 55555), (3, 5)),
                                   (12, [(1, 2, 3), (4, 5, 6), (pl.P - 1, pl.P - 2, 9)], (5, 0, pl.P - 1), (7, 64))):
        n, lb = 1 << log_n, 1
        N = n << lb
        tables = pl.Tables(n, lb)
        stmt = copy.deepcopy(pi)
        stmt.n_steps = n // 16
        root = pl.composition(n, pl.Hints.from_public_input(stmt, ch, n), ch, alpha, tables)
        program = ap.lower(root, pl.P, ext=True, symbols=tables.symbols)
        code, consts = np.array(program.code, dtype=np.uint32), np.array(program.consts, dtype=np.uint64)
        tvals, tdesc = tables.device_tables()
        lde = [rng.integers(0, 1 << 62, size=N, dtype=np.uint64) for _ in range(8)]
        lde[0][:4] = pl.P - 1
        want = oracle.gl3_eval_program(code, consts, program.n_slots, tvals, tdesc, lde, log_n, lb, pl.GENERATOR)
        w = pow(7, (pl.P - 1) >> (log_n + lb), pl.P)
        td = []
        for k in range(0, len(tdesc), 2):
            td += [tdesc[k], (1 << tdesc[k + 1]) - 1]
        fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<10Q", len(lde), N, len(tvals), len(td) // 2, len(consts), lb, pl.GENERATOR, w, grid[0], grid[1]))
            for c in lde:
                f.write(np.ascontiguousarray(c).tobytes())
            f.write(np.ascontiguousarray(tvals, dtype=np.uint64).tobytes())
            f.write(np.asarray(td, dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(consts.reshape(-1, 3)).tobytes())
        subprocess.check_call([exe, fin, fout], timeout=600)
        got = np.fromfile(fout, dtype=np.uint64).reshape(N, 3)
        assert np.array_equal(got, np.asarray(want).reshape(N, 3))


def test_generated_kernel_keeps_the_lazy_word_discipline():
    """The generated kernel leaves products, sums and differences as lazy words (any 64-bit word congruent to the value) and the
    generator tracks which coordinates are; a wrong track would only show on a value in [p, 2^64) - one random word in 2^32 - so
    the discipline is re-derived here from the EMITTED text alone, independently of the generator's bookkeeping: the second
    operand of every gl_add_lazy / gl_sub_lazy, the upper coordinates of gl3_mul's right factor, and everything written out must
    be canonical at that statement."""
    import re
    inc = open(os.path.join(ROOT, "sandstorm_amd", "csrc", "quotient_gen_plain_gl.inc")).read()
    body = inc[inc.index("Gl3 a0, a1, a2, a3;"):inc.index("#undef QG_K")]
    lazy = {}                                                   # "a1.c[0]" -> bool; anything absent (slots, constants, cells, x, 0) is canonical
    is_lazy = lambda atom: lazy.get(atom.strip(), False)
    var = r"[as]\d+\.c\[\d\]"                                  # a coordinate of an accumulator or of a scratch value
    n_checked = 0

    def split2(args):                                           # the two top-level arguments of a call
        depth = 0
        for i, ch in enumerate(args):
            depth += ch == "(" or ch == "{"
            depth -= ch == ")" or ch == "}"
            if ch == "," and depth == 0:
                return args[:i].strip(), args[i + 1:].strip()
        raise AssertionError(args)

    def assign(dst, expr):
        nonlocal n_checked
        m = re.fullmatch(r"(gl_\w+)\((.*)\)", expr)
        if not m:
            lazy[dst] = is_lazy(expr)                           # a copy
            return
        fn, args = m.group(1), m.group(2)
        if fn == "gl_canon":
            lazy[dst] = False
        elif fn in ("gl_add_lazy", "gl_sub_lazy"):
            _, b = split2(args)
            assert not is_lazy(b), "%s = %s: the second operand is a lazy word" % (dst, expr)
            n_checked += 1
            lazy[dst] = True
        elif fn == "gl_mul_lazy":
            lazy[dst] = True
        elif fn == "gl_pow":
            lazy[dst] = False
        else:
            raise AssertionError("unknown form: " + expr)

    for line in body.splitlines()[2:]:
        line = line.strip()
        if not line or line == "}":
            continue
        m = re.fullmatch(r"\{ const uint64_t f = (.+?); (.*) \}", line)
        if m:
            lazy["f"] = is_lazy(m.group(1))
            for part in m.group(2).split(";"):
                if part.strip():
                    dst, expr = part.strip().split(" = ", 1)
                    assign(dst, expr)
            continue
        m = re.fullmatch(r"(a\d) = gl3_mul\(a\d, Gl3\{\{(.+)\}\}\);", line)
        if m:
            comps = [c.strip() for c in re.split(r",\s*(?![^()]*\))", m.group(2))]
            assert len(comps) == 3 and not is_lazy(comps[1]) and not is_lazy(comps[2]), line
            n_checked += 1
            for t in range(3):
                lazy["%s.c[%d]" % (m.group(1), t)] = False
            continue
        m = re.fullmatch(r"(a\d) = gl3_inv\(a\d\);", line)
        if m:
            for t in range(3):
                lazy["%s.c[%d]" % (m.group(1), t)] = False
            continue
        m = re.fullmatch(r"QG_OUT\(\d, (.+)\);", line)
        if m:
            assert not is_lazy(m.group(1)), line
            n_checked += 1
            continue
        m = re.fullmatch(r"(%s) = (.+);" % var, line)
        assert m, "unrecognised statement: " + line
        assign(m.group(1), m.group(2))
    assert n_checked > 300


def _gl_twiddles(log_n, inverse, offset):
    """the plan csrc/capi.hip gl_get_plan uploads, from its definition: T_s[k] = h^(n / 2^(s+1)) r^(k n / 2^(s+1)) at (2^s - 1) + k"""
    import numpy as np
    P = 2**64 - 2**32 + 1
    n = 1 << log_n
    r, h = pow(7, (P - 1) >> log_n, P), offset
    if inverse:
        r, h = pow(r, P - 2, P), pow(h, P - 2, P)
    tw = np.zeros(max(1, n - 1), dtype=np.uint64)
    for s in range(log_n):
        hs, step, v = pow(h, n >> (s + 1), P), pow(r, n >> (s + 1), P), 1
        for k in range(1 << s):
            tw[(1 << s) - 1 + k] = hs * v % P
            v = v * step % P
    return tw[:n - 1]


def test_transform_passes_on_the_host(tmp_path, oracle):
    """csrc/gl_ntt.h - the pass code the device compiles (tile index arithmetic, register groups of 1-4 stages reading / writing
    HBM or the tile, twiddle indexing, lazy butterflies) and the plan that splits a transform into passes - compiled for the host:
    forward, inverse and extension (the zero-padded half never materialised) against the oracle for EVERY size up to 2^17 with the
    device's 2^13 tiles, and with 2^8 tiles so that small sizes run the multi-pass strided paths too; words at both ends of the
    field among the inputs."""
    import struct

    import numpy as np
    P = 2**64 - 2**32 + 1
    exe = str(tmp_path / "gl_ntt_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "sandstorm_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "gl_ntt_host_test.cpp")])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")

    def run(log_n, inverse, log_expand, scale, lt, src, tw):
        with open(fin, "wb") as f:
            f.write(struct.pack("<6Q", log_n, int(inverse), log_expand, scale, lt, len(src)))
            f.write(np.ascontiguousarray(src, dtype=np.uint64).tobytes())
            f.write(np.ascontiguousarray(tw, dtype=np.uint64).tobytes())
        subprocess.check_call([exe, fin, fout], timeout=600)
        return np.fromfile(fout, dtype=np.uint64)

    rng = np.random.default_rng(64)
    for lt, sizes in ((13, list(range(1, 18))), (8, list(range(1, 17)))):
        for log_n in sizes:
            n = 1 << log_n
            rev = np.array([int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)])
            x = rng.integers(0, P, size=n, dtype=np.uint64)
            edge = rng.integers(0, 4, size=n)
            x = np.where(edge == 0, rng.integers(0, 3, size=n, dtype=np.uint64), np.where(edge == 1, np.uint64(P - 1) - rng.integers(0, 3, size=n, dtype=np.uint64), x))
            off = 7 if log_n % 2 else 1
            fwd = run(log_n, False, 0, 1, lt, x[rev], _gl_twiddles(log_n, False, off))          # bit-reversed in, natural out
            assert np.array_equal(fwd, oracle.gl_ntt(x, offset=off)), (lt, log_n, "forward")
            inv = run(log_n, True, 0, pow(n, P - 2, P), lt, x, _gl_twiddles(log_n, True, off))   # natural in, bit-reversed out, 1/n on the last pass
            assert np.array_equal(inv[rev], oracle.gl_ntt(x, inverse=True, offset=off)), (lt, log_n, "inverse")
            for lb in (1, 2):                                                                    # capi.hip ss_lde_gl64: coefficients, then the expanding forward pass
                if log_n + lb > 17:
                    continue
                co = run(log_n, True, 0, pow(n, P - 2, P), lt, x, _gl_twiddles(log_n, True, 1))
                ev = run(log_n + lb, False, lb, 1, lt, co, _gl_twiddles(log_n + lb, False, 7))
                want_ev, want_co = oracle.gl_lde(x, lb, 7)
                assert np.array_equal(co[rev], want_co) and np.array_equal(ev, want_ev), (lt, log_n, lb, "extension")

"""BASELINE.json configs[4]: the `plain` layout over p = 2^64 - 2^32 + 1 with Fq3 challenges (layouts/plain.py) and the STARK
around it (sandstorm_amd/goldilocks.py).  PARITY UNPINNED (the reference's parts for this claim are un-vendored and it ships
no program, trace or proof for the field): what is held here is
  * the restated AIR against a run of a minimal Cairo machine over this field - every constraint vanishes on its domain, the
    permutation products close, and a corrupted cell is caught by the constraint that owns it;
  * the lowered composition program (oracle VM) against the expression DAG evaluated in Python integers;
  * on the device: a proof of the true statement verifies, and no tampered proof or statement does."""
import copy

import numpy as np
import pytest

from sandstorm_amd import air_program as ap
from sandstorm_amd.layouts import plain as pl

CH = [(11, 22, 33), (5, 6, 7), (9, 8, 7)]


@pytest.fixture(scope="module")
def run():
    prog = pl.example_program(10)
    states, memory = pl.run(prog, 64)
    pi = pl.public_input_of(prog, states, memory)
    return prog, states, memory, pi, pl.base_trace(states, memory, pi)


def test_the_machine_runs_the_example(run):
    prog, states, memory, pi, cols = run
    x = 3
    for _ in range(10):
        x = (x * x + 7) % pl.P
    assert x in memory and states[-1].pc == states[-2].pc               # the loop's result is in memory; the run ends in `jmp rel 0`
    assert states[-1].fp == states[0].fp and len(cols) == 5 and len(cols[0]) == 16 * 64


def test_plain_layout_trace_satisfies_the_constraints(run):
    prog, states, memory, pi, cols = run
    n = len(cols[0])
    ext, last = pl.extension_columns(cols, CH)
    hints = pl.Hints.from_public_input(pi, CH, n)
    assert last == hints.memory_quotient                                # the memory product ends at the public-memory quotient
    cs = pl.constraints(hints, CH)
    assert len(cs) == 47 and len(pl.mask(cs)) == 56
    for c in cs:
        assert not pl.failing_rows(c, cols + ext, n), c.name
    # a corrupted cell is caught by the constraints that read it
    bad = [list(c) for c in cols]
    bad[pl.COL_RANGE_CHECK][16 * 2 + pl.Auxiliary.RES[1]] += 1        # `res` of cycle 2 (the loop's multiplication)
    names = [c.name for c in cs if pl.failing_rows(c, bad + ext, n)]
    assert "cpu/operands/res" in names
    # a value outside the program's offsets breaks the range-check permutation
    bad = [list(c) for c in cols]
    bad[pl.COL_RANGE_CHECK][16 * 7 + pl.RangeCheck.OFF_DST] = 5
    with pytest.raises(ValueError):
        pl.extension_columns(bad, CH)


@pytest.mark.parametrize("run_steps,total", [(64, 64), (64, 256), (128, 1024)])
def test_vectorised_trace_generator_is_the_reference_one(run_steps, total):
    """base_trace_np (numpy columns, one decode per distinct state: what the 2^20-step GPU test below generates its trace with)
    writes base_trace's cells - a whole run, and runs padded with their final state (`jmp rel 0`)"""
    prog = pl.example_program(10)
    states, memory = pl.run(prog, run_steps)
    states = list(states) + [states[-1]] * (total - len(states))
    pi = pl.public_input_of(prog, states, memory)
    for a, b in zip(pl.base_trace(states, memory, pi), pl.base_trace_np(states, memory, pi)):
        assert np.array_equal(np.array(a, dtype=np.uint64), b)


def test_lowered_composition_is_the_dag(run, oracle):
    prog, states, memory, pi, cols = run
    n, lb = len(cols[0]), 1
    N, log_n = n << lb, n.bit_length() - 1
    ext, _ = pl.extension_columns(cols, CH)
    lde = [oracle.gl_lde(np.array(c, dtype=np.uint64), lb, pl.GENERATOR)[0] for c in cols + ext]
    hints = pl.Hints.from_public_input(pi, CH, n)
    tables = pl.Tables(n, lb)
    alpha = (123456789, 987654321, 55555)
    root = pl.composition(n, hints, CH, alpha, tables)
    program = ap.lower(root, pl.P, ext=True, symbols=tables.symbols)
    tvals, tdesc = tables.device_tables()
    got = oracle.gl3_eval_program(np.array(program.code, dtype=np.uint32), np.array(program.consts, dtype=np.uint64), program.n_slots, tvals, tdesc,
                                  lde, log_n, lb, pl.GENERATOR)
    wN = pl.root_of_unity(log_n + lb)
    for i in (0, 1, 2, 17, 1000, N - 1):
        x = pl.GENERATOR * pow(wN, i, pl.P) % pl.P
        want = ap.evaluate_ext(root, pl.P, (x, 0, 0), lambda c, o: (int(lde[c][(i + (o << lb)) % N]), 0, 0),
                               lambda t: (tables.host_values(tables.specs[t])[i % tables.length(tables.specs[t])], 0, 0), symbols=tables.symbols)
        assert tuple(int(v) for v in got[i]) == tuple(want), i
    # the trace satisfies the AIR: every quotient is a polynomial, the largest (a quadratic numerator over X - 1) of degree 2 n - 3
    for t in range(3):
        co = oracle.gl_ntt(got[:, t].copy(), inverse=True, offset=pl.GENERATOR)
        assert co.any() and not co[N - 2:].any()


@pytest.fixture(scope="module")
def proved(run):
    import torch
    from sandstorm_amd import backend as be, goldilocks as gs
    prog, states, memory, pi, cols = run
    dev = torch.device("cuda", 0)
    # ONE stream for torch's tensor ops and the C ABI's kernels (torch's default stream has handle 0, which ss_ctx_set_stream reads
    # as "the context's own stream": the two would not be ordered)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = be.Context(0, stream=stream.cuda_stream)
    tensor = lambda c: torch.from_numpy(np.array(c, dtype=np.uint64).view(np.int64)).to(dev)
    air = gs.plain_air()
    opt = gs.Options(num_queries=20, grinding=8)

    def prove(columns, options=opt):
        base = [tensor(c) for c in columns]
        return gs.Prover(ctx, air, options).prove(bytes(range(32)), base, lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0], statement=pi)
    prove.ctx = ctx
    yield gs, air, pi, cols, prove, opt
    ctx.close()


@pytest.mark.gpu
def test_extension_column_on_the_device(proved, run):
    import torch
    gs, air, pi, cols, prove, opt = proved
    ctx = prove.ctx
    tensor = lambda c: torch.from_numpy(np.array(c, dtype=np.uint64).view(np.int64)).cuda()
    dcols = [tensor(c) for c in cols]
    want, want_last = pl.extension_columns(cols, CH)
    got, last = gs.plain_extension_on_device(ctx, dcols, CH)
    assert last == want_last == pl.Hints.from_public_input(pi, CH, len(cols[0])).memory_quotient
    for g, w in zip(got, want):
        assert [int(v) for v in g.cpu().numpy().view(np.uint64)] == w
    # at scale: the quotient of two orderings of the same 2^22 (address, value) pairs closes to one; a changed value does not
    rng = np.random.default_rng(4)
    m = 1 << 22
    pairs = rng.integers(0, 2**62, size=(m, 2), dtype=np.uint64)
    a = torch.from_numpy(pairs.reshape(-1).view(np.int64).copy()).cuda()
    b = torch.from_numpy(pairs[rng.permutation(m)].reshape(-1).view(np.int64).copy()).cuda()
    out = [torch.zeros(2 * m, dtype=torch.int64, device="cuda") for _ in range(3)]
    assert ctx.running_product_gl64x3(a, a[1:], b, b[1:], 2, m, CH[0], CH[1], out, 2, 0) == (1, 0, 0)
    b[5] += 1
    assert ctx.running_product_gl64x3(a, a[1:], b, b[1:], 2, m, CH[0], CH[1], out, 2, 0) != (1, 0, 0)
    with pytest.raises(ValueError):
        bad = [list(c) for c in cols]
        bad[pl.COL_RANGE_CHECK][16 * 7 + pl.RangeCheck.OFF_DST] = 5
        gs.plain_extension_on_device(ctx, [tensor(c) for c in bad], CH)


@pytest.mark.gpu
def test_proof_of_the_example_verifies(proved):
    gs, air, pi, cols, prove, opt = proved
    proof = prove(cols)
    positions = gs.verify(proof, air, bytes(range(32)), statement=pi, expected_options=opt, required_security_bits=28)
    assert len(positions) >= 15 and len(proof.fri_layers) == 2 and proof.remainder.shape == (32, 3)
    assert gs.verify(prove(cols), air, bytes(range(32)), statement=pi, required_security_bits=28) == positions      # deterministic


@pytest.mark.gpu
def test_the_claim_with_the_parts_the_reference_names_for_it(proved):
    """Options(hash="sha256"): MatrixMerkleTreeImpl<Sha256HashFn> trees and a coin with SHA-256 inside (cli/src/main.rs:119-120) - the
    proof verifies, survives its array form, is not the Blake2s claim's proof, and neither verifies as the other"""
    import dataclasses
    gs, air, pi, cols, prove, opt = proved
    sha = dataclasses.replace(opt, hash="sha256")
    seed = bytes(range(32))
    proof = prove(cols, sha)
    positions = gs.verify(proof, air, seed, statement=pi, expected_options=sha, required_security_bits=28)
    assert len(positions) >= 15
    again = gs.proof_from_arrays(gs.proof_to_arrays(proof))
    assert again.options.hash == "sha256" and gs.verify(again, air, seed, statement=pi, required_security_bits=28) == positions
    other = prove(cols)
    assert other.base_root != proof.base_root and other.options.hash == "blake2s"
    for p, h in ((proof, "blake2s"), (other, "sha256")):
        forged = dataclasses.replace(p, options=dataclasses.replace(p.options, hash=h))
        with pytest.raises(gs.VerificationError):
            gs.verify(forged, air, seed, statement=pi, required_security_bits=28)
    bad = dataclasses.replace(proof, base=gs.Opening(proof.base.rows.copy(), proof.base.paths.copy()))
    bad.base.paths[0, 0, 0] ^= 1
    with pytest.raises(gs.VerificationError, match="authentication path"):
        gs.verify(bad, air, seed, statement=pi, required_security_bits=28)


@pytest.mark.gpu
@pytest.mark.parametrize("hash_name", ["blake2s", "sha256"])
def test_cpp_host_writes_the_python_hosts_proof(proved, hash_name):
    """hostlib.gl_prove (host/goldilocks_prover.cpp: the claim in the C++ host) against goldilocks.Prover on the MI355X: the same proof,
    array for array"""
    import dataclasses
    import torch
    from sandstorm_amd import hostlib
    gs, air, pi, cols, prove, opt = proved
    o = dataclasses.replace(opt, hash=hash_name)
    want = gs.proof_to_arrays(prove(cols, o))
    ctx = prove.ctx
    dev = torch.device("cuda", 0)
    base = [torch.from_numpy(np.array(c, dtype=np.uint64).view(np.int64)).to(dev) for c in cols]
    proof = hostlib.gl_prove(ctx, air, o, bytes(range(32)), base, lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0], statement=pi)
    got = gs.proof_to_arrays(proof)
    assert set(got) == set(want)
    for k in sorted(want):
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k
    gs.verify(proof, air, bytes(range(32)), statement=pi, expected_options=o, required_security_bits=28)


@pytest.mark.gpu
def test_tampered_proofs_and_statements_are_rejected(proved):
    gs, air, pi, cols, prove, opt = proved
    proof = prove(cols)
    seed = bytes(range(32))

    def rejected(mutate, **kw):
        p = copy.deepcopy(proof)
        mutate(p)
        with pytest.raises(gs.VerificationError):
            gs.verify(p, air, kw.get("seed", seed), statement=kw.get("statement", pi), expected_options=kw.get("expected", None), required_security_bits=28)
    rejected(lambda p: None, seed=bytes(32))                                               # another transcript
    other = copy.deepcopy(pi)
    other.public_memory[3] = (other.public_memory[3][0], other.public_memory[3][1] ^ 1)
    rejected(lambda p: None, statement=other)                                              # another program
    other = copy.deepcopy(pi)
    other.rc_max += 1
    rejected(lambda p: None, statement=other)
    rejected(lambda p: p.ood_trace.__setitem__((5, 0), (int(p.ood_trace[5, 0]) + 1) % pl.P))
    rejected(lambda p: p.ood_comp.__setitem__((2, 1), (int(p.ood_comp[2, 1]) + 1) % pl.P))
    rejected(lambda p: p.base.rows.__setitem__((0, 1), int(p.base.rows[0, 1]) ^ 1))
    rejected(lambda p: p.comp.paths.__setitem__((1, 2, 0), int(p.comp.paths[1, 2, 0]) ^ 1))
    rejected(lambda p: p.fri_layers[1].opening.rows.__setitem__((0, 4), int(p.fri_layers[1].opening.rows[0, 4]) ^ 1))
    rejected(lambda p: p.remainder.__setitem__((3, 0), (int(p.remainder[3, 0]) + 1) % pl.P))
    rejected(lambda p: setattr(p, "pow_nonce", p.pow_nonce + 1))
    rejected(lambda p: setattr(p, "comp_root", bytes(32)))
    rejected(lambda p: setattr(p.options, "num_queries", 19), expected=opt)
    rejected(lambda p: setattr(p.options, "grinding", 0))                                  # the transcript's nonce no longer counts
    # a trace that breaks a constraint: the quotient is then not the polynomial the out-of-domain identity asks for (on a domain
    # of 2 n points every function interpolates, so the prover cannot tell; the verifier's recomputation at z does)
    bad = [list(c) for c in cols]
    bad[pl.COL_AUXILIARY][16 * 9 + pl.Auxiliary.TMP0[1]] += 1
    with pytest.raises(gs.VerificationError, match="do not satisfy the AIR"):
        gs.verify(prove(bad), air, seed, statement=pi, required_security_bits=28)


@pytest.mark.gpu
def test_compiled_composition_kernel_is_the_interpreter(proved, run, oracle):
    """the plain layout's composition runs as generated straight-line code (tools/gen_quotient_gl.py); for other statements,
    sizes and transcripts it is still the program the library recognises, and its values are the interpreter's and the oracle's"""
    import os
    import torch
    gs, air, pi, cols, prove, opt = proved
    ctx = prove.ctx
    rng = np.random.default_rng(6)
    for log_n, ch, alpha in ((10, CH, (123456789, 987654321, 55555)), (13, [(1, 2, 3), (4, 5, 6), (7, 8, 9)], (5, 0, 1))):
        n, lb = 1 << log_n, 1
        N = n << lb
        tables = pl.Tables(n, lb)
        stmt = copy.deepcopy(pi)
        stmt.n_steps = n // 16
        root = pl.composition(n, pl.Hints.from_public_input(stmt, ch, n), ch, alpha, tables)
        program = ap.lower(root, pl.P, ext=True, symbols=tables.symbols)
        code, consts = np.array(program.code, dtype=np.uint32), np.array(program.consts, dtype=np.uint64)
        tvals, tdesc = tables.device_tables()
        lde = [(rng.integers(0, 1 << 62, size=N, dtype=np.uint64)) for _ in range(8)]
        d_lde, d_tab = [torch.from_numpy(c.view(np.int64)).cuda() for c in lde], torch.from_numpy(tvals.view(np.int64)).cuda()
        outs = []
        for interpret in (False, True):
            if interpret:
                os.environ["SS_QUOTIENT_INTERPRET"] = "1"
            try:
                out = torch.zeros((N, 3), dtype=torch.int64, device="cuda")
                ctx.eval_quotient_gl64x3(code, consts, program.n_slots, d_tab, tdesc, d_lde, log_n, lb, pl.GENERATOR, out)
                outs.append(out.cpu().numpy().view(np.uint64))
            finally:
                os.environ.pop("SS_QUOTIENT_INTERPRET", None)
        assert np.array_equal(outs[0], outs[1])
        assert np.array_equal(outs[0], oracle.gl3_eval_program(code, consts, program.n_slots, tvals, tdesc, lde, log_n, lb, pl.GENERATOR))
    # the library must in fact have taken the compiled kernel for this program: the generated hash is this program's
    import re
    inc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sandstorm_amd", "csrc", "quotient_gen_plain_gl.inc")).read()
    want = int(re.search(r"GL3_PLAIN_CODE_HASH = (0x[0-9a-f]+)ull", inc).group(1), 16)
    h = 0xcbf29ce484222325
    for w in program.code:
        for k in range(4):
            h = ((h ^ ((int(w) >> (8 * k)) & 0xff)) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    assert h == want, "regenerate csrc/quotient_gen_plain_gl.inc (python tools/gen_quotient_gl.py)"


def test_generated_kernel_is_current():
    """CPU: the committed generated kernel is the one the generator writes for today's layout and lowering"""
    import os
    import re
    import sys
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root_dir, "tools"))
    import gen_quotient_gl as g
    code, consts, n_slots, n_tables = g.template_program()
    inc = open(os.path.join(root_dir, "sandstorm_amd", "csrc", "quotient_gen_plain_gl.inc")).read()
    assert int(re.search(r"GL3_PLAIN_CODE_HASH = (0x[0-9a-f]+)ull", inc).group(1), 16) == g.code_hash(code)
    assert "GL3_PLAIN_N_INSTR = %du" % (len(code) // 2) in inc


def test_gpu_made_proof_verifies_on_the_cpu(run):
    """tests/golden/goldilocks_plain_proof.npz was written on the MI355X (tests/golden/make_goldilocks_proof.py); the verifier is host
    code: the statement is rebuilt here from the example run, the AIR identity is recomputed from the layout's DAG"""
    import os
    from sandstorm_amd import goldilocks as gs
    prog, states, memory, pi, cols = run
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goldilocks_plain_proof.npz")) as f:
        arrays = {k: f[k] for k in f.files}
    proof = gs.proof_from_arrays(arrays)
    air, opt = gs.plain_air(), gs.Options(num_queries=20, grinding=8)
    positions = gs.verify(proof, air, bytes(range(32)), statement=pi, expected_options=opt, required_security_bits=28)
    assert len(positions) >= 15 and proof.trace_len == len(cols[0])
    # the round trip through arrays is lossless
    again = gs.proof_to_arrays(proof)
    assert set(again) == set(arrays) and all(np.array_equal(again[k], arrays[k]) for k in arrays)
    # and the fixture does not verify as anything else
    for mutate in (lambda p: p.ood_trace.__setitem__((0, 0), (int(p.ood_trace[0, 0]) + 1) % pl.P),
                   lambda p: p.base.rows.__setitem__((3, 2), int(p.base.rows[3, 2]) ^ 1),
                   lambda p: p.fri_layers[0].opening.paths.__setitem__((0, 0, 0), int(p.fri_layers[0].opening.paths[0, 0, 0]) ^ 1),
                   lambda p: setattr(p, "pow_nonce", p.pow_nonce + 1)):
        p = gs.proof_from_arrays(arrays)
        mutate(p)
        with pytest.raises(gs.VerificationError):
            gs.verify(p, air, bytes(range(32)), statement=pi, required_security_bits=28)
    other = copy.deepcopy(pi)
    other.memory_segments["execution"] = (other.memory_segments["execution"][0], other.memory_segments["execution"][1] + 1)
    with pytest.raises(gs.VerificationError):
        gs.verify(gs.proof_from_arrays(arrays), air, bytes(range(32)), statement=other, required_security_bits=28)


def test_whole_pipeline_on_the_oracle_reproduces_the_gpu_made_proof(run):
    """the prover of sandstorm_amd/goldilocks.py with the oracle standing in for every kernel (oracle/gl_cpu_context.py) writes the
    proof the MI355X wrote (tests/golden/goldilocks_plain_proof.npz), array for array: transforms, row hashes and trees, the lowered
    constraint program, out-of-domain values, DEEP, every FRI layer, the remainder, the smallest proof-of-work nonce, the openings"""
    import os
    import torch
    from oracle.gl_cpu_context import GlCpuContext
    from sandstorm_amd import goldilocks as gs
    prog, states, memory, pi, cols = run
    ctx = GlCpuContext()
    base = [torch.from_numpy(np.array(c, dtype=np.uint64).view(np.int64)) for c in cols]
    air, opt = gs.plain_air(), gs.Options(num_queries=20, grinding=8)
    proof = gs.Prover(ctx, air, opt).prove(bytes(range(32)), base, lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0], statement=pi)
    got = gs.proof_to_arrays(proof)
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goldilocks_plain_proof.npz")) as f:
        want = {k: f[k] for k in f.files}
    assert set(got) == set(want)
    for k in sorted(want):
        assert np.array_equal(np.asarray(got[k]), want[k]), k


def test_verifier_defaults_and_malformed_proofs(run):
    """the proof's own options are untrusted - a default call requires the CLI's 80 conjectured bits (the 20-query
    fixture carries 28) -, the statement is part of the transcript, and whatever a malformed proof provokes is a VerificationError"""
    import os
    from sandstorm_amd import goldilocks as gs
    prog, states, memory, pi, cols = run
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goldilocks_plain_proof.npz")) as f:
        arrays = {k: f[k] for k in f.files}
    air, seed = gs.plain_air(), bytes(range(32))
    fresh = lambda: gs.proof_from_arrays(arrays)
    with pytest.raises(gs.VerificationError, match="bits of security"):
        gs.verify(fresh(), air, seed, statement=pi)                                         # default: 80 bits
    assert gs.conjectured_security_bits(gs.Options(), 1 << 24) == 81                        # the CLI's defaults clear it
    assert gs.conjectured_security_bits(gs.Options(num_queries=200), 1 << 24) == 128        # capped by the hash
    assert gs.conjectured_security_bits(gs.Options(num_queries=1, grinding=0), 1 << 10) == 1
    # the statement feeds the seed: the same bytes under another public input draw other challenges from the first one on
    assert gs.transcript_seed(seed, gs.Options(), 1 << 10, pi) != gs.transcript_seed(seed, gs.Options(), 1 << 10, None)
    other = copy.deepcopy(pi)
    other.public_memory[0] = (other.public_memory[0][0], other.public_memory[0][1] ^ 1)
    assert gs.statement_digest(other) != gs.statement_digest(pi)
    cases = [lambda p: setattr(p, "ood_comp", None),
             lambda p: setattr(p, "ood_trace", p.ood_trace.astype(np.int64)),
             lambda p: setattr(p.base, "rows", -p.base.rows.astype(np.int64)),
             lambda p: setattr(p, "pow_nonce", 1 << 64),
             lambda p: setattr(p, "pow_nonce", -1),
             lambda p: setattr(p.options, "grinding", 8.0),
             lambda p: setattr(p, "trace_len", 1 << 40),
             lambda p: setattr(p, "base", None),
             lambda p: setattr(p, "remainder", p.remainder[:, :2]),
             lambda p: setattr(p, "base_root", b"short"),
             lambda p: setattr(p.fri_layers[0], "opening", None)]
    for k, mutate in enumerate(cases):
        p = fresh()
        mutate(p)
        with pytest.raises(gs.VerificationError):
            gs.verify(p, air, seed, statement=pi, required_security_bits=28)


@pytest.mark.gpu
def test_plain_layout_at_2p20_steps_proves_and_verifies():
    """BASELINE.json configs[4] at its size: the example program's run padded with its final state to 2^20 steps (a real statement:
    2^24 rows x 5 base + 3 extension coordinate columns), CLI-default options (65 queries, 16 grinding bits), every stage a HIP
    kernel at the size `bench.py --workload goldilocks_plain_2p20` times - the proof verifies (80 conjectured bits), a flipped
    trace cell makes the prover's permutation check or the verifier fail.  PARITY UNPINNED, as everything about this field."""
    import torch
    from sandstorm_amd import backend as be, goldilocks as gs
    log_steps = 20
    prog = pl.example_program(10)
    states, memory = pl.run(prog, 64)
    states = list(states) + [states[-1]] * ((1 << log_steps) - len(states))
    pi = pl.public_input_of(prog, states, memory)
    cols = pl.base_trace_np(states, memory, pi)
    del states
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = be.Context(0, stream=stream.cuda_stream)
    try:
        air, opt = gs.plain_air(), gs.Options()
        base = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
        seed = bytes(range(32))
        proof = gs.Prover(ctx, air, opt).prove(seed, base, lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0], statement=pi)   # check on: the permutations close
        positions = gs.verify(proof, air, seed, statement=pi, expected_options=opt)
        assert len(positions) >= 60 and proof.trace_len == 16 << log_steps if hasattr(proof, "trace_len") else len(positions) >= 60
        bad = copy.deepcopy(proof)
        bad.ood_trace[3, 1] = (int(bad.ood_trace[3, 1]) + 1) % pl.P
        with pytest.raises(gs.VerificationError):
            gs.verify(bad, air, seed, statement=pi, expected_options=opt)
    finally:
        del base
        ctx.close()

"""Run ONLY against the host build of the device code (tests/test_device_code_on_host.py passes this file to pytest explicitly): every
entry point of include/sandstorm_hip.h called with arguments that cannot be served - a NULL context; a live context and everything
else zero / NULL; a live context and huge sizes with NULL data - must come back with an error status (and a message), never crash
and never launch anything.  Each call runs in a forked child, so a crash is reported as that call's failure.  (On the MI355X the
same validation code runs in front of the same kernels; here there is a context to hand in without a GPU.)"""
import ctypes as C
import os

import pytest

# calls that legitimately succeed with nothing to do
FINE_WITH_NOTHING = {"ss_ctx_sync", "ss_ctx_trim", "ss_ctx_set_stream", "ss_profile_enable", "ss_profile_reset", "ss_dev_zero", "ss_dev_free",
                     "ss_upload", "ss_download", "ss_dev_copy", "ss_dev_copy_2d", "ss_gather_batch"}
# ... and calls whose remaining arguments cannot be wrong: no stream / a NULL pointer to free / any flag / NULL outputs
ALWAYS_FINE = {"ss_ctx_sync", "ss_ctx_trim", "ss_profile_reset", "ss_ctx_set_stream", "ss_dev_free", "ss_profile_enable", "ss_profile_read", "ss_profile_read_clock"}
# the first argument of these is a communicator, not a context (a NULL one is refused; there is none to hand in without RCCL)
COMM_FIRST = {"ss_comm_exchange", "ss_comm_all_gather"}
NO_CTX = {"ss_last_error", "ss_abi_version", "ss_ctx_create", "ss_ctx_destroy", "ss_pedersen_hash_host", "ss_keccak256_host", "ss_comm_unique_id"}


def _zero(t):
    if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents"):
        return None
    return t(0)


def _huge(t):
    if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents"):
        return None
    if t in (C.c_int,):
        return t(0x7FFFFFFF)
    if t in (C.c_uint32,):
        return t(0xFFFFFFFF)
    return t(1 << 62)


def _in_child(fn):
    """-> (exit status, signal): run fn() in a forked child and report how it ended"""
    pid = os.fork()
    if pid == 0:
        try:
            os._exit(0 if fn() else 3)
        except BaseException:
            os._exit(4)
    _, status = os.waitpid(pid, 0)
    return (os.WEXITSTATUS(status) if os.WIFEXITED(status) else None), (os.WTERMSIG(status) if os.WIFSIGNALED(status) else None)


@pytest.mark.gpu
def test_every_entry_point_refuses_what_it_cannot_serve():
    from sandstorm_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.ss_ctx_create(0, C.byref(ctx)) == 0
    problems = []
    for name, (res, args) in sorted(_lib.SIGNATURES.items()):
        if name in NO_CTX or res is not C.c_int:
            continue
        fn = getattr(lib, name)
        for label, first, maker in (("NULL context", None, _zero), ("everything else zero / NULL", ctx, _zero), ("huge sizes, NULL data", ctx, _huge)):
            if first is not None and name in COMM_FIRST:
                continue
            argv = [first] + [maker(t) for t in args[1:]]

            def call():
                st = fn(*argv)
                if first is None:
                    return st != 0
                return st != 0 or (name in FINE_WITH_NOTHING and label != "huge sizes, NULL data") or name in ALWAYS_FINE
            code, sig = _in_child(call)
            if sig is not None:
                problems.append("%s(%s): killed by signal %d" % (name, label, sig))
            elif code != 0:
                problems.append("%s(%s): returned success" % (name, label) if code == 3 else "%s(%s): raised" % (name, label))
    lib.ss_ctx_destroy(ctx)
    assert lib.ss_comm_unique_id(None) != 0
    assert not problems, "\n".join(problems)


@pytest.mark.gpu
def test_inconsistent_parameters_are_refused():
    """live buffers, parameters that do not fit them or name nothing: sizes out of range, unknown directions / orders / hash / tree /
    leaf / coin kinds, a NULL entry in a column table, folds that are not 2, 4, 8 or 16, a mask cell or a program operand that names a
    column / constant / slot / table the call does not have, query indices beyond the tree - refused, each in a forked child"""
    from sandstorm_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.ss_ctx_create(0, C.byref(ctx)) == 0

    def alloc(nbytes):
        p = C.c_void_p()
        assert lib.ss_dev_alloc(ctx, nbytes, C.byref(p)) == 0
        lib.ss_dev_zero(ctx, p, nbytes)
        return p

    def cols(ptrs):
        return C.cast((C.c_void_p * len(ptrs))(*[p.value for p in ptrs]), C.POINTER(C.c_void_p))
    u64 = lambda *v: (C.c_uint64 * len(v))(*v)
    u32 = lambda *v: (C.c_uint32 * len(v))(*v)

    def child(f):
        code, sig = _in_child(lambda: f() != 0)
        return "killed by signal %d" % sig if sig is not None else {0: "refused", 3: "ACCEPTED", 4: "raised"}[code]
    n=16
    c0=alloc(32*64); c1=alloc(32*64); out=alloc(32*256); nodes=alloc(64*64); tags=alloc(2*64)
    one=u64(1,0,0,0)   # not Montgomery but a value
    R=[]
    R.append(("ntt log_n=0", child(lambda: lib.ss_ntt_fp252(ctx, cols([c0]), 1, 0, 0, None, 0, 0))))
    R.append(("ntt log_n=40", child(lambda: lib.ss_ntt_fp252(ctx, cols([c0]), 1, 40, 0, None, 0, 0))))
    R.append(("ntt ncols=17", child(lambda: lib.ss_ntt_fp252(ctx, cols([c0]*17), 17, 4, 0, None, 0, 0))))
    R.append(("ntt direction=7", child(lambda: lib.ss_ntt_fp252(ctx, cols([c0]), 1, 4, 7, None, 0, 0))))
    R.append(("ntt order=9", child(lambda: lib.ss_ntt_fp252(ctx, cols([c0]), 1, 4, 0, None, 9, 0))))
    R.append(("ntt NULL column", child(lambda: lib.ss_ntt_fp252(ctx, cols([C.c_void_p(0)]), 1, 4, 0, None, 0, 0))))
    R.append(("lde blowup=0", child(lambda: lib.ss_lde_fp252(ctx, cols([c0]), 1, 4, 0, one, cols([out]), None))))
    R.append(("lde blowup=30", child(lambda: lib.ss_lde_fp252(ctx, cols([c0]), 1, 4, 30, one, cols([out]), None))))
    R.append(("lde offset NULL", child(lambda: lib.ss_lde_fp252(ctx, cols([c0]), 1, 4, 1, None, cols([out]), None))))
    R.append(("hash kind=9", child(lambda: lib.ss_hash_rows(ctx, 9, cols([c0]), 1, 16, out))))
    R.append(("hash nrows=0", child(lambda: lib.ss_hash_rows(ctx, 0, cols([c0]), 1, 0, out))))
    R.append(("merkle tree=9", child(lambda: lib.ss_merkle_build(ctx, 9, 0, 0, c0, 16, nodes, tags, (C.c_uint8*32)()))))
    R.append(("merkle n=12", child(lambda: lib.ss_merkle_build(ctx, 0, 0, 0, c0, 12, nodes, tags, (C.c_uint8*32)()))))
    R.append(("merkle n=0", child(lambda: lib.ss_merkle_build(ctx, 0, 0, 0, c0, 0, nodes, tags, (C.c_uint8*32)()))))
    R.append(("merkle leaf_kind=5", child(lambda: lib.ss_merkle_build(ctx, 0, 0, 5, c0, 16, nodes, tags, (C.c_uint8*32)()))))
    R.append(("merkle_open idx>=n", child(lambda: lib.ss_merkle_open(ctx, nodes, tags, 16, u64(16), 1, (C.c_uint8*(32*4))(), (C.c_uint8*4)()))))
    R.append(("fri fold=3", child(lambda: lib.ss_fri_fold(ctx, c0, 4, 3, one, one, out))))
    R.append(("fri fold=32", child(lambda: lib.ss_fri_fold(ctx, c0, 6, 32, one, one, out))))
    R.append(("fri log_len<log_fold", child(lambda: lib.ss_fri_fold(ctx, c0, 1, 8, one, one, out))))
    R.append(("fri alpha>=p", child(lambda: lib.ss_fri_fold(ctx, c0, 4, 2, u64(2**64-1,2**64-1,2**64-1,2**64-1), one, out))))
    R.append(("pow bits=65", child(lambda: lib.ss_pow_grind(ctx, 0, bytes(32), 65, u64(0)))))
    R.append(("pow coin=9", child(lambda: lib.ss_pow_grind(ctx, 9, bytes(32), 8, u64(0)))))
    R.append(("ood cell_col>=ncols", child(lambda: lib.ss_ood_eval(ctx, cols([c0]), 1, 4, u32(3), u32(0), 1, one, (C.c_uint64*4)()))))
    R.append(("poly_eval ncols=0", child(lambda: lib.ss_poly_eval(ctx, cols([c0]), 0, 4, one, (C.c_uint64*4)()))))
    prog=_lib.AirProgram()
    code=u32(0x00003000 | 0, (5<<24)|0, 7, 0)   # MOV a0 <- trace col 5 (only 1 column); OUT
    prog.code=C.cast(code,C.POINTER(C.c_uint32)); prog.n_instr=2; prog.consts=None; prog.n_consts=0; prog.d_tables=None; prog.table_desc=None; prog.n_tables=0; prog.n_slots=0
    R.append(("quotient trace col>=ncols", child(lambda: lib.ss_eval_quotient(ctx, C.byref(prog), cols([c0]), 1, 4, 1, one, out))))
    code2=u32(0x00002000 | 0, 3, 7, 0)  # MOV a0 <- const 3 (n_consts = 0)
    prog2=_lib.AirProgram(); prog2.code=C.cast(code2,C.POINTER(C.c_uint32)); prog2.n_instr=2
    R.append(("quotient const>=n_consts", child(lambda: lib.ss_eval_quotient(ctx, C.byref(prog2), cols([c0]), 1, 4, 1, one, out))))
    code3=u32(0x00001000 | 0, 9, 7, 0)  # MOV a0 <- slot 9 (n_slots=0)
    prog3=_lib.AirProgram(); prog3.code=C.cast(code3,C.POINTER(C.c_uint32)); prog3.n_instr=2
    R.append(("quotient slot>=n_slots", child(lambda: lib.ss_eval_quotient(ctx, C.byref(prog3), cols([c0]), 1, 4, 1, one, out))))
    code4=u32(0x00004000 | 0, 2, 7, 0)  # table 2 (n_tables=0)
    prog4=_lib.AirProgram(); prog4.code=C.cast(code4,C.POINTER(C.c_uint32)); prog4.n_instr=2
    R.append(("quotient table>=n_tables", child(lambda: lib.ss_eval_quotient(ctx, C.byref(prog4), cols([c0]), 1, 4, 1, one, out))))
    code5=u32(0x00000033, 0, 7, 0)  # opcode 0x33
    prog5=_lib.AirProgram(); prog5.code=C.cast(code5,C.POINTER(C.c_uint32)); prog5.n_instr=2
    R.append(("quotient bad opcode", child(lambda: lib.ss_eval_quotient(ctx, C.byref(prog5), cols([c0]), 1, 4, 1, one, out))))
    # 64-bit field
    g0=alloc(8*64)
    R.append(("gl ntt offset=0", child(lambda: lib.ss_ntt_gl64(ctx, cols([g0]), 1, 4, 0, 0, 0, 0))))
    R.append(("gl ntt offset>=p", child(lambda: lib.ss_ntt_gl64(ctx, cols([g0]), 1, 4, 0, 2**64-1, 0, 0))))
    R.append(("gl lde blowup=20", child(lambda: lib.ss_lde_gl64(ctx, cols([g0]), 1, 4, 20, 7, cols([out]), None))))
    R.append(("gl fri fold=3", child(lambda: lib.ss_fri_fold_gl64x3(ctx, g0, 4, 3, u64(1,2,3), 7, 0, out))))
    R.append(("gl hash kind=9", child(lambda: lib.ss_hash_rows_gl64(ctx, 9, cols([g0]), 1, 1, 16, out))))
    R.append(("gl gather idx>=nrows", child(lambda: lib.ss_gather_rows_gl64(ctx, cols([g0]), 1, 1, 16, u64(99), 1, (C.c_uint64*4)()))))
    lib.ss_ctx_destroy(ctx)
    # accepted on purpose: more columns than one launch's table (served in batches), an extension factor of 1, a NULL offset (= 1),
    # nothing to hash / evaluate.  (Not checkable and not probed: ss_gather_rows is not told the column length - ss_gather_rows_gl64 is, and checks.)
    fine = {"ntt ncols=17", "lde blowup=0", "lde offset NULL", "hash nrows=0", "poly_eval ncols=0", "fri alpha>=p"}
    bad = ["%s: %s" % (k, v) for k, v in R if v != "refused" and k not in fine]
    assert not bad, "\n".join(bad)

#!/usr/bin/env python3
"""Compile the `plain` layout's composition constraint over the 64-bit field into straight-line HIP.

The 64-bit counterpart of tools/gen_quotient.py.  layouts/plain.py builds the composition with every statement- or
transcript-dependent value as a NAMED constant (air_program.Sym), so air_program.lower gives the same program, word for word,
for every statement, trace length and transcript: this tool takes it once and writes
sandstorm_amd/csrc/quotient_gen_plain_gl.inc (included by goldilocks.hip): one kernel whose body is the program unrolled with

  * STATIC typing of every accumulator / slot / constant as base-field (one coordinate) or extension (three): most of the
    program is base-field arithmetic (the CPU constraints before their alpha^k), which the interpreter discovers per lane
    and per instruction at run time;
  * accumulators and scratch slots in registers (no slot file in HBM), operands as immediates, constants through scalar loads.

ss_eval_quotient_gl64x3 recognises the program by the FNV-1a hash of its code words (and checks that every constant typed
base-field here has zero upper coordinates in the table it is given); any other program runs on the interpreter.

Usage (build container):  python tools/gen_quotient_gl.py     - the generated file is committed; the build does not run this.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OP_MOV, OP_ADD, OP_SUB, OP_RSUB, OP_MUL, OP_INV, OP_ST, OP_OUT = range(8)
SRC_ACC, SRC_SLOT, SRC_CONST, SRC_TRACE, SRC_TABLE, SRC_X = range(6)
B, E = "base", "ext"


def code_hash(code):
    h = 0xcbf29ce484222325
    for w in code:
        for k in range(4):
            h = ((h ^ ((int(w) >> (8 * k)) & 0xff)) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def template_program():
    from sandstorm_amd import air_program as ap
    from sandstorm_amd.layouts import plain as pl
    prog = pl.example_program(10)
    states, memory = pl.run(prog, 64)
    pi = pl.public_input_of(prog, states, memory)
    ch = [(11, 22, 33), (5, 6, 7), (9, 8, 7)]                 # generic: upper coordinates non-zero, so an extension constant shows
    n = 1 << 10
    tables = pl.Tables(n, 1)
    root = pl.composition(n, pl.Hints.from_public_input(pi, ch, n), ch, (123, 456, 789), tables)
    p = ap.lower(root, pl.P, ext=True, symbols=tables.symbols)
    return [int(w) for w in p.code], p.consts, p.n_slots, len(tables.specs)


def generate():
    code, consts, n_slots, n_tables = template_program()
    n_instr = len(code) // 2
    const_type = [B if (c[1] == 0 and c[2] == 0) else E for c in consts]
    acc_t, slot_t = [B] * 4, [B] * max(1, n_slots)
    # coordinates of accumulators and scratch values held as lazy words (gl64.h), by variable name; constants, cells, x: canonical
    lazy = {"a%d" % k: [False] * 3 for k in range(4)}
    lazy.update({"s%d" % k: [False] * 3 for k in range(max(1, n_slots))})
    out = []
    emit = out.append
    stats = {"mul1": 0, "mul3": 0, "mul9": 0, "inv": 0, "canon": 0}
    comps = lambda t: (0,) if t == B else (0, 1, 2)

    def canon(var, t):
        """make coordinate t of an accumulator or scratch value canonical, in place"""
        if var is not None and lazy[var][t]:
            emit("    %s.c[%d] = gl_canon(%s.c[%d]);" % (var, t, var, t))
            lazy[var][t] = False
            stats["canon"] += 1

    for pc in range(n_instr):
        w0, w1 = code[2 * pc], code[2 * pc + 1]
        op, d, kind = w0 & 0xff, (w0 >> 8) & 0xf, (w0 >> 12) & 0xf
        a = "a%d" % d
        src_var = None                                            # the source, when it is a variable that may hold lazy words
        if op <= OP_MUL:
            if kind == SRC_ACC:
                src_var = "a%d" % (w1 & 3)
                st, s = acc_t[w1 & 3], lambda t, k=w1 & 3: "a%d.c[%d]" % (k, t)
            elif kind == SRC_SLOT:
                src_var = "s%d" % w1
                st, s = slot_t[w1], lambda t, k=w1: "s%d.c[%d]" % (k, t)
            elif kind == SRC_CONST:
                st, s = const_type[w1], lambda t, k=w1: "QG_K(%d, %d)" % (k, t)
            elif kind == SRC_TRACE:
                st, s = B, lambda t, c=w1 >> 24, o=w1 & 0xffffff: "QG_T(%d, %du)" % (c, o)
            elif kind == SRC_TABLE:
                st, s = B, lambda t, k=w1: "QG_TAB(%d)" % k
            else:
                st, s = B, lambda t: "x"
        src_lazy = lambda t: src_var is not None and lazy[src_var][t]
        dt = acc_t[d]
        if op == OP_MOV:
            if src_var != a:
                for t in comps(st):
                    emit("    %s.c[%d] = %s;" % (a, t, s(t)))
                    lazy[a][t] = src_lazy(t)
            acc_t[d] = st
        elif op in (OP_ADD, OP_SUB, OP_RSUB):
            # gl_add_lazy / gl_sub_lazy (lazy or canonical, CANONICAL) -> lazy: five instructions where the canonical forms take nine / seven
            nt = E if E in (dt, st) else B
            for t in comps(nt):
                has_a, has_s = t in comps(dt), t in comps(st)
                x_, y_ = "%s.c[%d]" % (a, t), (s(t) if has_s else None)
                if op == OP_ADD:
                    if not has_s:
                        continue
                    if not has_a:
                        emit("    %s = %s;" % (x_, y_))
                        lazy[a][t] = src_lazy(t)
                    elif src_var == a:                               # a + a
                        canon(a, t)
                        emit("    %s = gl_add_lazy(%s, %s);" % (x_, x_, x_))
                        lazy[a][t] = True
                    elif not src_lazy(t):
                        emit("    %s = gl_add_lazy(%s, %s);" % (x_, x_, y_))
                        lazy[a][t] = True
                    elif not lazy[a][t]:
                        emit("    %s = gl_add_lazy(%s, %s);" % (x_, y_, x_))
                        lazy[a][t] = True
                    else:
                        canon(src_var, t)
                        emit("    %s = gl_add_lazy(%s, %s);" % (x_, x_, y_))
                elif op == OP_SUB:                                   # a - s: s canonical
                    if not has_s:
                        continue
                    canon(src_var, t)
                    emit("    %s = gl_sub_lazy(%s, %s);" % (x_, x_ if has_a else "0", y_))
                    lazy[a][t] = True
                else:                                                # s - a: a canonical
                    if not has_a:
                        emit("    %s = %s;" % (x_, y_))
                        lazy[a][t] = src_lazy(t)
                        continue
                    canon(a, t)
                    emit("    %s = gl_sub_lazy(%s, %s);" % (x_, y_ if has_s else "0", x_))
                    lazy[a][t] = True
            acc_t[d] = nt
        elif op == OP_MUL:
            # a product takes any words; its result stays lazy (gl_mul_lazy: no final comparison) until something needs the residue itself
            if dt == B and st == B:
                emit("    %s.c[0] = gl_mul_lazy(%s.c[0], %s);" % (a, a, s(0)))
                lazy[a][0] = True
                stats["mul1"] += 1
            elif st == B:
                emit("    { const uint64_t f = %s; %s }" % (s(0), " ".join("%s.c[%d] = gl_mul_lazy(%s.c[%d], f);" % (a, t, a, t) for t in range(3))))
                lazy[a] = [True] * 3
                stats["mul3"] += 1
            elif dt == B:
                emit("    { const uint64_t f = %s.c[0]; %s }" % (a, " ".join("%s.c[%d] = gl_mul_lazy(%s, f);" % (a, t, s(t)) for t in range(3))))
                lazy[a] = [True] * 3
                stats["mul3"] += 1
                acc_t[d] = E
            else:
                canon(src_var, 1), canon(src_var, 2)                 # gl3_mul doubles its right factor's upper coordinates: canonical
                emit("    %s = gl3_mul(%s, Gl3{{%s, %s, %s}});" % (a, a, s(0), s(1), s(2)))
                lazy[a] = [False] * 3
                stats["mul9"] += 1
        elif op == OP_INV:
            emit("    %s.c[0] = gl_pow(%s.c[0], GL_P - 2);" % (a, a) if dt == B else "    %s = gl3_inv(%s);" % (a, a))
            for t in comps(dt):
                lazy[a][t] = False
            stats["inv"] += 1
        elif op == OP_ST:
            for t in comps(dt):
                emit("    s%d.c[%d] = %s.c[%d];" % (w1, t, a, t))
                lazy["s%d" % w1][t] = lazy[a][t]
            slot_t[w1] = dt
        else:
            for t in range(3):
                if t in comps(dt):
                    canon(a, t)
                emit("    QG_OUT(%d, %s);" % (t, "%s.c[%d]" % (a, t) if t in comps(dt) else "0"))
    base_consts = [k for k, t in enumerate(const_type) if t == B]
    h = code_hash(code)
    body = "\n".join(out)
    src = '''// GENERATED by tools/gen_quotient_gl.py - DO NOT EDIT; regenerate with `python tools/gen_quotient_gl.py`.
//
// The composition constraint of the `plain` layout over the 64-bit field (layouts/src/plain/air.rs; sandstorm_amd/layouts/plain.py
// + air_program.lower) as straight-line code: %(n_instr)d program instructions - %(mul1)d base-field products, %(mul3)d products of
// an extension value with a base-field one, %(mul9)d extension products, %(inv)d inversion(s) - %(n_slots)d scratch values and the four
// accumulators in registers, typed base-field / extension at generation time; products, sums and differences are left as lazy
// words (csrc/gl64.h) and made canonical only where a residue is needed (%(canon)d places).  Included by goldilocks.hip; compiled for the host
// (over csrc/gl64.h, the arithmetic both sides share) and held to the oracle by tests/test_gl64_host.py.
// Code hash (FNV-1a of the program's code words) 0x%(hash)016x: ss_eval_quotient_gl64x3 launches this kernel for exactly that program
// (after checking that the constants typed base-field here are base-field in its table) and interprets any other.
static constexpr uint64_t GL3_PLAIN_CODE_HASH = 0x%(hash)016xull;
static constexpr uint32_t GL3_PLAIN_N_INSTR = %(n_instr)du, GL3_PLAIN_N_CONSTS = %(n_consts)du, GL3_PLAIN_N_TABLES = %(n_tables)du;
static const uint16_t GL3_PLAIN_BASE_CONSTS[] = {%(base_consts)s};

__global__ __launch_bounds__(256) void gl3_plain_kernel(Gl3VmArgs a) {
    typedef const uint32_t __attribute__((address_space(4))) *const_u32;
    typedef const uint64_t __attribute__((address_space(4))) *const_u64;
    const_u32 tdesc = (const_u32)(uintptr_t)a.tdesc;
    const_u64 consts = (const_u64)(uintptr_t)a.consts;
    const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x, lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t wstep = gl_pow(a.w, lanes), maskN = a.N - 1;
    const uint32_t lb = a.log_blowup;
    uint64_t x = gl_mul(a.offset, gl_pow(a.w, lane));
#define QG_K(k, t) consts[3 * (k) + (t)]
#define QG_T(col, off) a.cols[col][(i + ((uint64_t)(off) << lb)) & maskN]
#define QG_TAB(t) a.tables[tdesc[2 * (t)] + (i & tdesc[2 * (t) + 1])]
#define QG_OUT(t, v) a.out[3 * i + (t)] = (v)
    for (uint64_t i = lane; i < a.N; i += lanes, x = gl_mul(x, wstep)) {
        Gl3 a0, a1, a2, a3;
        Gl3 %(slots)s;
%(body)s
    }
#undef QG_K
#undef QG_T
#undef QG_TAB
#undef QG_OUT
}
''' % dict(n_instr=n_instr, n_slots=n_slots, hash=h, n_consts=len(consts), n_tables=n_tables, body=body,
           base_consts=", ".join(str(k) for k in base_consts), slots=", ".join("s%d" % k for k in range(max(1, n_slots))), **stats)
    path = os.path.join(ROOT, "sandstorm_amd", "csrc", "quotient_gen_plain_gl.inc")
    with open(path, "w") as f:
        f.write(src)
    print("plain (64-bit field): %d instructions, products %d base / %d mixed / %d extension, %d inversion(s), %d slots, hash 0x%016x -> %s"
          % (n_instr, stats["mul1"], stats["mul3"], stats["mul9"], stats["inv"], n_slots, h, os.path.relpath(path, ROOT)))


if __name__ == "__main__":
    generate()

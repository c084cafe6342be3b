"""The `plain` layout over the 64-bit field p = 2^64 - 2^32 + 1 with challenges in Fq3 = Fp[X]/(X^3 - 2): the claim the
reference instantiates behind its `experimental_claims` feature (cli/src/main.rs:103-133: layouts::plain::AirConfig<Fp, Fq3>;
BASELINE.json configs[4]).

Restated from layouts/src/plain/air.rs (47 constraints: the CPU of the Cairo paper, the memory and 16-bit range-check
permutation arguments; 5 base columns, one Fq3 extension column) and layouts/src/plain/trace.rs (ExecutionTrace::new,
build_extension_columns).  The CPU constraints are the expressions of the other layouts over this layout's cells
(recursive.cpu_constraints with this module as the cell map; air.rs:50-379), the two permutation arguments are written here
because their challenges and running products live in Fq3.

PARITY UNPINNED, and necessarily so: the field crate (ministark-gpu), ministark's generic coin / SHA-256 trees and its default
`composition_constraint` are un-vendored, and the reference ships no program, trace or proof for this field.  What is fixed
here where the reference is silent: the composition is sum_k alpha^k * constraint_k (one Fq3 coefficient, as the other layouts'
air.rs do), an Fq3 column is committed as its three Fp coordinate columns, the multiplicative generator is 7.
Everything is host Python on integers: sizes for tests (the bench uses synthetic columns for the kernels' timing)."""
from dataclasses import dataclass
from typing import List

from .. import air_program as ap
from .. import binary as bn
from . import recursive as _rec
from .recursive import (ALL_CYCLES, EVERY_2ND_EXCEPT_LAST, EVERY_4TH_EXCEPT_LAST, FIRST_ROW, FOURTH_LAST_ROW, SECOND_LAST_ROW, Constraint,
                        _every)

P = 2**64 - 2**32 + 1
GENERATOR = 7
CYCLE_HEIGHT, PUBLIC_MEMORY_STEP, MEMORY_STEP, RANGE_CHECK_STEP = 16, 8, 2, 4            # plain/mod.rs
NUM_BASE_COLUMNS, NUM_EXTENSION_COLUMNS = 5, 1
COL_FLAGS, COL_NPC, COL_MEMORY, COL_RANGE_CHECK, COL_AUXILIARY, COL_PERMUTATION = range(6)
NUM_COMPONENT_COLUMNS = NUM_BASE_COLUMNS + 3 * NUM_EXTENSION_COLUMNS                       # what is committed: 5 + 3 Fp columns
MEM_Z, MEM_A, RC_Z = range(3)                                                            # air.rs:801-823
NUM_CHALLENGES = 3
EVERY_8TH_ROW = _every(8, "every 8th row")


# ---- Fq3 on the host ---------------------------------------------------------------------------------------------------------
def f3(v):
    return v if isinstance(v, tuple) else (v % P, 0, 0)


def add3(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def sub3(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def mul3(a, b):
    d0, d1 = a[0] * b[0], a[0] * b[1] + a[1] * b[0]
    d2, d3, d4 = a[0] * b[2] + a[1] * b[1] + a[2] * b[0], a[1] * b[2] + a[2] * b[1], a[2] * b[2]
    return ((d0 + 2 * d3) % P, (d1 + 2 * d4) % P, d2 % P)


def scale3(a, s):
    return tuple(x * s % P for x in a)


def inv3(a):
    adj = ((a[0] * a[0] - 2 * a[1] * a[2]) % P, (2 * a[2] * a[2] - a[0] * a[1]) % P, (a[1] * a[1] - a[0] * a[2]) % P)
    norm = (a[0] * adj[0] + 2 * a[2] * adj[1] + 2 * a[1] * adj[2]) % P
    return scale3(adj, pow(norm, -1, P)) if norm else (0, 0, 0)


def pow3(a, e):
    r = (1, 0, 0)
    while e:
        if e & 1:
            r = mul3(r, a)
        a, e = mul3(a, a), e >> 1
    return r


def root_of_unity(log_n):
    return pow(GENERATOR, (P - 1) >> log_n, P)


# ---- virtual columns (air.rs:560-760) --------------------------------------------------------------------------------------------
class Npc:
    PC, INSTRUCTION, PUB_MEM_ADDR, PUB_MEM_VAL, MEM_OP0_ADDR, MEM_OP0 = 0, 1, 2, 3, 4, 5
    MEM_DST_ADDR, MEM_DST, MEM_OP1_ADDR, MEM_OP1, GAP_ADDR, GAP_VAL = 8, 9, 12, 13, 14, 15


class Mem:
    ADDRESS, VALUE = 0, 1


class RangeCheck:
    OFF_DST, ORDERED, OFF_OP1, OFF_OP0, UNUSED = 0, 2, 4, 8, 12


class Auxiliary:
    """(column, offset in the cycle): ap, fp, op0*op1 and res sit in the range-check column's free cells (RangeCheck::Ap = 3,
    Op0MulOp1 = 7, Fp = 11, Res = 15), tmp0 / tmp1 in the auxiliary column (air.rs:700-735)"""
    AP, OP0_MUL_OP1, FP, RES = (COL_RANGE_CHECK, 3), (COL_RANGE_CHECK, 7), (COL_RANGE_CHECK, 11), (COL_RANGE_CHECK, 15)
    TMP0, TMP1 = (COL_AUXILIARY, 0), (COL_AUXILIARY, 8)


def flag(f, cycle_offset=0):
    o = CYCLE_HEIGHT * cycle_offset + f
    return ap.Trace(COL_FLAGS, o) - (ap.Trace(COL_FLAGS, o + 1) + ap.Trace(COL_FLAGS, o + 1))


def npc(cell, cycle_offset=0):
    return ap.Trace(COL_NPC, CYCLE_HEIGHT * cycle_offset + cell)


def rc(cell, cycle_offset=0):
    return ap.Trace(COL_RANGE_CHECK, CYCLE_HEIGHT * cycle_offset + cell)


def aux(cell, cycle_offset=0):
    return ap.Trace(cell[0], CYCLE_HEIGHT * cycle_offset + cell[1])


def mem(cell, mem_offset=0):
    return ap.Trace(COL_MEMORY, MEMORY_STEP * mem_offset + cell)


def rc_ordered(step_offset=0):
    return ap.Trace(COL_RANGE_CHECK, RANGE_CHECK_STEP * step_offset + RangeCheck.ORDERED)


_X1, _X2 = ap.Const3(0, 1, 0), ap.Const3(0, 0, 1)


def perm(offset):
    """the Fq3 cell of the permutation column at a row offset: its three committed coordinate columns recombined"""
    c = NUM_BASE_COLUMNS
    return ap.Trace(c, offset) + _X1 * ap.Trace(c + 1, offset) + _X2 * ap.Trace(c + 2, offset)


def perm_memory(step_offset=0):                      # Permutation::Memory: offset 0, step MEMORY_STEP
    return perm(MEMORY_STEP * step_offset)


def perm_range_check(step_offset=0):                 # Permutation::RangeCheck: offset 1, step 4
    return perm(4 * step_offset + 1)


@dataclass
class Hints:
    """PublicInputHint (air.rs:771-799); the products are elements of Fq3"""
    initial_ap: int
    initial_pc: int
    final_ap: int
    final_pc: int
    range_check_min: int = 0
    range_check_max: int = 0
    memory_quotient: tuple = (1, 0, 0)
    range_check_product: tuple = (1, 0, 0)

    @classmethod
    def from_public_input(cls, pi, challenges=None, trace_len=None):
        prog, exe = pi.memory_segments["program"], pi.memory_segments["execution"]
        h = cls(initial_ap=exe[0], initial_pc=prog[0], final_ap=exe[1], final_pc=prog[1], range_check_min=pi.rc_min, range_check_max=pi.rc_max)
        if challenges is not None:
            h.memory_quotient = public_memory_quotient(challenges[MEM_Z], challenges[MEM_A], trace_len or CYCLE_HEIGHT * pi.n_steps, pi)
        return h


def public_memory_quotient(z, alpha, trace_len, pi):
    """compute_public_memory_quotient (layouts/src/utils.rs:14-46) over Fq3: z^S / (prod (z - (a_i + alpha v_i)) * padding^(S - N))"""
    s, count = trace_len // PUBLIC_MEMORY_STEP, len(pi.public_memory)
    den = (1, 0, 0)
    for a, v in pi.public_memory:
        den = mul3(den, sub3(z, add3(scale3(alpha, v % P), f3(a))))
    pad_a, pad_v = pi.public_memory_padding()
    den = mul3(den, pow3(sub3(z, add3(scale3(alpha, pad_v % P), f3(pad_a))), s - count))
    return mul3(pow3(z, s), inv3(den))


def constraints(hints: Hints, challenges=None) -> List[Constraint]:
    """air.rs:381-446, in the reference's order: 33 CPU constraints, 8 for the memory, 6 for the 16-bit range check"""
    import sys
    me = sys.modules[__name__]
    ch = challenges or [(0, 0, 0)] * NUM_CHALLENGES
    c3 = lambda t: t if isinstance(t, ap.Expr) else ap.Const3(*t)          # values, or named constants (composition())
    z, a, zr = c3(ch[MEM_Z]), c3(ch[MEM_A]), c3(ch[RC_Z])
    one = ap.Const(1)
    address_diff = mem(Mem.ADDRESS, 1) - mem(Mem.ADDRESS)
    diff = rc_ordered(1) - rc_ordered()
    return _rec.cpu_constraints(hints, me) + [
        Constraint("memory/multi_column_perm/perm/init0",
                   (z - (mem(Mem.ADDRESS) + a * mem(Mem.VALUE))) * perm_memory() + npc(Npc.PC) + a * npc(Npc.INSTRUCTION) - z, FIRST_ROW),
        # Npc::PubMemAddr.curr() is Trace(1, 2): seen from row 2k it is the NEXT (address, value) pair of the pool
        Constraint("memory/multi_column_perm/perm/step0",
                   (z - (mem(Mem.ADDRESS, 1) + a * mem(Mem.VALUE, 1))) * perm_memory(1)
                   - (z - (ap.Trace(COL_NPC, 2) + a * ap.Trace(COL_NPC, 3))) * perm_memory(), EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/multi_column_perm/perm/last", perm_memory() - c3(hints.memory_quotient), SECOND_LAST_ROW),
        Constraint("memory/diff_is_bit", address_diff * address_diff - address_diff, EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/is_func", (address_diff - one) * (mem(Mem.VALUE) - mem(Mem.VALUE, 1)), EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/initial_addr", mem(Mem.ADDRESS) - one, FIRST_ROW),
        Constraint("public_memory_addr_zero", npc(Npc.PUB_MEM_ADDR), EVERY_8TH_ROW),
        Constraint("public_memory_value_zero", npc(Npc.PUB_MEM_VAL), EVERY_8TH_ROW),
        Constraint("rc16/perm/init0", (zr - rc_ordered()) * perm_range_check() + rc(RangeCheck.OFF_DST) - zr, FIRST_ROW),
        # RangeCheck::OffOp1.curr() is Trace(3, 4): seen from row 4k it is the NEXT unordered value
        Constraint("rc16/perm/step0", (zr - rc_ordered(1)) * perm_range_check(1) - (zr - ap.Trace(COL_RANGE_CHECK, 4)) * perm_range_check(),
                   EVERY_4TH_EXCEPT_LAST),
        Constraint("rc16/perm/last", perm_range_check() - c3(hints.range_check_product), FOURTH_LAST_ROW),
        Constraint("rc16/diff_is_bit", diff * diff - diff, EVERY_4TH_EXCEPT_LAST),
        Constraint("rc16/minimum", rc_ordered() - hints.range_check_min, FIRST_ROW),
        Constraint("rc16/maximum", rc_ordered() - hints.range_check_max, FOURTH_LAST_ROW),
    ]


# ---- zerofier multipliers: periodic factors as tables, single-row factors as expressions ---------------------------------------
class Tables:
    """A domain's multiplier is prod(X^p - g^e over num) / prod(over den).  X^p is periodic in the LDE index with period
    N / gcd(N, p): factors with p > 1 become table operands (a few dozen entries), a factor X - g^e is an expression
    (its inverse a VM INV: four of them per point)."""

    def __init__(self, n, log_blowup=1, offset=GENERATOR, single_rows=None):
        """single_rows: the rows r whose factor X - g^r appears as a DENOMINATOR (first row, last cycle, ...): their inverses share
        ONE inversion per point - 1 / prod (X - g^r), times the other factors - instead of one each"""
        self.n, self.lb, self.offset = n, log_blowup, offset
        self.N = n << log_blowup
        self.g = root_of_unity(n.bit_length() - 1)
        self.specs, self._ix = [], {}
        self.single_rows = sorted(set(r % n for r in (single_rows if single_rows is not None else (0, n - CYCLE_HEIGHT, n - 2, n - 4))))
        self._single_inv = {}
        self.symbols = {}               # named constants of the DAG built with these tables: name -> value (composition() fills it)

    def _row_point(self, r):
        """g^r as an expression: a literal for row 0, a NAMED constant for a row counted from the end (its value depends on n)"""
        r %= self.n
        if r == 0:
            return ap.Const(1)
        name = "g^(n-%d)" % (self.n - r)
        self.symbols[name] = pow(self.g, r, P)
        return ap.Sym(name)

    def _table(self, p_, e, inverse):
        key = (p_, e, inverse)
        if key not in self._ix:
            self._ix[key] = len(self.specs)
            self.specs.append(key)
        return ap.Table(self._ix[key])

    def factor(self, p_, e, inverse=False):
        if p_ == 1:
            f = ap.X - self._row_point(e)
            if not inverse:
                return f
            if e % self.n not in self.single_rows or len(self.single_rows) < 2:
                return f.inverse()
            if not self._single_inv:
                fs = [ap.X - self._row_point(r) for r in self.single_rows]
                prod = fs[0]
                for g_ in fs[1:]:
                    prod = prod * g_
                inv_all = prod.inverse()
                for k, r in enumerate(self.single_rows):
                    others = None
                    for j, g_ in enumerate(fs):
                        if j != k:
                            others = g_ if others is None else others * g_
                    self._single_inv[r] = inv_all * others
            return self._single_inv[e % self.n]
        return self._table(p_, e, inverse)

    def multiplier(self, domain):
        m = None
        for p_, e in domain.num(self.n):
            f = self.factor(p_, e)
            m = f if m is None else m * f
        for p_, e in domain.den(self.n):
            f = self.factor(p_, e, True)
            m = f if m is None else m * f
        return m

    def length(self, spec):
        import math
        return self.N // math.gcd(self.N, spec[0])

    def host_values(self, spec):
        p_, e, inverse = spec
        wN = root_of_unity(self.N.bit_length() - 1)
        step, cur, ge = pow(wN, p_, P), pow(self.offset, p_, P), pow(self.g, e, P)
        out = []
        for _ in range(self.length(spec)):
            v = (cur - ge) % P
            out.append(pow(v, -1, P) if inverse and v else v)
            cur = cur * step % P
        return out

    def value_at(self, spec, x):
        """the factor at an arbitrary Fq3 point (the verifier's side)"""
        p_, e, inverse = spec
        v = sub3(pow3(x, p_), f3(pow(self.g, e, P)))
        return inv3(v) if inverse else v

    def device_tables(self):
        """-> (concatenated values as uint64, table_desc [offset, log2 length] per table)"""
        import numpy as np
        vals, desc = [], []
        for spec in self.specs:
            t = self.host_values(spec)
            desc += [len(vals), len(t).bit_length() - 1]
            vals += t
        return np.array(vals or [0], dtype=np.uint64), desc


def composition(n, hints: Hints, challenges, alpha, tables: Tables):
    """sum_k alpha^k * numerator_k * multiplier(domain_k), alpha in Fq3.  Everything a statement or a transcript decides - hints,
    challenges, alpha^k, the powers of the trace generator - enters the DAG as a NAMED constant (air_program.Sym) whose value
    goes into tables.symbols: every statement of the layout lowers to the same program, word for word."""
    sym = tables.symbols
    names = ("initial_ap", "initial_pc", "final_ap", "final_pc", "range_check_min", "range_check_max", "memory_quotient", "range_check_product")
    for name in names:
        sym["hint:" + name] = getattr(hints, name)
    for k, c in enumerate(challenges):
        sym["challenge:%d" % k] = tuple(c)
    shints = Hints(*[ap.Sym("hint:" + name) for name in names])
    root, coeff = None, (1, 0, 0)
    for k, c in enumerate(constraints(shints, [ap.Sym("challenge:%d" % j) for j in range(NUM_CHALLENGES)])):
        sym["alpha^%d" % k] = coeff
        term = ap.Sym("alpha^%d" % k) * (c.numerator * tables.multiplier(c.domain))
        root = term if root is None else root + term
        coeff = mul3(coeff, alpha)
    return root


def mask(constraint_list=None):
    """the (component column, row offset) cells the constraints read, sorted"""
    cells = set()
    h = Hints(0, 0, 0, 0)
    for c in (constraint_list or constraints(h)):
        stack, seen = [c.numerator], set()
        while stack:
            e = stack.pop()
            if e._id in seen:
                continue
            seen.add(e._id)
            if e.kind == "trace":
                cells.add(tuple(e.args))
            stack += [a for a in e.args if isinstance(a, ap.Expr)]
    return sorted(cells)


# ---- the machine over this field, and the trace --------------------------------------------------------------------------------
def _res(w, pc, ap_, fp, memory):
    if w.pc_update == 4:
        d = memory[w.dst_addr(ap_, fp)] % P
        return pow(d, -1, P) if d else 0
    op0, op1 = memory[w.op0_addr(ap_, fp)], memory[w.op1_addr(pc, ap_, fp, memory)]
    return op1 % P if w.res_logic == 0 else (op0 + op1) % P if w.res_logic == 1 else op0 * op1 % P


def run(program, n_steps, program_base=1):
    """A minimal Cairo machine over this field (the state transition of the Cairo paper, 4.5, with the memory filled in as
    assert_eq / call write it): program = instruction words and immediates; runs n_steps steps (the program is expected to
    end in `jmp rel 0`).  -> (register states, memory list indexed by address, None = never accessed)"""
    execution_base = program_base + len(program)
    memory = [None] * program_base + [w % P for w in program] + [None] * (4 * n_steps + 64)
    ap_, fp, pc = execution_base + 2, execution_base + 2, program_base
    memory[execution_base], memory[execution_base + 1] = execution_base + 2, 0      # the caller's frame of `main`: saved fp, return pc
    states = []
    for _ in range(n_steps):
        states.append(bn.RegisterState(ap_, fp, pc))
        w = bn.Word(memory[pc])
        size = 1 + w.flag(bn.OP1_IMM)
        dst_addr, op0_addr = w.dst_addr(ap_, fp), w.op0_addr(ap_, fp)
        if w.flag(bn.OPCODE_CALL):
            memory[dst_addr], memory[op0_addr] = fp, pc + size
        if memory[op0_addr] is None:
            memory[op0_addr] = 0                                                      # an operand the instruction ignores
        op1_addr = w.op1_addr(pc, ap_, fp, memory)
        if memory[op1_addr] is None:
            memory[op1_addr] = 0
        if w.flag(bn.OPCODE_ASSERT_EQ) and memory[dst_addr] is None:
            memory[dst_addr] = _res(w, pc, ap_, fp, memory)
        if memory[dst_addr] is None:
            memory[dst_addr] = 0
        dst = memory[dst_addr]
        res = _res(w, pc, ap_, fp, memory)
        if w.flag(bn.OPCODE_ASSERT_EQ) and dst != res:
            raise ValueError("assert_eq fails at pc %d" % pc)
        if w.pc_update == 4:
            npc_ = pc + size if dst == 0 else (pc + memory[op1_addr]) % P
        else:
            npc_ = pc + size if w.pc_update == 0 else res if w.pc_update == 1 else (pc + res) % P
        nap = ap_ + (res if w.ap_update == 1 else w.ap_update // 2) + (2 if w.flag(bn.OPCODE_CALL) else 0)
        nfp = ap_ + 2 if w.flag(bn.OPCODE_CALL) else dst if w.flag(bn.OPCODE_RET) else fp
        ap_, fp, pc = nap % P, nfp, npc_
    top = max(a for a, v in enumerate(memory) if v is not None)
    return states, memory[:top + 1]


def instruction(off_dst=0, off_op0=-1, off_op1=-1, flags=()):
    """one instruction word: three biased 16-bit offsets and the flag bits (binary/src/lib.rs:565-600)"""
    w = (off_dst + bn.HALF_OFFSET) | ((off_op0 + bn.HALF_OFFSET) << 16) | ((off_op1 + bn.HALF_OFFSET) << 32)
    for f in flags:
        w |= 1 << (bn.FLAGS_BIT_OFFSET + f)
    return w


def example_program(loops=10):
    """x -> x^2 + 7 in a counted loop (assert_eq with add / mul / immediates, jnz), one call / ret of a multiplying function,
    then the `jmp rel 0` every Cairo program ends in"""
    F = bn
    AE, AP1 = F.OPCODE_ASSERT_EQ, F.AP_ADD1
    push_imm = lambda v: [instruction(0, -1, 1, (F.OP0_REG, F.OP1_IMM, AE, AP1)), v % P]
    prog = push_imm(3) + push_imm(loops)                                                   # x, counter
    loop = (
        [instruction(0, -2, -2, (F.OP1_AP, F.RES_MUL, AE, AP1))]                           # [ap] = [ap-2] * [ap-2]; ap++
        + [instruction(0, -1, 1, (F.OP1_IMM, F.RES_ADD, AE, AP1)), 7]                      # [ap] = [ap-1] + 7; ap++
        + [instruction(0, -3, 1, (F.OP1_IMM, F.RES_ADD, AE, AP1)), P - 1]                  # [ap] = [ap-3] - 1; ap++
    )
    prog += loop
    prog += [instruction(-1, -1, 1, (F.OP0_REG, F.OP1_IMM, F.PC_JNZ)), (P - len(loop)) % P]          # jmp rel -len(loop) if [ap-1] != 0
    prog += [instruction(0, 1, 1, (F.OP1_IMM, F.PC_JUMP_REL, F.OPCODE_CALL)), 4]          # call rel 4 (over this word pair and the jmp below)
    prog += [instruction(-1, -1, 1, (F.DST_REG, F.OP0_REG, F.OP1_IMM, F.PC_JUMP_REL)), 0]  # jmp rel 0
    prog += [instruction(0, -4, -3, (F.OP0_REG, F.OP1_FP, F.RES_MUL, AE, AP1))]            # f: [ap] = [fp-4] * [fp-3]; ap++
    prog += [instruction(-2, -1, -1, (F.DST_REG, F.OP0_REG, F.OP1_FP, F.PC_JUMP_ABS, F.OPCODE_RET))]   # ret
    return prog


@dataclass
class PublicInput:
    """what AirPublicInput carries for this layout (binary/src/lib.rs: rc_min, rc_max, n_steps, memory_segments, public_memory)"""
    n_steps: int
    rc_min: int
    rc_max: int
    memory_segments: dict
    public_memory: list

    def public_memory_padding(self):
        return next((a, v) for a, v in self.public_memory if a == 1)


def public_input_of(program, states, memory, program_base=1):
    pool = [o for st in states for w in [bn.Word(memory[st.pc])] for o in (w.off_dst, w.off_op0, w.off_op1)]
    exe = program_base + len(program) + 2
    return PublicInput(len(states), min(pool), max(pool), {"program": (program_base, states[-1].pc), "execution": (exe, states[-1].ap)},
                       [(program_base + k, w % P) for k, w in enumerate(program)])


def _rc_ordered_with_padding(values):
    ordered = sorted(values)
    padding = [v for a, b in zip(ordered, ordered[1:]) for v in range(a + 1, b)]
    return sorted(ordered + padding), padding


def base_trace(states, memory, pi):
    """ExecutionTrace::new (trace.rs:60-262) -> the 5 base columns as lists of integers"""
    num_cycles = len(states)
    if num_cycles & (num_cycles - 1):
        raise ValueError("the number of cycles must be a power of two")
    n = num_cycles * CYCLE_HEIGHT
    pad_addr, pad_value = pi.public_memory_padding()
    cols = [[0] * n for _ in range(NUM_BASE_COLUMNS)]
    flags, npc_col, rc_col, aux_col = cols[COL_FLAGS], cols[COL_NPC], cols[COL_RANGE_CHECK], cols[COL_AUXILIARY]
    npc_col[0::2], npc_col[1::2] = [pad_addr] * (n // 2), [pad_value] * (n // 2)
    # addresses nothing reads or writes get one gap pair each (pair 7 of every 8 of the pool, trace.rs:92-98; the reference takes
    # the holes of the memory file, which are the same set for a run that touches every cell it wrote)
    touched = {a for a, _ in pi.public_memory}
    for st in states:
        w = bn.Word(memory[st.pc])
        touched |= {st.pc, w.dst_addr(st.ap, st.fp), w.op0_addr(st.ap, st.fp), w.op1_addr(st.pc, st.ap, st.fp, memory)}
    holes = [a for a in range(1, max(touched) + 1) if a not in touched]
    if len(holes) > n // 16:
        raise ValueError("more memory holes than gap cells")
    for k, a in enumerate(holes):
        npc_col[16 * k + Npc.GAP_ADDR], npc_col[16 * k + Npc.GAP_VAL] = a, 0
    pool = []
    for st in states:
        w = bn.Word(memory[st.pc])
        pool += [w.off_dst, w.off_op0, w.off_op1]
    ordered_vals, padding_vals = _rc_ordered_with_padding(pool)
    rc_max = max(pool)
    rc_col[:] = [rc_max] * n
    for cycle, st in enumerate(states):
        r, (pc, ap_, fp) = cycle * CYCLE_HEIGHT, (st.pc, st.ap, st.fp)
        w = bn.Word(memory[pc])
        if w.flag(bn.ZERO):
            raise ValueError("instruction at pc %d has bit 63 set" % pc)
        dst_addr, op0_addr, op1_addr = w.dst_addr(ap_, fp), w.op0_addr(ap_, fp), w.op1_addr(pc, ap_, fp, memory)
        dst, op0, op1 = memory[dst_addr] % P, memory[op0_addr] % P, memory[op1_addr] % P
        res = _res(w, pc, ap_, fp, memory)
        tmp0 = dst if w.flag(bn.PC_JNZ) else 0
        for f in range(16):
            flags[r + f] = w.flag_prefix(f)
        npc_col[r + Npc.PC], npc_col[r + Npc.INSTRUCTION] = pc, memory[pc] % P
        npc_col[r + Npc.MEM_OP0_ADDR], npc_col[r + Npc.MEM_OP0] = op0_addr, op0
        npc_col[r + Npc.MEM_DST_ADDR], npc_col[r + Npc.MEM_DST] = dst_addr, dst
        npc_col[r + Npc.MEM_OP1_ADDR], npc_col[r + Npc.MEM_OP1] = op1_addr, op1
        for o in range(0, CYCLE_HEIGHT, PUBLIC_MEMORY_STEP):
            npc_col[r + o + Npc.PUB_MEM_ADDR] = npc_col[r + o + Npc.PUB_MEM_VAL] = 0
        rc_col[r + RangeCheck.OFF_DST], rc_col[r + RangeCheck.OFF_OP1], rc_col[r + RangeCheck.OFF_OP0] = w.off_dst, w.off_op1, w.off_op0
        rc_col[r + Auxiliary.AP[1]], rc_col[r + Auxiliary.FP[1]] = ap_, fp
        rc_col[r + Auxiliary.OP0_MUL_OP1[1]], rc_col[r + Auxiliary.RES[1]] = op0 * op1 % P, res
        aux_col[r + Auxiliary.TMP0[1]], aux_col[r + Auxiliary.TMP1[1]] = tmp0, tmp0 * res % P
    pad_it, ord_it = iter(padding_vals), iter(ordered_vals)
    for r in range(0, n, CYCLE_HEIGHT):
        rc_col[r + RangeCheck.UNUSED] = next(pad_it, rc_max)
        for o in range(0, CYCLE_HEIGHT, RANGE_CHECK_STEP):
            rc_col[r + o + RangeCheck.ORDERED] = next(ord_it, rc_max)
    if next(pad_it, None) is not None or next(ord_it, None) is not None:
        raise ValueError("range-check values do not fit the trace")
    # sorted memory (get_ordered_memory_accesses, layouts/src/utils.rs:112-152): the pool's accesses with the public-memory
    # cells (address 0 in the pool) replaced by the public memory and its padding
    acc = [(npc_col[2 * k], npc_col[2 * k + 1]) for k in range(n // 2)]
    cells = n // PUBLIC_MEMORY_STEP
    if len(pi.public_memory) > cells:
        raise ValueError("public memory does not fit")
    acc += [(pad_addr, pad_value)] * (cells - len(pi.public_memory)) + [(a, v % P) for a, v in pi.public_memory]
    acc.sort(key=lambda e: e[0])
    if any(a != 0 for a, _ in acc[:cells]) or acc[cells][0] != 1:
        raise ValueError("the public-memory cells must be the only accesses of address 0, and memory starts at 1")
    for (a0, v0), (a1, v1) in zip(acc[cells:], acc[cells + 1:]):
        if not ((a0 == a1 and v0 == v1) or a0 + 1 == a1):
            raise ValueError("memory is not continuous and single-valued at address %d" % a0)
    mem_col = cols[COL_MEMORY]
    mem_col[0::2], mem_col[1::2] = [a for a, _ in acc[cells:]], [v for _, v in acc[cells:]]
    return cols


def base_trace_np(states, memory, pi):
    """base_trace for runs of millions of steps: the same cells as numpy uint64 arrays.  The instruction of a state is decoded once
    per DISTINCT (pc, ap, fp) - a run that idles in `jmp rel 0` (every padded one) has few - and every column is filled by
    index arithmetic; the pools are sorted by numpy.  (tests/test_goldilocks_stark.py holds it to base_trace cell for cell.)"""
    import numpy as np
    num_cycles = len(states)
    if num_cycles & (num_cycles - 1):
        raise ValueError("the number of cycles must be a power of two")
    n = num_cycles * CYCLE_HEIGHT
    u64 = np.uint64
    pad_addr, pad_value = pi.public_memory_padding()
    st = np.array([(s.pc, s.ap, s.fp) for s in states], dtype=u64)
    uniq, inv = np.unique(st, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    mult = np.bincount(inv, minlength=len(uniq))
    U = len(uniq)
    fl = np.zeros((U, 16), dtype=u64)
    cell = {k: np.zeros(U, dtype=u64) for k in ("pc", "inst", "op0a", "op0", "dsta", "dst", "op1a", "op1", "offd", "off0", "off1", "ap", "fp", "mul", "res", "tmp0", "tmp1")}
    touched = {a for a, _ in pi.public_memory}
    for k, (pc, ap_, fp) in enumerate(uniq.tolist()):
        w = bn.Word(memory[pc])
        if w.flag(bn.ZERO):
            raise ValueError("instruction at pc %d has bit 63 set" % pc)
        dst_addr, op0_addr, op1_addr = w.dst_addr(ap_, fp), w.op0_addr(ap_, fp), w.op1_addr(pc, ap_, fp, memory)
        dst, op0, op1 = memory[dst_addr] % P, memory[op0_addr] % P, memory[op1_addr] % P
        res = _res(w, pc, ap_, fp, memory)
        tmp0 = dst if w.flag(bn.PC_JNZ) else 0
        touched |= {pc, dst_addr, op0_addr, op1_addr}
        for f in range(16):
            fl[k, f] = w.flag_prefix(f)
        for name, v in (("pc", pc), ("inst", memory[pc] % P), ("op0a", op0_addr), ("op0", op0), ("dsta", dst_addr), ("dst", dst), ("op1a", op1_addr),
                        ("op1", op1), ("offd", w.off_dst), ("off0", w.off_op0), ("off1", w.off_op1), ("ap", ap_), ("fp", fp), ("mul", op0 * op1 % P),
                        ("res", res), ("tmp0", tmp0), ("tmp1", tmp0 * res % P)):
            cell[name][k] = v
    cols = [np.zeros(n, dtype=u64) for _ in range(NUM_BASE_COLUMNS)]
    flags, npc_col, rc_col, aux_col = cols[COL_FLAGS], cols[COL_NPC], cols[COL_RANGE_CHECK], cols[COL_AUXILIARY]
    npc_col[0::2], npc_col[1::2] = u64(pad_addr), u64(pad_value)
    holes = [a for a in range(1, max(touched) + 1) if a not in touched]
    if len(holes) > n // 16:
        raise ValueError("more memory holes than gap cells")
    hk = np.arange(len(holes), dtype=np.int64) * 16
    npc_col[hk + Npc.GAP_ADDR], npc_col[hk + Npc.GAP_VAL] = np.array(holes, dtype=u64), u64(0)
    # the range-check pool: every cycle's three offsets; ordered with the gaps between the smallest and the largest filled
    cnt = np.zeros(1 << 16, dtype=np.int64)
    for name in ("offd", "off0", "off1"):
        np.add.at(cnt, cell[name].astype(np.int64), mult)
    present = np.nonzero(cnt)[0]
    lo, hi = int(present[0]), int(present[-1])
    rng_vals = np.arange(lo, hi + 1, dtype=np.int64)
    padding_vals = rng_vals[cnt[lo:hi + 1] == 0].astype(u64)
    ordered_vals = np.repeat(rng_vals, np.maximum(cnt[lo:hi + 1], 1)).astype(u64)
    rc_max = u64(hi)
    rc_col[:] = rc_max
    r = np.arange(num_cycles, dtype=np.int64) * CYCLE_HEIGHT
    for f in range(16):
        flags[r + f] = fl[inv, f]
    for off, name in ((Npc.PC, "pc"), (Npc.INSTRUCTION, "inst"), (Npc.MEM_OP0_ADDR, "op0a"), (Npc.MEM_OP0, "op0"), (Npc.MEM_DST_ADDR, "dsta"),
                      (Npc.MEM_DST, "dst"), (Npc.MEM_OP1_ADDR, "op1a"), (Npc.MEM_OP1, "op1")):
        npc_col[r + off] = cell[name][inv]
    for o in range(0, CYCLE_HEIGHT, PUBLIC_MEMORY_STEP):
        npc_col[r + o + Npc.PUB_MEM_ADDR] = npc_col[r + o + Npc.PUB_MEM_VAL] = u64(0)
    for off, name in ((RangeCheck.OFF_DST, "offd"), (RangeCheck.OFF_OP1, "off1"), (RangeCheck.OFF_OP0, "off0"), (Auxiliary.AP[1], "ap"), (Auxiliary.FP[1], "fp"),
                      (Auxiliary.OP0_MUL_OP1[1], "mul"), (Auxiliary.RES[1], "res")):
        rc_col[r + off] = cell[name][inv]
    aux_col[r + Auxiliary.TMP0[1]], aux_col[r + Auxiliary.TMP1[1]] = cell["tmp0"][inv], cell["tmp1"][inv]
    if len(padding_vals) > num_cycles or len(ordered_vals) > n // RANGE_CHECK_STEP:
        raise ValueError("range-check values do not fit the trace")
    unused = np.full(num_cycles, rc_max, dtype=u64)
    unused[:len(padding_vals)] = padding_vals
    rc_col[r + RangeCheck.UNUSED] = unused
    ordered = np.full(n // RANGE_CHECK_STEP, rc_max, dtype=u64)
    ordered[:len(ordered_vals)] = ordered_vals
    rc_col[np.arange(n // RANGE_CHECK_STEP, dtype=np.int64) * RANGE_CHECK_STEP + RangeCheck.ORDERED] = ordered
    # sorted memory: the pool's accesses with the public-memory cells (address 0 in the pool) replaced by the public memory and its padding
    cells = n // PUBLIC_MEMORY_STEP
    if len(pi.public_memory) > cells:
        raise ValueError("public memory does not fit")
    extra = cells - len(pi.public_memory)
    addr = np.concatenate([npc_col[0::2], np.full(extra, pad_addr, dtype=u64), np.array([a for a, _ in pi.public_memory], dtype=u64)])
    val = np.concatenate([npc_col[1::2], np.full(extra, pad_value, dtype=u64), np.array([v % P for _, v in pi.public_memory], dtype=u64)])
    order = np.argsort(addr, kind="stable")
    addr, val = addr[order], val[order]
    if addr[:cells].any() or addr[cells] != 1:
        raise ValueError("the public-memory cells must be the only accesses of address 0, and memory starts at 1")
    a0, a1, v0, v1 = addr[cells:-1], addr[cells + 1:], val[cells:-1], val[cells + 1:]
    bad = ~(((a0 == a1) & (v0 == v1)) | (a0 + u64(1) == a1))
    if bad.any():
        raise ValueError("memory is not continuous and single-valued at address %d" % int(a0[np.nonzero(bad)[0][0]]))
    mem_col = cols[COL_MEMORY]
    mem_col[0::2], mem_col[1::2] = addr[cells:], val[cells:]
    return cols


def _batch_inv3(vals):
    pre, run = [], (1, 0, 0)
    for v in vals:
        pre.append(run)
        run = mul3(run, v)
    inv = inv3(run)
    out = [None] * len(vals)
    for k in range(len(vals) - 1, -1, -1):
        out[k] = mul3(inv, pre[k])
        inv = mul3(inv, vals[k])
    return out


def extension_columns(cols, challenges, check=True):
    """Trace::build_extension_columns (trace.rs:274-330): the permutation column - memory running product on the even rows,
    range-check running product on rows 1 mod 4 - as its three coordinate columns"""
    n = len(cols[0])
    npc_col, mem_col, rc_col = cols[COL_NPC], cols[COL_MEMORY], cols[COL_RANGE_CHECK]
    z, alpha, zr = challenges[MEM_Z], challenges[MEM_A], challenges[RC_Z]
    out = [[0] * n for _ in range(3)]

    def running(nums, dens, first_row, step):
        num_acc, den_acc, ns, ds = (1, 0, 0), (1, 0, 0), [], []
        for a, b in zip(nums, dens):
            num_acc, den_acc = mul3(num_acc, a), mul3(den_acc, b)
            ns.append(num_acc)
            ds.append(den_acc)
        for i, (a, b) in enumerate(zip(ns, _batch_inv3(ds))):
            v = mul3(a, b)
            for t in range(3):
                out[t][first_row + i * step] = v[t]
        return mul3(ns[-1], inv3(ds[-1]))
    last_mem = running([sub3(z, add3(scale3(alpha, npc_col[2 * k + 1]), f3(npc_col[2 * k]))) for k in range(n // 2)],
                       [sub3(z, add3(scale3(alpha, mem_col[2 * k + 1]), f3(mem_col[2 * k]))) for k in range(n // 2)], 0, MEMORY_STEP)
    last_rc = running([sub3(zr, f3(rc_col[4 * k + RangeCheck.OFF_DST])) for k in range(n // 4)],
                      [sub3(zr, f3(rc_col[4 * k + RangeCheck.ORDERED])) for k in range(n // 4)], 1, RANGE_CHECK_STEP)
    if check and last_rc != (1, 0, 0):
        raise ValueError("the range-check permutation does not close")
    return out, last_mem


def failing_rows(constraint, cols8, n, limit=5):
    """rows of the constraint's domain where its numerator does not vanish on the trace (cols8: the 8 component columns)"""
    bad = []
    for r in constraint.domain.rows(n):
        v = ap.evaluate_ext(constraint.numerator, P, (0, 0, 0), lambda c, o: (cols8[c][(r + o) % n], 0, 0), None)       # concrete constants
        if any(v):
            bad.append(r)
            if len(bad) >= limit:
                break
    return bad

"""The `recursive` layout (layouts/src/recursive/{mod,air,trace}.rs): base-trace generation from a `cairo-run`
output and the AIR's constraints as air_program expressions.

STATUS (round 1): the CPU component is restated — the 33 `cpu/*` + initial/final register constraints
(air.rs:82-443) and the trace cells they read (trace.rs:172-232): flags (column 0), the instruction / operand
cells of the memory pool (column 3), the offset cells of the range-check column (column 5) and the auxiliary column
(column 6).  The two restatements validate each other: every constraint vanishes on its domain on the trace
generated from the reference's own example run (tests/test_layout_recursive.py).  Memory, range-check permutation,
Pedersen, bitwise and diluted-check constraints (air.rs:444-1200) and their trace cells are NOT restated yet
(DESIGN.md §8 item 4); `constraints()` lists what exists.

Column map (air.rs:1324-1729): 0 flags | 1 diluted unordered / bitwise | 2 diluted ordered | 3 memory pool ("npc") |
4 sorted memory | 5 range check / Pedersen partial sums | 6 auxiliary / Pedersen suffixes, slopes |
extension: 7 diluted aggregate | 8 diluted permutation | 9 memory + range-check permutation.
"""
from dataclasses import dataclass
from typing import Callable, List

from .. import air_program as ap
from .. import binary as bn

P = bn.P
CYCLE_HEIGHT = 16                   # recursive/mod.rs:16
PUBLIC_MEMORY_STEP, MEMORY_STEP, RANGE_CHECK_STEP, DILUTED_CHECK_STEP = 16, 2, 4, 1
NUM_BASE_COLUMNS, NUM_EXTENSION_COLUMNS = 7, 3
COL_FLAGS, COL_DILUTED_UNORDERED, COL_DILUTED_ORDERED, COL_NPC, COL_MEMORY, COL_RANGE_CHECK, COL_AUXILIARY = range(7)


# ---- virtual columns (air.rs:1324-1695): (column, offset inside the step, step) --------------------------------
class Npc:
    PC, INSTRUCTION, PUB_MEM_ADDR, PUB_MEM_VAL, MEM_OP0_ADDR, MEM_OP0 = 0, 1, 2, 3, 4, 5
    MEM_DST_ADDR, MEM_DST, MEM_OP1_ADDR, MEM_OP1, UNUSED_ADDR, UNUSED_VAL = 8, 9, 12, 13, 14, 15


class RangeCheck:
    OFF_DST, ORDERED, OFF_OP1, OFF_OP0, UNUSED = 0, 2, 4, 8, 12


class Auxiliary:
    AP, TMP0, OP0_MUL_OP1, FP, TMP1, RES = 1, 3, 5, 9, 11, 13


def flag(f, cycle_offset=0):
    """Flag::f.offset(k): the BIT, i.e. prefix_f - 2 prefix_{f+1} (air.rs:1329-1337)"""
    o = CYCLE_HEIGHT * cycle_offset + f
    return ap.Trace(COL_FLAGS, o) - (ap.Trace(COL_FLAGS, o + 1) + ap.Trace(COL_FLAGS, o + 1))


def npc(cell, cycle_offset=0):
    return ap.Trace(COL_NPC, CYCLE_HEIGHT * cycle_offset + cell)


def rc(cell, cycle_offset=0):
    return ap.Trace(COL_RANGE_CHECK, CYCLE_HEIGHT * cycle_offset + cell)


def aux(cell, cycle_offset=0):
    return ap.Trace(COL_AUXILIARY, CYCLE_HEIGHT * cycle_offset + cell)


# ---- domains: where a constraint's numerator must vanish, and the inverse of the zerofier the reference divides by
@dataclass
class Domain:
    name: str
    rows: Callable                  # trace length -> iterable of rows
    zerofier_inv: Callable          # (trace length, g = trace-domain generator) -> Expr in X


def _x_pow(k):
    return ap.X ** k if k > 1 else ap.X


ALL_CYCLES = Domain("every cycle", lambda n: range(0, n, CYCLE_HEIGHT),
                    lambda n, g: (_x_pow(n // CYCLE_HEIGHT) - 1).inverse())
ALL_CYCLES_EXCEPT_LAST = Domain("every cycle but the last", lambda n: range(0, n - CYCLE_HEIGHT, CYCLE_HEIGHT),
                                lambda n, g: (ap.X - pow(g, n - CYCLE_HEIGHT, P)) * (_x_pow(n // CYCLE_HEIGHT) - 1).inverse())
# (X^n - 1) / (X^(n/16) - g^(15 n/16)): every row whose index is not 15 mod 16 (air.rs:140-145)
FLAG_ROWS = Domain("every row holding a flag", lambda n: (r for r in range(n) if r % CYCLE_HEIGHT != 15),
                   lambda n, g: (_x_pow(n // CYCLE_HEIGHT) - pow(g, 15 * n // CYCLE_HEIGHT, P)) * (_x_pow(n) - 1).inverse())
FLAG_ZERO_ROWS = Domain("the 16th row of every cycle", lambda n: range(15, n, CYCLE_HEIGHT),
                        lambda n, g: (_x_pow(n // CYCLE_HEIGHT) - pow(g, 15 * n // CYCLE_HEIGHT, P)).inverse())
FIRST_ROW = Domain("first row", lambda n: [0], lambda n, g: (ap.X - 1).inverse())
LAST_CYCLE = Domain("first row of the last cycle", lambda n: [n - CYCLE_HEIGHT],
                    lambda n, g: (ap.X - pow(g, n - CYCLE_HEIGHT, P)).inverse())


@dataclass
class Constraint:
    name: str                       # StarkWare's name, as the reference's variable (air.rs)
    numerator: object               # air_program.Expr over trace cells, challenges and hints
    domain: Domain


@dataclass
class Hints:
    """PublicInputHint values the CPU constraints use (air.rs:1216-1260)"""
    initial_ap: int
    initial_pc: int
    final_ap: int
    final_pc: int

    @classmethod
    def from_public_input(cls, pi):
        prog, exe = pi.memory_segments["program"], pi.memory_segments["execution"]
        return cls(initial_ap=exe[0], initial_pc=prog[0], final_ap=exe[1], final_pc=prog[1])


def cpu_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:82-443, in the reference's order"""
    F = bn
    one, two, four = ap.Const(1), ap.Const(2), ap.Const(4)
    offset_size, half_offset_size = ap.Const(1 << 16), ap.Const(1 << 15)
    flag_op1_base_op0_0 = one - (flag(F.OP1_IMM) + flag(F.OP1_AP) + flag(F.OP1_FP))
    flag_res_op1_0 = one - (flag(F.RES_ADD) + flag(F.RES_MUL) + flag(F.PC_JNZ))
    flag_pc_update_regular_0 = one - (flag(F.PC_JUMP_ABS) + flag(F.PC_JUMP_REL) + flag(F.PC_JNZ))
    fp_update_regular_0 = one - (flag(F.OPCODE_CALL) + flag(F.OPCODE_RET))
    npc_reg_0 = npc(Npc.PC) + flag(F.OP1_IMM) + one                    # pc + instruction size
    whole_flag_prefix = ap.Trace(COL_FLAGS, 0)
    c = []

    def add(name, numerator, domain):
        c.append(Constraint(name, numerator, domain))

    add("cpu/decode/opcode_rc/bit", flag(F.DST_REG) * flag(F.DST_REG) - flag(F.DST_REG), FLAG_ROWS)
    add("cpu/decode/opcode_rc/zero", whole_flag_prefix, FLAG_ZERO_ROWS)
    add("cpu/decode/opcode_rc_input",
        npc(Npc.INSTRUCTION) - (((whole_flag_prefix * offset_size + rc(RangeCheck.OFF_OP1)) * offset_size
                                 + rc(RangeCheck.OFF_OP0)) * offset_size + rc(RangeCheck.OFF_DST)), ALL_CYCLES)
    for name, e in (("cpu/decode/flag_op1_base_op0_bit", flag_op1_base_op0_0), ("cpu/decode/flag_res_op1_bit", flag_res_op1_0),
                    ("cpu/decode/flag_pc_update_regular_bit", flag_pc_update_regular_0),
                    ("cpu/decode/fp_update_regular_bit", fp_update_regular_0)):
        add(name, e * e - e, ALL_CYCLES)
    add("cpu/operands/mem_dst_addr",
        npc(Npc.MEM_DST_ADDR) + half_offset_size
        - (flag(F.DST_REG) * aux(Auxiliary.FP) + (one - flag(F.DST_REG)) * aux(Auxiliary.AP) + rc(RangeCheck.OFF_DST)), ALL_CYCLES)
    add("cpu/operands/mem0_addr",
        npc(Npc.MEM_OP0_ADDR) + half_offset_size
        - (flag(F.OP0_REG) * aux(Auxiliary.FP) + (one - flag(F.OP0_REG)) * aux(Auxiliary.AP) + rc(RangeCheck.OFF_OP0)), ALL_CYCLES)
    add("cpu/operands/mem1_addr",
        npc(Npc.MEM_OP1_ADDR) + half_offset_size
        - (flag(F.OP1_IMM) * npc(Npc.PC) + flag(F.OP1_AP) * aux(Auxiliary.AP) + flag(F.OP1_FP) * aux(Auxiliary.FP)
           + flag_op1_base_op0_0 * npc(Npc.MEM_OP0) + rc(RangeCheck.OFF_OP1)), ALL_CYCLES)
    add("cpu/operands/ops_mul", aux(Auxiliary.OP0_MUL_OP1) - npc(Npc.MEM_OP0) * npc(Npc.MEM_OP1), ALL_CYCLES)
    add("cpu/operands/res",
        (one - flag(F.PC_JNZ)) * aux(Auxiliary.RES)
        - (flag(F.RES_ADD) * (npc(Npc.MEM_OP0) + npc(Npc.MEM_OP1)) + flag(F.RES_MUL) * aux(Auxiliary.OP0_MUL_OP1)
           + flag_res_op1_0 * npc(Npc.MEM_OP1)), ALL_CYCLES)
    add("cpu/update_registers/update_pc/tmp0", aux(Auxiliary.TMP0) - flag(F.PC_JNZ) * npc(Npc.MEM_DST), ALL_CYCLES_EXCEPT_LAST)
    add("cpu/update_registers/update_pc/tmp1", aux(Auxiliary.TMP1) - aux(Auxiliary.TMP0) * aux(Auxiliary.RES), ALL_CYCLES_EXCEPT_LAST)
    add("cpu/update_registers/update_pc/pc_cond_negative",
        (one - flag(F.PC_JNZ)) * npc(Npc.PC, 1) + aux(Auxiliary.TMP0) * (npc(Npc.PC, 1) - (npc(Npc.PC) + npc(Npc.MEM_OP1)))
        - (flag_pc_update_regular_0 * npc_reg_0 + flag(F.PC_JUMP_ABS) * aux(Auxiliary.RES)
           + flag(F.PC_JUMP_REL) * (npc(Npc.PC) + aux(Auxiliary.RES))), ALL_CYCLES_EXCEPT_LAST)
    add("cpu/update_registers/update_pc/pc_cond_positive",
        (aux(Auxiliary.TMP1) - flag(F.PC_JNZ)) * (npc(Npc.PC, 1) - npc_reg_0), ALL_CYCLES_EXCEPT_LAST)
    add("cpu/update_registers/update_ap/ap_update",
        aux(Auxiliary.AP, 1) - (aux(Auxiliary.AP) + flag(F.AP_ADD) * aux(Auxiliary.RES) + flag(F.AP_ADD1) + flag(F.OPCODE_CALL) * two),
        ALL_CYCLES_EXCEPT_LAST)
    add("cpu/update_registers/update_fp/fp_update",
        aux(Auxiliary.FP, 1) - (fp_update_regular_0 * aux(Auxiliary.FP) + flag(F.OPCODE_RET) * npc(Npc.MEM_DST)
                                + flag(F.OPCODE_CALL) * (aux(Auxiliary.AP) + two)), ALL_CYCLES_EXCEPT_LAST)
    add("cpu/opcodes/call/push_fp", flag(F.OPCODE_CALL) * (npc(Npc.MEM_DST) - aux(Auxiliary.FP)), ALL_CYCLES)
    add("cpu/opcodes/call/push_pc", flag(F.OPCODE_CALL) * (npc(Npc.MEM_OP0) - (npc(Npc.PC) + flag(F.OP1_IMM) + one)), ALL_CYCLES)
    add("cpu/opcodes/call/off0", flag(F.OPCODE_CALL) * (rc(RangeCheck.OFF_DST) - half_offset_size), ALL_CYCLES)
    add("cpu/opcodes/call/off1", flag(F.OPCODE_CALL) * (rc(RangeCheck.OFF_OP0) - (half_offset_size + one)), ALL_CYCLES)
    add("cpu/opcodes/call/flags",
        flag(F.OPCODE_CALL) * (flag(F.OPCODE_CALL) + flag(F.OPCODE_CALL) + one + one - (flag(F.DST_REG) + flag(F.OP0_REG) + four)), ALL_CYCLES)
    add("cpu/opcodes/ret/off0", flag(F.OPCODE_RET) * (rc(RangeCheck.OFF_DST) + two - half_offset_size), ALL_CYCLES)
    add("cpu/opcodes/ret/off2", flag(F.OPCODE_RET) * (rc(RangeCheck.OFF_OP1) + one - half_offset_size), ALL_CYCLES)
    add("cpu/opcodes/ret/flags",
        flag(F.OPCODE_RET) * (flag(F.PC_JUMP_ABS) + flag(F.DST_REG) + flag(F.OP1_FP) + flag_res_op1_0 - four), ALL_CYCLES)
    add("cpu/opcodes/assert_eq/assert_eq", flag(F.OPCODE_ASSERT_EQ) * (npc(Npc.MEM_DST) - aux(Auxiliary.RES)), ALL_CYCLES)
    add("initial_ap", aux(Auxiliary.AP) - hints.initial_ap, FIRST_ROW)
    add("initial_fp", aux(Auxiliary.FP) - hints.initial_ap, FIRST_ROW)
    add("initial_pc", npc(Npc.PC) - hints.initial_pc, FIRST_ROW)
    add("final_ap", aux(Auxiliary.AP) - hints.final_ap, LAST_CYCLE)
    add("final_fp", aux(Auxiliary.FP) - hints.initial_ap, LAST_CYCLE)
    add("final_pc", npc(Npc.PC) - hints.final_pc, LAST_CYCLE)
    return c


def constraints(hints: Hints) -> List[Constraint]:
    """what is restated so far (see the module docstring)"""
    return cpu_constraints(hints)


# ---- base trace (trace.rs:95-232), CPU cells ----------------------------------------------------------------------
def cpu_trace(register_states, memory, public_input):
    """-> the 7 base columns (lists of canonical ints, 16 rows per cycle) with the CPU cells filled:
    flags, the instruction/operand cells of the memory pool, the offset cells of the range-check column and the
    auxiliary column.  Cells owned by components that are not restated yet keep the reference's initial fill
    (memory pool: the public-memory padding entry; range check: rc_max; everything else 0)."""
    num_cycles = len(register_states)
    if num_cycles & (num_cycles - 1):
        raise ValueError("the number of cycles must be a power of two")
    n = num_cycles * CYCLE_HEIGHT
    pad_addr, pad_value = public_input.public_memory_padding()
    cols = [[0] * n for _ in range(NUM_BASE_COLUMNS)]
    flags, npc_col, rc_col, aux_col = cols[COL_FLAGS], cols[COL_NPC], cols[COL_RANGE_CHECK], cols[COL_AUXILIARY]
    npc_col[0::2] = [pad_addr] * (n // 2)                   # trace.rs:118-128
    npc_col[1::2] = [pad_value] * (n // 2)
    rc_col[:] = [public_input.rc_max] * n                   # range_check_padding_value (trace.rs:157-159)
    for cycle, st in enumerate(register_states):
        r = cycle * CYCLE_HEIGHT
        pc, ap_, fp = st.pc, st.ap, st.fp
        w = bn.Word(memory[pc])
        if w.flag(bn.ZERO):
            raise ValueError("instruction at pc %d has bit 63 set" % pc)
        dst_addr, op0_addr = w.dst_addr(ap_, fp), w.op0_addr(ap_, fp)
        op1_addr = w.op1_addr(pc, ap_, fp, memory)
        dst, op0, op1 = memory[dst_addr] % P, memory[op0_addr] % P, memory[op1_addr] % P
        res = w.res(pc, ap_, fp, memory)
        tmp0 = dst if w.flag(bn.PC_JNZ) else 0              # get_tmp0 / get_tmp1 (binary/src/lib.rs:705-716)
        for f in range(16):
            flags[r + f] = w.flag_prefix(f)
        npc_col[r + Npc.PC], npc_col[r + Npc.INSTRUCTION] = pc, memory[pc] % P
        npc_col[r + Npc.MEM_OP0_ADDR], npc_col[r + Npc.MEM_OP0] = op0_addr, op0
        npc_col[r + Npc.MEM_DST_ADDR], npc_col[r + Npc.MEM_DST] = dst_addr, dst
        npc_col[r + Npc.MEM_OP1_ADDR], npc_col[r + Npc.MEM_OP1] = op1_addr, op1
        npc_col[r + Npc.PUB_MEM_ADDR] = npc_col[r + Npc.PUB_MEM_VAL] = 0
        rc_col[r + RangeCheck.OFF_DST], rc_col[r + RangeCheck.OFF_OP1], rc_col[r + RangeCheck.OFF_OP0] = w.off_dst, w.off_op1, w.off_op0
        aux_col[r + Auxiliary.TMP0], aux_col[r + Auxiliary.TMP1] = tmp0, tmp0 * res % P
        aux_col[r + Auxiliary.AP], aux_col[r + Auxiliary.FP] = ap_, fp
        aux_col[r + Auxiliary.OP0_MUL_OP1], aux_col[r + Auxiliary.RES] = op0 * op1 % P, res
    return cols


def failing_rows(constraint: Constraint, cols, rows=None, limit=5):
    """rows of the constraint's domain (or of `rows`) where its numerator does not vanish on the trace"""
    n = len(cols[0])
    bad = []
    for r in (constraint.domain.rows(n) if rows is None else rows):
        v = ap.evaluate(constraint.numerator, P, None, lambda c, o: cols[c][(r + o) % n], lambda t: 0)
        if v:
            bad.append(r)
            if len(bad) >= limit:
                break
    return bad

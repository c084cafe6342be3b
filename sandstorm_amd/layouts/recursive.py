"""The `recursive` layout (layouts/src/recursive/{mod,air,trace}.rs): base-trace generation from a `cairo-run`
output and the AIR's constraints as air_program expressions.

STATUS (round 1): all 93 constraints are restated, in the reference's order (air.rs:1083-1180) — CPU (33), memory and
public memory (8), 16-bit range check (6), diluted check (7), Pedersen builtin (25), range-check builtin (3), bitwise
builtin (11) — together with the base-trace generation they are checked against (trace.rs:95-660, builtins/src/
{pedersen,bitwise,range_check}/mod.rs, layouts/src/utils.rs).  The two restatements validate each other
(tests/test_layout_recursive.py): every constraint vanishes on its domain on the trace generated from the reference's
own example run and on traces with real builtin instances; the memory product closes to the public-memory quotient,
the range-check and diluted products to one, the diluted aggregate to its closed form, every Pedersen partial sum ends
at the library's Pedersen hash; and the set of trace cells the constraints read is exactly the 133-cell mask whose
size the reference's shipped proof confirms.

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
PEDERSEN_BUILTIN_RATIO, RANGE_CHECK_BUILTIN_RATIO, RANGE_CHECK_BUILTIN_PARTS, BITWISE_RATIO = 128, 8, 8, 8
NUM_BASE_COLUMNS, NUM_EXTENSION_COLUMNS = 7, 3
COL_DILUTED_AGGREGATE, COL_DILUTED_PERMUTATION, COL_MEM_RC_PERMUTATION = 7, 8, 9
DILUTED_CHECK_N_BITS, DILUTED_CHECK_SPACING = 16, 4
# challenge indices (air.rs:1759-1801)
MEM_Z, MEM_A, RC_Z, DC_Z, AGG_Z, AGG_A = range(6)
COL_FLAGS, COL_DILUTED_UNORDERED, COL_DILUTED_ORDERED, COL_NPC, COL_MEMORY, COL_RANGE_CHECK, COL_AUXILIARY = range(7)


# ---- virtual columns (air.rs:1324-1695): (column, offset inside the step, step) --------------------------------
class Npc:
    PC, INSTRUCTION, PUB_MEM_ADDR, PUB_MEM_VAL, MEM_OP0_ADDR, MEM_OP0 = 0, 1, 2, 3, 4, 5
    MEM_DST_ADDR, MEM_DST, MEM_OP1_ADDR, MEM_OP1, UNUSED_ADDR, UNUSED_VAL = 8, 9, 12, 13, 14, 15


    # builtin cells (air.rs:1489-1507): offset inside the builtin's own step
    PEDERSEN_INPUT0_ADDR, PEDERSEN_INPUT1_ADDR, PEDERSEN_OUTPUT_ADDR = 10, 1034, 522
    RANGE_CHECK128_ADDR, BITWISE_POOL_ADDR, BITWISE_X_OR_Y_ADDR = 74, 26, 42


class Mem:
    ADDRESS, VALUE = 0, 1


class RangeCheck:
    OFF_DST, ORDERED, OFF_OP1, OFF_OP0, UNUSED = 0, 2, 4, 8, 12
    RC16_COMPONENT = 12             # RangeCheckBuiltin::Rc16Component: one 16-bit part per cycle


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


# ---- domains: where a constraint's numerator must vanish, and the zerofier quotient the reference multiplies by ----
# A factor is (p, e): X^p - g^e with g the trace-domain generator; a domain's multiplier is prod(num) / prod(den).
@dataclass
class Domain:
    name: str
    rows: Callable                  # trace length -> iterable of rows where the numerator must vanish
    num: Callable                   # trace length -> list of factors (p, e)
    den: Callable

    def multiplier_at(self, n, x):
        """prod(num) / prod(den) at an arbitrary point x (the verifier's side of the out-of-domain identity)"""
        g = pow(3, (P - 1) // n, P)
        val = 1
        for p_, e in self.num(n):
            val = val * (pow(x, p_, P) - pow(g, e, P)) % P
        d = 1
        for p_, e in self.den(n):
            d = d * (pow(x, p_, P) - pow(g, e, P)) % P
        return val * pow(d, -1, P) % P


def _every(k, name):
    """rows 0, k, 2k, ...: 1 / (X^(n/k) - 1)"""
    return Domain(name, lambda n: range(0, n, k), lambda n: [], lambda n: [(n // k, 0)])


def _every_except_last(k, name):
    return Domain(name, lambda n: range(0, n - k, k), lambda n: [(1, n - k)], lambda n: [(n // k, 0)])


def _row_from_end(k, name):
    """the single row n - k"""
    return Domain(name, lambda n: [n - k], lambda n: [], lambda n: [(1, n - k)])


ALL_CYCLES, ALL_CYCLES_EXCEPT_LAST = _every(CYCLE_HEIGHT, "every cycle"), _every_except_last(CYCLE_HEIGHT, "every cycle but the last")
# (X^n - 1) / (X^(n/16) - g^(15 n/16)): every row whose index is not 15 mod 16 (air.rs:140-145)
FLAG_ROWS = Domain("every row holding a flag", lambda n: (r for r in range(n) if r % CYCLE_HEIGHT != 15),
                   lambda n: [(n // CYCLE_HEIGHT, 15 * n // CYCLE_HEIGHT)], lambda n: [(n, 0)])
FLAG_ZERO_ROWS = Domain("the 16th row of every cycle", lambda n: range(15, n, CYCLE_HEIGHT),
                        lambda n: [], lambda n: [(n // CYCLE_HEIGHT, 15 * n // CYCLE_HEIGHT)])
FIRST_ROW = Domain("first row", lambda n: [0], lambda n: [], lambda n: [(1, 0)])
LAST_CYCLE = _row_from_end(CYCLE_HEIGHT, "first row of the last cycle")
EVERY_2ND_EXCEPT_LAST, SECOND_LAST_ROW = _every_except_last(2, "every 2nd row but the last"), _row_from_end(2, "row n-2")
EVERY_4TH_EXCEPT_LAST, FOURTH_LAST_ROW = _every_except_last(4, "every 4th row but the last"), _row_from_end(4, "row n-4")
EVERY_32 = _every(32, "every 32nd row")
EVERY_ROW_EXCEPT_LAST, LAST_ROW = _every_except_last(1, "every row but the last"), _row_from_end(1, "last row")
# (X^(n/32) - 1) / (X^(n/128) - g^(3n/4)): rows 0, 32, 64 (not 96) of every 128 (air.rs:926-928)
BITWISE_TRANSITION = Domain("rows 0, 32, 64 of every 128", lambda n: (r for r in range(0, n, 32) if r % 128 != 96),
                            lambda n: [(n // 128, 3 * n // 4)], lambda n: [(n // 32, 0)])
# air.rs:961-978: 1 / prod_{k<16} (X^(n/128) - g^(k n/64))
EVERY_16_BIT_SEGMENT = Domain("rows 0, 2, ..., 30 of every 128", lambda n: (r for r in range(0, n, 2) if r % 128 < 32),
                              lambda n: [], lambda n: [(n // 128, k * n // 64) for k in range(16)])
EVERY_128, EVERY_128_EXCEPT_LAST = _every(128, "every 128th row"), _every_except_last(128, "every 128th row but the last")


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
    range_check_min: int = 0
    range_check_max: int = 0
    initial_rc_addr: int = 0
    memory_quotient: int = 0        # needs the challenges
    range_check_product: int = 1
    diluted_check_product: int = 1
    diluted_check_first: int = 0
    diluted_check_cumulative_value: int = 0     # needs the challenges
    initial_bitwise_addr: int = 0
    initial_pedersen_addr: int = 0

    @classmethod
    def from_public_input(cls, pi, challenges=None, trace_len=None):
        prog, exe = pi.memory_segments["program"], pi.memory_segments["execution"]
        h = cls(initial_ap=exe[0], initial_pc=prog[0], final_ap=exe[1], final_pc=prog[1], range_check_min=pi.rc_min,
                range_check_max=pi.rc_max, initial_rc_addr=pi.memory_segments["range_check"][0],
                initial_bitwise_addr=pi.memory_segments["bitwise"][0], initial_pedersen_addr=pi.memory_segments["pedersen"][0])
        if challenges is not None:
            h.memory_quotient = public_memory_quotient(challenges[MEM_Z], challenges[MEM_A], trace_len or 16 * pi.n_steps, pi)
            h.diluted_check_cumulative_value = diluted_cumulative_value(challenges[AGG_Z], challenges[AGG_A])
        return h


def diluted_cumulative_value(z, alpha, n_bits=DILUTED_CHECK_N_BITS, spacing=DILUTED_CHECK_SPACING):
    """compute_diluted_cumulative_value (layouts/src/utils.rs:48-108): the aggregate after running over every diluted
    n_bits-bit value once, in log steps"""
    diff_multiplier, diff_x = 1 << spacing, (1 << spacing) - 2
    p, q, x = (z + 1) % P, 1, 1
    for _ in range(1, n_bits):
        x = (x + diff_x) % P
        diff_x = diff_x * diff_multiplier % P
        xp = x * p % P
        y = (p + z * xp) % P
        q = (q + q * y + x * xp) % P
        p = p * y % P
    return (p + q * alpha) % P


def public_memory_quotient(z, alpha, trace_len, pi, public_memory_step=PUBLIC_MEMORY_STEP):
    """compute_public_memory_quotient (layouts/src/utils.rs:14-46): z^S / (prod (z - (a_i + alpha v_i)) * padding^(S-N))"""
    s, count = trace_len // public_memory_step, len(pi.public_memory)
    den = 1
    for a, v in pi.public_memory:
        den = den * (z - (alpha * v + a)) % P
    pad_a, pad_v = pi.public_memory_padding()
    den = den * pow((z - (alpha * pad_v + pad_a)) % P, s - count, P) % P
    return pow(z, s, P) * pow(den, -1, P) % P


def cpu_constraints(hints: Hints, L=None) -> List[Constraint]:
    """air.rs:82-443, in the reference's order.  L: the layout module that supplies the cell helpers (this one by default;
    layouts/starknet.py passes itself - its CPU constraints are the same expressions over other cells, starknet/air.rs:128-557)"""
    import sys
    L = L or sys.modules[__name__]
    flag, npc, rc, aux, Npc, RangeCheck, Auxiliary, COL_FLAGS = L.flag, L.npc, L.rc, L.aux, L.Npc, L.RangeCheck, L.Auxiliary, L.COL_FLAGS
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


def npc_at(offset):
    return ap.Trace(COL_NPC, offset)


def mem(cell, mem_offset=0):
    return ap.Trace(COL_MEMORY, MEMORY_STEP * mem_offset + cell)


def rc_ordered(step_offset=0):
    return ap.Trace(COL_RANGE_CHECK, RANGE_CHECK_STEP * step_offset + RangeCheck.ORDERED)


def perm_memory(step_offset=0):                  # Permutation::Memory -> (9, 0), step MEMORY_STEP
    return ap.Trace(COL_MEM_RC_PERMUTATION, MEMORY_STEP * step_offset)


def perm_range_check(step_offset=0):             # Permutation::RangeCheck -> (9, 1), step 4
    return ap.Trace(COL_MEM_RC_PERMUTATION, 4 * step_offset + 1)


def memory_constraints(hints: Hints, challenges) -> List[Constraint]:
    """air.rs:444-497"""
    z, a = ap.Const(challenges[MEM_Z]), ap.Const(challenges[MEM_A])
    one = ap.Const(1)
    address_diff = mem(Mem.ADDRESS, 1) - mem(Mem.ADDRESS)
    return [
        Constraint("memory/multi_column_perm/perm/init0",
                   (z - (mem(Mem.ADDRESS) + a * mem(Mem.VALUE))) * perm_memory() + npc(Npc.PC) + a * npc(Npc.INSTRUCTION) - z, FIRST_ROW),
        # Npc::PubMemAddr.curr() is Trace(3, 2): seen from row 2k it is the NEXT (address, value) pair of the pool
        Constraint("memory/multi_column_perm/perm/step0",
                   (z - (mem(Mem.ADDRESS, 1) + a * mem(Mem.VALUE, 1))) * perm_memory(1)
                   - (z - (npc_at(2) + a * npc_at(3))) * perm_memory(), EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/multi_column_perm/perm/last", perm_memory() - hints.memory_quotient, SECOND_LAST_ROW),
        Constraint("memory/diff_is_bit", address_diff * address_diff - address_diff, EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/is_func", (address_diff - one) * (mem(Mem.VALUE) - mem(Mem.VALUE, 1)), EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/initial_addr", mem(Mem.ADDRESS) - one, FIRST_ROW),
        Constraint("public_memory_addr_zero", npc(Npc.PUB_MEM_ADDR), ALL_CYCLES),
        Constraint("public_memory_value_zero", npc(Npc.PUB_MEM_VAL), ALL_CYCLES),
    ]


def range_check_constraints(hints: Hints, challenges) -> List[Constraint]:
    """air.rs:499-538 (16-bit range check) and 899-918 (the 128-bit range-check builtin)"""
    z = ap.Const(challenges[RC_Z])
    diff = rc_ordered(1) - rc_ordered()
    offset_size = ap.Const(1 << 16)
    value = None                                   # rc_builtin_value7_0: the 8 parts, most significant first (air.rs:105-119)
    for k in range(RANGE_CHECK_BUILTIN_PARTS):
        part = ap.Trace(COL_RANGE_CHECK, CYCLE_HEIGHT * k + RangeCheck.RC16_COMPONENT)
        value = part if value is None else value * offset_size + part
    step = CYCLE_HEIGHT * RANGE_CHECK_BUILTIN_RATIO
    return [
        Constraint("rc16/perm/init0", (z - rc_ordered()) * perm_range_check() + rc(RangeCheck.OFF_DST) - z, FIRST_ROW),
        # RangeCheck::OffOp1.curr() is Trace(5, 4): seen from row 4k it is the NEXT unordered value
        Constraint("rc16/perm/step0", (z - rc_ordered(1)) * perm_range_check(1) - (z - ap.Trace(COL_RANGE_CHECK, 4)) * perm_range_check(),
                   EVERY_4TH_EXCEPT_LAST),
        Constraint("rc16/perm/last", perm_range_check() - hints.range_check_product, FOURTH_LAST_ROW),
        Constraint("rc16/diff_is_bit", diff * diff - diff, EVERY_4TH_EXCEPT_LAST),
        Constraint("rc16/minimum", rc_ordered() - hints.range_check_min, FIRST_ROW),
        Constraint("rc16/maximum", rc_ordered() - hints.range_check_max, FOURTH_LAST_ROW),
        Constraint("rc_builtin/value", value - npc_at(Npc.RANGE_CHECK128_ADDR + 1), EVERY_128),
        Constraint("rc_builtin/addr_step", npc_at(step + Npc.RANGE_CHECK128_ADDR) - (npc_at(Npc.RANGE_CHECK128_ADDR) + 1), EVERY_128_EXCEPT_LAST),
        Constraint("rc_builtin/init_addr", npc_at(Npc.RANGE_CHECK128_ADDR) - hints.initial_rc_addr, FIRST_ROW),
    ]


def diluted_check_constraints(hints: Hints, challenges) -> List[Constraint]:
    """air.rs:540-603; DILUTED_CHECK_STEP = 1: every row"""
    z, za, aa = ap.Const(challenges[DC_Z]), ap.Const(challenges[AGG_Z]), ap.Const(challenges[AGG_A])
    un, od = (lambda o=0: ap.Trace(COL_DILUTED_UNORDERED, o)), (lambda o=0: ap.Trace(COL_DILUTED_ORDERED, o))
    perm, agg = (lambda o=0: ap.Trace(COL_DILUTED_PERMUTATION, o)), (lambda o=0: ap.Trace(COL_DILUTED_AGGREGATE, o))
    diff = od(1) - od()
    return [
        Constraint("diluted_check/permutation/init0", (z - od()) * perm() + un() - z, FIRST_ROW),
        Constraint("diluted_check/permutation/step0", (z - od(1)) * perm(1) - (z - un(1)) * perm(), EVERY_ROW_EXCEPT_LAST),
        Constraint("diluted_check/permutation/last", perm() - hints.diluted_check_product, LAST_ROW),
        Constraint("diluted_check/init", agg() - 1, FIRST_ROW),
        Constraint("diluted_check/first_element", od() - hints.diluted_check_first, FIRST_ROW),
        Constraint("diluted_check/step", agg(1) - (agg() * (1 + za * diff) + aa * diff * diff), EVERY_ROW_EXCEPT_LAST),
        Constraint("diluted_check/last", agg() - hints.diluted_check_cumulative_value, LAST_ROW),
    ]


def bitwise_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:920-1081.  A bitwise instance spans 128 rows: four 32-row partitions (x, y, x&y, x^y) of column 1, whose
    even cells 0..30 hold the 16 diluted 16-bit segments (chunk c, stream s at cell 8c + 2s)."""
    bw = lambda o: ap.Trace(COL_DILUTED_UNORDERED, o)
    pool_addr = lambda k: npc_at(32 * k + Npc.BITWISE_POOL_ADDR)
    pool_val = lambda k: npc_at(32 * k + Npc.BITWISE_POOL_ADDR + 1)
    two = ap.Const(2)
    sum_var = None                                   # bitwise_sum_var_0_0 + bitwise_sum_var_8_0 (air.rs:121-138)
    for chunk in range(4):
        for stream in range(4):
            term = bw(8 * chunk + 2 * stream)
            shift = 64 * chunk + stream
            term = term * (1 << shift) if shift else term
            sum_var = term if sum_var is None else sum_var + term
    out = [
        Constraint("bitwise/init_var_pool_addr", pool_addr(0) - hints.initial_bitwise_addr, FIRST_ROW),
        Constraint("bitwise/step_var_pool_addr", pool_addr(1) - (pool_addr(0) + 1), BITWISE_TRANSITION),
        Constraint("bitwise/x_or_y_addr", npc_at(Npc.BITWISE_X_OR_Y_ADDR) - (pool_addr(3) + 1), EVERY_128),
        Constraint("bitwise/next_var_pool_addr", pool_addr(4) - (npc_at(Npc.BITWISE_X_OR_Y_ADDR) + 1), EVERY_128_EXCEPT_LAST),
        Constraint("bitwise/partition", sum_var - pool_val(0), EVERY_32),
        Constraint("bitwise/or_is_and_plus_xor", npc_at(Npc.BITWISE_X_OR_Y_ADDR + 1) - (pool_val(2) + pool_val(3)), EVERY_128),
        Constraint("bitwise/addition_is_xor_with_and", bw(0) + bw(32) - (bw(96) + bw(64) + bw(64)), EVERY_16_BIT_SEGMENT),
    ]
    # unique unpacking of the top chunk: (and + xor) segments shifted (air.rs:1052-1081); cells 1, 65, 33, 97 hold the results
    for k, (cell, shift) in enumerate(((1, 4), (65, 4), (33, 4), (97, 8))):
        seg = 24 + 2 * k
        out.append(Constraint("bitwise/unique_unpacking%d" % (192 + k), (bw(64 + seg) + bw(96 + seg)) * (1 << shift) - bw(cell), EVERY_128))
    return out


def pedersen_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:605-895.  A hash spans 2048 rows = 512 steps of 4 rows (256 for each input): suffix at row 4k of column 6,
    slope at 4k + 2, partial sum x / y at rows 4k + 1 / 4k + 3 of column 5; the two flag cells of an input at rows 7 and
    1022 of its 1024 rows."""
    suffix = lambda k=0: ap.Trace(COL_AUXILIARY, 4 * k)
    slope = lambda k=0: ap.Trace(COL_AUXILIARY, 4 * k + 2)
    sum_x = lambda k=0: ap.Trace(COL_RANGE_CHECK, 4 * k + 1)
    sum_y = lambda k=0: ap.Trace(COL_RANGE_CHECK, 4 * k + 3)
    bit_251_196_192, bit_251_196 = ap.Trace(COL_AUXILIARY, 7), ap.Trace(COL_AUXILIARY, 1022)
    point_x, point_y = ap.Table(TABLE_PEDERSEN_X), ap.Table(TABLE_PEDERSEN_Y)
    one = ap.Const(1)
    bit = lambda k: suffix(k) - (suffix(k + 1) + suffix(k + 1))
    b0 = bit(0)
    b0_negate = one - b0
    every_1024, every_2048, every_2048_except_last = _every(1024, "every 1024th row"), _every(2048, "every 2048th row"), \
        _every_except_last(2048, "every 2048th row but the last")
    # (X^(n/4) - 1) / (X^(n/1024) - g^(255 n/256)): every 4th row except step 255 of an input (air.rs:652-654)
    transition = Domain("steps 0..254 of every input", lambda n: (r for r in range(0, n, 4) if r % 1024 != 1020),
                        lambda n: [(n // 1024, 255 * n // 256)], lambda n: [(n // 4, 0)])
    step_252 = Domain("step 252 of every input", lambda n: range(1008, n, 1024), lambda n: [], lambda n: [(n // 1024, 63 * n // 64)])
    step_255 = Domain("step 255 of every input", lambda n: range(1020, n, 1024), lambda n: [], lambda n: [(n // 1024, 255 * n // 256)])
    px, py = PEDERSEN_POINTS[0]
    H = "pedersen/hash0/ec_subset_sum/"
    return [
        Constraint(H + "bit_unpacking/last_one_is_zero", bit_251_196_192 * bit(0), every_1024),
        Constraint(H + "bit_unpacking/zeroes_between_ones0", bit_251_196_192 * (suffix(1) - suffix(192) * (1 << 191)), every_1024),
        Constraint(H + "bit_unpacking/cumulative_bit192", bit_251_196_192 - bit_251_196 * bit(192), every_1024),
        Constraint(H + "bit_unpacking/zeroes_between_ones192", bit_251_196 * (suffix(193) - suffix(196) * (1 << 3)), every_1024),
        Constraint(H + "bit_unpacking/cumulative_bit196", bit_251_196 - bit(251) * bit(196), every_1024),
        Constraint(H + "bit_unpacking/zeroes_between_ones196", bit(251) * (suffix(197) - suffix(251) * (1 << 54)), every_1024),
        Constraint(H + "booleanity_test", b0 * (b0 - one), transition),
        Constraint(H + "bit_extraction_end", suffix(), step_252),
        Constraint(H + "zeros_tail", suffix(), step_255),
        Constraint(H + "add_points/slope", b0 * (sum_y() - point_y) - slope() * (sum_x() - point_x), transition),
        Constraint(H + "add_points/x", slope() * slope() - b0 * (sum_x() + point_x + sum_x(1)), transition),
        Constraint(H + "add_points/y", b0 * (sum_y() + sum_y(1)) - slope() * (sum_x() - sum_x(1)), transition),
        Constraint(H + "copy_point/x", b0_negate * (sum_x(1) - sum_x()), transition),
        Constraint(H + "copy_point/y", b0_negate * (sum_y(1) - sum_y()), transition),
        Constraint("pedersen/hash0/copy_point/x", sum_x(256) - sum_x(255), every_2048),
        Constraint("pedersen/hash0/copy_point/y", sum_y(256) - sum_y(255), every_2048),
        Constraint("pedersen/hash0/init/x", sum_x() - px, every_2048),
        Constraint("pedersen/hash0/init/y", sum_y() - py, every_2048),
        Constraint("pedersen/input0_value0", npc_at(Npc.PEDERSEN_INPUT0_ADDR + 1) - suffix(), every_2048),
        Constraint("pedersen/input0_addr", npc_at(2048 + Npc.PEDERSEN_INPUT0_ADDR) - (npc_at(Npc.PEDERSEN_OUTPUT_ADDR) + 1), every_2048_except_last),
        Constraint("pedersen/init_addr", npc_at(Npc.PEDERSEN_INPUT0_ADDR) - hints.initial_pedersen_addr, FIRST_ROW),
        Constraint("pedersen/input1_value0", npc_at(Npc.PEDERSEN_INPUT1_ADDR + 1) - suffix(256), every_2048),
        Constraint("pedersen/input1_addr", npc_at(Npc.PEDERSEN_INPUT1_ADDR) - (npc_at(Npc.PEDERSEN_INPUT0_ADDR) + 1), every_2048),
        Constraint("pedersen/output_value0", npc_at(Npc.PEDERSEN_OUTPUT_ADDR + 1) - sum_x(511), every_2048),
        Constraint("pedersen/output_addr", npc_at(Npc.PEDERSEN_OUTPUT_ADDR) - (npc_at(Npc.PEDERSEN_INPUT1_ADDR) + 1), every_2048),
    ]


def periodic_value(table, row):
    """value of a periodic column at a trace row it is read at (rows = 0 mod 4): the Pedersen point of step row / 4"""
    if table in (TABLE_PEDERSEN_X, TABLE_PEDERSEN_Y):
        return pedersen_constant_points()[(row // 4) % 512][table]
    raise ValueError("unknown periodic column %d" % table)


def constraints(hints: Hints, challenges=None) -> List[Constraint]:
    """the 93 constraints in the reference's order (air.rs:1083-1180) - the order fixes which power of alpha each one gets
    in the composition; without challenges only the 36 + 36 constraints that do not involve a permutation argument"""
    if challenges is None:
        return cpu_constraints(hints) + pedersen_constraints(hints) + range_check_constraints(hints, [0] * 6)[6:] + bitwise_constraints(hints)
    rc = range_check_constraints(hints, challenges)
    return (cpu_constraints(hints) + memory_constraints(hints, challenges) + rc[:6] + diluted_check_constraints(hints, challenges)
            + pedersen_constraints(hints) + rc[6:] + bitwise_constraints(hints))


# ---- the Pedersen builtin (builtins/src/pedersen/{mod,constants}.rs) ----------------------------------------------
# P0 (the shift point) and the four base points, builtins/src/pedersen/constants.rs:5-30 (StarkWare's pedersen_params)
PEDERSEN_POINTS = (
    (2089986280348253421170679821480865132823066470938446095505822317253594081284, 1713931329540660377023406109199410414810705867260802078187082345529207694986),
    (996781205833008774514500082376783249102396023663454813447423147977397232763, 1668503676786377725805489344771023921079126552019160156920634619255970485781),
    (2251563274489750535117886426533222435294046428347329203627021249169616184184, 1798716007562728905295480679789526322175868328062420237419143593021674992973),
    (2138414695194151160943305727036575959195309218611738193261179310511854807447, 113410276730064486255102093846540133784865286929052426931474106396135072156),
    (2379962749567351885752724891227938183011949129833673362440656643086021394946, 776496453633298175483985398648758586525933812536653089401905292063708816422),
)
TABLE_PEDERSEN_X, TABLE_PEDERSEN_Y = 0, 1      # air_program.Table indices of the periodic columns


def _ec_double(pt):
    x, y = pt
    lam = (3 * x * x + 1) * pow(2 * y, -1, P) % P            # curve y^2 = x^3 + x + beta
    x3 = (lam * lam - 2 * x) % P
    return x3, (lam * (x - x3) - y) % P


def _ec_add(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if y1 != y2:
            raise ValueError("point at infinity in a Pedersen partial sum")
        return _ec_double(p1)
    lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


_PEDERSEN_CONSTANT_POINTS = None


def pedersen_constant_points():
    """the 512 points of the periodic columns: 2^i P1 (i < 248), 2^i P2 (i < 4), the last one repeated up to 256, then
    the same for P3, P4 (gen_element_steps' constant_points; builtins/src/pedersen/periodic.rs:1211-1250)"""
    global _PEDERSEN_CONSTANT_POINTS
    if _PEDERSEN_CONSTANT_POINTS is None:
        pts = []
        for lo, hi in ((PEDERSEN_POINTS[1], PEDERSEN_POINTS[2]), (PEDERSEN_POINTS[3], PEDERSEN_POINTS[4])):
            half, acc = [], lo
            for _ in range(248):
                half.append(acc)
                acc = _ec_double(acc)
            acc = hi
            for _ in range(4):
                half.append(acc)
                acc = _ec_double(acc)
            pts += half + [half[-1]] * 4
        _PEDERSEN_CONSTANT_POINTS = pts
    return _PEDERSEN_CONSTANT_POINTS


def pedersen_element_steps(x, start, which):
    """gen_element_steps (builtins/src/pedersen/mod.rs:121-163): 256 steps (partial sum BEFORE bit i is added, suffix
    x >> i, slope of the addition or 0) and the partial sum after them.  which: 0 for input a (P1, P2), 1 for b (P3, P4)"""
    consts = pedersen_constant_points()[256 * which: 256 * which + 256]
    point, steps = start, []
    for i in range(256):
        suffix = x >> i
        slope = 0
        nxt = point
        if suffix & 1:
            cx, cy = consts[i]
            if cx == point[0]:
                if cy != point[1]:
                    raise ValueError("point at infinity in a Pedersen partial sum")
                slope = (3 * cx * cx + 1) * pow(2 * cy, -1, P) % P           # calculate_slope's tangent case
            else:
                slope = (point[1] - cy) * pow(point[0] - cx, -1, P) % P      # calculate_slope(constant_point, partial_point)
            nxt = _ec_add(point, consts[i])
        steps.append((point, suffix % P, slope))
        point = nxt
    return steps, point


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


def _dilute(v, spacing=DILUTED_CHECK_SPACING):
    """bit i -> bit i * spacing (builtins/src/bitwise/mod.rs dilute)"""
    out, i = 0, 0
    while v:
        out |= (v & 1) << (i * spacing)
        v >>= 1
        i += 1
    return out


def _undilute(v, spacing=DILUTED_CHECK_SPACING, n_bits=DILUTED_CHECK_N_BITS):
    """DilutedCheckPool::push_diluted (layouts/src/utils.rs:255-271)"""
    out = 0
    for i in range(n_bits):
        out |= ((v >> (i * spacing)) & 1) << i
    if _dilute(out, spacing) != v:
        raise ValueError("value %#x is not in diluted form" % v)
    return out


def _partition64(v, spacing=DILUTED_CHECK_SPACING):
    """Partition64::new (builtins/src/bitwise/mod.rs): stream s = the bits at positions = s mod spacing, left in place"""
    segs = [0] * spacing
    for b in range(64 // spacing):
        for s_ in range(spacing):
            segs[s_] |= ((v >> (b * spacing + s_)) & 1) << (b * spacing)
    return segs


def _rc_ordered_with_padding(values):
    """RangeCheckPool::get_ordered_values_with_padding (layouts/src/utils.rs:357-380)"""
    ordered = sorted(values)
    padding = [v for a, b in zip(ordered, ordered[1:]) for v in range(a + 1, b)]
    return sorted(ordered + padding), padding


def base_trace(register_states, memory, public_input, private_input=None):
    """ExecutionTrace::new (trace.rs:95-660) for the components restated so far: the CPU cells (cpu_trace), the whole
    memory pool with the builtins' memory cells and the gap fillers, the range-check column and the sorted memory
    column.  private_input: {"pedersen": [(index, a, b)], "range_check": [(index, value)], "bitwise": [(index, x, y)]}
    (air-private-input.json); missing instances are the reference's empty dummies."""
    from .. import backend as be                     # host Pedersen (ss_pedersen_hash_host): no device involved
    from ..coin import canonical
    private_input = private_input or {}
    cols = cpu_trace(register_states, memory, public_input)
    n = len(cols[0])
    num_cycles = n // CYCLE_HEIGHT
    npc_col, rc_col = cols[COL_NPC], cols[COL_RANGE_CHECK]
    seg = public_input.memory_segments

    # ---- range-check pool (trace.rs:131-160, 236-284)
    pool = []
    for st in register_states:
        w = bn.Word(memory[st.pc])
        pool += [w.off_dst, w.off_op0, w.off_op1]
    rc128 = [(int(i), int(v)) for i, v in private_input.get("range_check", [])]
    parts_of = lambda v: [(v >> (16 * (RANGE_CHECK_BUILTIN_PARTS - 1 - k))) & 0xFFFF for k in range(RANGE_CHECK_BUILTIN_PARTS)]
    for _, v in rc128:
        pool += parts_of(v)
    ordered_vals, padding_vals = _rc_ordered_with_padding(pool)
    rc_min, rc_max = min(pool), max(pool)
    padding_iter, ordered_iter = iter(padding_vals), iter(ordered_vals)
    for index in range(len(rc128), num_cycles // RANGE_CHECK_BUILTIN_RATIO):        # dummies made of padding values
        value = 0
        for _ in range(RANGE_CHECK_BUILTIN_PARTS):
            value = (value << 16) + next(padding_iter, rc_max)
        rc128.append((index, value))
    for cycle in range(num_cycles):
        r = cycle * CYCLE_HEIGHT
        if cycle % 2 == 1:
            rc_col[r + RangeCheck.UNUSED] = next(padding_iter, rc_max)
        for o in range(0, CYCLE_HEIGHT, RANGE_CHECK_STEP):
            rc_col[r + o + RangeCheck.ORDERED] = next(ordered_iter, rc_max)
    if next(padding_iter, None) is not None or next(ordered_iter, None) is not None:
        raise ValueError("range-check values do not fit the trace")

    # ---- builtin memory cells (trace.rs:300-420, 540-570)
    ped = {int(i): (int(a), int(b)) for i, a, b in private_input.get("pedersen", [])}
    step = PEDERSEN_BUILTIN_RATIO * CYCLE_HEIGHT
    ped_begin = seg["pedersen"][0]
    aux_col = cols[COL_AUXILIARY]
    cache = {}
    for i in range(n // step):
        a, b = ped.get(i, (0, 0))
        if (a, b) not in cache:                      # pedersen::InstanceTrace::new (builtins/src/pedersen/mod.rs:81-118)
            a_steps, mid = pedersen_element_steps(a % P, PEDERSEN_POINTS[0], 0)
            b_steps, end = pedersen_element_steps(b % P, mid, 1)
            if canonical(be.pedersen_hash_host(be.felt(a), be.felt(b))) != b_steps[-1][0][0]:
                raise ValueError("Pedersen partial sums do not end at the hash")        # the reference's own assert
            cache[(a, b)] = (a_steps + b_steps, b_steps[-1][0][0])
        steps, out = cache[(a, b)]
        base, addr = i * step, ped_begin + 3 * i
        for j, (point, suffix, slope) in enumerate(steps):
            r = base + 4 * j
            rc_col[r + 1], rc_col[r + 3] = point
            aux_col[r], aux_col[r + 2] = suffix, slope
        for half, v in ((0, a), (1, b)):             # the flags that make the bit decomposition unique (trace.rs:383-392)
            b251, b196, b192 = (v >> 251) & 1, (v >> 196) & 1, (v >> 192) & 1
            aux_col[base + 1024 * half + 1022] = b251 & b196
            aux_col[base + 1024 * half + 7] = b251 & b196 & b192
        for off, (ad, val) in ((Npc.PEDERSEN_INPUT0_ADDR, (addr, a)), (Npc.PEDERSEN_INPUT1_ADDR, (addr + 1, b)),
                               (Npc.PEDERSEN_OUTPUT_ADDR, (addr + 2, out))):
            npc_col[base + off], npc_col[base + off + 1] = ad, val % P
    step = RANGE_CHECK_BUILTIN_RATIO * CYCLE_HEIGHT
    rc_begin = seg["range_check"][0]
    for block, (index, value) in enumerate(rc128):
        base = block * step
        for k, part in enumerate(parts_of(value)):
            rc_col[base + CYCLE_HEIGHT * k + RangeCheck.RC16_COMPONENT] = part
        npc_col[base + Npc.RANGE_CHECK128_ADDR], npc_col[base + Npc.RANGE_CHECK128_ADDR + 1] = rc_begin + index, value
    step = BITWISE_RATIO * CYCLE_HEIGHT
    bw = {int(i): (int(x), int(y)) for i, x, y in private_input.get("bitwise", [])}
    bw_begin = seg["bitwise"][0]
    for i in range(n // step):
        x, y = bw.get(i, (0, 0))
        base, addr = i * step, bw_begin + 5 * i
        for k, val in enumerate((x, y, x & y, x ^ y)):
            o = base + Npc.BITWISE_POOL_ADDR + k * (step // 4)
            npc_col[o], npc_col[o + 1] = addr + k, val
        npc_col[base + Npc.BITWISE_X_OR_Y_ADDR], npc_col[base + Npc.BITWISE_X_OR_Y_ADDR + 1] = addr + 4, x | y

    # ---- bitwise partitions and the diluted check (trace.rs:432-588)
    un_col, od_col = cols[COL_DILUTED_UNORDERED], cols[COL_DILUTED_ORDERED]
    mask64 = (1 << 64) - 1
    diluted_pool = []
    shifted_cells = (1, 65, 33, 97)                  # Bits16Chunk3Offset{0,1,2,3}ResShifted
    for i in range(n // step):
        x, y = bw.get(i, (0, 0))
        base = i * step
        parts = [[_partition64((v >> (64 * c)) & mask64) for c in range(4)] for v in (x, y, x & y, x ^ y)]
        for k in range(4):                           # shifts that make the unpacking unique (trace.rs:447-473)
            v = parts[2][3][k] + parts[3][3][k]
            sh = 8 if k == 3 else 4
            if (v << sh) >> sh != v or (v << sh) >= 1 << 64:
                raise ValueError("bitwise instance %d: top segment does not fit" % i)
            un_col[base + shifted_cells[k]] = v << sh
            diluted_pool.append(_undilute(v << sh))
        for pidx, part in enumerate(parts):          # x, y, x&y, x^y: 32 rows each
            for c in range(4):
                for st_ in range(4):
                    un_col[base + 32 * pidx + 8 * c + 2 * st_] = part[c][st_]
                    diluted_pool.append(_undilute(part[c][st_]))
    lo, hi = 0, (1 << DILUTED_CHECK_N_BITS) - 1
    ordered = sorted(diluted_pool)
    if ordered and (ordered[0] < lo or ordered[-1] > hi):
        raise ValueError("diluted value out of range")
    present = set(ordered)
    padding = [v for v in range(lo, hi + 1) if v not in present]    # get_ordered_values_with_padding (utils.rs:296-333)
    ordered = sorted(ordered + padding)
    pad_iter = iter(padding)
    done = False
    for blk in range(n // step):                     # padding goes to the free odd cells of column 1 (trace.rs:547-571)
        for off in range(1, step, 2):
            if off in shifted_cells:
                continue
            v = next(pad_iter, None)
            if v is None:
                done = True
                break
            un_col[blk * step + off] = _dilute(v)
        if done:
            break
    if next(pad_iter, None) is not None or len(ordered) > n:
        raise ValueError("diluted-check values do not fit the trace")
    od_col[n - len(ordered):] = [_dilute(v) for v in ordered]

    # ---- gap fillers (trace.rs:594-625): every address between two accessed ones gets an (address, 0) access
    accessed = sorted(set(npc_col[0::2]) | {a for a, _ in public_input.public_memory})
    gaps = [v for a, b in zip(accessed, accessed[1:]) for v in range(a + 1, b)]
    if len(gaps) > num_cycles:
        raise ValueError("more memory gaps than cycles to hold them")
    for cycle, addr in enumerate(gaps):
        r = cycle * CYCLE_HEIGHT
        npc_col[r + Npc.UNUSED_ADDR], npc_col[r + Npc.UNUSED_VAL] = addr, 0

    # ---- sorted memory (get_ordered_memory_accesses, layouts/src/utils.rs:112-152)
    cells = n // PUBLIC_MEMORY_STEP
    accesses = list(zip(npc_col[0::2], npc_col[1::2]))
    accesses += [public_input.public_memory_padding()] * (cells - len(public_input.public_memory)) + list(public_input.public_memory)
    accesses.sort(key=lambda e: e[0])
    zeros, ordered = accesses[:cells], accesses[cells:]
    if any(a != 0 for a, _ in zeros) or ordered[0][0] != 1:
        raise ValueError("the public-memory cells of the pool must be the only accesses of address 0")
    for (a0, v0), (a1, v1) in zip(ordered, ordered[1:]):
        if not ((a0, v0) == (a1, v1) or a0 + 1 == a1):
            raise ValueError("memory is not continuous and single-valued at address %d" % a0)
    mem_col = cols[COL_MEMORY]
    mem_col[0::2] = [a for a, _ in ordered]
    mem_col[1::2] = [v for _, v in ordered]
    return cols


def failing_rows(constraint: Constraint, cols, rows=None, limit=5):
    """rows of the constraint's domain (or of `rows`) where its numerator does not vanish on the trace"""
    n = len(cols[0])
    bad = []
    for r in (constraint.domain.rows(n) if rows is None else rows):
        v = ap.evaluate(constraint.numerator, P, None, lambda c, o: cols[c][(r + o) % n], lambda t: periodic_value(t, r))
        if v:
            bad.append(r)
            if len(bad) >= limit:
                break
    return bad


# ---- the composition constraint and its tables (composition_constraint, air.rs:1183-1199) ---------------------------
class Tables:
    """The wave-uniform data the composition reads besides the trace: the two Pedersen periodic columns and the zerofier
    multipliers.  On the LDE coset x_i = offset * w_N^i a power X^(n/k) has period blowup * k in i, so every multiplier
    prod(X^p - c) / prod(X^p - c') with p > 1 is a short periodic table; a denominator X - c (first / last rows) is a
    full-length table of inverses (ss_inverse_table); a numerator X - c is evaluated in the program."""

    def __init__(self, n, blowup=2, offset=3):
        self.n, self.N, self.offset = n, n * blowup, offset
        self.g = pow(3, (P - 1) // n, P)
        self.w = pow(3, (P - 1) // self.N, P)
        self.specs = [("pedersen", TABLE_PEDERSEN_X), ("pedersen", TABLE_PEDERSEN_Y)]
        self._index = {}

    def _table(self, spec):
        if spec not in self._index:
            self._index[spec] = len(self.specs)
            self.specs.append(spec)
        return ap.Table(self._index[spec])

    def multiplier(self, domain: Domain):
        num, den = domain.num(self.n), domain.den(self.n)
        periodic = (tuple(f for f in num if f[0] > 1), tuple(f for f in den if f[0] > 1))
        expr = self._table(("periodic",) + periodic) if periodic != ((), ()) else None
        for p_, e in num:
            if p_ == 1:
                lin = ap.X - pow(self.g, e, P)
                expr = lin if expr is None else expr * lin
        for p_, e in den:
            if p_ == 1:
                t = self._table(("inverse", e))
                expr = t if expr is None else expr * t
        return expr

    # -- values
    def length(self, spec):
        if spec[0] == "pedersen":
            return 2048 * (self.N // self.n)
        if spec[0] == "inverse":
            return self.N
        return max(self.N // p_ for fs in spec[1:] for p_, _ in fs)

    def value_at(self, spec, x):
        """the table's underlying function at an arbitrary point (the verifier needs it at z)"""
        if spec[0] == "pedersen":
            return _poly_eval(_pedersen_coefficients(spec[1]), pow(x, self.n // 2048, P))
        if spec[0] == "inverse":
            return pow(x - pow(self.g, spec[1], P), -1, P)
        val, d = 1, 1
        for p_, e in spec[1]:
            val = val * (pow(x, p_, P) - pow(self.g, e, P)) % P
        for p_, e in spec[2]:
            d = d * (pow(x, p_, P) - pow(self.g, e, P)) % P
        return val * pow(d, -1, P) % P

    def host_values(self, spec):
        """the table over the LDE coset as python ints (tests, small traces; the device builds the long ones itself)"""
        length = self.length(spec)
        if spec[0] == "pedersen":
            coeffs = _pedersen_coefficients(spec[1])
            base, step = pow(self.offset, self.n // 2048, P), pow(self.w, self.n // 2048, P)
            out, x = [], base
            for _ in range(length):
                out.append(_poly_eval(coeffs, x))
                x = x * step % P
            return out
        if spec[0] == "inverse":
            c = pow(self.g, spec[1], P)
            vals, x = [], self.offset
            for _ in range(length):
                vals.append((x - c) % P)
                x = x * self.w % P
            return _batch_inverse(vals)
        # prod(X^p - c) / prod(X^p - c') along x_i = offset * w^i: every power advances by its own step, one batch inversion
        def running(factors):
            state = [[pow(self.offset, p_, P), pow(self.w, p_, P), pow(self.g, e, P)] for p_, e in factors]
            vals = []
            for _ in range(length):
                v = 1
                for f in state:
                    v = v * (f[0] - f[2]) % P
                    f[0] = f[0] * f[1] % P
                vals.append(v)
            return vals
        nums, dens = running(spec[1]), _batch_inverse(running(spec[2]))
        return [a * b % P for a, b in zip(nums, dens)]


def _poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


_PEDERSEN_COEFFS = {}


def _pedersen_coefficients(which):
    """coefficients of the degree < 512 interpolant with value points[j][which] at w_512^j (the reference stores them as
    HASH_POINTS_{X,Y}_COEFFS; its periodic_*_evals_match tests pin exactly this relation)"""
    if which not in _PEDERSEN_COEFFS:
        vals = [pt[which] for pt in pedersen_constant_points()]
        m = len(vals)
        w_inv = pow(pow(3, (P - 1) // m, P), -1, P)
        m_inv = pow(m, -1, P)
        # radix-2 inverse transform in plain python (512 points)
        a = list(vals)
        rev = [int(format(i, "0%db" % (m.bit_length() - 1))[::-1], 2) for i in range(m)]
        a = [a[rev[i]] for i in range(m)]
        length = 2
        while length <= m:
            wl = pow(w_inv, m // length, P)
            for start in range(0, m, length):
                wcur = 1
                for k in range(length // 2):
                    u, v = a[start + k], a[start + k + length // 2] * wcur % P
                    a[start + k], a[start + k + length // 2] = (u + v) % P, (u - v) % P
                    wcur = wcur * wl % P
            length *= 2
        _PEDERSEN_COEFFS[which] = [v * m_inv % P for v in a]
    return _PEDERSEN_COEFFS[which]


def _batch_inverse(vals):
    prefix, run = [], 1
    for v in vals:
        prefix.append(run)
        run = run * v % P
    inv = pow(run, -1, P)
    out = [0] * len(vals)
    for i in range(len(vals) - 1, -1, -1):
        out[i] = inv * prefix[i] % P
        inv = inv * vals[i] % P
    return out


def composition(n, hints: Hints, challenges, alpha, tables: Tables, constraint_list=None):
    """sum_i alpha^i * constraint_i (air.rs:1183-1199), constraint_i = numerator_i * multiplier(domain_i); constraints that
    share a domain are summed before the one multiplication by its multiplier.  constraint_list: another layout's
    constraints (layouts/starknet.py)"""
    groups, order, apow = {}, [], 1
    for c in (constraints(hints, challenges) if constraint_list is None else constraint_list):
        term = c.numerator * ap.Const(apow) if apow != 1 else c.numerator
        key = c.domain.name                      # a domain's name identifies it
        if key not in groups:
            groups[key] = [c.domain, term]
            order.append(key)
        else:
            groups[key][1] = groups[key][1] + term
        apow = apow * alpha % P
    total = None
    for key in order:
        domain, partial = groups[key]
        term = partial * tables.multiplier(domain)
        total = term if total is None else total + term
    return total


def mask(hints=None, constraint_list=None):
    """trace_arguments(): the sorted (column, row offset) cells the constraints read - the order of the OOD vector"""
    cells, seen = set(), set()

    def walk(e):
        if e._id in seen:
            return
        seen.add(e._id)
        if e.kind == "trace":
            cells.add(tuple(e.args))
        elif e.kind in ("add", "sub", "mul", "inv"):
            for a in e.args:
                walk(a)
    h = hints or Hints(0, 0, 0, 0)
    for c in (constraints(h, [2, 3, 5, 7, 11, 13]) if constraint_list is None else constraint_list):
        walk(c.numerator)
    return sorted(cells)


# ---- the layout as the prover and the verifier see it ------------------------------------------------------------------
_TABLES = {}


def tables_for(n, blowup=2, offset=3, L=None):
    """the table registry of trace length n with every table the composition refers to registered (their set and order
    do not depend on the challenges).  L: the layout module (this one by default, layouts/starknet.py passes itself)"""
    import sys
    L = L or sys.modules[__name__]
    key = (L.__name__, n, blowup, offset)
    if key not in _TABLES:
        t = L.Tables(n, blowup, offset)
        L.composition(n, L.Hints(0, 0, 0, 0), [2, 3, 5, 7, 11, 13], 17, t)
        _TABLES[key] = t
    return _TABLES[key]


def make_air(ctx, public_input, n, log_blowup=1, lde_offset=3, L=None):
    """-> prover.Air for this public input and trace length.  The tables are built once: the periodic ones on the host
    (up to 2^16 entries each), the full-length inverse tables on the device (ss_inverse_table)."""
    import sys
    from .. import backend as be
    from ..coin import canonical
    from ..prover import Air
    L = L or sys.modules[__name__]
    tables = tables_for(n, 1 << log_blowup, lde_offset, L)
    lengths = [tables.length(s) for s in tables.specs]
    desc, off = [], 0
    for ln in lengths:
        desc += [off, ln.bit_length() - 1]
        off += ln
    buf = ctx.alloc(32 * off)
    g = be.felt(lde_offset)
    for spec, ln, start in zip(tables.specs, lengths, desc[0::2]):
        view = be.DeviceView(buf, 32 * start, 32 * ln)
        if spec[0] == "inverse":
            ctx.inverse_table((n << log_blowup).bit_length() - 1, g, be.felt(pow(tables.g, spec[1], P)), view)
        else:
            import numpy as np
            host = np.stack([be.felt(v) for v in tables.host_values(spec)])
            be.check(ctx.lib.ss_upload(ctx.handle, view.ptr, host.ctypes.data, host.nbytes))

    def build_program(n_, challenges, comp_coeff):
        if n_ != n:
            raise ValueError("this Air was built for trace length %d" % n)
        ch = [canonical(c) for c in challenges]
        hints = L.Hints.from_public_input(public_input, ch, n)
        expr = L.composition(n, hints, ch, canonical(comp_coeff), tables)
        return ap.lower(expr, P), buf, desc

    air = Air(L.__name__.rsplit(".", 1)[-1], L.NUM_BASE_COLUMNS, L.NUM_EXTENSION_COLUMNS, 6, L.mask(), build_program)
    air.table_buffer = buf
    return air


def verifier_air(public_input, log_blowup=1, lde_offset=3, L=None):
    """-> verifier.VerifierAir: the same composition, with the tables' underlying functions evaluated at the
    out-of-domain point"""
    import sys
    from ..verifier import VerifierAir
    L = L or sys.modules[__name__]

    def comp(n, challenges, alpha):
        return L.composition(n, L.Hints.from_public_input(public_input, challenges, n), challenges, alpha, tables_for(n, 1 << log_blowup, lde_offset, L))

    def table_at(n, x, t):
        tables = tables_for(n, 1 << log_blowup, lde_offset, L)
        return tables.value_at(tables.specs[t], x)
    return VerifierAir(L.NUM_BASE_COLUMNS, L.NUM_EXTENSION_COLUMNS, 6, L.mask(), comp, table_at)


def trace_columns(ctx, base_cols_device, n):
    """extension.TraceColumns of a base trace resident in HBM (columns 3, 4, 5, 1, 2: trace.rs:652-660)"""
    from ..extension import TraceColumns
    return TraceColumns(npc=base_cols_device[COL_NPC], memory=base_cols_device[COL_MEMORY], range_check=base_cols_device[COL_RANGE_CHECK],
                        trace_len=n, diluted_unordered=base_cols_device[COL_DILUTED_UNORDERED],
                        diluted_ordered=base_cols_device[COL_DILUTED_ORDERED])


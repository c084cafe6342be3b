"""The `starknet` layout (layouts/src/starknet/{mod,air,trace}.rs; builtins/src/{pedersen,range_check,ecdsa,bitwise,
ec_op,poseidon}): the AIR's 195 constraints as air_program expressions, in the reference's order (air.rs:2188-2386),
and the base-trace generation they are checked against.

Column map (air.rs:2539-3242): 0 flags | 1, 2 Pedersen partial sum x, y | 3 Pedersen suffix | 4 Pedersen slope |
5 memory pool ("npc") | 6 sorted memory | 7 range check / diluted check / bitwise / Poseidon partial rounds |
8 auxiliary / ECDSA / EC op / Poseidon full rounds | extension: 9 the four permutation and aggregation products.

Everything the CPU, memory, range-check, diluted-check, Pedersen and bitwise components do is the recursive layout's
(layouts/recursive.py) over other cells and periods; the ECDSA, EC-op and Poseidon builtins are new here.  The Poseidon
round keys are StarkWare's Hades constants (sha256("Hades<i>") mod p); the keys of the AIR's optimised partial rounds
and of the margins between full and partial rounds are DERIVED from them (poseidon_air_keys) - the tests pin the derived
values to the literals of air.rs:2052-2160 and to the reference's periodic-column polynomials."""
from dataclasses import dataclass
from hashlib import sha256
from typing import List

from .. import air_program as ap
from .. import binary as bn
from . import recursive as rec
from .recursive import (Constraint, Domain, P, ALL_CYCLES, ALL_CYCLES_EXCEPT_LAST, FLAG_ROWS, FLAG_ZERO_ROWS, FIRST_ROW, LAST_CYCLE,  # noqa: F401
                        EVERY_2ND_EXCEPT_LAST, SECOND_LAST_ROW, EVERY_4TH_EXCEPT_LAST, FOURTH_LAST_ROW, PEDERSEN_POINTS,
                        _every, _every_except_last, _row_from_end, _ec_add, _ec_double)

CYCLE_HEIGHT = 16                   # starknet/mod.rs:14-48
PUBLIC_MEMORY_STEP, MEMORY_STEP, RANGE_CHECK_STEP, DILUTED_CHECK_STEP = 8, 2, 4, 8
PEDERSEN_BUILTIN_RATIO, RANGE_CHECK_BUILTIN_RATIO, RANGE_CHECK_BUILTIN_PARTS, BITWISE_RATIO = 32, 16, 8, 64
ECDSA_BUILTIN_RATIO, EC_OP_BUILTIN_RATIO, EC_OP_SCALAR_HEIGHT, POSEIDON_RATIO = 2048, 1024, 256, 32
POSEIDON_ROUNDS_FULL, POSEIDON_ROUNDS_PARTIAL = 8, 83
NUM_BASE_COLUMNS, NUM_EXTENSION_COLUMNS = 9, 1
DILUTED_CHECK_N_BITS, DILUTED_CHECK_SPACING = 16, 4
MEM_Z, MEM_A, RC_Z, DC_Z, AGG_Z, AGG_A = range(6)                  # challenge indices (air.rs:3274-3322)
(COL_FLAGS, COL_PEDERSEN_X, COL_PEDERSEN_Y, COL_PEDERSEN_SUFFIX, COL_PEDERSEN_SLOPE, COL_NPC, COL_MEMORY, COL_RANGE_CHECK,
 COL_AUXILIARY, COL_PERMUTATION) = range(10)
# the curve y^2 = x^3 + x + beta, its generator and its order (builtins/src/utils.rs:134-158)
CURVE_BETA = 3141592653589793238462643383279502884197169399375105820974944592307816406665
CURVE_ORDER = 3618502788666131213697322783095070105526743751716087489154079457884512865583
GENERATOR = (874739451078007766457464989774322083649278607533249481151382481072868806602,
             152666792071518830868575557812948353041420400780739481342941381225525861407)
SHIFT_POINT = PEDERSEN_POINTS[0]                                   # ecdsa::SHIFT_POINT = pedersen P0 (ecdsa/mod.rs:21)


# ---- virtual columns (air.rs:2482-3242) ---------------------------------------------------------------------------
class Npc:
    PC, INSTRUCTION, PUB_MEM_ADDR, PUB_MEM_VAL, MEM_OP0_ADDR, MEM_OP0 = 0, 1, 2, 3, 4, 5
    MEM_DST_ADDR, MEM_DST, MEM_OP1_ADDR, MEM_OP1, UNUSED_ADDR, UNUSED_VAL = 8, 9, 12, 13, 14, 15
    # builtin cells: the offset inside the builtin's own period (the value cell is the next row)
    PEDERSEN_INPUT0_ADDR, PEDERSEN_INPUT1_ADDR, PEDERSEN_OUTPUT_ADDR = 6, 262, 134
    RANGE_CHECK128_ADDR, ECDSA_PUBKEY_ADDR, ECDSA_MESSAGE_ADDR = 70, 390, 16774
    BITWISE_POOL_ADDR, BITWISE_X_OR_Y_ADDR = 198, 902
    EC_OP_P_X_ADDR, EC_OP_P_Y_ADDR, EC_OP_Q_X_ADDR, EC_OP_Q_Y_ADDR, EC_OP_M_ADDR, EC_OP_R_X_ADDR, EC_OP_R_Y_ADDR = \
        8582, 4486, 12678, 2438, 10630, 6534, 14726
    POSEIDON_ADDRS = (38, 102, 166, 230, 294, 358)                  # input 0, 1, 2, output 0, 1, 2


class Mem:
    ADDRESS, VALUE = 0, 1


class RangeCheck:
    OFF_DST, ORDERED, OFF_OP1, OFF_OP0, UNUSED = 0, 2, 4, 8, 12
    RC16_COMPONENT = 12             # RangeCheckBuiltin::Rc16Component: one 16-bit part every 32 rows


class Auxiliary:
    AP, TMP0, OP0_MUL_OP1, FP, TMP1, RES = 0, 2, 4, 8, 10, 12


class DilutedCheck:                 # column 7, step 8; the aggregate lives in the permutation column
    UNORDERED, ORDERED, AGGREGATE = 1, 5, 3


class Permutation:                  # column 9: memory every 2 rows, range check at 4k + 1, diluted check at 8k + 7
    MEMORY, RANGE_CHECK, DILUTED_CHECK = 0, 1, 7


class Ecdsa:                        # column 8 (air.rs:2691-2783)
    PUBKEY_DOUBLING_X, PUBKEY_DOUBLING_Y, PUBKEY_DOUBLING_SLOPE = 1, 33, 35                 # step 64
    PUBKEY_PARTIAL_SUM_X, PUBKEY_PARTIAL_SUM_Y, PUBKEY_PARTIAL_SUM_X_DIFF_INV, PUBKEY_PARTIAL_SUM_SLOPE, R_SUFFIX = 17, 49, 51, 19, 9
    MESSAGE_SUFFIX, GENERATOR_PARTIAL_SUM_Y, GENERATOR_PARTIAL_SUM_X = 59, 91, 27           # step 128
    GENERATOR_PARTIAL_SUM_X_DIFF_INV, GENERATOR_PARTIAL_SUM_SLOPE = 7, 123
    R_POINT_SLOPE, R_POINT_X_DIFF_INV, R_INV, W_INV, MESSAGE_INV = 16331, 32715, 16355, 32739, 16363    # one per instance
    PUBKEY_X_SQUARED, B_SLOPE, B_X_DIFF_INV = 32747, 32763, 32647


class EcOp:                         # column 8, step 64 (air.rs:2636-2689)
    Q_DOUBLING_X, Q_DOUBLING_Y, Q_DOUBLING_SLOPE = 41, 25, 57
    R_PARTIAL_SUM_X, R_PARTIAL_SUM_Y, R_PARTIAL_SUM_SLOPE, R_PARTIAL_SUM_X_DIFF_INV, M_SUFFIX = 5, 37, 11, 43, 21
    M_BIT251_AND_BIT196_AND_BIT192, M_BIT251_AND_BIT196 = 16371, 16339


class Poseidon:                     # (column, shift, step) (air.rs:2576-2634)
    FULL_STATE = ((8, 53, 64), (8, 13, 64), (8, 45, 64))
    FULL_STATE_SQUARED = ((8, 29, 64), (8, 61, 64), (8, 3, 64))
    PARTIAL_STATE = ((7, 3, 8), (8, 6, 16))
    PARTIAL_STATE_SQUARED = ((7, 7, 8), (8, 14, 16))


class Bitwise:                      # column 7
    SHIFTED_CELLS = (9, 521, 265, 777)          # Bits16Chunk3Offset{0,1,2,3}ResShifted

    @staticmethod
    def cell(chunk, stream):        # Bits16Chunk<c>Offset<s>: 1, 17, 33, ... inside a 256-row partition
        return 16 * (4 * chunk + stream) + 1


def flag(f, cycle_offset=0):
    o = CYCLE_HEIGHT * cycle_offset + f
    return ap.Trace(COL_FLAGS, o) - (ap.Trace(COL_FLAGS, o + 1) + ap.Trace(COL_FLAGS, o + 1))


def npc(cell, cycle_offset=0):
    return ap.Trace(COL_NPC, CYCLE_HEIGHT * cycle_offset + cell)


def rc(cell, cycle_offset=0):
    return ap.Trace(COL_RANGE_CHECK, CYCLE_HEIGHT * cycle_offset + cell)


def aux(cell, cycle_offset=0):
    return ap.Trace(COL_AUXILIARY, CYCLE_HEIGHT * cycle_offset + cell)


def npc_at(offset):
    return ap.Trace(COL_NPC, offset)


def c7(offset):
    return ap.Trace(COL_RANGE_CHECK, offset)


def c8(offset):
    return ap.Trace(COL_AUXILIARY, offset)


# ---- domains --------------------------------------------------------------------------------------------------------
def F(d, k=0, m=1):
    """the factor X^(n/d) - g^(k n / m)"""
    return lambda n: (n // d, k * n // m)


def _domain(name, num=(), den=()):
    """rows: where the denominator vanishes and the numerator does not"""
    def rows(n):
        numf, denf = [f(n) for f in num], [f(n) for f in den]

        def roots(p, e):                   # p r = e (mod n), p | n
            return range(0) if e % p else range((e // p) % (n // p), n, n // p)
        cands = roots(*denf[0]) if len(denf) == 1 else sorted(set(r for f in denf for r in roots(*f)))
        return (r for r in cands if not any((p * r - e) % n == 0 for p, e in numf))
    return Domain(name, rows, lambda n: [f(n) for f in num], lambda n: [f(n) for f in den])


EVERY_8, EVERY_8_EXCEPT_LAST, EIGHTH_LAST_ROW = _every(8, "every 8th row"), _every_except_last(8, "every 8th row but the last"), _row_from_end(8, "row n-8")
EVERY_64, EVERY_256, EVERY_512 = _every(64, "every 64th row"), _every(256, "every 256th row"), _every(512, "every 512th row")
EVERY_256_EXCEPT_LAST, EVERY_512_EXCEPT_LAST = _every_except_last(256, "every 256th row but the last"), _every_except_last(512, "every 512th row but the last")
ALL_BITWISE, ALL_BITWISE_EXCEPT_LAST = _every(1024, "every 1024th row"), _every_except_last(1024, "every 1024th row but the last")
ALL_EC_OP, ALL_EC_OP_EXCEPT_LAST = _every(16384, "every 16384th row"), _every_except_last(16384, "every 16384th row but the last")
ALL_ECDSA, ALL_ECDSA_EXCEPT_LAST = _every(32768, "every 32768th row"), _every_except_last(32768, "every 32768th row but the last")
# air.rs:862-864: (X^(n/256) - g^(255n/256)) / (X^n - 1)
PEDERSEN_TRANSITION = _domain("steps 0..254 of every Pedersen input", [F(256, 255, 256)], [F(1)])
PEDERSEN_STEP_252 = _domain("step 252 of every Pedersen input", [], [F(256, 63, 64)])
PEDERSEN_STEP_255 = _domain("step 255 of every Pedersen input", [], [F(256, 255, 256)])
# air.rs:947-949: (X^(n/512) - g^(n/2)) / (X^(n/256) - 1): the 256-row steps that start a hash
PEDERSEN_HASH_START = _domain("every 512th row, as every 256th but the odd ones", [F(512, 1, 2)], [F(256)])
EC_OP_TRANSITION = _domain("steps 0..254 of every 16384 rows", [F(16384, 255, 256)], [F(64)])            # air.rs:1045-1047
ECDSA_TRANSITION = _domain("steps 0..254 of every 32768 rows", [F(32768, 255, 256)], [F(128)])           # air.rs:1075-1077
ECDSA_STEP_251, ECDSA_STEP_255 = _domain("step 251 of every 32768 rows", [], [F(32768, 251, 256)]), _domain("step 255 of every 32768 rows", [], [F(32768, 255, 256)])
EC_OP_STEP_251 = _domain("step 251 of every 16384 rows", [], [F(16384, 251, 256)])
EC_OP_STEP_252 = _domain("step 252 of every 16384 rows", [], [F(16384, 63, 64)])
EC_OP_STEP_255 = _domain("step 255 of every 16384 rows", [], [F(16384, 255, 256)])
BITWISE_TRANSITION = _domain("rows 0, 256, 512 of every 1024", [F(1024, 3, 4)], [F(256)])                 # air.rs:1512-1514
EVERY_16_BIT_SEGMENT = _domain("rows 0, 16, ..., 240 of every 1024", [], [F(1024, k, 64) for k in range(1, 16)] + [F(1024)])    # air.rs:1546-1580
# Poseidon (air.rs:1869-1905): factors over X^(n/512), i.e. over the 64-row (or 16-, 8-row) steps of one 512-row instance
_D14 = [F(512, 3, 4), F(512, 7, 8)]
_D15 = [F(512, 5, 8)] + _D14
_D16 = [F(512, 31, 32)]
_D17 = [F(512, 11, 16), F(512, 23, 32), F(512, 25, 32), F(512, 13, 16), F(512, 27, 32), F(512, 29, 32), F(512, 15, 16)] + _D16
_D19 = [F(512, 61, 64), F(512, 63, 64)] + _D16
_D20 = [F(512, 19, 32), F(512, 21, 32)] + _D15 + _D17
POSEIDON_ADDR_STEP = _domain("64-row steps 0..4 of every Poseidon instance", _D15, [F(64)])
POSEIDON_PARTIAL1_SQUARING = _domain("16-row steps 0..21 of every Poseidon instance", _D14 + _D17, [F(16)])
POSEIDON_HALF_FULL_ROUND_TRANSITION = _domain("64-row steps 0, 1, 2 of every 256", [F(256, 3, 4)], [F(64)])
POSEIDON_PARTIAL_ROUND0 = _domain("8-row steps 0..60 of every Poseidon instance", _D19, [F(8)])
POSEIDON_PARTIAL_ROUND1 = _domain("16-row steps 0..18 of every Poseidon instance", _D20, [F(16)])


@dataclass
class Hints:
    """PublicInputHint (air.rs:3244-3272)"""
    initial_ap: int
    initial_pc: int
    final_ap: int
    final_pc: int
    range_check_min: int = 0
    range_check_max: int = 0
    memory_quotient: int = 0
    range_check_product: int = 1
    diluted_check_product: int = 1
    diluted_check_first: int = 0
    diluted_check_cumulative_value: int = 0
    initial_pedersen_addr: int = 0
    initial_rc_addr: int = 0
    initial_ecdsa_addr: int = 0
    initial_bitwise_addr: int = 0
    initial_ec_op_addr: int = 0
    initial_poseidon_addr: int = 0

    @classmethod
    def from_public_input(cls, pi, challenges=None, trace_len=None):
        """gen_hints (air.rs:2408-2479)"""
        seg = pi.memory_segments
        h = cls(initial_ap=seg["execution"][0], initial_pc=seg["program"][0], final_ap=seg["execution"][1], final_pc=seg["program"][1],
                range_check_min=pi.rc_min, range_check_max=pi.rc_max, initial_pedersen_addr=seg["pedersen"][0],
                initial_rc_addr=seg["range_check"][0], initial_ecdsa_addr=seg["ecdsa"][0], initial_bitwise_addr=seg["bitwise"][0],
                initial_ec_op_addr=seg["ec_op"][0], initial_poseidon_addr=seg["poseidon"][0])
        if challenges is not None:
            h.memory_quotient = rec.public_memory_quotient(challenges[MEM_Z], challenges[MEM_A], trace_len or 16 * pi.n_steps, pi, PUBLIC_MEMORY_STEP)
            h.diluted_check_cumulative_value = rec.diluted_cumulative_value(challenges[AGG_Z], challenges[AGG_A])
        return h


# ---- periodic columns -------------------------------------------------------------------------------------------------
(TABLE_PEDERSEN_X, TABLE_PEDERSEN_Y, TABLE_ECDSA_GENERATOR_X, TABLE_ECDSA_GENERATOR_Y, TABLE_POSEIDON_FULL_KEY0, TABLE_POSEIDON_FULL_KEY1,
 TABLE_POSEIDON_FULL_KEY2, TABLE_POSEIDON_PARTIAL_KEY0, TABLE_POSEIDON_PARTIAL_KEY1) = range(9)
NUM_PERIODIC = 9


def poseidon_round_keys():
    """StarkWare's Hades round constants (builtins/src/poseidon/params.rs ROUND_KEYS): sha256("Hades" + index) mod p"""
    return [[int(sha256(("Hades%d" % (3 * i + j)).encode()).hexdigest(), 16) % P for j in range(3)] for i in range(POSEIDON_ROUNDS_FULL + POSEIDON_ROUNDS_PARTIAL)]


def _mds(s):                                       # params.rs MDS_MATRIX: [[3, 1, 1], [1, -1, 1], [1, 1, -2]]
    return [(3 * s[0] + s[1] + s[2]) % P, (s[0] - s[1] + s[2]) % P, (s[0] + s[1] - 2 * s[2]) % P]


_POSEIDON_KEYS = None


def poseidon_states(inp, rk=None):
    """the s-box inputs of the permutation (poseidon/mod.rs:45-98, permute 121-146): the 8 full-round states after the round
    keys are added, the 83 partial-round values of the third element after its key is added, and the output"""
    rk = rk or poseidon_round_keys()
    st, full, partial, r = [v % P for v in inp], [], [], 0
    for phase in range(3):
        for _ in range(POSEIDON_ROUNDS_PARTIAL if phase == 1 else POSEIDON_ROUNDS_FULL // 2):
            st = [(a + k) % P for a, k in zip(st, rk[r])]
            if phase == 1:
                partial.append(st[2])
                st[2] = pow(st[2], 3, P)
            else:
                full.append(tuple(st))
                st = [pow(a, 3, P) for a in st]
            st = _mds(st)
            r += 1
    return full, partial, st


def poseidon_air_keys():
    """The constants of the AIR's Poseidon constraints, derived from the round keys.  The AIR never materialises the first
    two state elements of a partial round: it writes the next s-box input of the third element as a fixed linear
    combination of the previous three inputs and their cubes plus a key (air.rs:2084-2118); being an identity of the
    permutation, the key is what is left when the combination is subtracted on any input (0, 0, 0 here; the tests
    repeat it on another input).  -> dict: partial (80 keys: s[k+3] from s[k..k+2]), margin_full_to_partial (3),
    margin_partial_to_full (3), full (periodic values: 8 per state element)."""
    global _POSEIDON_KEYS
    if _POSEIDON_KEYS is None:
        rk = poseidon_round_keys()
        full, s, _ = poseidon_states((0, 0, 0), rk)
        c = [pow(v, 3, P) for v in s]
        f3 = [pow(v, 3, P) for v in full[3]]
        keys = {"partial": [(s[k + 3] - (8 * c[k] + 4 * s[k + 1] + 6 * c[k + 1] + 2 * s[k + 2] - 2 * c[k + 2])) % P for k in range(80)]}
        keys["margin_full_to_partial"] = [
            rk[4][2],                                                                       # PARTIAL_ROUND_KEYS[0][2] (air.rs:2047)
            (s[1] - (-4 * f3[1] + 10 * f3[2] + 4 * s[0] - 2 * c[0])) % P,
            (s[2] - (8 * f3[2] + 4 * s[0] + 6 * c[0] + 2 * s[1] - 2 * c[1])) % P]
        keys["margin_partial_to_full"] = [
            (full[4][0] - (16 * c[80] + 8 * s[81] + 16 * c[81] + 6 * s[82] + c[82])) % P,
            (full[4][1] - (4 * c[81] + 2 * s[82] + c[82])) % P,
            (full[4][2] - (8 * c[80] + 4 * s[81] + 6 * c[81] + 2 * s[82] - 2 * c[82])) % P]
        # the key added on the way to full round i + 1 sits at step i; nothing at the last step of each half
        keys["full"] = [[rk[1][j], rk[2][j], rk[3][j], 0, rk[88][j], rk[89][j], rk[90][j], 0] for j in range(3)]
        _POSEIDON_KEYS = keys
    return _POSEIDON_KEYS


_ECDSA_GENERATOR_POINTS = None


def ecdsa_generator_points():
    """the 256 points of the ECDSA periodic columns: 2^i G for i <= 250, the last one repeated (what the reference's
    GENERATOR_POINTS_{X,Y}_COEFFS evaluate to; gen_ec_mad_steps::<250>, ecdsa/mod.rs:103, 157-190)"""
    global _ECDSA_GENERATOR_POINTS
    if _ECDSA_GENERATOR_POINTS is None:
        pts, acc = [], GENERATOR
        for i in range(256):
            pts.append(acc)
            if i < 250:
                acc = _ec_double(acc)
        _ECDSA_GENERATOR_POINTS = pts
    return _ECDSA_GENERATOR_POINTS


def periodic_columns():
    """table index -> (values over one period, the period in trace rows); the column is the polynomial of degree < len(values)
    in X^(n / period) that takes values[j] at w^j (air.rs:47-104)"""
    k = poseidon_air_keys()
    ped, gen = rec.pedersen_constant_points(), ecdsa_generator_points()
    return {
        TABLE_PEDERSEN_X: ([p[0] for p in ped], PEDERSEN_BUILTIN_RATIO * CYCLE_HEIGHT),
        TABLE_PEDERSEN_Y: ([p[1] for p in ped], PEDERSEN_BUILTIN_RATIO * CYCLE_HEIGHT),
        TABLE_ECDSA_GENERATOR_X: ([p[0] for p in gen], ECDSA_BUILTIN_RATIO * CYCLE_HEIGHT),
        TABLE_ECDSA_GENERATOR_Y: ([p[1] for p in gen], ECDSA_BUILTIN_RATIO * CYCLE_HEIGHT),
        TABLE_POSEIDON_FULL_KEY0: (k["full"][0], POSEIDON_RATIO * CYCLE_HEIGHT),
        TABLE_POSEIDON_FULL_KEY1: (k["full"][1], POSEIDON_RATIO * CYCLE_HEIGHT),
        TABLE_POSEIDON_FULL_KEY2: (k["full"][2], POSEIDON_RATIO * CYCLE_HEIGHT),
        TABLE_POSEIDON_PARTIAL_KEY0: (k["partial"][:61] + [0, 0, 0], POSEIDON_RATIO * CYCLE_HEIGHT),
        TABLE_POSEIDON_PARTIAL_KEY1: (k["partial"][61:80] + [0] * 13, POSEIDON_RATIO * CYCLE_HEIGHT),
    }


def periodic_value(table, row):
    """value of a periodic column at trace row `row`"""
    values, period = periodic_columns()[table]
    step = period // len(values)
    if row % step:
        raise ValueError("periodic column %d is not read at row %d" % (table, row))
    return values[(row % period) // step]


# ---- constraints --------------------------------------------------------------------------------------------------------
def memory_constraints(hints: Hints, challenges) -> List[Constraint]:
    """air.rs:560-600"""
    z, a, one = ap.Const(challenges[MEM_Z]), ap.Const(challenges[MEM_A]), ap.Const(1)
    mem = lambda cell, k=0: ap.Trace(COL_MEMORY, MEMORY_STEP * k + cell)
    perm = lambda k=0: ap.Trace(COL_PERMUTATION, MEMORY_STEP * k + Permutation.MEMORY)
    diff = mem(Mem.ADDRESS, 1) - mem(Mem.ADDRESS)
    return [
        Constraint("memory/multi_column_perm/perm/init0",
                   (z - (mem(Mem.ADDRESS) + a * mem(Mem.VALUE))) * perm() + npc(Npc.PC) + a * npc(Npc.INSTRUCTION) - z, FIRST_ROW),
        Constraint("memory/multi_column_perm/perm/step0",
                   (z - (mem(Mem.ADDRESS, 1) + a * mem(Mem.VALUE, 1))) * perm(1) - (z - (npc_at(2) + a * npc_at(3))) * perm(), EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/multi_column_perm/perm/last", perm() - hints.memory_quotient, SECOND_LAST_ROW),
        Constraint("memory/diff_is_bit", diff * diff - diff, EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/is_func", (diff - one) * (mem(Mem.VALUE) - mem(Mem.VALUE, 1)), EVERY_2ND_EXCEPT_LAST),
        Constraint("memory/initial_addr", mem(Mem.ADDRESS) - one, FIRST_ROW),
        Constraint("public_memory_addr_zero", npc_at(Npc.PUB_MEM_ADDR), EVERY_8),
        Constraint("public_memory_value_zero", npc_at(Npc.PUB_MEM_VAL), EVERY_8),
    ]


def range_check_constraints(hints: Hints, challenges) -> List[Constraint]:
    """air.rs:602-630"""
    z = ap.Const(challenges[RC_Z])
    ordered = lambda k=0: c7(RANGE_CHECK_STEP * k + RangeCheck.ORDERED)
    perm = lambda k=0: ap.Trace(COL_PERMUTATION, 4 * k + Permutation.RANGE_CHECK)
    diff = ordered(1) - ordered()
    return [
        Constraint("rc16/perm/init0", (z - ordered()) * perm() + rc(RangeCheck.OFF_DST) - z, FIRST_ROW),
        Constraint("rc16/perm/step0", (z - ordered(1)) * perm(1) - (z - c7(4)) * perm(), EVERY_4TH_EXCEPT_LAST),
        Constraint("rc16/perm/last", perm() - hints.range_check_product, FOURTH_LAST_ROW),
        Constraint("rc16/diff_is_bit", diff * diff - diff, EVERY_4TH_EXCEPT_LAST),
        Constraint("rc16/minimum", ordered() - hints.range_check_min, FIRST_ROW),
        Constraint("rc16/maximum", ordered() - hints.range_check_max, FOURTH_LAST_ROW),
    ]


def diluted_check_constraints(hints: Hints, challenges) -> List[Constraint]:
    """air.rs:632-690"""
    z, za, aa = ap.Const(challenges[DC_Z]), ap.Const(challenges[AGG_Z]), ap.Const(challenges[AGG_A])
    un, od = (lambda k=0: c7(8 * k + DilutedCheck.UNORDERED)), (lambda k=0: c7(8 * k + DilutedCheck.ORDERED))
    perm = lambda k=0: ap.Trace(COL_PERMUTATION, 8 * k + Permutation.DILUTED_CHECK)
    agg = lambda k=0: ap.Trace(COL_PERMUTATION, 8 * k + DilutedCheck.AGGREGATE)
    diff = od(1) - od()
    return [
        Constraint("diluted_check/permutation/init0", (z - od()) * perm() + un() - z, FIRST_ROW),
        Constraint("diluted_check/permutation/step0", (z - od(1)) * perm(1) - (z - un(1)) * perm(), EVERY_8_EXCEPT_LAST),
        Constraint("diluted_check/permutation/last", perm() - hints.diluted_check_product, EIGHTH_LAST_ROW),
        Constraint("diluted_check/init", agg() - 1, FIRST_ROW),
        Constraint("diluted_check/first_element", od() - hints.diluted_check_first, FIRST_ROW),
        Constraint("diluted_check/step", agg(1) - (agg() * (1 + za * diff) + aa * diff * diff), EVERY_8_EXCEPT_LAST),
        Constraint("diluted_check/last", agg() - hints.diluted_check_cumulative_value, EIGHTH_LAST_ROW),
    ]


def _bit_unpacking(prefix, suffix, bit_251_196_192, bit_251_196, domain):
    """the six constraints that make a 252-bit decomposition unique (air.rs:694-745, 1700-1738)"""
    bit = lambda k: suffix(k) - (suffix(k + 1) + suffix(k + 1))
    return [
        Constraint(prefix + "bit_unpacking/last_one_is_zero", bit_251_196_192 * bit(0), domain),
        Constraint(prefix + "bit_unpacking/zeroes_between_ones0", bit_251_196_192 * (suffix(1) - suffix(192) * (1 << 191)), domain),
        Constraint(prefix + "bit_unpacking/cumulative_bit192", bit_251_196_192 - bit_251_196 * bit(192), domain),
        Constraint(prefix + "bit_unpacking/zeroes_between_ones192", bit_251_196 * (suffix(193) - suffix(196) * (1 << 3)), domain),
        Constraint(prefix + "bit_unpacking/cumulative_bit196", bit_251_196 - bit(251) * bit(196), domain),
        Constraint(prefix + "bit_unpacking/zeroes_between_ones196", bit(251) * (suffix(197) - suffix(251) * (1 << 54)), domain),
    ]


def _subset_sum(prefix, bit, sum_x, sum_y, slope, point_x, point_y, transition, x_diff_inv=None):
    """add the fixed point where the bit is set, copy the partial sum where it is not (air.rs:766-803, 1090-1135, ...)"""
    one = ap.Const(1)
    out = [
        Constraint(prefix + "add_points/slope", bit * (sum_y() - point_y) - slope() * (sum_x() - point_x), transition),
        Constraint(prefix + "add_points/x", slope() * slope() - bit * (sum_x() + point_x + sum_x(1)), transition),
        Constraint(prefix + "add_points/y", bit * (sum_y() + sum_y(1)) - slope() * (sum_x() - sum_x(1)), transition),
    ]
    if x_diff_inv is not None:
        out.append(Constraint(prefix + "add_points/x_diff_inv", x_diff_inv() * (sum_x() - point_x) - one, transition))
    out += [
        Constraint(prefix + "copy_point/x", (one - bit) * (sum_x(1) - sum_x()), transition),
        Constraint(prefix + "copy_point/y", (one - bit) * (sum_y(1) - sum_y()), transition),
    ]
    return out


def _doubling(prefix, x, y, slope, transition):
    """the chain of doublings of a point (air.rs:1050-1072, 1668-1690)"""
    x2 = x() * x()
    return [
        Constraint(prefix + "slope", x2 + x2 + x2 + ap.Const(1) - (y() + y()) * slope(), transition),        # curve alpha = 1
        Constraint(prefix + "x", slope() * slope() - (x() + x() + x(1)), transition),
        Constraint(prefix + "y", y() + y(1) - slope() * (x() - x(1)), transition),
    ]


def pedersen_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:692-1025.  A hash spans 512 rows: 256 steps for each input, one per row of columns 1-4"""
    suffix = lambda k=0: ap.Trace(COL_PEDERSEN_SUFFIX, k)
    slope = lambda k=0: ap.Trace(COL_PEDERSEN_SLOPE, k)
    sum_x, sum_y = (lambda k=0: ap.Trace(COL_PEDERSEN_X, k)), (lambda k=0: ap.Trace(COL_PEDERSEN_Y, k))
    b0 = suffix() - (suffix(1) + suffix(1))
    H = "pedersen/hash0/ec_subset_sum/"
    out = _bit_unpacking(H, suffix, c8(71), ap.Trace(COL_PEDERSEN_SLOPE, 255), EVERY_256)
    out += [Constraint(H + "booleanity_test", b0 * (b0 - ap.Const(1)), PEDERSEN_TRANSITION),
            Constraint(H + "bit_extraction_end", suffix(), PEDERSEN_STEP_252),
            Constraint(H + "zeros_tail", suffix(), PEDERSEN_STEP_255)]
    out += _subset_sum(H, b0, sum_x, sum_y, slope, ap.Table(TABLE_PEDERSEN_X), ap.Table(TABLE_PEDERSEN_Y), PEDERSEN_TRANSITION)
    px, py = PEDERSEN_POINTS[0]
    out += [
        Constraint("pedersen/hash0/copy_point/x", sum_x(256) - sum_x(255), PEDERSEN_HASH_START),
        Constraint("pedersen/hash0/copy_point/y", sum_y(256) - sum_y(255), PEDERSEN_HASH_START),
        Constraint("pedersen/hash0/init/x", sum_x() - px, EVERY_512),
        Constraint("pedersen/hash0/init/y", sum_y() - py, EVERY_512),
        Constraint("pedersen/input0_value0", npc_at(Npc.PEDERSEN_INPUT0_ADDR + 1) - suffix(), EVERY_512),
        Constraint("pedersen/input0_addr", npc_at(512 + Npc.PEDERSEN_INPUT0_ADDR) - (npc_at(Npc.PEDERSEN_OUTPUT_ADDR) + 1), EVERY_512_EXCEPT_LAST),
        Constraint("pedersen/init_addr", npc_at(Npc.PEDERSEN_INPUT0_ADDR) - hints.initial_pedersen_addr, FIRST_ROW),
        Constraint("pedersen/input1_value0", npc_at(Npc.PEDERSEN_INPUT1_ADDR + 1) - suffix(256), EVERY_512),
        Constraint("pedersen/input1_addr", npc_at(Npc.PEDERSEN_INPUT1_ADDR) - (npc_at(Npc.PEDERSEN_INPUT0_ADDR) + 1), EVERY_512),
        Constraint("pedersen/output_value0", npc_at(Npc.PEDERSEN_OUTPUT_ADDR + 1) - sum_x(511), EVERY_512),
        Constraint("pedersen/output_addr", npc_at(Npc.PEDERSEN_OUTPUT_ADDR) - (npc_at(Npc.PEDERSEN_INPUT1_ADDR) + 1), EVERY_512),
    ]
    return out


def range_check_builtin_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:1027-1040"""
    value = None
    for k in range(RANGE_CHECK_BUILTIN_PARTS):
        part = c7(32 * k + RangeCheck.RC16_COMPONENT)
        value = part if value is None else value * (1 << 16) + part
    return [
        Constraint("rc_builtin/value", value - npc_at(Npc.RANGE_CHECK128_ADDR + 1), EVERY_256),
        Constraint("rc_builtin/addr_step", npc_at(256 + Npc.RANGE_CHECK128_ADDR) - (npc_at(Npc.RANGE_CHECK128_ADDR) + 1), EVERY_256_EXCEPT_LAST),
        Constraint("rc_builtin/init_addr", npc_at(Npc.RANGE_CHECK128_ADDR) - hints.initial_rc_addr, FIRST_ROW),
    ]


def ecdsa_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:1042-1503.  An instance spans 32768 rows of column 8: two 256-step scalar multiplications of 64-row steps
    (r * Q on the doublings of the public key, then w * B on the doublings of B = z G + r Q) beside one 256-step
    multiplication z * G of 128-row steps on the periodic generator points."""
    E, one = Ecdsa, ap.Const(1)
    key = lambda cell: (lambda k=0: c8(64 * k + cell))
    gen = lambda cell: (lambda k=0: c8(128 * k + cell))
    dx, dy, dslope = key(E.PUBKEY_DOUBLING_X), key(E.PUBKEY_DOUBLING_Y), key(E.PUBKEY_DOUBLING_SLOPE)
    kx, ky, kslope, kinv, rsuffix = (key(E.PUBKEY_PARTIAL_SUM_X), key(E.PUBKEY_PARTIAL_SUM_Y), key(E.PUBKEY_PARTIAL_SUM_SLOPE),
                                     key(E.PUBKEY_PARTIAL_SUM_X_DIFF_INV), key(E.R_SUFFIX))
    gx, gy, gslope, ginv, msuffix = (gen(E.GENERATOR_PARTIAL_SUM_X), gen(E.GENERATOR_PARTIAL_SUM_Y), gen(E.GENERATOR_PARTIAL_SUM_SLOPE),
                                     gen(E.GENERATOR_PARTIAL_SUM_X_DIFF_INV), gen(E.MESSAGE_SUFFIX))
    gen_b0, key_b0 = msuffix() - (msuffix(1) + msuffix(1)), rsuffix() - (rsuffix(1) + rsuffix(1))
    shift_x, shift_y = SHIFT_POINT
    S = "ecdsa/signature0/"
    out = _doubling(S + "doubling_key/", dx, dy, dslope, EC_OP_TRANSITION)
    out += [Constraint(S + "exponentiate_generator/booleanity_test", gen_b0 * (gen_b0 - one), ECDSA_TRANSITION),
            Constraint(S + "exponentiate_generator/bit_extraction_end", msuffix(), ECDSA_STEP_251),
            Constraint(S + "exponentiate_generator/zeros_tail", msuffix(), ECDSA_STEP_255)]
    out += _subset_sum(S + "exponentiate_generator/", gen_b0, gx, gy, gslope, ap.Table(TABLE_ECDSA_GENERATOR_X), ap.Table(TABLE_ECDSA_GENERATOR_Y),
                       ECDSA_TRANSITION, ginv)
    out += [Constraint(S + "exponentiate_key/booleanity_test", key_b0 * (key_b0 - one), EC_OP_TRANSITION),
            Constraint(S + "exponentiate_key/bit_extraction_end", rsuffix(), EC_OP_STEP_251),
            Constraint(S + "exponentiate_key/zeros_tail", rsuffix(), EC_OP_STEP_255)]
    out += _subset_sum(S + "exponentiate_key/", key_b0, kx, ky, kslope, dx(), dy(), EC_OP_TRANSITION, kinv)
    b_slope, b_inv = c8(E.B_SLOPE), c8(E.B_X_DIFF_INV)
    r_slope, r_inv_x = c8(E.R_POINT_SLOPE), c8(E.R_POINT_X_DIFF_INV)
    out += [
        Constraint(S + "init_gen/x", gx() - shift_x, ALL_ECDSA),
        Constraint(S + "init_gen/y", gy() + shift_y, ALL_ECDSA),
        Constraint(S + "init_key/x", kx() - shift_x, ALL_EC_OP),
        Constraint(S + "init_key/y", ky() - shift_y, ALL_EC_OP),
        # B = z G + r Q: the start of the second doubling chain (air.rs:1347-1400)
        Constraint(S + "add_results/slope", gy(255) - (ky(255) + b_slope * (gx(255) - kx(255))), ALL_ECDSA),
        Constraint(S + "add_results/x", b_slope * b_slope - (gx(255) + kx(255) + dx(256)), ALL_ECDSA),
        Constraint(S + "add_results/y", gy(255) + dy(256) - b_slope * (gx(255) - dx(256)), ALL_ECDSA),
        Constraint(S + "add_results/x_diff_inv", b_inv * (gx(255) - kx(255)) - one, ALL_ECDSA),
        # r = x(w B - shift point) (air.rs:1402-1420)
        Constraint(S + "extract_r/slope", ky(511) + shift_y - r_slope * (kx(511) - shift_x), ALL_ECDSA),
        Constraint(S + "extract_r/x", r_slope * r_slope - (kx(511) + shift_x + rsuffix()), ALL_ECDSA),
        Constraint(S + "extract_r/x_diff_inv", r_inv_x * (kx(511) - shift_x) - one, ALL_ECDSA),
        Constraint(S + "z_nonzero", msuffix() * c8(E.MESSAGE_INV) - one, ALL_ECDSA),
        Constraint(S + "r_and_w_nonzero", rsuffix() * dslope(255) - one, ALL_EC_OP),
        Constraint(S + "q_on_curve/x_squared", c8(E.PUBKEY_X_SQUARED) - dx() * dx(), ALL_ECDSA),
        Constraint(S + "q_on_curve/on_curve", dy() * dy() - (dx() * c8(E.PUBKEY_X_SQUARED) + dx() * ap.Const(1) + CURVE_BETA), ALL_ECDSA),
        Constraint("ecdsa/init_addr", npc_at(Npc.ECDSA_PUBKEY_ADDR) - hints.initial_ecdsa_addr, FIRST_ROW),
        Constraint("ecdsa/message_addr", npc_at(Npc.ECDSA_MESSAGE_ADDR) - (npc_at(Npc.ECDSA_PUBKEY_ADDR) + 1), ALL_ECDSA),
        Constraint("ecdsa/pubkey_addr", npc_at(32768 + Npc.ECDSA_PUBKEY_ADDR) - (npc_at(Npc.ECDSA_MESSAGE_ADDR) + 1), ALL_ECDSA_EXCEPT_LAST),
        Constraint("ecdsa/message_value0", npc_at(Npc.ECDSA_MESSAGE_ADDR + 1) - msuffix(), ALL_ECDSA),
        Constraint("ecdsa/pubkey_value0", npc_at(Npc.ECDSA_PUBKEY_ADDR + 1) - dx(), ALL_ECDSA),
    ]
    return out


def bitwise_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:1505-1660.  An instance spans 1024 rows: four 256-row partitions (x, y, x&y, x^y) of column 7, whose cells
    16j + 1 hold the 16 diluted 16-bit segments."""
    bw = c7
    pool_addr = lambda k: npc_at(256 * k + Npc.BITWISE_POOL_ADDR)
    pool_val = lambda k: npc_at(256 * k + Npc.BITWISE_POOL_ADDR + 1)
    sum_var = None                                   # bitwise_sum_var_0_0 + bitwise_sum_var_8_0 (air.rs:232-262)
    for chunk in range(4):
        for stream in range(4):
            term = bw(Bitwise.cell(chunk, stream))
            shift = 64 * chunk + stream
            term = term * (1 << shift) if shift else term
            sum_var = term if sum_var is None else sum_var + term
    out = [
        Constraint("bitwise/init_var_pool_addr", pool_addr(0) - hints.initial_bitwise_addr, FIRST_ROW),
        Constraint("bitwise/step_var_pool_addr", pool_addr(1) - (pool_addr(0) + 1), BITWISE_TRANSITION),
        Constraint("bitwise/x_or_y_addr", npc_at(Npc.BITWISE_X_OR_Y_ADDR) - (pool_addr(3) + 1), ALL_BITWISE),
        Constraint("bitwise/next_var_pool_addr", pool_addr(4) - (npc_at(Npc.BITWISE_X_OR_Y_ADDR) + 1), ALL_BITWISE_EXCEPT_LAST),
        Constraint("bitwise/partition", sum_var - pool_val(0), EVERY_256),
        Constraint("bitwise/or_is_and_plus_xor", npc_at(Npc.BITWISE_X_OR_Y_ADDR + 1) - (pool_val(2) + pool_val(3)), ALL_BITWISE),
        Constraint("bitwise/addition_is_xor_with_and", bw(1) + bw(257) - (bw(769) + bw(513) + bw(513)), EVERY_16_BIT_SEGMENT),
    ]
    for k, shift in enumerate((4, 4, 4, 8)):         # air.rs:1608-1660
        seg = Bitwise.cell(3, k)
        out.append(Constraint("bitwise/unique_unpacking%d" % (192 + k), (bw(512 + seg) + bw(768 + seg)) * (1 << shift) - bw(Bitwise.SHIFTED_CELLS[k]),
                              ALL_BITWISE))
    return out


def ec_op_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:1662-1866.  An instance (R = P + m Q) spans 16384 rows of column 8 in 256 steps of 64 rows"""
    E, one = EcOp, ap.Const(1)
    cell = lambda c: (lambda k=0: c8(64 * k + c))
    qx, qy, qslope = cell(E.Q_DOUBLING_X), cell(E.Q_DOUBLING_Y), cell(E.Q_DOUBLING_SLOPE)
    rx, ry, rslope, rinv, msuffix = (cell(E.R_PARTIAL_SUM_X), cell(E.R_PARTIAL_SUM_Y), cell(E.R_PARTIAL_SUM_SLOPE),
                                     cell(E.R_PARTIAL_SUM_X_DIFF_INV), cell(E.M_SUFFIX))
    N = Npc
    b0 = msuffix() - (msuffix(1) + msuffix(1))
    out = [
        Constraint("ec_op/init_addr", npc_at(N.EC_OP_P_X_ADDR) - hints.initial_ec_op_addr, FIRST_ROW),
        Constraint("ec_op/p_x_addr", npc_at(16384 + N.EC_OP_P_X_ADDR) - (npc_at(N.EC_OP_P_X_ADDR) + 7), ALL_EC_OP_EXCEPT_LAST),
        Constraint("ec_op/p_y_addr", npc_at(N.EC_OP_P_Y_ADDR) - (npc_at(N.EC_OP_P_X_ADDR) + 1), ALL_EC_OP),
        Constraint("ec_op/q_x_addr", npc_at(N.EC_OP_Q_X_ADDR) - (npc_at(N.EC_OP_P_Y_ADDR) + 1), ALL_EC_OP),
        Constraint("ec_op/q_y_addr", npc_at(N.EC_OP_Q_Y_ADDR) - (npc_at(N.EC_OP_Q_X_ADDR) + 1), ALL_EC_OP),
        Constraint("ec_op/m_addr", npc_at(N.EC_OP_M_ADDR) - (npc_at(N.EC_OP_Q_Y_ADDR) + 1), ALL_EC_OP),
        Constraint("ec_op/r_x_addr", npc_at(N.EC_OP_R_X_ADDR) - (npc_at(N.EC_OP_M_ADDR) + 1), ALL_EC_OP),
        Constraint("ec_op/r_y_addr", npc_at(N.EC_OP_R_Y_ADDR) - (npc_at(N.EC_OP_R_X_ADDR) + 1), ALL_EC_OP),
    ]
    out += _doubling("ec_op/doubling_q/", qx, qy, qslope, EC_OP_TRANSITION)
    out += [Constraint("ec_op/get_q_x", npc_at(N.EC_OP_Q_X_ADDR + 1) - qx(), ALL_EC_OP),
            Constraint("ec_op/get_q_y", npc_at(N.EC_OP_Q_Y_ADDR + 1) - qy(), ALL_EC_OP)]
    H = "ec_op/ec_subset_sum/"
    out += _bit_unpacking(H, msuffix, c8(E.M_BIT251_AND_BIT196_AND_BIT192), c8(E.M_BIT251_AND_BIT196), ALL_EC_OP)
    out += [Constraint(H + "booleanity_test", b0 * (b0 - one), EC_OP_TRANSITION),
            Constraint(H + "bit_extraction_end", msuffix(), EC_OP_STEP_252),
            Constraint(H + "zeros_tail", msuffix(), EC_OP_STEP_255)]
    out += _subset_sum(H, b0, rx, ry, rslope, qx(), qy(), EC_OP_TRANSITION, rinv)
    out += [
        Constraint("ec_op/get_m", msuffix() - npc_at(N.EC_OP_M_ADDR + 1), ALL_EC_OP),
        Constraint("ec_op/get_p_x", npc_at(N.EC_OP_P_X_ADDR + 1) - rx(), ALL_EC_OP),
        Constraint("ec_op/get_p_y", npc_at(N.EC_OP_P_Y_ADDR + 1) - ry(), ALL_EC_OP),
        Constraint("ec_op/set_r_x", npc_at(N.EC_OP_R_X_ADDR + 1) - rx(255), ALL_EC_OP),
        Constraint("ec_op/set_r_y", npc_at(N.EC_OP_R_Y_ADDR + 1) - ry(255), ALL_EC_OP),
    ]
    return out


def poseidon_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:1868-2186.  An instance spans 512 rows: the 8 full rounds every 64 rows of column 8 (state and squares), the
    partial rounds 0..63 every 8 rows of column 7 and 61..82 every 16 rows of column 8."""
    keys = poseidon_air_keys()
    cell = lambda spec: (lambda k=0: ap.Trace(spec[0], spec[2] * k + spec[1]))
    full = [cell(s) for s in Poseidon.FULL_STATE]
    full_sq = [cell(s) for s in Poseidon.FULL_STATE_SQUARED]
    part, part_sq = [cell(s) for s in Poseidon.PARTIAL_STATE], [cell(s) for s in Poseidon.PARTIAL_STATE_SQUARED]
    cubed = lambda j, k: full[j](k) * full_sq[j](k)
    pcubed = lambda which, k: part[which](k) * part_sq[which](k)
    addr = lambda i: npc_at(Npc.POSEIDON_ADDRS[i])
    val = lambda i: npc_at(Npc.POSEIDON_ADDRS[i] + 1)
    rk0 = poseidon_round_keys()[0]
    Pn = "poseidon/poseidon/"
    out = [
        Constraint("poseidon/init_input_output_addr", addr(0) - hints.initial_poseidon_addr, FIRST_ROW),
        Constraint("poseidon/addr_input_output_step_inner", addr(1) - (addr(0) + 1), POSEIDON_ADDR_STEP),
        Constraint("poseidon/addr_input_output_step_outter", npc_at(512 + Npc.POSEIDON_ADDRS[0]) - (addr(5) + 1), EVERY_512_EXCEPT_LAST),
    ]
    out += [Constraint(Pn + "full_rounds_state%d_squaring" % j, full[j]() * full[j]() - full_sq[j](), EVERY_64) for j in range(3)]
    out += [Constraint(Pn + "partial_rounds_state0_squaring", part[0]() * part[0]() - part_sq[0](), EVERY_8),
            Constraint(Pn + "partial_rounds_state1_squaring", part[1]() * part[1]() - part_sq[1](), POSEIDON_PARTIAL1_SQUARING)]
    out += [Constraint(Pn + "add_first_round_key%d" % j, val(j) + rk0[j] - full[j](), EVERY_512) for j in range(3)]
    c0, c1, c2 = cubed(0, 0), cubed(1, 0), cubed(2, 0)
    fk = [ap.Table(TABLE_POSEIDON_FULL_KEY0 + j) for j in range(3)]
    out += [
        Constraint(Pn + "full_round0", full[0](1) - (c0 + c0 + c0 + c1 + c2 + fk[0]), POSEIDON_HALF_FULL_ROUND_TRANSITION),
        Constraint(Pn + "full_round1", full[1](1) + c1 - (c0 + c2 + fk[1]), POSEIDON_HALF_FULL_ROUND_TRANSITION),
        Constraint(Pn + "full_round2", full[2](1) + c2 + c2 - (c0 + c1 + fk[2]), POSEIDON_HALF_FULL_ROUND_TRANSITION),
    ]
    l0, l1, l2 = cubed(0, 7), cubed(1, 7), cubed(2, 7)
    out += [
        Constraint(Pn + "last_full_round0", val(3) - (l0 + l0 + l0 + l1 + l2), EVERY_512),
        Constraint(Pn + "last_full_round1", val(4) + l1 - (l0 + l2), EVERY_512),
        Constraint(Pn + "last_full_round2", val(5) + l2 + l2 - (l0 + l1), EVERY_512),
    ]
    out += [Constraint(Pn + "copy_partial_rounds0_i%d" % i, part[0](61 + i) - part[1](i), EVERY_512) for i in range(3)]
    m0, m1, m2 = cubed(0, 3), cubed(1, 3), cubed(2, 3)
    mk = keys["margin_full_to_partial"]
    p0c = [pcubed(0, k) for k in range(3)]
    out += [
        Constraint(Pn + "margin_full_to_partial0", part[0](0) + m2 + m2 - (m0 + m1 + mk[0]), EVERY_512),
        Constraint(Pn + "margin_full_to_partial1", part[0](1) - (m1 * (P - 4) + m2 * 10 + part[0](0) * 4 + p0c[0] * (P - 2) + mk[1]), EVERY_512),
        Constraint(Pn + "margin_full_to_partial2",
                   part[0](2) - (m2 * 8 + part[0](0) * 4 + p0c[0] * 6 + part[0](1) + part[0](1) + p0c[1] * (P - 2) + mk[2]), EVERY_512),
    ]
    for which, table, domain in ((0, TABLE_POSEIDON_PARTIAL_KEY0, POSEIDON_PARTIAL_ROUND0), (1, TABLE_POSEIDON_PARTIAL_KEY1, POSEIDON_PARTIAL_ROUND1)):
        pc = [pcubed(which, k) for k in range(3)]
        s = part[which]
        out.append(Constraint(Pn + "partial_round%d" % which,
                              s(3) - (pc[0] * 8 + s(1) * 4 + pc[1] * 6 + s(2) + s(2) + pc[2] * (P - 2) + ap.Table(table)), domain))
    q19, q20, q21 = pcubed(1, 19), pcubed(1, 20), pcubed(1, 21)
    s1 = part[1]
    fk2 = keys["margin_partial_to_full"]
    out += [
        Constraint(Pn + "margin_partial_to_full0", full[0](4) - (q19 * 16 + s1(20) * 8 + q20 * 16 + s1(21) * 6 + q21 + fk2[0]), EVERY_512),
        Constraint(Pn + "margin_partial_to_full1", full[1](4) - (q20 * 4 + s1(21) + s1(21) + q21 + fk2[1]), EVERY_512),
        Constraint(Pn + "margin_partial_to_full2",
                   full[2](4) - (q19 * 8 + s1(20) * 4 + q20 * 6 + s1(21) + s1(21) + q21 * (P - 2) + fk2[2]), EVERY_512),
    ]
    return out


def cpu_constraints(hints: Hints) -> List[Constraint]:
    """air.rs:128-557: the recursive layout's expressions over this layout's cells"""
    import sys
    return rec.cpu_constraints(hints, sys.modules[__name__])


def constraints(hints: Hints, challenges=None) -> List[Constraint]:
    """the 195 constraints in the reference's order (air.rs:2188-2386); without challenges the ones that involve no
    permutation argument"""
    local = (pedersen_constraints(hints) + range_check_builtin_constraints(hints) + ecdsa_constraints(hints) + bitwise_constraints(hints)
             + ec_op_constraints(hints) + poseidon_constraints(hints))
    if challenges is None:
        return cpu_constraints(hints) + local
    return (cpu_constraints(hints) + memory_constraints(hints, challenges) + range_check_constraints(hints, challenges)
            + diluted_check_constraints(hints, challenges) + local)


# ---- builtin instance traces (builtins/src/{ecdsa,ec_op,poseidon}/mod.rs) ------------------------------------------------
def _ec_neg(pt):
    return pt[0], (-pt[1]) % P


def _slope(p1, p2):
    """calculate_slope (builtins/src/utils.rs:163-181): the chord through two points, the tangent when they coincide"""
    (x1, y1), (x2, y2) = p1, p2
    if (x1, y1) == (x2, y2):
        return (3 * x1 * x1 + 1) * pow(2 * y1, -1, P) % P
    if x1 == x2:
        raise ValueError("vertical chord")
    return (y1 - y2) * pow(x1 - x2, -1, P) % P


def _ec_mul(k, pt):
    acc, addend = None, pt
    while k:
        if k & 1:
            acc = addend if acc is None else _ec_add(acc, addend)
        addend = _ec_double(addend)
        k >>= 1
    return acc


def _sqrt(a):
    """Tonelli-Shanks (p - 1 = 2^192 * odd)"""
    a %= P
    if a == 0:
        return 0
    if pow(a, (P - 1) // 2, P) != 1:
        return None
    s, q = 192, (P - 1) >> 192
    z = 3                                            # a generator of the multiplicative group, hence a non-residue
    m, c, t, r = s, pow(z, q, P), pow(a, q, P), pow(a, (q + 1) // 2, P)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % P
            i += 1
        b = pow(c, 1 << (m - i - 1), P)
        m, c = i, b * b % P
        t, r = t * c % P, r * b % P
    return r


def doubling_steps(point, count=256):
    """ecdsa/mod.rs:192-206: (point, slope of its tangent) for point, 2 point, 4 point, ..."""
    out = []
    for _ in range(count):
        out.append((point, _slope(point, point)))
        point = _ec_double(point)
    return out


def ec_mad_steps(x, point, start, max_doublings=255):
    """gen_ec_mad_steps (ecdsa/mod.rs:157-190, ec_op/mod.rs:98-130): 256 steps of start + x * point, least significant bit
    first.  -> [(partial sum before the step, fixed point of the step, suffix, slope or 0, 1 / (partial.x - fixed.x))]"""
    partial, out = start, []
    for i in range(256):
        suffix = x >> i
        slope, nxt = 0, partial
        if suffix & 1:
            slope = _slope(point, partial)
            nxt = _ec_add(partial, point)
        if partial[0] == point[0]:
            raise ValueError("a partial sum meets the fixed point")          # the reference's inverse().unwrap()
        out.append((partial, point, suffix % P, slope, pow(partial[0] - point[0], -1, P)))
        partial = nxt
        if i < max_doublings:
            point = _ec_double(point)
    return out


def _mimic_ec_mad(m, point, start):
    """mimic_ec_mad_air (ecdsa/mod.rs:278-301): start + m * point, None if a partial sum meets the doubled point"""
    if not 1 <= m.bit_length() < 252:
        return None
    partial = start
    while m:
        if partial[0] == point[0]:
            return None
        if m & 1:
            partial = _ec_add(partial, point)
        point = _ec_double(point)
        m >>= 1
    return partial


class EcdsaInstanceTrace:
    """ecdsa::InstanceTrace::new (ecdsa/mod.rs:61-140)"""

    def __init__(self, pubkey_x, message, r, w):
        self.pubkey_x, self.message, self.r, self.w = pubkey_x, message, r, w
        y = _sqrt((pow(pubkey_x, 3, P) + pubkey_x + CURVE_BETA) % P)
        if y is None:
            raise ValueError("the public key is not on the curve")
        neg_shift = _ec_neg(SHIFT_POINT)
        self.pubkey = None
        for cand in sorted((y, P - y), reverse=True):                       # verify (ecdsa/mod.rs:250-276)
            q = (pubkey_x, cand)
            zg, rq = _mimic_ec_mad(message, GENERATOR, neg_shift), _mimic_ec_mad(r, q, SHIFT_POINT)
            if zg is None or rq is None:
                continue
            wb = _mimic_ec_mad(w, _ec_add(zg, rq), SHIFT_POINT)
            if wb is not None and _ec_add(wb, neg_shift)[0] == r:
                self.pubkey, self.zg, self.rq, self.wb = q, zg, rq, wb
                break
        if self.pubkey is None:
            raise ValueError("signature is invalid")
        self.b = _ec_add(self.zg, self.rq)
        self.b_slope = _slope(self.zg, self.rq)
        self.b_x_diff_inv = pow(self.zg[0] - self.rq[0], -1, P)
        self.zg_steps = ec_mad_steps(message, GENERATOR, neg_shift, 250)
        self.rq_steps = ec_mad_steps(r, self.pubkey, SHIFT_POINT)
        self.wb_steps = ec_mad_steps(w, self.b, SHIFT_POINT)
        assert self.zg_steps[-1][0] == self.zg and self.rq_steps[-1][0] == self.rq and self.wb_steps[-1][0] == self.wb
        self.pubkey_doubling, self.b_doubling = doubling_steps(self.pubkey), doubling_steps(self.b)
        self.w_inv, self.r_inv, self.message_inv = pow(w, -1, P), pow(r, -1, P), pow(message, -1, P)
        self.r_point_slope = _slope(self.wb, neg_shift)
        self.r_point_x_diff_inv = pow(self.wb[0] - neg_shift[0], -1, P)


_DUMMIES = {}


def ecdsa_dummy_instance():
    """gen_dummy_instance (ecdsa/mod.rs:208-248): private key 1, message pedersen(1, 0), the first nonce k = 1, 2, ... that
    gives r and w below 2^251.  -> (pubkey_x, message, r, w)"""
    if "ecdsa" not in _DUMMIES:
        from .. import backend as be
        from ..coin import canonical
        message = canonical(be.pedersen_hash_host(be.felt(1), be.felt(0)))
        assert 0 < message < 1 << 251
        k = 0
        while True:
            k += 1
            r = _ec_mul(k, GENERATOR)[0]
            if r == 0 or r >= 1 << 251 or (message + r) % CURVE_ORDER == 0:
                continue
            w = k * pow(message + r, -1, CURVE_ORDER) % CURVE_ORDER
            if w == 0 or w >= 1 << 251:
                continue
            _DUMMIES["ecdsa"] = (GENERATOR[0], message, r, w)
            break
    return _DUMMIES["ecdsa"]


class EcOpInstanceTrace:
    """ec_op::InstanceTrace::new (ec_op/mod.rs:36-70): R = P + m Q"""

    def __init__(self, p, q, m):
        self.p, self.q, self.m = p, q, m
        self.q_doubling = doubling_steps(q)
        self.r_steps = ec_mad_steps(m, q, p)
        self.r = self.r_steps[-1][0]
        b251, b196, b192 = (m >> 251) & 1, (m >> 196) & 1, (m >> 192) & 1
        self.bit251_and_bit196, self.bit251_and_bit196_and_bit192 = b251 & b196, b251 & b196 & b192


class PoseidonInstanceTrace:
    """poseidon::InstanceTrace::new (poseidon/mod.rs:45-98)"""

    def __init__(self, inputs):
        self.inputs = tuple(v % P for v in inputs)
        self.full, self.partial, out = poseidon_states(self.inputs)
        self.outputs = tuple(out)


# ---- base trace (trace.rs:98-987) -----------------------------------------------------------------------------------------
def base_trace(register_states, memory, public_input, private_input=None):
    """ExecutionTrace::new (starknet/trace.rs:98-987) -> the 9 base columns as lists of canonical ints.
    private_input: {"pedersen": [(index, a, b)], "range_check": [(index, value)], "ecdsa": [(index, pubkey_x, message, r, w)],
    "bitwise": [(index, x, y)], "ec_op": [(index, p_x, p_y, q_x, q_y, m)], "poseidon": [(index, s0, s1, s2)]}; missing
    instances are the reference's dummies."""
    from .. import backend as be
    from ..coin import canonical
    private_input = private_input or {}
    num_cycles = len(register_states)
    if num_cycles & (num_cycles - 1):
        raise ValueError("the number of cycles must be a power of two")
    n = num_cycles * CYCLE_HEIGHT
    if n < ECDSA_BUILTIN_RATIO * CYCLE_HEIGHT:
        raise ValueError("the starknet layout needs at least %d cycles" % ECDSA_BUILTIN_RATIO)
    seg = public_input.memory_segments
    pad_addr, pad_value = public_input.public_memory_padding()
    cols = [[0] * n for _ in range(NUM_BASE_COLUMNS)]
    flags, npc_col, rc_col, aux_col = cols[COL_FLAGS], cols[COL_NPC], cols[COL_RANGE_CHECK], cols[COL_AUXILIARY]
    npc_col[0::2] = [pad_addr] * (n // 2)
    npc_col[1::2] = [pad_value] * (n // 2)

    def set_pair(row, address, value):
        npc_col[row], npc_col[row + 1] = address, value % P

    # ---- range-check pool (trace.rs:142-165)
    pool = []
    for st in register_states:
        w = bn.Word(memory[st.pc])
        pool += [w.off_dst, w.off_op0, w.off_op1]
    rc128 = [(int(i), int(v)) for i, v in private_input.get("range_check", [])]
    parts_of = lambda v: [(v >> (16 * (RANGE_CHECK_BUILTIN_PARTS - 1 - k))) & 0xFFFF for k in range(RANGE_CHECK_BUILTIN_PARTS)]
    for _, v in rc128:
        pool += parts_of(v)
    ordered_vals, padding_vals = rec._rc_ordered_with_padding(pool)
    rc_max = max(pool)
    rc_col[:] = [rc_max] * n
    padding_iter, ordered_iter = iter(padding_vals), iter(ordered_vals)

    # ---- CPU cells (trace.rs:177-244)
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
        tmp0 = dst if w.flag(bn.PC_JNZ) else 0
        for f in range(16):
            flags[r + f] = w.flag_prefix(f)
        set_pair(r + Npc.PC, pc, memory[pc])
        set_pair(r + Npc.MEM_OP0_ADDR, op0_addr, op0)
        set_pair(r + Npc.MEM_DST_ADDR, dst_addr, dst)
        set_pair(r + Npc.MEM_OP1_ADDR, op1_addr, op1)
        for o in range(0, CYCLE_HEIGHT, PUBLIC_MEMORY_STEP):
            set_pair(r + o + Npc.PUB_MEM_ADDR, 0, 0)
        rc_col[r + RangeCheck.OFF_DST], rc_col[r + RangeCheck.OFF_OP1], rc_col[r + RangeCheck.OFF_OP0] = w.off_dst, w.off_op1, w.off_op0
        aux_col[r + Auxiliary.TMP0], aux_col[r + Auxiliary.TMP1] = tmp0, tmp0 * res % P
        aux_col[r + Auxiliary.AP], aux_col[r + Auxiliary.FP] = ap_, fp
        aux_col[r + Auxiliary.OP0_MUL_OP1], aux_col[r + Auxiliary.RES] = op0 * op1 % P, res

    # ---- range-check padding and the ordered values (trace.rs:246-292)
    for index in range(len(rc128), num_cycles // RANGE_CHECK_BUILTIN_RATIO):
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
    for k in range(n // DILUTED_CHECK_STEP):                             # trace.rs:294-302
        rc_col[8 * k + DilutedCheck.UNORDERED] = rc_col[8 * k + DilutedCheck.ORDERED] = 0

    # ---- Pedersen (trace.rs:304-386)
    ped = {int(i): (int(a), int(b)) for i, a, b in private_input.get("pedersen", [])}
    step, begin, cache = PEDERSEN_BUILTIN_RATIO * CYCLE_HEIGHT, seg["pedersen"][0], {}
    xs, ys, suffixes, slopes = cols[COL_PEDERSEN_X], cols[COL_PEDERSEN_Y], cols[COL_PEDERSEN_SUFFIX], cols[COL_PEDERSEN_SLOPE]
    for i in range(n // step):
        a, b = ped.get(i, (0, 0))
        if (a, b) not in cache:
            a_steps, mid = rec.pedersen_element_steps(a % P, PEDERSEN_POINTS[0], 0)
            b_steps, _ = rec.pedersen_element_steps(b % P, mid, 1)
            if canonical(be.pedersen_hash_host(be.felt(a), be.felt(b))) != b_steps[-1][0][0]:
                raise ValueError("Pedersen partial sums do not end at the hash")
            cache[(a, b)] = a_steps + b_steps
        steps = cache[(a, b)]
        base, addr = i * step, begin + 3 * i
        xs[base:base + 512] = [s[0][0] for s in steps]
        ys[base:base + 512] = [s[0][1] for s in steps]
        suffixes[base:base + 512] = [s[1] for s in steps]
        slopes[base:base + 512] = [s[2] for s in steps]
        for half, v in ((0, a), (1, b)):
            b251, b196, b192 = (v >> 251) & 1, (v >> 196) & 1, (v >> 192) & 1
            slopes[base + 256 * half + 255] = b251 & b196
            aux_col[base + 256 * half + 71] = b251 & b196 & b192
        set_pair(base + Npc.PEDERSEN_INPUT0_ADDR, addr, a)
        set_pair(base + Npc.PEDERSEN_INPUT1_ADDR, addr + 1, b)
        set_pair(base + Npc.PEDERSEN_OUTPUT_ADDR, addr + 2, steps[-1][0][0])

    # ---- range-check builtin (trace.rs:388-426)
    step, begin = RANGE_CHECK_BUILTIN_RATIO * CYCLE_HEIGHT, seg["range_check"][0]
    for block, (index, value) in enumerate(rc128):
        base = block * step
        for k, part in enumerate(parts_of(value)):
            rc_col[base + 32 * k + RangeCheck.RC16_COMPONENT] = part
        set_pair(base + Npc.RANGE_CHECK128_ADDR, begin + index, value)

    # ---- ECDSA (trace.rs:428-523)
    given = {int(e[0]): tuple(int(v) for v in e[1:]) for e in private_input.get("ecdsa", [])}
    step, begin, cache = ECDSA_BUILTIN_RATIO * CYCLE_HEIGHT, seg["ecdsa"][0], {}
    E = Ecdsa
    for i in range(n // step):
        inst = given.get(i) or ecdsa_dummy_instance()
        if inst not in cache:
            cache[inst] = EcdsaInstanceTrace(*inst)
        t = cache[inst]
        base = i * step
        for half, (mad, dbl) in enumerate(((t.rq_steps, t.pubkey_doubling), (t.wb_steps, t.b_doubling))):
            for j in range(256):
                r = base + 64 * (256 * half + j)
                (point, slope), (partial, _, suffix, pslope, inv) = dbl[j], mad[j]
                aux_col[r + E.PUBKEY_DOUBLING_X], aux_col[r + E.PUBKEY_DOUBLING_Y], aux_col[r + E.PUBKEY_DOUBLING_SLOPE] = point[0], point[1], slope
                aux_col[r + E.PUBKEY_PARTIAL_SUM_X], aux_col[r + E.PUBKEY_PARTIAL_SUM_Y] = partial
                aux_col[r + E.PUBKEY_PARTIAL_SUM_SLOPE], aux_col[r + E.PUBKEY_PARTIAL_SUM_X_DIFF_INV], aux_col[r + E.R_SUFFIX] = pslope, inv, suffix
        for j, (partial, _, suffix, pslope, inv) in enumerate(t.zg_steps):
            r = base + 128 * j
            aux_col[r + E.GENERATOR_PARTIAL_SUM_X], aux_col[r + E.GENERATOR_PARTIAL_SUM_Y] = partial
            aux_col[r + E.GENERATOR_PARTIAL_SUM_SLOPE], aux_col[r + E.GENERATOR_PARTIAL_SUM_X_DIFF_INV], aux_col[r + E.MESSAGE_SUFFIX] = pslope, inv, suffix
        for cell, value in ((E.B_SLOPE, t.b_slope), (E.B_X_DIFF_INV, t.b_x_diff_inv), (E.W_INV, t.w_inv), (E.R_INV, t.r_inv),
                            (E.R_POINT_SLOPE, t.r_point_slope), (E.R_POINT_X_DIFF_INV, t.r_point_x_diff_inv), (E.MESSAGE_INV, t.message_inv),
                            (E.PUBKEY_X_SQUARED, t.pubkey[0] * t.pubkey[0] % P)):
            aux_col[base + cell] = value
        set_pair(base + Npc.ECDSA_PUBKEY_ADDR, begin + 2 * i, t.pubkey[0])
        set_pair(base + Npc.ECDSA_MESSAGE_ADDR, begin + 2 * i + 1, t.message)

    # ---- bitwise and the diluted check (trace.rs:525-705)
    step, begin = BITWISE_RATIO * CYCLE_HEIGHT, seg["bitwise"][0]
    bw = {int(i): (int(x), int(y)) for i, x, y in private_input.get("bitwise", [])}
    mask64, diluted_pool = (1 << 64) - 1, []
    for i in range(n // step):
        x, y = bw.get(i, (0, 0))
        base, addr = i * step, begin + 5 * i
        parts = [[rec._partition64((v >> (64 * c)) & mask64) for c in range(4)] for v in (x, y, x & y, x ^ y)]
        for k in range(4):
            v = parts[2][3][k] + parts[3][3][k]
            sh = 8 if k == 3 else 4
            if (v << sh) >> sh != v or (v << sh) >= 1 << 64:
                raise ValueError("bitwise instance %d: top segment does not fit" % i)
            rc_col[base + Bitwise.SHIFTED_CELLS[k]] = v << sh
            diluted_pool.append(rec._undilute(v << sh))
        for pidx, part in enumerate(parts):
            for c in range(4):
                for s_ in range(4):
                    rc_col[base + 256 * pidx + Bitwise.cell(c, s_)] = part[c][s_]
                    diluted_pool.append(rec._undilute(part[c][s_]))
        for k, val in enumerate((x, y, x & y, x ^ y)):
            set_pair(base + Npc.BITWISE_POOL_ADDR + 256 * k, addr + k, val)
        set_pair(base + Npc.BITWISE_X_OR_Y_ADDR, addr + 4, x | y)
    lo, hi = 0, (1 << DILUTED_CHECK_N_BITS) - 1
    ordered = sorted(diluted_pool)
    present = set(ordered)
    padding = [v for v in range(lo, hi + 1) if v not in present]
    ordered = sorted(ordered + padding)
    pad_iter, done = iter(padding), False
    for blk in range(n // 1024):                     # padding goes to the free cells 8i + 1, i odd (trace.rs:668-693)
        for i in range(1, 1024 // DILUTED_CHECK_STEP, 2):
            off = 8 * i + DilutedCheck.UNORDERED
            if off in Bitwise.SHIFTED_CELLS:
                continue
            v = next(pad_iter, None)
            if v is None:
                done = True
                break
            rc_col[blk * 1024 + off] = rec._dilute(v)
        if done:
            break
    slots = n // DILUTED_CHECK_STEP
    if next(pad_iter, None) is not None or len(ordered) > slots:
        raise ValueError("diluted-check values do not fit the trace")
    for k, v in enumerate(ordered, slots - len(ordered)):
        rc_col[8 * k + DilutedCheck.ORDERED] = rec._dilute(v)

    # ---- EC op (trace.rs:707-777)
    given = {int(e[0]): tuple(int(v) for v in e[1:]) for e in private_input.get("ec_op", [])}
    step, begin, cache = EC_OP_BUILTIN_RATIO * CYCLE_HEIGHT, seg["ec_op"][0], {}
    O = EcOp
    for i in range(n // step):
        inst = given.get(i) or (SHIFT_POINT[0], SHIFT_POINT[1], GENERATOR[0], GENERATOR[1], 1)      # gen_dummy_instance (ec_op/mod.rs:84-96)
        if inst not in cache:
            cache[inst] = EcOpInstanceTrace((inst[0], inst[1]), (inst[2], inst[3]), inst[4])
        t = cache[inst]
        base, addr = i * step, begin + 7 * i
        for j in range(256):
            r = base + 64 * j
            (point, slope), (partial, _, suffix, pslope, inv) = t.q_doubling[j], t.r_steps[j]
            aux_col[r + O.Q_DOUBLING_X], aux_col[r + O.Q_DOUBLING_Y], aux_col[r + O.Q_DOUBLING_SLOPE] = point[0], point[1], slope
            aux_col[r + O.R_PARTIAL_SUM_X], aux_col[r + O.R_PARTIAL_SUM_Y], aux_col[r + O.M_SUFFIX] = partial[0], partial[1], suffix
            if j != 255:                             # the last step's cells belong to the ECDSA builtin
                aux_col[r + O.R_PARTIAL_SUM_SLOPE], aux_col[r + O.R_PARTIAL_SUM_X_DIFF_INV] = pslope, inv
        aux_col[base + O.M_BIT251_AND_BIT196], aux_col[base + O.M_BIT251_AND_BIT196_AND_BIT192] = t.bit251_and_bit196, t.bit251_and_bit196_and_bit192
        for k, (cell, value) in enumerate(((Npc.EC_OP_P_X_ADDR, t.p[0]), (Npc.EC_OP_P_Y_ADDR, t.p[1]), (Npc.EC_OP_Q_X_ADDR, t.q[0]),
                                          (Npc.EC_OP_Q_Y_ADDR, t.q[1]), (Npc.EC_OP_M_ADDR, t.m), (Npc.EC_OP_R_X_ADDR, t.r[0]),
                                          (Npc.EC_OP_R_Y_ADDR, t.r[1]))):
            set_pair(base + cell, addr + k, value)

    # ---- Poseidon (trace.rs:779-888)
    given = {int(e[0]): tuple(int(v) for v in e[1:]) for e in private_input.get("poseidon", [])}
    step, begin, cache = POSEIDON_RATIO * CYCLE_HEIGHT, seg["poseidon"][0], {}
    for i in range(n // step):
        inst = given.get(i, (0, 0, 0))
        if inst not in cache:
            cache[inst] = PoseidonInstanceTrace(inst)
        t = cache[inst]
        base, addr = i * step, begin + 6 * i
        for rnd, state in enumerate(t.full):
            for j in range(3):
                (c, sh, st_), (c2, sh2, _) = Poseidon.FULL_STATE[j], Poseidon.FULL_STATE_SQUARED[j]
                cols[c][base + st_ * rnd + sh] = state[j]
                cols[c2][base + st_ * rnd + sh2] = state[j] * state[j] % P
        for which, values in ((0, t.partial[:64]), (1, t.partial[61:])):
            (c, sh, st_), (c2, sh2, _) = Poseidon.PARTIAL_STATE[which], Poseidon.PARTIAL_STATE_SQUARED[which]
            for k, v in enumerate(values):
                cols[c][base + st_ * k + sh] = v
                cols[c2][base + st_ * k + sh2] = v * v % P
        for k, value in enumerate(t.inputs + t.outputs):
            set_pair(base + Npc.POSEIDON_ADDRS[k], addr + k, value)

    # ---- gap fillers and the sorted memory column (trace.rs:890-960; layouts/src/utils.rs:112-152)
    accessed = sorted(set(npc_col[0::2]) | {a for a, _ in public_input.public_memory})
    gaps = [v for a, b in zip(accessed, accessed[1:]) for v in range(a + 1, b)]
    if len(gaps) > num_cycles:
        raise ValueError("more memory gaps than cycles to hold them")
    for cycle, addr in enumerate(gaps):
        set_pair(cycle * CYCLE_HEIGHT + Npc.UNUSED_ADDR, addr, 0)
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


def example_public_input(pi):
    """The reference ships one Cairo run (example/trace.bin, memory.bin - no builtin is used).  Its public input names the
    recursive layout; this re-declares the same run for the starknet layout: every builtin segment sized for n_steps and
    laid out back to back after the execution segment, so that the memory stays continuous."""
    import copy
    out = copy.deepcopy(pi)
    out.layout = "starknet"
    n_steps, seg = pi.n_steps, dict(pi.memory_segments)
    addr = seg["execution"][1]
    seg["output"] = (addr, addr)
    for name, ratio, cells in (("pedersen", PEDERSEN_BUILTIN_RATIO, 3), ("range_check", RANGE_CHECK_BUILTIN_RATIO, 1), ("ecdsa", ECDSA_BUILTIN_RATIO, 2),
                               ("bitwise", BITWISE_RATIO, 5), ("ec_op", EC_OP_BUILTIN_RATIO, 7), ("poseidon", POSEIDON_RATIO, 6)):
        size = cells * (n_steps // ratio)
        seg[name] = (addr, addr)                     # begin and stop pointer: nothing of the segment is used by the program
        addr += size
    out.memory_segments = seg
    return out


# ---- the composition constraint and its tables (composition_constraint, air.rs:2390-2406) -------------------------------
_COEFFS = {}


def _interpolate(vals):
    """coefficients of the polynomial of degree < len(vals) with value vals[j] at w^j, w the len(vals)-th root of unity"""
    m = len(vals)
    w_inv, m_inv = pow(pow(3, (P - 1) // m, P), -1, P), pow(m, -1, P)
    bits = m.bit_length() - 1
    a = [vals[int(format(i, "0%db" % bits)[::-1], 2) if bits else 0] for i in range(m)]
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
    return [v * m_inv % P for v in a]


def periodic_coefficients(table):
    if table not in _COEFFS:
        _COEFFS[table] = _interpolate(periodic_columns()[table][0])
    return _COEFFS[table]


class Tables(rec.Tables):
    """rec.Tables with this layout's nine periodic columns in front (table index = TABLE_*)"""

    def __init__(self, n, blowup=2, offset=3):
        super().__init__(n, blowup, offset)
        self.specs = [("column", t) for t in range(NUM_PERIODIC)]

    def length(self, spec):
        if spec[0] == "column":
            return periodic_columns()[spec[1]][1] * (self.N // self.n)
        return super().length(spec)

    def value_at(self, spec, x):
        if spec[0] == "column":
            return rec._poly_eval(periodic_coefficients(spec[1]), pow(x, self.n // periodic_columns()[spec[1]][1], P))
        return super().value_at(spec, x)

    def host_values(self, spec):
        if spec[0] != "column":
            return super().host_values(spec)
        coeffs, period = periodic_coefficients(spec[1]), periodic_columns()[spec[1]][1]
        x, step, out = pow(self.offset, self.n // period, P), pow(self.w, self.n // period, P), []
        for _ in range(self.length(spec)):
            out.append(rec._poly_eval(coeffs, x))
            x = x * step % P
        return out


def composition(n, hints: Hints, challenges, alpha, tables: Tables):
    """sum_i alpha^i * constraint_i (air.rs:2390-2406)"""
    return rec.composition(n, hints, challenges, alpha, tables, constraints(hints, challenges))


def mask(hints=None):
    """the 269 trace cells the constraints read, sorted: the order of the out-of-domain vector"""
    return rec.mask(None, constraints(hints or Hints(0, 0, 0, 0), [2, 3, 5, 7, 11, 13]))


# ---- the layout as the prover and the verifier see it -------------------------------------------------------------------
def make_air(ctx, public_input, n, log_blowup=1, lde_offset=3):
    """-> prover.Air (the tables: 9 periodic columns and the periodic multipliers from the host, 5 full-length inverse
    tables built on the device)"""
    import sys
    return rec.make_air(ctx, public_input, n, log_blowup, lde_offset, sys.modules[__name__])


def verifier_air(public_input, log_blowup=1, lde_offset=3):
    import sys
    return rec.verifier_air(public_input, log_blowup, lde_offset, sys.modules[__name__])


def trace_columns(ctx, base_cols_device, n):
    """extension.TraceColumns of a base trace resident in HBM: the diluted check lives in the range-check column
    (trace.rs:997-1056)"""
    from ..extension import TraceColumns
    return TraceColumns(npc=base_cols_device[COL_NPC], memory=base_cols_device[COL_MEMORY], range_check=base_cols_device[COL_RANGE_CHECK], trace_len=n)

"""Instruction encoding of the constraint-evaluation program (ss_air_program,
include/sandstorm_hip.h): a 4-accumulator two-address register machine that the
host lowers an AIR's composition constraint into (reference: the `Expr` DAG of
layouts/src/recursive/air.rs:61-1200 evaluated by ministark's
AirConfig::eval_constraint)."""


class OP:
    MOV, ADD, SUB, RSUB, MUL, INV, ST, OUT = range(8)


class SRC:
    ACC, SLOT, CONST, TRACE, TABLE, X = range(6)


def instr(op, dst, kind, payload):
    """-> [word0, word1]"""
    assert 0 <= dst < 4 and 0 <= payload < 2**32
    return [op | (dst << 8) | (kind << 12), payload]


def trace_payload(col, row_offset):
    assert 0 <= col < 256 and 0 <= row_offset < (1 << 24)
    return (col << 24) | row_offset

"""The `cairo-run` artefacts `sandstorm-cli prove` reads (the data formats on the input side of
the path): `trace.bin` (register states), `memory.bin` (partial memory) and the instruction word layout
(binary/src/lib.rs:33-56, 147-222, 565-721).  Host-only.

  trace.bin    records of three little-endian u64: ap, fp, pc                       (RegisterStates::from_reader)
  memory.bin   records of a little-endian u64 address + a 32-byte little-endian word (Memory::from_reader)
  instruction  off_dst | off_op0 << 16 | off_op1 << 32 | flags << 48, offsets biased by 2^15 (Word)
  air-private-input.json   where the two files are, and the builtin instances of the run          (AirPrivateInput)
"""
import json
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

P = 2**251 + 17 * 2**192 + 1
HALF_OFFSET = 1 << 15
OFF_DST_BIT_OFFSET, OFF_OP0_BIT_OFFSET, OFF_OP1_BIT_OFFSET, FLAGS_BIT_OFFSET = 0, 16, 32, 48
# enum Flag (binary/src/lib.rs:740-772)
(DST_REG, OP0_REG, OP1_IMM, OP1_FP, OP1_AP, RES_ADD, RES_MUL, PC_JUMP_ABS, PC_JUMP_REL, PC_JNZ, AP_ADD, AP_ADD1,
 OPCODE_CALL, OPCODE_RET, OPCODE_ASSERT_EQ, ZERO) = range(16)


@dataclass(frozen=True)
class RegisterState:
    ap: int
    fp: int
    pc: int


def read_register_states(data: bytes) -> List[RegisterState]:
    if len(data) % 24:
        raise ValueError("trace file is not a sequence of (ap, fp, pc) u64 triples")
    return [RegisterState(*struct.unpack_from("<QQQ", data, o)) for o in range(0, len(data), 24)]


def read_memory(data: bytes) -> List[Optional[int]]:
    """-> memory[address] = word or None (cells the run never touched)"""
    if len(data) % 40:
        raise ValueError("memory file is not a sequence of (u64 address, 32-byte word) records")
    cells = [(struct.unpack_from("<Q", data, o)[0], int.from_bytes(data[o + 8:o + 40], "little")) for o in range(0, len(data), 40)]
    memory = [None] * (max(a for a, _ in cells) + 1 if cells else 0)
    for a, w in cells:
        memory[a] = w
    return memory


def write_register_states(states) -> bytes:
    """the inverse of read_register_states (the `trace.bin` cairo-run writes)"""
    return b"".join(struct.pack("<QQQ", s.ap, s.fp, s.pc) for s in states)


def write_memory(memory) -> bytes:
    """the inverse of read_memory: one record per touched cell, in address order"""
    return b"".join(struct.pack("<Q", a) + int(w).to_bytes(32, "little") for a, w in enumerate(memory) if w is not None)


def _u256(text) -> int:
    """deserialize_hex_str (binary/src/utils.rs:30-33): the string through ruint's FromStr - "0x..." as every cairo-run file writes it,
    a bare decimal number as well"""
    if not isinstance(text, str):
        raise ValueError("a 256-bit value of the private input must be a string, got %r" % (text,))
    value = int(text, 0)
    if not 0 <= value < 1 << 256:
        raise ValueError("%s does not fit 256 bits" % text)
    return value


@dataclass
class AirPrivateInput:
    """`air-private-input.json` (binary/src/lib.rs:522-536; the instance records :343-520): the paths of trace.bin / memory.bin and the
    builtin instances `cairo-run` recorded.  `instances` holds them as the trace generators take them (layouts/*.py base_trace,
    hostlib.*_base_trace, hostlib.prove_files): tuples that start with the instance's index -
      pedersen (index, x, y) | range_check (index, value) | ecdsa (index, pubkey, msg, r, w) | bitwise (index, x, y) |
      ec_op (index, p_x, p_y, q_x, q_y, m) | poseidon (index, input_s0, input_s1, input_s2).
    `pedersen` and `range_check` must be there; the other four default to none (the reference's #[serde(default)])."""
    trace_path: str
    memory_path: str
    instances: Dict[str, List[tuple]] = field(default_factory=dict)

    _FIELDS = (("pedersen", ("x", "y"), True), ("range_check", ("value",), True), ("ecdsa", ("pubkey", "msg", ("signature_input", "r"), ("signature_input", "w")), False),
               ("bitwise", ("x", "y"), False), ("ec_op", ("p_x", "p_y", "q_x", "q_y", "m"), False), ("poseidon", ("input_s0", "input_s1", "input_s2"), False))

    @classmethod
    def from_dict(cls, doc) -> "AirPrivateInput":
        for key in ("trace_path", "memory_path"):
            if not isinstance(doc.get(key), str):
                raise ValueError("private input: `%s` is missing" % key)
        instances = {}
        for name, keys, required in cls._FIELDS:
            if name not in doc:
                if required:
                    raise ValueError("private input: `%s` is missing" % name)
                instances[name] = []
                continue
            rows = []
            for rec in doc[name]:
                try:
                    index = rec["index"]
                    if not isinstance(index, int) or isinstance(index, bool) or not 0 <= index < 1 << 32:
                        raise ValueError("index %r is not a u32" % (index,))
                    rows.append((index,) + tuple(_u256(rec[k[0]][k[1]] if isinstance(k, tuple) else rec[k]) for k in keys))
                except (KeyError, TypeError) as e:
                    raise ValueError("private input: a `%s` instance lacks %s" % (name, e))
            instances[name] = rows
        return cls(doc["trace_path"], doc["memory_path"], instances)

    @classmethod
    def from_json(cls, path) -> "AirPrivateInput":
        with open(path) as f:
            return cls.from_dict(json.load(f))


class Word:
    """One memory word read as an instruction (binary/src/lib.rs:565-721)"""

    def __init__(self, value: int):
        self.value = int(value)

    def flag(self, f) -> int:
        return (self.value >> (FLAGS_BIT_OFFSET + f)) & 1

    def flag_prefix(self, f) -> int:
        """~f_i of the Cairo paper: the flag bits from f upwards (get_flag_prefix)"""
        return 0 if f == ZERO else (self.value >> (FLAGS_BIT_OFFSET + f)) & ((1 << (15 - f)) - 1)

    @property
    def off_dst(self):
        return (self.value >> OFF_DST_BIT_OFFSET) & 0xFFFF

    @property
    def off_op0(self):
        return (self.value >> OFF_OP0_BIT_OFFSET) & 0xFFFF

    @property
    def off_op1(self):
        return (self.value >> OFF_OP1_BIT_OFFSET) & 0xFFFF

    # flag groups (get_flag_group)
    @property
    def op1_src(self):
        return self.flag(OP1_IMM) + 2 * self.flag(OP1_FP) + 4 * self.flag(OP1_AP)

    @property
    def res_logic(self):
        return self.flag(RES_ADD) + 2 * self.flag(RES_MUL)

    @property
    def pc_update(self):
        return self.flag(PC_JUMP_ABS) + 2 * self.flag(PC_JUMP_REL) + 4 * self.flag(PC_JNZ)

    @property
    def ap_update(self):
        return self.flag(AP_ADD) + 2 * self.flag(AP_ADD1)

    @property
    def opcode(self):
        return self.flag(OPCODE_CALL) + 2 * self.flag(OPCODE_RET) + 4 * self.flag(OPCODE_ASSERT_EQ)

    def dst_addr(self, ap, fp):
        return self.off_dst + (fp if self.flag(DST_REG) else ap) - HALF_OFFSET

    def op0_addr(self, ap, fp):
        return self.off_op0 + (fp if self.flag(OP0_REG) else ap) - HALF_OFFSET

    def op1_addr(self, pc, ap, fp, memory):
        src = self.op1_src
        if src not in (0, 1, 2, 4):
            raise ValueError("invalid op1 source %d" % src)
        base = memory[self.op0_addr(ap, fp)] if src == 0 else pc if src == 1 else fp if src == 2 else ap
        return self.off_op1 + base - HALF_OFFSET

    def res(self, pc, ap, fp, memory):
        """get_res: op1, op0 + op1 or op0 * op1; on a jnz the slot holds dst^-1 (or 0)"""
        if self.pc_update == 4:
            d = memory[self.dst_addr(ap, fp)] % P
            return pow(d, -1, P) if d else 0
        op0, op1 = memory[self.op0_addr(ap, fp)], memory[self.op1_addr(pc, ap, fp, memory)]
        logic = self.res_logic
        if logic > 2:
            raise ValueError("invalid res logic")
        return op1 % P if logic == 0 else (op0 + op1) % P if logic == 1 else op0 * op1 % P


def next_state(state: RegisterState, memory) -> RegisterState:
    """One step of the Cairo machine (the state transition of the Cairo paper, section 4.5) - what the
    cpu/update_registers constraints of the AIR enforce between consecutive register states."""
    ap, fp, pc = state.ap, state.fp, state.pc
    w = Word(memory[pc])
    size = 1 + w.flag(OP1_IMM)
    dst = memory[w.dst_addr(ap, fp)]
    if w.pc_update == 4:
        npc = pc + size if dst % P == 0 else (pc + memory[w.op1_addr(pc, ap, fp, memory)]) % P
    else:
        res = w.res(pc, ap, fp, memory)
        npc = pc + size if w.pc_update == 0 else res if w.pc_update == 1 else (pc + res) % P
    nap = ap + (w.res(pc, ap, fp, memory) if w.ap_update == 1 else w.ap_update // 2) + (2 if w.flag(OPCODE_CALL) else 0)
    nfp = ap + 2 if w.flag(OPCODE_CALL) else dst if w.flag(OPCODE_RET) else fp
    return RegisterState(nap % P, nfp, npc)

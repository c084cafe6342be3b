"""`cairo-run` artefacts on the input side of the path (sandstorm_amd/binary.py; binary/src/lib.rs:147-222, 565-721),
pinned by the files the reference ships for its own `array-sum` example (tests/golden/example/, data): every register
state follows from the previous one by executing the instruction at pc, the range-check bounds and the public memory
of `air-public-input.json` are what the run implies, and the first/last states are the segment bounds."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "example")


@pytest.fixture(scope="module")
def run():
    from sandstorm_amd import binary, public_input
    with open(os.path.join(EX, "trace.bin"), "rb") as f:
        states = binary.read_register_states(f.read())
    with open(os.path.join(EX, "memory.bin"), "rb") as f:
        memory = binary.read_memory(f.read())
    pi = public_input.AirPublicInput.from_json(os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json"))
    return states, memory, pi


def test_register_states_follow_from_execution(run):
    from sandstorm_amd import binary
    states, memory, pi = run
    assert len(states) == pi.n_steps == 16384
    for i in range(len(states) - 1):
        assert binary.next_state(states[i], memory) == states[i + 1], i


def test_public_input_is_what_the_run_implies(run):
    from sandstorm_amd import binary
    states, memory, pi = run
    prog, exe = pi.memory_segments["program"], pi.memory_segments["execution"]
    assert (states[0].pc, states[0].ap, states[0].fp) == (prog[0], exe[0], exe[0])          # initial_pc / initial_ap
    assert (states[-1].pc, states[-1].ap) == (prog[1], exe[1])                               # final_pc / final_ap
    offs = [o for s in states for w in [binary.Word(memory[s.pc])] for o in (w.off_dst, w.off_op0, w.off_op1)]
    assert (min(offs), max(offs)) == (pi.rc_min, pi.rc_max)          # RangeCheckPool over the instruction offsets (trace.rs:134-140)
    assert all(memory[a] == v for a, v in pi.public_memory)
    assert pi.public_memory_padding() == (1, memory[1])


def test_word_decoding():
    from sandstorm_amd import binary as b
    # the first instruction of the example program (memory[1] = 0x40780017fff7fff: `ap += <imm>`)
    w = b.Word(0x40780017fff7fff)
    assert (w.off_dst, w.off_op0, w.off_op1) == (0x7fff, 0x7fff, 0x8001)
    assert (w.op1_src, w.res_logic, w.pc_update, w.ap_update, w.opcode) == (1, 0, 0, 1, 0)
    assert w.flag_prefix(b.ZERO) == 0 and w.flag_prefix(b.DST_REG) == 0x407 and w.flag_prefix(b.OP1_IMM) == 0x407 >> 2
    assert w.flag(b.DST_REG) == 1 and w.dst_addr(100, 200) == 199 and w.op0_addr(100, 200) == 199 and w.op1_addr(7, 100, 200, []) == 8
    with pytest.raises(ValueError):
        b.read_register_states(bytes(25))
    with pytest.raises(ValueError):
        b.read_memory(bytes(41))


def test_private_input_is_read_as_the_reference_reads_it():
    """sandstorm_amd/binary.py AirPrivateInput against binary/src/lib.rs:343-536: the two files the reference ships (the example's lacks
    the three lists #[serde(default)] allows to be absent; the bootloader's holds two real Pedersen instances), a document with an
    instance of every builtin (the signature's r and w sit one level down, values are strings ruint's FromStr takes), and what serde
    would refuse"""
    from sandstorm_amd.binary import AirPrivateInput
    ex = AirPrivateInput.from_json(os.path.join(EX, "air-private-input.json"))
    assert (ex.trace_path, ex.memory_path) == ("example/trace.bin", "example/memory.bin")
    assert ex.instances == {k: [] for k in ("pedersen", "range_check", "ecdsa", "bitwise", "ec_op", "poseidon")}
    boot = AirPrivateInput.from_json(os.path.join(ROOT, "tests", "golden", "bootloader", "air-private-input.json"))
    assert boot.instances["pedersen"] == [(0, 0, 0x706bd57414b57145b118dd7b92e0d1f040e1a6b6987b842ffb08135699a5ae4),
                                          (1, 0x3bf6a6baa7dad79b6ec242e70a2524309ae0c108c280943859a8b61562cb167, 1)]
    doc = {"trace_path": "t", "memory_path": "m",
           "pedersen": [{"index": 3, "x": "0x10", "y": "17"}], "range_check": [{"index": 0, "value": "0xffff0001"}],
           "ecdsa": [{"index": 1, "pubkey": "0x5", "msg": "0x6", "signature_input": {"r": "0x7", "w": "0x8"}}],
           "bitwise": [{"index": 2, "x": "0x0", "y": "0x%x" % ((1 << 251) - 1)}],
           "ec_op": [{"index": 0, "p_x": "0x1", "p_y": "0x2", "q_x": "0x3", "q_y": "0x4", "m": "0x5"}],
           "poseidon": [{"index": 9, "input_s0": "0x1", "input_s1": "0x2", "input_s2": "0x3"}]}
    got = AirPrivateInput.from_dict(doc).instances
    assert got == {"pedersen": [(3, 16, 17)], "range_check": [(0, 0xffff0001)], "ecdsa": [(1, 5, 6, 7, 8)], "bitwise": [(2, 0, (1 << 251) - 1)],
                   "ec_op": [(0, 1, 2, 3, 4, 5)], "poseidon": [(9, 1, 2, 3)]}
    for broken, what in (({k: v for k, v in doc.items() if k != "pedersen"}, "pedersen"),
                         ({k: v for k, v in doc.items() if k != "trace_path"}, "trace_path"),
                         (dict(doc, ecdsa=[{"index": 1, "pubkey": "0x5", "msg": "0x6", "signature_input": {"r": "0x7"}}]), "ecdsa"),
                         (dict(doc, bitwise=[{"index": 2, "x": 0, "y": "0x1"}]), "string"),
                         (dict(doc, poseidon=[{"index": -1, "input_s0": "0x1", "input_s1": "0x2", "input_s2": "0x3"}]), "u32"),
                         (dict(doc, range_check=[{"index": 0, "value": "0x1" + "0" * 64}]), "256 bits")):
        with pytest.raises(ValueError, match=what):
            AirPrivateInput.from_dict(broken)
